#!/usr/bin/env python3
"""Second, independent restatement of the reference path in pure Python (floats are IEEE doubles,
math.sin/cos/atan2/sqrt are the platform libm, no FMA) -> tests/golden/golden_small.json.

Written directly from the reference sources (not from oracle/svsdf_oracle.c), function by function:
  SWM = src/swept_volume/include/swept_volume/sw_manager.hpp
  BEO = src/planner_algorithm/include/planner_algorithm/back_end_optimizer.hpp
  SHP = src/utils/include/utils/Shape.hpp      TRJ = src/utils/include/utils/trajectory.hpp
  MNC = src/utils/include/utils/minco.hpp
The C oracle must reproduce these vectors (tests/test_golden_vectors.py); the HIP path is checked
against the oracle.  Three implementations, two authorship passes over the reference.

Run:  python tests/golden/make_golden.py      (about a minute; pure-Python loops, small cases only)
"""
import json
import math
import sys
import os

PI = 3.14159265358979323846  # SHP:31


# ----------------------------------------------------------------------------- trajectory (TRJ)
class Traj:
    """Trajectory<5> built by MINCO::getTrajectory (MNC:515-528): piece i keeps rows 6i..6i+5 of b
    (row k = coefficient of s^k)."""

    def __init__(self, rows, durs):
        self.N = len(durs)
        self.T = list(durs)
        self.c = [[list(rows[6 * i + k]) for k in range(6)] for i in range(self.N)]

    def total(self):  # TRJ:410-419
        s = 0.0
        for d in self.T:
            s += d
        return s

    def locate(self, t):  # TRJ:498-516
        idx = 0
        while idx < self.N and t > self.T[idx]:
            t -= self.T[idx]
            idx += 1
        if idx == self.N:
            idx -= 1
            t += self.T[idx]
        return idx, t

    def pos(self, t):  # TRJ:518-522 + Piece::getPos TRJ:104-114 (constant term first, tn *= t)
        i, s = self.locate(t)
        p = [0.0, 0.0, 0.0]
        tn = 1.0
        for k in range(6):
            for d in range(3):
                p[d] += tn * self.c[i][k][d]
            tn *= s
        return p

    def vel(self, t):  # TRJ:524-528 + Piece::getVel TRJ:116-128
        i, s = self.locate(t)
        v = [0.0, 0.0, 0.0]
        tn = 1.0
        n = 1
        for k in range(1, 6):
            for d in range(3):
                v[d] += n * tn * self.c[i][k][d]
            tn *= s
            n += 1
        return v


# ----------------------------------------------------------------------------- shapes (SHP)
def _norm(x, y):
    return math.sqrt(x * x + y * y)


def _clip(v, lo, hi):
    return max(min(v, hi), lo)


def _cs(v):  # std::copysign(1.0, v)
    return math.copysign(1.0, v)


class Shape:
    def __init__(self, name, poly_params=(0.0, 0.0, 0.0), verts=None):
        self.name = name
        self.tx, self.ty = poly_params[0], poly_params[1]
        yaw = poly_params[2] * PI / 180.0  # SHP:287
        self.R = [[math.cos(yaw), -math.sin(yaw)], [math.sin(yaw), math.cos(yaw)]]
        self.verts = verts

    def local(self, x, y):  # ((pos_rel - trans) * Rotate).head(2): row vector times matrix
        dx, dy = x - self.tx, y - self.ty
        return dx * self.R[0][0] + dy * self.R[1][0], dx * self.R[0][1] + dy * self.R[1][1]

    def sdf(self, x, y):
        if self.name == "Polygon":
            return self.polygon(x, y)[0]
        px, py = self.local(x, y)
        return getattr(self, "f_" + self.name)(px, py)

    # SHP:584-601
    def f_star(self, px, py):
        r, rf = 2.8, 0.6
        k1 = (0.809016994375, -0.587785252292)
        k2 = (-k1[0], k1[1])
        px = abs(px)
        m = 2.0 * max(k1[0] * px + k1[1] * py, 0.0)
        px, py = px - m * k1[0], py - m * k1[1]
        m = 2.0 * max(k2[0] * px + k2[1] * py, 0.0)
        px, py = px - m * k2[0], py - m * k2[1]
        px = abs(px)
        py -= r
        ba = (rf * (-k1[1]) - 0.0, rf * k1[0] - 1.0)
        h = _clip((px * ba[0] + py * ba[1]) / (ba[0] * ba[0] + ba[1] * ba[1]), 0.0, r)
        return _norm(px - ba[0] * h, py - ba[1] * h) * _cs(py * ba[0] - px * ba[1])

    # SHP:870-891 (angle 20.5 is in radians)
    def f_sdHorseshoe(self, px, py):
        r, w = 1.5, (1.55, 0.20)
        c = (math.cos(20.5), math.sin(20.5))
        px = abs(px)
        l = _norm(px, py)
        nx = -c[0] * px + c[1] * py
        ny = c[1] * px + c[0] * py
        first = nx
        if first <= 0 and ny <= 0:
            nx = l * _cs(-c[0])
        if first <= 0:
            ny = l
        nx = nx - w[0]
        ny = abs(ny - r) - w[1]
        return _norm(max(nx, 0.0), max(ny, 0.0)) + min(0.0, max(nx, ny))

    # SHP:939-952
    def f_sdHeart(self, px, py):
        px, py = px / 4.0, py / 4.0
        px = abs(px)
        if py + px > 1.0:
            ax, ay = px - 0.25, py - 0.75
            return 4 * (math.sqrt(ax * ax + ay * ay) - math.sqrt(2.0) / 4.0)
        v1 = (px - 0.0) * (px - 0.0) + (py - 1.0) * (py - 1.0)
        t = max(px + py, 0.0)
        cx, cy = px - 0.5 * t, py - 0.5 * t
        v2 = cx * cx + cy * cy
        return 4 * (math.sqrt(min(v1, v2)) * _cs(px - py))

    # SHP:698-711
    def f_sdCutDisk(self, px, py):
        r, h = 5.0, 2.0
        w = math.sqrt(r * r - h * h)
        px = abs(px)
        s = max((h - r) * px * px + w * w * (h + r - 2.0 * py), h * px - w * py)
        if s < 0.0:
            return _norm(px, py) - r
        if px < w:
            return h - py
        return _norm(px - w, py - h)

    # SHP:1448-1476 with the edge helpers SHP:1370-1401 (no trans / Rotate)
    def polygon(self, x, y):
        dmin, cmin, rs = 1e9, (0.0, 0.0), 0
        n = len(self.verts)
        for i in range(n):
            s, e = self.verts[i], self.verts[(i + 1) % n]
            vx, vy = e[0] - s[0], e[1] - s[1]
            wx, wy = x - s[0], y - s[1]
            t = (wx * vx + wy * vy) / (vx * vx + vy * vy)
            t = 0.0 if t < 0.0 else (1.0 if t > 1.0 else t)
            c = (s[0] + t * vx, s[1] + t * vy)
            d = _norm(x - c[0], y - c[1])
            if d < dmin:
                dmin, cmin = d, c
            ths = math.atan2(s[1] - y, s[0] - x)
            the = math.atan2(e[1] - y, e[0] - x)
            ths = ths + 2 * PI if ths < 0.0 else ths
            the = the + 2 * PI if the < 0.0 else the
            if not abs(ths - the) < PI:
                rs += 1
        return (dmin if rs % 2 == 0 else -dmin), cmin, rs

    def grad(self, x, y):
        """getonlyGrad1: FD macro SHP:35-53; Polygon: analytic SHP:1505-1531."""
        if self.name == "Polygon":
            _, c, rs = self.polygon(x, y)
            vx, vy = x - c[0], y - c[1]
            z = vx * vx + vy * vy
            if z > 0.0:
                nn = math.sqrt(z)
                vx, vy = vx / nn, vy / nn
            return (vx, vy) if rs % 2 == 0 else (-vx, -vy)
        dx = 0.000001
        t0 = x - dx
        old = self.sdf(t0, y)
        t0 += 2 * dx
        gx = self.sdf(t0, y) - old
        t1 = y - dx
        old = self.sdf(x, t1)
        t1 += 2 * dx
        gy = self.sdf(x, t1) - old
        return gx / (2 * dx), gy / (2 * dx)


# ----------------------------------------------------------------------------- swept-volume SDF (SWM)
class Swept:
    def __init__(self, shape, traj):
        self.shape, self.traj = shape, traj
        self.dur = traj.total()  # updateTraj SWM:376-385 (total < 300 here)
        self.n_solves = 0

    def sdf_at(self, px, py, t):  # SWM:741-750 with SWM:465-474, 521-526
        xt = self.traj.pos(t)
        s, c = math.sin(xt[2]), math.cos(xt[2])
        dx, dy = px - xt[0], py - xt[1]
        return self.shape.sdf(c * dx + s * dy, (-s) * dx + c * dy)

    def grad_at(self, px, py, t):  # SWM:779-788
        xt = self.traj.pos(t)
        s, c = math.sin(xt[2]), math.cos(xt[2])
        dx, dy = px - xt[0], py - xt[1]
        return self.shape.grad(c * dx + s * dy, (-s) * dx + c * dy)

    def sdf_dot(self, px, py, t):  # SWM:799-806
        t1 = max(0.0, t - 0.000001)
        t2 = min(self.dur, t + 0.000001)
        return (self.sdf_at(px, py, t2) - self.sdf_at(px, py, t1)) * 500000

    def choice_t_init(self, px, py, dt):  # SWM:538-581
        min_dis, seed = 1e9, 0.0
        layer, terminal, t = 1, self.dur, 0.0
        while layer <= 4:
            if layer == 1:
                t = 0.0
            if layer > 1:
                t = max(0.0, seed - 10 * dt)
                terminal = min(self.dur, seed + 10 * dt)
            while t <= terminal:
                d = self.sdf_at(px, py, t)
                if d < min_dis:
                    seed, min_dis = t, d
                t += dt
            dt *= 0.1
            layer += 1
        return seed

    def gradient_descent(self, tmin, tmax, x0, px, py):  # SWM:1249-1325
        alpha, tol = 0.01, 1e-16
        x, prev, it, stop, fx = x0, 10000000.0, 0, False, 0.0
        while it < 1000 and not stop and abs(x - prev) > tol:
            if it == 0:
                fx = self.sdf_at(px, py, x)
            self.sdf_dot(px, py, x)  # SWM:1291 (value unused, no side effect)
            tau = alpha
            prev = x
            for div in range(1, 30):
                it += 1
                g = self.sdf_dot(px, py, x)
                change = -tau * (int(g > 0) - int(g < 0))
                xc = max(min(x + change, tmax), tmin)
                fc = self.sdf_at(px, py, xc)
                if (fc - fx) < 0:
                    x, fx = xc, fc
                    break
                tau = 0.5 * tau
                if div == 29:
                    stop = True
        return fx, x

    def solve(self, px, py):  # getSDFofSweptVolume<false,true> SWM:844-866
        self.n_solves += 1
        ts = self.choice_t_init(px, py, 0.15)
        tmin, tmax = max(0.0, ts - 3.4), min(ts + 3.4, self.dur)
        f, t = self.gradient_descent(tmin, tmax, ts, px, py)
        g = self.grad_at(px, py, t)
        return f, t, (g[0], g[1])

    def true_sdf(self, px, py):  # getTrueSDFofSweptVolume<true> SWM:916-1018
        f, t, g = self.solve(px, py)
        if f > 0:
            return f, t, g
        vel = self.traj.vel(t)
        nv = lambda v: math.sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2])
        if nv(vel) < 0.01:
            if t < 0.1:
                ts = t
                while ts <= self.dur:
                    vel = self.traj.vel(ts)
                    if nv(vel) >= 0.01:
                        break
                    ts += 0.1
            elif t > self.dur - 0.1:
                ts = t
                while ts >= 0:
                    vel = self.traj.vel(ts)
                    if nv(vel) >= 0.01:
                        break
                    ts -= 0.1
        # SampleSet2D::initSet SWM:73-103
        r = 10
        theta0 = math.atan2(vel[0], -vel[1])
        if theta0 < 0:
            theta0 += 2 * PI
        theta_res, real_t, star_th, it = PI + 0.1, t, 0.0, 1
        while True:
            max_g = -100000
            th = theta0
            while th < theta0 + 2 * PI:  # getElements SWM:60-71, single ring rk = 1.0
                yx, yy = px + 1.0 * r * math.cos(th), py + 1.0 * r * math.sin(th)
                cg, ct, _ = self.solve(yx, yy)
                if cg > max_g:
                    max_g, real_t, star_th = cg, ct, th
                th += theta_res
            r_star = r - max_g
            r = r_star
            if it > 8:
                break
            if abs(max_g) < 0.1:
                break
            theta_res = max(0.3, theta_res / (2 + 1))  # expandSet SWM:105-110
            theta0 = star_th
            it += 1
        cx, cy = px + 1.0 * r_star * math.cos(star_th), py + 1.0 * r_star * math.sin(star_th)
        gx, gy = cx - px, cy - py
        z = gx * gx + gy * gy
        if z > 0.0:
            nn = math.sqrt(z)
            gx, gy = gx / nn, gy / nn
        return -r_star, real_t, (gx, gy)


# ----------------------------------------------------------------------------- penalty (BEO)
def smoothed_l1(x, mu):  # BEO:316-340
    if x < 0.0:
        return False, None, None
    if x > mu:
        return True, x - 0.5 * mu, 1.0
    xd = x / mu
    sq = xd * xd
    mm = mu - 0.5 * x
    return True, mm * sq * xd, sq * ((-0.5) * xd + 3.0 * mm / mu)


def penalty(sw, rows, durs, points, safety_hor, weight_p):
    """addSaftyPenaOnSweptVolume (serial twin BEO:624-772 == loop body BEO:786-865), summed in order."""
    N = len(durs)
    cost, gradT = 0.0, [0.0] * N
    gradC = [[0.0, 0.0, 0.0] for _ in range(6 * N)]
    per = []
    for p in points:
        px, py = p[0], p[1]
        sdf, tstar, g = sw.true_sdf(px, py)
        per.append([sdf, tstar, g[0], g[1]])
        i, s1 = sw.traj.locate(tstar)
        s2 = s1 * s1; s3 = s2 * s1; s4 = s2 * s2; s5 = s4 * s1
        b0 = [1.0, s1, s2, s3, s4, s5]
        b1 = [0.0, 1.0, 2.0 * s1, 3.0 * s2, 4.0 * s3, 5.0 * s4]
        pos = [sum(rows[6 * i + k][d] * b0[k] for k in range(6)) for d in range(3)]
        vel = [sum(rows[6 * i + k][d] * b1[k] for k in range(6)) for d in range(3)]
        yaw = pos[2]
        sy, cy = math.sin(yaw), math.cos(yaw)
        gr = [g[0], g[1]]
        if sdf < 0:  # BEO:832
            gr = [cy * g[0] + sy * g[1], (-sy) * g[0] + cy * g[1]]
        ok, L, dL = smoothed_l1(safety_hor - sdf, 0.01)  # grad_cost_p_sw BEO:1031-1066
        gx = gy = gyaw = pena = 0.0
        if ok and L > 0:
            sgx = -dL * ((-cy) * gr[0] + sy * gr[1])
            sgy = -dL * ((-sy) * gr[0] + (-cy) * gr[1])
            dx, dy = px - pos[0], py - pos[1]
            v0 = (-sy) * dx + cy * dy
            v1 = (-cy) * dx + (-sy) * dy
            gyv = (-dL * gr[0]) * v0 + (-dL * gr[1]) * v1
            gx, gy, gyaw, pena = weight_p * sgx, weight_p * sgy, weight_p * gyv, weight_p * L
        cost += pena
        for k in range(6):
            gradC[6 * i + k][0] += b0[k] * gx
            gradC[6 * i + k][1] += b0[k] * gy
            gradC[6 * i + k][2] += b0[k] * gyaw
        gdT = -((gx * vel[0] + gy * vel[1]) + gyaw * vel[2])
        for j in range(i):
            gradT[j] += gdT
    return cost, gradT, gradC, per


# ----------------------------------------------------------------------------- MINCO S3NU (MNC)
def minco(head, tail, q, T):
    """setParameters MNC:435-513 with a dense no-pivot LU (identical arithmetic to the banded one:
    fill-in stays inside the band and skipped zero multipliers change nothing).  head/tail: rows
    pos, vel, acc; q: N-1 waypoints; returns rows (6N x 3), A (LU), T powers."""
    N = len(T)
    n = 6 * N
    A = [[0.0] * n for _ in range(n)]
    b = [[0.0, 0.0, 0.0] for _ in range(n)]
    T1 = list(T); T2 = [t * t for t in T1]; T3 = [T2[i] * T1[i] for i in range(N)]
    T4 = [T2[i] * T2[i] for i in range(N)]; T5 = [T4[i] * T1[i] for i in range(N)]
    A[0][0] = 1.0; A[1][1] = 1.0; A[2][2] = 2.0
    b[0], b[1], b[2] = list(head[0]), list(head[1]), list(head[2])
    for i in range(N - 1):
        r = 6 * i
        A[r + 3][r + 3] = 6.0; A[r + 3][r + 4] = 24.0 * T1[i]; A[r + 3][r + 5] = 60.0 * T2[i]; A[r + 3][r + 9] = -6.0
        A[r + 4][r + 4] = 24.0; A[r + 4][r + 5] = 120.0 * T1[i]; A[r + 4][r + 10] = -24.0
        A[r + 5][r:r + 6] = [1.0, T1[i], T2[i], T3[i], T4[i], T5[i]]
        A[r + 6][r:r + 6] = [1.0, T1[i], T2[i], T3[i], T4[i], T5[i]]; A[r + 6][r + 6] = -1.0
        A[r + 7][r + 1:r + 6] = [1.0, 2 * T1[i], 3 * T2[i], 4 * T3[i], 5 * T4[i]]; A[r + 7][r + 7] = -1.0
        A[r + 8][r + 2:r + 6] = [2.0, 6 * T1[i], 12 * T2[i], 20 * T3[i]]; A[r + 8][r + 8] = -2.0
        b[r + 5] = list(q[i])
    e = N - 1
    A[n - 3][n - 6:n] = [1.0, T1[e], T2[e], T3[e], T4[e], T5[e]]
    A[n - 2][n - 5:n] = [1.0, 2 * T1[e], 3 * T2[e], 4 * T3[e], 5 * T4[e]]
    A[n - 1][n - 4:n] = [2, 6 * T1[e], 12 * T2[e], 20 * T3[e]]
    b[n - 3], b[n - 2], b[n - 1] = list(tail[0]), list(tail[1]), list(tail[2])
    lo = up = 6
    for k in range(n - 1):  # factorizeLU MNC:96-128
        iM = min(k + lo, n - 1)
        piv = A[k][k]
        for i in range(k + 1, iM + 1):
            if A[i][k] != 0.0:
                A[i][k] /= piv
        jM = min(k + up, n - 1)
        for j in range(k + 1, jM + 1):
            c = A[k][j]
            if c != 0.0:
                for i in range(k + 1, iM + 1):
                    if A[i][k] != 0.0:
                        A[i][j] -= A[i][k] * c
    for j in range(n):  # solve MNC:133-163
        for i in range(j + 1, min(j + lo, n - 1) + 1):
            if A[i][j] != 0.0:
                for d in range(3):
                    b[i][d] -= A[i][j] * b[j][d]
    for j in range(n - 1, -1, -1):
        for d in range(3):
            b[j][d] /= A[j][j]
        for i in range(max(0, j - up), j):
            if A[i][j] != 0.0:
                for d in range(3):
                    b[i][d] -= A[i][j] * b[j][d]
    return b, A, (T1, T2, T3, T4, T5)


def dot3(a, b):
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def cost_function(shape_name, poly_params, verts, head, tail, x, points, safety_hor, weight_p, rho):
    """costFunctionLmbmParallel BEO:344-408."""
    n = len(x)
    N = (n + 3) // 4
    tau = x[:N]
    T = [((0.5 * t + 1.0) * t + 1.0) if t > 0.0 else 1.0 / ((0.5 * t - 1.0) * t + 1.0) for t in tau]  # BEO:213-226
    q = [x[N + 3 * i:N + 3 * i + 3] for i in range(N - 1)]
    b, A, (T1, T2, T3, T4, T5) = minco(head, tail, q, T)
    energy = 0.0  # MNC:530-543
    for i in range(N):
        c3, c4, c5 = b[6 * i + 3], b[6 * i + 4], b[6 * i + 5]
        energy += (36.0 * dot3(c3, c3) * T1[i] + 144.0 * dot3(c4, c3) * T2[i] + 192.0 * dot3(c4, c4) * T3[i] +
                   240.0 * dot3(c5, c3) * T3[i] + 720.0 * dot3(c5, c4) * T4[i] + 720.0 * dot3(c5, c5) * T5[i])
    gC = [[0.0, 0.0, 0.0] for _ in range(6 * N)]  # MNC:550-567
    gT = [0.0] * N  # MNC:569-582
    for i in range(N):
        c3, c4, c5 = b[6 * i + 3], b[6 * i + 4], b[6 * i + 5]
        for d in range(3):
            gC[6 * i + 5][d] = 240.0 * c3[d] * T3[i] + 720.0 * c4[d] * T4[i] + 1440.0 * c5[d] * T5[i]
            gC[6 * i + 4][d] = 144.0 * c3[d] * T2[i] + 384.0 * c4[d] * T3[i] + 720.0 * c5[d] * T4[i]
            gC[6 * i + 3][d] = 72.0 * c3[d] * T1[i] + 144.0 * c4[d] * T2[i] + 240.0 * c5[d] * T3[i]
        gT[i] = (36.0 * dot3(c3, c3) + 288.0 * dot3(c4, c3) * T1[i] + 576.0 * dot3(c4, c4) * T2[i] +
                 720.0 * dot3(c5, c3) * T2[i] + 2880.0 * dot3(c5, c4) * T3[i] + 3600.0 * dot3(c5, c5) * T4[i])
    sw = Swept(Shape(shape_name, poly_params, verts), Traj(b, T))
    pc, pT, pC, per = penalty(sw, b, T, points, safety_hor, weight_p)
    cost = energy + pc
    for r in range(6 * N):
        for d in range(3):
            gC[r][d] += pC[r][d]
    for i in range(N):
        gT[i] += pT[i]
    pos_cost = cost - energy
    # propogateGrad MNC:584-654: solveAdj MNC:168-197
    n6 = 6 * N
    adj = [list(r) for r in gC]
    lo = up = 6
    for j in range(n6):
        for d in range(3):
            adj[j][d] /= A[j][j]
        for i in range(j + 1, min(j + up, n6 - 1) + 1):
            if A[j][i] != 0.0:
                for d in range(3):
                    adj[i][d] -= A[j][i] * adj[j][d]
    for j in range(n6 - 1, -1, -1):
        for i in range(max(0, j - lo), j):
            if A[j][i] != 0.0:
                for d in range(3):
                    adj[i][d] -= A[j][i] * adj[j][d]
    gradP = [list(adj[6 * i + 5]) for i in range(N - 1)]
    gradTimes = [0.0] * N
    for i in range(N - 1):
        c = [b[6 * i + k] for k in range(6)]
        B1 = [None] * 6
        B1[2] = [-(c[1][d] + 2.0 * T1[i] * c[2][d] + 3.0 * T2[i] * c[3][d] + 4.0 * T3[i] * c[4][d] + 5.0 * T4[i] * c[5][d]) for d in range(3)]
        B1[3] = list(B1[2])
        B1[4] = [-(2.0 * c[2][d] + 6.0 * T1[i] * c[3][d] + 12.0 * T2[i] * c[4][d] + 20.0 * T3[i] * c[5][d]) for d in range(3)]
        B1[5] = [-(6.0 * c[3][d] + 24.0 * T1[i] * c[4][d] + 60.0 * T2[i] * c[5][d]) for d in range(3)]
        B1[0] = [-(24.0 * c[4][d] + 120.0 * T1[i] * c[5][d]) for d in range(3)]
        B1[1] = [-120.0 * c[5][d] for d in range(3)]
        s = 0.0
        for d in range(3):
            for r in range(6):
                s += B1[r][d] * adj[6 * i + 3 + r][d]
        gradTimes[i] = s
    e = N - 1
    c = [b[6 * e + k] for k in range(6)]
    B2 = [[-(c[1][d] + 2.0 * T1[e] * c[2][d] + 3.0 * T2[e] * c[3][d] + 4.0 * T3[e] * c[4][d] + 5.0 * T4[e] * c[5][d]) for d in range(3)],
          [-(2.0 * c[2][d] + 6.0 * T1[e] * c[3][d] + 12.0 * T2[e] * c[4][d] + 20.0 * T3[e] * c[5][d]) for d in range(3)],
          [-(6.0 * c[3][d] + 24.0 * T1[e] * c[4][d] + 60.0 * T2[e] * c[5][d]) for d in range(3)]]
    s = 0.0
    for d in range(3):
        for r in range(3):
            s += B2[r][d] * adj[6 * N - 3 + r][d]
    gradTimes[e] = s
    for i in range(N):
        gradTimes[i] += gT[i]
    tsum = 0.0
    for t in T:
        tsum += t
    cost += rho * tsum
    g = [0.0] * n
    for i in range(N):  # backwardGradT BEO:268-289
        gt = gradTimes[i] + rho
        if tau[i] > 0:
            g[i] = gt * (tau[i] + 1.0)
        else:
            den = (0.5 * tau[i] - 1.0) * tau[i] + 1.0
            g[i] = gt * (1.0 - tau[i]) / (den * den)
    for i in range(N - 1):
        g[N + 3 * i:N + 3 * i + 3] = gradP[i]
    return cost, g, [pos_cost, cost - pos_cost, cost], b, T, per, sw.n_solves


# ----------------------------------------------------------------------------- cases
def main():
    cases = []
    head = [[2.0, 1.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]   # rows: pos, vel, acc
    tail = [[14.0, 9.0, 0.0], [0.0, 0.0, 0.0], [0.0, 0.0, 0.0]]
    N = 4
    q = [[5.0, 2.5, 0.4], [8.5, 6.0, -0.3], [11.0, 7.0, 0.5]]
    T = [2.2, 2.6, 2.4, 2.1]
    tau = [(math.sqrt(2.0 * t - 1.0) - 1.0) if t > 1.0 else (1.0 - math.sqrt(2.0 / t - 1.0)) for t in T]  # BEO:228-241
    x = tau + [v for w in q for v in w]
    sets = {
        "star": dict(pp=(0.0, 0.0, 0.0), sh=0.7, pts=[(4.0, 5.5), (9.0, 2.0), (12.5, 10.5), (1.0, -2.5), (6.0, 3.2),
                                                          (10.0, 6.6), (7.2, 8.9), (13.2, 5.1), (3.0, 3.4)]),
        "sdHorseshoe": dict(pp=(0.0, 0.0, 0.0), sh=0.7, pts=[(4.0, 4.6), (9.5, 3.5), (12.0, 9.5), (6.4, 3.6), (10.2, 6.9),
                                                                 (2.4, 2.9), (8.0, 7.8)]),
        "sdHeart": dict(pp=(0.0, 0.0, 0.0), sh=0.8, pts=[(3.0, 6.0), (9.0, 1.0), (13.0, 12.5), (6.0, 5.0), (10.5, 9.0),
                                                             (7.5, 3.0), (1.5, 3.5)]),
        "sdCutDisk": dict(pp=(0.0, -3.0, 0.0), sh=0.87, pts=[(3.0, 8.0), (10.0, 0.5), (15.5, 13.0), (6.0, 3.0), (12.0, 3.0)]),
        "Polygon": dict(pp=(0.0, 0.0, 0.0), sh=0.7, pts=[(4.0, 3.3), (9.0, 4.4), (12.5, 9.0), (6.0, 2.9), (2.0, 6.5)],
                        verts=[(6, -0.1), (6, 0.1), (-6, 0.1), (-6, -0.1)]),
    }
    for name, cfg in sets.items():
        pts = [[p[0], p[1], 0.0] for p in cfg["pts"]]
        f, g, c3, rows, Tv, per, nsolves = cost_function(name, cfg["pp"], cfg.get("verts"), head, tail, x, pts,
                                                         cfg["sh"], 60.0, 3.8)
        print(name, "cost", f, "solves", nsolves, "interior", sum(1 for p in per if p[0] <= 0))
        cases.append(dict(shape=name, poly_params=list(cfg["pp"]), polygon=cfg.get("verts"), safety_hor=cfg["sh"],
                          weight_p=60.0, rho=3.8, head=head, tail=tail, N=N, x=x, T=Tv, coeffs=rows, points=pts,
                          per_point=per, n_solves=nsolves, cost=f, costs3=c3, grad=g))
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_small.json")
    json.dump(dict(generator="tests/golden/make_golden.py (pure-Python restatement of the reference path)",
                   cases=cases), open(out, "w"))
    print("wrote", out, os.path.getsize(out), "bytes")

# ---- front-end collision check + shape kernels ---------------------------------------
def check_sub_sw_collision(shape, father, child, pts):  # SWM:1171-1211
    dt = 0.02
    for (px, py) in pts:
        min_sdf = 1e9
        kt = 0.0
        while kt <= 1.0:
            lx = kt * child[0] + (1 - kt) * father[0]
            ly = kt * child[1] + (1 - kt) * father[1]
            yaw = kt * child[2] + (1 - kt) * father[2]
            s, c = math.sin(yaw), math.cos(yaw)
            dx, dy = px - lx, py - ly
            temp = shape.sdf(c * dx + s * dy, (-s) * dx + c * dy)  # posEva2Rel SWM:521-526
            if temp < min_sdf:
                min_sdf = temp
            if min_sdf < 0:
                return False
            kt += dt
    return True


def shape_kernels(shape, ks, count, resu, safemargin):  # SHP:386-430 with the (pos_rel, R_obj) overloads
    size_side = int(0.5 * (ks - 1))
    yaw_res = 2 * PI / count
    maps, yaws = [], []
    yaw, ind = -PI, 0
    while yaw < PI:
        if ind < count:
            c, s = math.cos(yaw), math.sin(yaw)
            rows = []
            for a in range(ks):
                row = []
                for b in range(ks):
                    x = resu * a - size_side * resu
                    y = resu * b - size_side * resu
                    qx, qy = shape.local(x, y)
                    px, py = qx * c + qy * s, qx * (-s) + qy * c  # row vector times R_obj = AngleAxisd(yaw, Z)
                    row.append(1 if getattr(shape, "f_" + shape.name)(px, py) <= safemargin else 0)
                rows.append(row)
            maps.append(rows)
            yaws.append(yaw)
        yaw += yaw_res
        ind += 1
    return maps, yaws, ind


def main_frontend():
    import random
    rng = random.Random(20240807)
    cases = []
    shapes = {"star": (0.0, 0.0, 0.0), "sdHorseshoe": (0.0, 0.0, 0.0), "sdHeart": (0.0, 0.0, 0.0),
              "sdCutDisk": (0.0, -3.0, 0.0), "Polygon": (0.0, 0.0, 0.0)}
    for name, pp in shapes.items():
        verts = [(6, -0.1), (6, 0.1), (-6, 0.1), (-6, -0.1)] if name == "Polygon" else None
        sh = Shape(name, pp, verts)
        edges = []
        for _ in range(24):
            fx, fy = rng.uniform(5, 25), rng.uniform(5, 25)
            father = [fx, fy, rng.uniform(-PI, PI)]
            child = [fx + rng.choice([-1.0, 0.0, 1.0]), fy + rng.choice([-1.0, 0.0, 1.0]),
                     father[2] + rng.uniform(-0.7, 0.7)]
            n = rng.randrange(1, 6)
            pts = [[fx + rng.uniform(-6, 6), fy + rng.uniform(-6, 6)] for _ in range(n)]
            edges.append(dict(father=father, child=child, points=pts,
                              free=check_sub_sw_collision(sh, father, child, pts)))
        case = dict(shape=name, poly_params=list(pp), polygon=verts, edges=edges)
        if name != "Polygon":
            maps, yaws, ind = shape_kernels(sh, 17, 18, 1.0, 0.5)
            case.update(kernel_size=17, kernel_count=18, resolution=1.0, safemargin=0.5, kernels=maps, yaws=yaws,
                        loop_count=ind)
        print(name, "free edges", sum(e["free"] for e in edges), "/", len(edges))
        cases.append(case)
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_frontend.json")
    json.dump(dict(generator="tests/golden/make_golden.py main_frontend() (pure-Python restatement of "
                             "checkSubSWCollision SWM:1171-1211 and initShape SHP:386-430)", cases=cases),
              open(out, "w"))
    print("wrote", out, os.path.getsize(out), "bytes")


if __name__ == "__main__":
    if "--frontend-only" not in sys.argv:
        main()
    main_frontend()
