"""Shape constants as divisors (round 5): svsdf_shapes.hpp divides by three compile-time constants -- sdStar's |ba|^2
(Shape.hpp:597), sdTrapezoid's |k2|^2 = 20 (:763), sdRhombus's |b|^2 = 21.25 (:816) -- and obtains the quotient from the
constant's correctly rounded reciprocal by poly_quot's two residual steps instead of the division's instruction sequence.
The quotient must be the division's own rounding for every operand the kernels can meet: checked here on the host-compiled
product function (tests/cpp/poly_host.cpp) against IEEE division for random operands over 40 decades, exact multiples of
the constant and their neighbours, quotients with short significands and their neighbours (the hard cases for a residual
correction are quotients next to a rounding boundary), signs, and the operands that take the division itself."""
import ctypes as C

import numpy as np

from test_polygon_accel import polyhost, _dp   # noqa: F401  (fixture: builds tests/cpp/libpoly_host.so)


def _constants():
    r, rf = 2.8, 0.6
    k1x, k1y = 0.809016994375, -0.587785252292
    bax = rf * (-k1y) - 0.0
    bay = rf * k1x - 1.0
    return {"star": bax * bax + bay * bay, "trapezoid": 2.0 * 2.0 + 4.0 * 4.0, "rhombus": 1.0 * 1.0 + 4.5 * 4.5}


def test_constant_divisions_round_like_the_division(polyhost):
    rng = np.random.default_rng(2718)
    for name, b in _constants().items():
        a = [rng.standard_normal(400000) * 10.0 ** rng.uniform(-20, 20, 400000)]
        k = rng.integers(-2 ** 40, 2 ** 40, 200000).astype(np.float64)
        mult = k * b                                     # (near-)exact multiples: quotients next to integers
        a += [mult, np.nextafter(mult, np.inf), np.nextafter(mult, -np.inf)]
        q = (rng.integers(1, 2 ** 12, 200000) * 2.0 ** rng.integers(-30, 30, 200000)) * rng.choice([-1.0, 1.0], 200000)
        prod = q * b                                     # quotients with short significands, and halfway-ish neighbours
        a += [prod, np.nextafter(prod, np.inf), np.nextafter(prod, -np.inf), prod * (1 + 2.0 ** -52), prod * (1 - 2.0 ** -53)]
        qh = (rng.integers(2 ** 52, 2 ** 53, 200000) + 0.5) * 2.0 ** rng.integers(-80, 20, 200000)   # (.5 is lost: odd / even significands)
        a += [qh * b, np.nextafter(qh * b, np.inf)]
        a += [np.array([0.0, -0.0, 1e-160, -1e-160, 1e160, np.inf, -np.inf, np.nan, 5e-324, 1e-150, 1e150, 2.2250738585072014e-308])]
        a = np.ascontiguousarray(np.concatenate(a))
        bb = np.full_like(a, b)
        out = np.zeros_like(a)
        used = np.zeros(len(a), dtype=np.int32)
        polyhost.polyhost_quot(_dp(a), _dp(bb), C.c_size_t(len(a)), _dp(out), used.ctypes.data_as(C.POINTER(C.c_int)))
        with np.errstate(all="ignore"):
            ref = a / bb
        same = (out.view(np.int64) == ref.view(np.int64)) | (np.isnan(out) & np.isnan(ref))
        assert same.all(), (name, a[~same][:5], out[~same][:5], ref[~same][:5])
        assert used.mean() > 0.99 and used[-12:].sum() <= 4      # the refinement serves the in-range operands, the rest divide
