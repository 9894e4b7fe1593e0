"""CPU emulation of the layer-1 table scan: evaluations with today's circle bound vs an added pose-Lipschitz test."""
import sys, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
from oracle import orc
import svsdf_amd
from svsdf_amd import workload
cfg=sys.argv[1]; NQ=int(sys.argv[2]) if len(sys.argv)>2 else 1500
w=workload.make(cfg, P=20000, minco=orc.minco_coeffs)
o=orc.Oracle(w["shape"], safety_hor=w["safety_hor"], weight_p=w["weight_p"], rho=w["rho"], poly_params=w["poly_params"], polygon=w["polygon"], head_state=w["head_state"], tail_state=w["tail_state"])
o.set_traj(w["coeffs"], w["T"])
dur=o.duration()
tk=[]; t=0.0
while t<=dur: tk.append(t); t+=0.15
tk=np.array(tk); K=len(tk)
pose=np.array([o.pos(t) for t in tk])  # x,y,yaw
# R_shape: sampled circumradius (max |q| - sdf(q) on a grid), +1e-6
so=orc._shape_oracle(w["shape"], w["poly_params"], w["polygon"])
R=0.0
for r in np.linspace(0,12,49):
    for a in np.linspace(0,2*np.pi,181):
        q=(r*np.cos(a), r*np.sin(a)); R=max(R, r-so.sdf_at_time(q[0],q[1],0.0))
R*=1.0+1e-6
nch=(K+7)//8
ch=[]
for c in range(nch):
    k0=c*8; k1=min(k0+8,K)
    xs=pose[k0:k1,0]; ys=pose[k0:k1,1]
    cx=0.5*(xs.min()+xs.max()); cy=0.5*(ys.min()+ys.max())
    r=np.hypot(xs-cx,ys-cy).max()
    kc=min(k0+3,k1-1)   # centre pose of the chunk
    dx=np.hypot(xs-pose[kc,0], ys-pose[kc,1]).max()
    sm=(2*np.abs(np.sin(0.5*(pose[k0:k1,2]-pose[kc,2])))).max()
    ch.append((cx,cy,r+R,kc,dx,sm))
rng=np.random.default_rng(5)
pts=w["points"][rng.choice(len(w["points"]),NQ,replace=False),:2]
# GSIP-like samples: circles around interior points
def table(q):
    return np.array([o.sdf_at_time(q[0],q[1],t) for t in tk])
tot_old=tot_new=tot_new_steps4=tot_old_steps4=0; bad=0; n=0
queries=[]
for p in pts:
    queries.append(p)
for p in pts[:NQ//3]:
    D=table(p)
    if D.min()<0:
        for r in (10.0, 3.0, 1.0):
            a=rng.uniform(0,2*np.pi); queries.append(p+r*np.array([np.cos(a),np.sin(a)]))
for q in queries:
    D=table(q)
    lb=np.array([np.hypot(q[0]-c[0],q[1]-c[1])-c[2] for c in ch])
    c0=int(np.argmin(lb))
    # old
    best=D[c0*8:c0*8+8].min(); ev=min(8,K-c0*8)
    bestk=c0*8+int(np.argmin(D[c0*8:c0*8+8]))
    for c in range(nch):
        if c==c0: continue
        if not (lb[c]>best):
            seg=D[c*8:c*8+8]; ev+=len(seg)
            m=seg.min()
            if m<best or (m==best and c*8+int(np.argmin(seg))<bestk): best=m; bestk=c*8+int(np.argmin(seg))
    # new
    best2=D[c0*8:c0*8+8].min(); ev2=min(8,K-c0*8); bestk2=c0*8+int(np.argmin(D[c0*8:c0*8+8]))
    for c in range(nch):
        if c==c0: continue
        if not (lb[c]>best2):
            cx,cy,rb,kc,dx,sm=ch[c]
            fc=D[kc]; ev2+=1
            dist=np.hypot(q[0]-pose[kc,0], q[1]-pose[kc,1])
            lb2=fc-(dx+sm*dist)*(1+1e-9)-1e-9
            if fc<best2 or (fc==best2 and kc<bestk2): best2=fc; bestk2=kc
            if lb2>best2: continue
            seg=D[c*8:c*8+8]; ev2+=len(seg)-1
            m=seg.min()
            if m<best2 or (m==best2 and c*8+int(np.argmin(seg))<bestk2): best2=m; bestk2=c*8+int(np.argmin(seg))
    kt=int(np.argmin(D))  # earliest minimal
    if bestk!=kt or bestk2!=kt: bad+=1
    tot_old+=ev; tot_new+=ev2; n+=1
print(cfg, "K",K,"queries",n,"evals/query old %.1f new %.1f  wrong seeds %d"%(tot_old/n, tot_new/n, bad))
