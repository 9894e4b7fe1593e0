import sys, numpy as np
import os; ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [ROOT, os.path.join(ROOT, "implicit-svsdf-planner_amd")]
from oracle import orc
from svsdf_amd import workload
from svsdf_amd.binding import SHAPES
rng=np.random.default_rng(1)
N=150000
for sh in SHAPES:
    poly = workload.star_outline() if sh=="Polygon" else None
    o=orc._shape_oracle(sh,(0.0,0.0,0.0),poly)
    p=rng.uniform(-9,9,(N,2))
    d=10**rng.uniform(-7,0.5,N)
    a=rng.uniform(0,2*np.pi,N)
    q=p+np.stack([d*np.cos(a),d*np.sin(a)],1)
    worst=0.0; wi=-1
    fp=np.array([o.sdf_at_time(x,y,0.0) for x,y in p]); fq=np.array([o.sdf_at_time(x,y,0.0) for x,y in q])
    dist=np.hypot(*(p-q).T)
    r=np.abs(fp-fq)/dist
    # ignore pairs where difference is at rounding level
    ok=np.abs(fp-fq)>1e-9
    i=np.argmax(np.where(ok,r,0))
    print(f"{sh:18s} max ratio {r[ok].max():.9f} at p={p[i]} d={dist[i]:.3e} f={fp[i]:.6f},{fq[i]:.6f}  frac>1+1e-9: {(r[ok]>1+1e-9).mean():.2e}")
