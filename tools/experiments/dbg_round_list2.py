import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0]=[ROOT, os.path.join(ROOT,"implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from svsdf_amd import workload
from oracle import orc
cfg,P="C5",20000
w=workload.make(cfg,P=P,minco=svsdf_amd.minco_coeffs)
o=orc.Oracle(w["shape"],safety_hor=w["safety_hor"],weight_p=w["weight_p"],rho=w["rho"],polygon=w["polygon"],head_state=w["head_state"],tail_state=w["tail_state"])
o.set_traj(w["coeffs"],w["T"]); o.set_trig_mode(1)
osdf,ots,og=o.query(w["points"],nthreads=os.cpu_count())
for mode in (0,1):
  for lst in (3,):
    os.environ["SVSDF_UB_FULL"]=str(mode); os.environ["SVSDF_ROUND_LIST"]=str(lst)
    c=svsdf_amd.SvsdfContext(shape=w["shape"],polygon=w["polygon"],safety_hor=w["safety_hor"],weight_p=w["weight_p"],rho=w["rho"],head_state=w["head_state"],tail_state=w["tail_state"],device=0)
    c.set_points(w["points"])
    sdf,ts,g,_=c.query_points(w["coeffs"],w["T"])
    bad=np.nonzero((sdf!=osdf)|(ts!=ots))[0]
    print("mode",mode,"list",lst,"differs from oracle at",len(bad),"points", bad[:5], [(sdf[i],osdf[i],ts[i],ots[i]) for i in bad[:3]], flush=True)
    c.close()
