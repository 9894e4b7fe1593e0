import os, sys
ROOT="/root/repo"
sys.path[:0]=[ROOT, os.path.join(ROOT,"implicit-svsdf-planner_amd")]
import numpy as np, svsdf_amd
from svsdf_amd import workload
for cfg,P in (("C5",20000),):
    w=workload.make(cfg,P=P,minco=svsdf_amd.minco_coeffs)
    ref=None
    for mode in (1,):
        for lst in (0,4,1,2,3):
            os.environ["SVSDF_UB_FULL"]=str(mode); os.environ["SVSDF_ROUND_LIST"]=str(lst)
            c=svsdf_amd.SvsdfContext(shape=w["shape"],polygon=w["polygon"],safety_hor=w["safety_hor"],weight_p=w["weight_p"],rho=w["rho"],head_state=w["head_state"],tail_state=w["tail_state"],device=0)
            c.set_points(w["points"])
            out=c.eval_penalty(w["coeffs"],w["T"]); st=c.stats()
            q=c.query_points(w["coeffs"],w["T"])
            if ref is None: ref=q
            same=all(np.array_equal(a,b) for a,b in zip(q[:3],ref[:3]))
            print(cfg,"mode",mode,"list",lst,"solves",st["solves"],"samples",st["gsip_samples"],"round_scan",st["round_scan_evals"],"scan",st["scan_evals"],"evals",st["sdf_evals"],"iters",st["gsip_iterations"],"same",same, flush=True)
            c.close()
