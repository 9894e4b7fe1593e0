"""Timing of library variants (SVSDF_LIB_VARIANT) in subprocesses (GPU)."""
import os, sys, subprocess
cfg, P = sys.argv[1], sys.argv[2]
for v in sys.argv[3:]:
    env = dict(os.environ); env["SVSDF_LIB_VARIANT"] = "" if v == "default" else v
    out = subprocess.run([sys.executable, "-u", os.path.join(os.path.dirname(__file__), "sweep.py"), cfg, P, "SVSDF_G=4", "SVSDF_G=8"],
                         env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout.decode()
    for l in out.strip().split("\n"): print(f"[{v}] {l}")
