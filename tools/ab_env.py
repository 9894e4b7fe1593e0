"""A/B of environment settings on ONE library build: same child as tools/exp_variants.py, different env per run.

usage: python tools/ab_env.py <variant or -> "<ENV=V[,ENV=V]>;<ENV=V...>;..." [config[,config...]] [points] [steps]
The first spec is the baseline (may be empty); every run is its own process with a hard timeout."""
import json
import os
import subprocess
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from exp_variants import CHILD, ROOT  # noqa: E402


def run(variant, spec, cfg, P, steps):
    env = dict(os.environ)
    if variant and variant != "-":
        env["SVSDF_LIB_VARIANT"] = variant
    for kv in [x for x in spec.split(",") if x]:
        k, v = kv.split("=", 1)
        env[k] = v
    try:
        out = subprocess.run([sys.executable, "-c", CHILD % dict(root=ROOT), cfg, str(P), str(steps)], env=env,
                             stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    except subprocess.TimeoutExpired:
        return dict(error="timeout")
    if out.returncode:
        return dict(error=out.stderr.decode()[-600:])
    r = json.loads(out.stdout.decode().strip().splitlines()[-1])
    err = out.stderr.decode().strip()
    if err:
        r["stderr"] = err[-200:]
    return r


def main():
    variant = sys.argv[1]
    specs = sys.argv[2].split(";")
    cfgs = sys.argv[3].split(",") if len(sys.argv) > 3 else ["C3", "NS"]
    P = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    steps = int(sys.argv[5]) if len(sys.argv) > 5 else 10
    from svsdf_cfg import default_points
    for cfg in cfgs:
        base = None
        for spec in specs:
            r = run(variant, spec, cfg, P or default_points(cfg), steps)
            if base is None:
                base = r
            same = r.get("hash") == base.get("hash")
            sp = base.get("ms", 0) / r["ms"] if "ms" in r else 0
            print(f"{cfg:4s} [{spec or 'baseline'}] {json.dumps(r)} identical={same} speedup={sp:.3f}", flush=True)


if __name__ == "__main__":
    main()
