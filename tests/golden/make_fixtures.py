"""Extract DATA fixtures from the reference's demo assets (run in the build container only;
/root/reference does not exist on the GPU box).  Output: tests/golden/reference_assets.json

  * shapes/<name>.obj        -> vertex lists (the reference's own meshes of each analytic shape;
                                every vertex lies on/inside the zero level set of that shape's SDF) and
                                triangle lists ("mesh_faces", zero-based): the 13 meshes are the inputs of
                                BASELINE config 5 (arbitrary .obj mesh -> z = 0 outline -> Polygon SDF)
  * pcds/map_<name>.pcd      -> obstacle point clouds (ASCII PCD v0.7, FIELDS x y z)
  * pcds/trajectory_<name>.txt, config/<name>.yaml -> start/end poses and the hot-path constants

These are inputs/expected properties, not code.
"""
import json
import os
import re

REF = "/root/reference/src/plan_manager"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_assets.json")
KEYS = ["safety_hor", "weight_p", "rho", "inittime", "kernel_size", "occupancy_resolution", "threads_num"]


def read_obj(path):
    return [[round(float(v), 6) for v in l.split()[1:4]] for l in open(path) if l.startswith("v ")]


def read_obj_faces(path):
    out = []
    for l in open(path):
        if l.startswith("f "):
            idx = [int(t.split("/")[0]) - 1 for t in l.split()[1:]]
            out += [[idx[0], idx[k], idx[k + 1]] for k in range(1, len(idx) - 1)]
    return out


def read_pcd(path):
    lines = open(path).read().split("\n")
    i = [k for k, l in enumerate(lines) if l.startswith("DATA")][0]
    return [[float(v) for v in l.split()] for l in lines[i + 1:] if l.strip()]


def read_yaml_consts(path):
    out = {}
    txt = open(path).read()
    for k in KEYS:
        m = re.search(r"^%s:\s*([-0-9.eE+]+)" % k, txt, re.M)
        if m:
            out[k] = float(m.group(1))
    m = re.search(r"^poly_params:\s*\[([^\]]*)\]", txt, re.M)
    out["poly_params"] = [float(v) for v in m.group(1).split(",")]
    m = re.search(r'^inputdata:\s*"([^"]+)"', txt, re.M)
    out["inputdata"] = m.group(1)
    return out


def main():
    names = sorted(f[:-5] for f in os.listdir(os.path.join(REF, "config")) if f.endswith(".yaml"))
    assets = {"source": "ZJU-FAST-Lab/Implicit-SVSDF-Planner @ 2024_08_07, src/plan_manager/{shapes,pcds,config}",
              "shapes": {}, "mesh_faces": {}, "scenarios": {}, "maps": {}}
    for n in names:
        assets["shapes"][n] = read_obj(os.path.join(REF, "shapes", n + ".obj"))
        assets["mesh_faces"][n] = read_obj_faces(os.path.join(REF, "shapes", n + ".obj"))
        sc = read_yaml_consts(os.path.join(REF, "config", n + ".yaml"))
        se = open(os.path.join(REF, "pcds", "trajectory_%s.txt" % n)).read().split("\n")
        sc["start"] = [float(v) for v in se[0].split()[1:4]]
        sc["end"] = [float(v) for v in se[1].split()[1:4]]
        assets["scenarios"][n] = sc
    for n in ["star", "sdHorseshoe", "sdHeart"]:
        assets["maps"][n] = read_pcd(os.path.join(REF, "pcds", "map_%s.pcd" % n))
    json.dump(assets, open(OUT, "w"), separators=(",", ":"))
    print(OUT, os.path.getsize(OUT), "bytes;", len(names), "shapes")


if __name__ == "__main__":
    main()
