"""Build the gfx950 shared library (C ABI of include/svsdf_c.h) in-tree with hipcc.

hipcc cross-compiles for gfx950 without a GPU.  The .so is git-ignored but travels to the GPU
box with the gpurun snapshot.

Eight translation units, compiled in parallel and incrementally: the host layer in four (csrc/svsdf_pipeline.hip: one
device's pipeline + the shape-independent kernels; svsdf_group.hip: in-process multi-GPU; svsdf_capi.hip: the hot
path's C ABI; svsdf_extras.hip: front end, map / mesh / outline helpers, L-BFGS driver) and
csrc/svsdf_shape_slice.hip four times (-DSVSDF_SLICE=0..3: the kernels specialised per shape id, shapes with
id % 4 == slice).  A host-side edit recompiles one small unit; a kernel edit the pipeline unit and the slices.
"""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
DEPS = sorted(glob.glob(os.path.join(CSRC, "*"))) + [os.path.join(HERE, "..", "include", "svsdf_c.h"),
                                                     os.path.abspath(__file__)]
OUT = os.path.join(HERE, "libsvsdf_hip.so")
OBJDIR = os.path.join(HERE, "build")
NSLICES = 4
# what a slice object depends on: its own source and EVERY header of csrc/ (a hand-kept include closure went stale the
# moment a kernel header gained an include -- stale slice objects would then link silently against a changed SolveLaunch /
# GsipState layout); the host objects depend on everything
SLICE_DEPS = [os.path.join(CSRC, "svsdf_shape_slice.hip")] + sorted(glob.glob(os.path.join(CSRC, "*.hpp"))) + \
             [os.path.join(HERE, "..", "include", "svsdf_c.h")]

# -ffp-contract=off: the parity build rounds every operation like the reference's x86-64 build
# (no FMA contraction); see DESIGN.md "Floating-point policy".
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-Wall", "-Wno-unused-result",
          "-pthread"]
LDFLAGS = ["--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-ldl"]


def needs_build(out=OUT):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in DEPS)


def _hipcc():
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    return hipcc if os.path.exists(hipcc) else "hipcc"


def build(force=False, verbose=False, out=OUT, extra_flags=(), tag=""):
    """Compile the five objects concurrently and link them.  `extra_flags` / `tag` / `out` make variant builds
    (tools/: e.g. extra_flags=["-DSVSDF_FAST_BUILD"], tag="fast", out=".../libsvsdf_hip_fast.so")."""
    if not force and not needs_build(out):
        return out
    hipcc = _hipcc()
    os.makedirs(OBJDIR, exist_ok=True)
    units = [(h + tag, os.path.join(CSRC, f"svsdf_{h}.hip"), []) for h in ("pipeline", "group", "capi", "extras")]
    units += [(f"slice{k}{tag}", os.path.join(CSRC, "svsdf_shape_slice.hip"), [f"-DSVSDF_SLICE={k}"]) for k in range(NSLICES)]
    procs = []
    objs_kept = []
    flags_key = " ".join(CFLAGS + list(extra_flags))
    for name, src, defs in units:
        obj = os.path.join(OBJDIR, name + ".o")
        # incremental: an object whose sources (and flags) did not change since it was compiled is kept -- a host-side
        # edit then costs one small translation unit
        deps = SLICE_DEPS if name.startswith("slice") else DEPS
        stamp = obj + ".flags"
        if (not force and not tag and os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == flags_key and
                all(os.path.getmtime(d) <= os.path.getmtime(obj) for d in deps if os.path.exists(d))):
            objs_kept.append(obj)
            continue
        # compile to a temporary name; object and flags stamp appear only after the compiler succeeded (an interrupted
        # compile must not leave a truncated object that looks newer than its sources)
        for stale in (stamp, obj + ".tmp"):
            if os.path.exists(stale):
                os.remove(stale)
        cmd = [hipcc] + CFLAGS + list(extra_flags) + defs + ["-c", src, "-o", obj + ".tmp"]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, obj, subprocess.Popen(cmd, cwd=HERE)))
    objs = list(objs_kept)
    for cmd, obj, p in procs:
        if p.wait() != 0:
            for _, _, q in procs:
                if q.poll() is None:
                    q.kill()
            raise subprocess.CalledProcessError(p.returncode, cmd)
        os.replace(obj + ".tmp", obj)
        open(obj + ".flags", "w").write(flags_key)
        objs.append(obj)
    link = [hipcc] + LDFLAGS + objs + ["-o", out]
    if verbose:
        print(" ".join(link), file=sys.stderr)
    subprocess.check_call(link, cwd=HERE)
    if tag:   # variant builds: the objects (30 MB a set) would travel to the GPU box with every gpurun snapshot
        for obj in objs:
            os.remove(obj)
    return out


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if a != "--force"]
    if args and args[0] == "--variant":   # python build.py --variant <tag> [-Dflags...]: libsvsdf_hip_<tag>.so (fast build)
        tag = args[1]
        print(build(force=True, verbose=True, out=os.path.join(HERE, f"libsvsdf_hip_{tag}.so"),
                    extra_flags=["-DSVSDF_FAST_BUILD"] + args[2:], tag="_" + tag))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
