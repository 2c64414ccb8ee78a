"""Build recipe: hipcc --offload-arch=gfx950 -> gpu-fluid-simulation_amd/libfluidsim_hip.so (in-tree).

-ffp-contract=off keeps the kernels' f32 arithmetic in the written association
(no FMA contraction) so results match the CPU oracle bit for bit.
-fno-slp-vectorize (float kernels, see NO_SLP): at -O3 the SLP vectoriser packs adjacent scalar f32 mul/add into v_pk_* ops;
on gfx950 those issue at half rate and cost extra v_mov to line the operands up (measured at 16M:
force 0.93 -> 0.85 ms, density 0.30 -> 0.29 ms without it; same bits either way).
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "libfluidsim_hip.so")
SOURCES = ["engine.hip", "comm.hip", "buffer.hip", "kernels_step.hip", "kernels_sort.hip", "kernels_slab.hip", "kernels_csort.hip", "kernels_field.hip", "sim3d.hip"]
HEADERS = ["fs_device.h", "fs_kernels.h", "fs_scan.h", "sort_policy.h", os.path.join("..", "..", "include", "fluidsim.h")]
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
]


# float-heavy kernels only: the integer sort kernels measured ~1 % faster with the vectoriser on
NO_SLP = {"kernels_step.hip", "sim3d.hip", "kernels_slab.hip"}


def _flags(src):
    return FLAGS + (["-fno-slp-vectorize"] if os.path.basename(src) in NO_SLP else [])


def _stale(out, deps):
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def is_stale():
    """True when the library is missing or older than any source/header (what build() would act on)."""
    deps = [os.path.join(CSRC, s) for s in SOURCES] + [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    return _stale(OUT, deps)


def build(force=False, verbose=True, out=None, extra_flags=()):
    """Compile and link.  `out`/`extra_flags` build an experimental variant next to the default library."""
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if out is not None:
        return _build_variant(hipcc, out, list(extra_flags), verbose)
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    hdrs = [os.path.join(CSRC, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs = []
    procs = []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [sp] + hdrs):
            cmd = [hipcc] + _flags(sp) + ["-c", sp, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    if force or procs or _stale(OUT, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl"]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return OUT


def _build_variant(hipcc, out, extra, verbose):
    """Same per-source flags as the default build plus `extra`; objects go to build/<variant>/."""
    objdir = os.path.join(HERE, "build", os.path.splitext(os.path.basename(out))[0])
    os.makedirs(objdir, exist_ok=True)
    procs, objs = [], []
    for src in SOURCES:
        sp = os.path.join(CSRC, src)
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        cmd = [hipcc] + _flags(sp) + extra + ["-c", sp, "-o", obj]
        if verbose:
            print(" ".join(cmd), flush=True)
        procs.append((src, subprocess.Popen(cmd)))
    for src, p in procs:
        if p.wait() != 0:
            raise RuntimeError(f"hipcc failed on {src}")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs + ["-ldl"])
    return out


if __name__ == "__main__":
    build(force="--force" in sys.argv)
