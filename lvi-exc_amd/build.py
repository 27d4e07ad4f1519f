"""Build liblvx.so (HIP, gfx950 only) in-tree: python lvi-exc_amd/build.py [--force] [--verbose]."""
import glob
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OUT = os.path.join(HERE, "liblvx.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-fno-gpu-rdc", "-Wno-unused-result"] + os.environ.get("LVX_DEFINES", "").split()   # e.g. LVX_DEFINES=-DLVX_KTIME: cycle counters of the MFMA family kernel (debug print)
# FP contraction: the upstream kernels reproduce the reference's float arithmetic bit for bit (no FMA formation); the FP64
# residual / Jacobian / solver kernels are compared at 1e-12 relative and use FMAs.
CONTRACT = {"lvx_upstream.hip": "off"}
DEFAULT_CONTRACT = os.environ.get("LVX_CONTRACT", "fast")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(os.path.join(HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    if not force and not stale():
        return OUT
    objs = []
    procs = []
    for src in sources():
        obj = os.path.join(CSRC, os.path.basename(src)[:-4] + ".o")
        cmd = [HIPCC] + FLAGS + ["-ffp-contract=" + CONTRACT.get(os.path.basename(src), DEFAULT_CONTRACT), "-c", src, "-o", obj]
        if verbose:
            cmd.insert(1, "-Rpass-analysis=kernel-resource-usage")
            print(" ".join(cmd), flush=True)
        procs.append((subprocess.Popen(cmd), cmd))
        objs.append(obj)
    for p, cmd in procs:
        if p.wait() != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd))
    cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs + ["-ldl", "-Wl,-rpath,/opt/rocm/lib"]   # rocBLAS / rocSOLVER are dlopen'ed on demand (bands wider than 208: lvx_bcr.hip), librccl likewise
    subprocess.check_call(cmd)
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
