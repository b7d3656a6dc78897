"""Build libpclean_hip.so (gfx950) in-tree with hipcc.

No CPU fallback exists: importing `pclean_amd` works without the library, but
every compute entry point raises if the shared object or a GPU is missing.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libpclean_hip.so")
SOURCES = ["api.hip", "comm.hip", "commit.hip", "dist_kernels.hip", "enum_kernels.hip", "root_wave.hip", "random_kernels.hip", "sweep.hip", "eval.hip", "latent.hip"]
HEADERS = ["ctx.h", "../../include/pclean_hip.h", "../../include/pclean_detmath.h", "../../include/pclean_philox.h"]
# -ffp-contract=off: the parity contract (include/pclean_detmath.h) needs plain
# IEEE mul/add on device, identical to the gcc-built oracle.
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
         "-Wall", "-Wno-unused-function"] + os.environ.get("PCLEAN_EXTRA_HIPCC_FLAGS", "").split()


def _stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    for extra in os.listdir(CSRC):
        if extra.endswith((".h", ".hip")):
            deps.append(os.path.join(CSRC, extra))
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def _obj_stale(src, obj):
    if not os.path.exists(obj):
        return True
    t = os.path.getmtime(obj)
    deps = [src, os.path.abspath(__file__)] + [os.path.join(CSRC, h) for h in HEADERS]
    deps += [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    return any(os.path.exists(d) and os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=True):
    if not force and not _stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objs, procs = [], []
    for s in SOURCES:  # one hipcc per stale translation unit, all in parallel
        o = os.path.join(CSRC, s.replace(".hip", ".o"))
        objs.append(o)
        if not force and not _obj_stale(os.path.join(CSRC, s), o):
            continue
        cmd = [hipcc] + [f for f in FLAGS if f] + ["-c", os.path.join(CSRC, s), "-o", o]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((cmd, subprocess.Popen(cmd)))
    for cmd, pr in procs:
        if pr.wait() != 0:
            raise subprocess.CalledProcessError(pr.returncode, cmd)
    cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs + ["-ldl"]
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
