"""Builds libs2c_hip.so (the gfx950 kernels behind include/s2c_ops.h) with hipcc.

The library is pure HIP + a C ABI -- no torch headers -- so it cross-compiles in
seconds on a machine without a GPU and travels in-tree to the GPU box.
"""
import glob
import os
import subprocess
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_DIR = os.path.join(_HERE, "lib")
LIB_PATH = os.path.join(LIB_DIR, "libs2c_hip.so")

HIPCC_FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17",
    # canonical arithmetic: IEEE binary32, source order, no FMA contraction
    "-ffp-contract=off",
    # hardware float atomics (global_atomic_add_f32) instead of CAS loops in the
    # scatter-add (grad) kernels
    "-munsafe-fp-atomics",
    "-fPIC",
]
OBJ_DIR = os.path.join(LIB_DIR, "obj")


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(
        os.path.join(_HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def _headers_mtime():
    deps = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(
        os.path.join(_HERE, "..", "include", "*.h"))
    return max(os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False):
    """Compile every .hip source (one object per source, in parallel, only the stale ones
    unless `force`) and link lib/libs2c_hip.so.  Returns the path."""
    if not force and not _stale():
        return LIB_PATH
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "hipcc")
    os.makedirs(OBJ_DIR, exist_ok=True)
    hdr = _headers_mtime()
    jobs, objs = [], []
    for src in sources():
        obj = os.path.join(OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if force or not os.path.exists(obj) or \
                os.path.getmtime(obj) < max(os.path.getmtime(src), hdr):
            jobs.append([hipcc] + HIPCC_FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
