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

# S2C_NVCC_CONTRACT = 1 | 2: the variant of the library whose index-producing ops (FPS, ball query,
# three_nn) form a*a + b*b + c*c with the fused multiply-adds an nvcc build of the reference may
# use (csrc/s2c_common.h: sq3) -> lib/libs2c_hip_nvcc<k>.so.  Only the sources that contain
# those ops are recompiled; every other object is shared with the canonical build.
CONTRACT = int(os.environ.get("S2C_NVCC_CONTRACT", "0") or 0)
INDEX_SOURCES = ("s2c_ops.hip", "s2c_fps_small.hip", "s2c_fps_bucket.hip", "s2c_fps_cells.hip",
                 "s2c_bq_grid.hip")


def lib_path(contract=None):
    k = CONTRACT if contract is None else int(contract)
    return LIB_PATH if k == 0 else os.path.join(LIB_DIR, "libs2c_hip_nvcc%d.so" % k)


def sources():
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def _stale(path=None):
    path = path or LIB_PATH
    if not os.path.exists(path):
        return True
    t = os.path.getmtime(path)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(
        os.path.join(_HERE, "..", "include", "*.h"))
    return any(os.path.getmtime(d) > t for d in deps)


def _headers_mtime():
    deps = glob.glob(os.path.join(CSRC, "*.h")) + glob.glob(
        os.path.join(_HERE, "..", "include", "*.h"))
    return max(os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False, contract=None):
    """Compile every .hip source (one object per source, in parallel, only the stale ones
    unless `force`) and link lib/libs2c_hip.so (contract = 1 | 2: the nvcc-contraction variant,
    default from the environment).  Returns the path."""
    k = CONTRACT if contract is None else int(contract)
    out = lib_path(k)
    if k:
        build(force=False, verbose=verbose, contract=0)      # the shared objects
    if not force and not _stale(out):
        return out
    from concurrent.futures import ThreadPoolExecutor
    hipcc = os.environ.get("HIPCC", "hipcc")
    vdir = OBJ_DIR if k == 0 else OBJ_DIR + "_nvcc%d" % k
    os.makedirs(OBJ_DIR, exist_ok=True)
    os.makedirs(vdir, exist_ok=True)
    hdr = _headers_mtime()
    jobs, objs = [], []
    for src in sources():
        variant = k != 0 and os.path.basename(src) in INDEX_SOURCES
        obj = os.path.join(vdir if variant else OBJ_DIR, os.path.basename(src)[:-4] + ".o")
        objs.append(obj)
        if (force and (k == 0 or variant)) or not os.path.exists(obj) or \
                os.path.getmtime(obj) < max(os.path.getmtime(src), hdr):
            extra = ["-DS2C_NVCC_CONTRACT=%d" % k] if variant else []
            jobs.append([hipcc] + HIPCC_FLAGS + extra + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as pool:
        list(pool.map(run, jobs))
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out] + objs)
    return out


PROBES_DIR = os.path.join(_HERE, "..", "tools", "probes")
PROBES_PATH = os.path.join(PROBES_DIR, "libs2c_probes.so")


def build_probes(force=False):
    """tools/probes/s2c_probe.hip -> tools/probes/libs2c_probes.so: streaming-copy probes and the
    CU-hogging kernel tests/test_fused_gpu.py provokes the persistent decoder's give-up path with.
    Test / diagnostic code: NOT part of libs2c_hip.so."""
    src = os.path.join(PROBES_DIR, "s2c_probe.hip")
    if not force and os.path.exists(PROBES_PATH) and \
            os.path.getmtime(PROBES_PATH) >= os.path.getmtime(src):
        return PROBES_PATH
    subprocess.check_call([os.environ.get("HIPCC", "hipcc")] + HIPCC_FLAGS +
                          ["-shared", "-o", PROBES_PATH, src])
    return PROBES_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
