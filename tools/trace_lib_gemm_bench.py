"""Where does the DEFAULT bench.py command (hipGraph capture, replay, instrumented pass, evidence runs)
call a library GEMM?  Wraps torch.mm / addmm / bmm / baddbmm / matmul / F.linear, runs bench.main()
with the given arguments and prints every call site with its count and shapes.

    python tools/trace_lib_gemm_bench.py [bench.py arguments ...]        (default: --no-cpu-baseline --no-fed)

(round-5 review: `profiles/r05_bench_default_kernel_stats.csv` holds a `Cijk_*_Bias_*` row of 120 calls
although the eager cfg3 step makes no library call.)"""
import collections
import os
import sys
import traceback

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

log = collections.Counter()
shapes_of = {}


def site():
    out = []
    for fr in reversed(traceback.extract_stack()[:-2]):
        if ROOT in fr.filename and "trace_lib_gemm_bench" not in fr.filename:
            out.append("%s:%d" % (os.path.relpath(fr.filename, ROOT), fr.lineno))
        if len(out) == 3:
            break
    return " <- ".join(out) or "?"


def wrap(name, fn):
    def f(*a, **k):
        if any(torch.is_tensor(x) and x.is_cuda for x in a):
            key = (name, site())
            log[key] += 1
            shapes_of.setdefault(key, tuple(tuple(x.shape) for x in a if torch.is_tensor(x)))
        return fn(*a, **k)
    return f


for n in ("mm", "addmm", "bmm", "baddbmm", "matmul"):
    setattr(torch, n, wrap(n, getattr(torch, n)))
torch.nn.functional.linear = wrap("linear", torch.nn.functional.linear)
torch.Tensor.matmul = wrap("Tensor.matmul", torch.Tensor.matmul)
torch.Tensor.__matmul__ = wrap("Tensor.__matmul__", torch.Tensor.__matmul__)

import bench  # noqa: E402

sys.argv = ["bench.py"] + (sys.argv[1:] or ["--no-cpu-baseline", "--no-fed"])
try:
    bench.main()
finally:
    print("library GEMM call sites of `%s`:" % " ".join(sys.argv), file=sys.stderr)
    for (name, where), c in sorted(log.items(), key=lambda kv: -kv[1]):
        print("%5d x %-16s %s   %s" % (c, name, where, shapes_of[(name, where)]), file=sys.stderr)
    print("total %d calls" % sum(log.values()), file=sys.stderr)
