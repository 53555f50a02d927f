"""Summarise a rocprofv3 *_kernel_stats.csv: top kernels by total time."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
rows.sort(key=lambda r: -float(r["TotalDurationNs"]))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time %.2f ms (%.2f ms/step over %g steps)" % (tot / 1e6, tot / 1e6 / steps, steps))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%9.3f ms/step %7.1f calls/step %9.1f us avg  %s" % (
        float(r["TotalDurationNs"]) / 1e6 / steps, float(r["Calls"]) / steps,
        float(r["AverageNs"]) / 1e3, r["Name"][:110]))

# ---- category roll-up ------------------------------------------------------
CATS = [
    ("geometry: FPS (side stream)", ("fps_",)),
    ("geometry: ball query / 3-NN", ("ball_query", "three_nn")),
    ("decoder kernels", ("small_linear", "gru_", "attn_", "decoder_fwd_persist", "decoder_bwd_persist")),
    ("BN stats/apply/pool (fwd+bwd)", ("bn_finalize", "bn_bwd_finalize", "bn_bwd_stats", "bn_bwd_apply", "bn_relu", "col_stats", "pool_bwd", "pool_select")),
    ("hand MFMA GEMM", ("rows_gemm", "rows_stream_gemm", "dw_x3", "sa_fused_eval", "planes_gemm", "dw_private_kernel",
                        "sgemm_kernel", "point_gemm_kernel", "bn_bwd_dx_dw64")),
    ("hand fp32 multi-GEMM", ("mgemm_kernel",)),
    ("gather/scatter rows, interpolate", ("sa_gather", "sa_scatter", "three_interpolate", "gather_points", "group_points")),
    ("library GEMM (Tensile)", ("Cijk_",)),
    ("torch elementwise/reduce/copy/fill", ("elementwise", "reduce_kernel", "FillFunctor", "copyBuffer", "fillBuffer", "CatArray", "scatter_gather", "index", "sort", "topk", "gatherTopK", "bitonic", "radix", "softmax", "log_softmax", "nll", "cunn_", "arange", "masked", "where", "clamp", "argmax", "compare", "bucketize")),
    ("optimizer", ("multi_tensor", "adam", "Adam")),
]
agg = {}
for r in rows:
    name = r["Name"]
    cat = "other"
    for c, keys in CATS:
        if any(k in name for k in keys):
            cat = c
            break
    a = agg.setdefault(cat, [0.0, 0.0])
    a[0] += float(r["TotalDurationNs"]) / 1e6 / steps
    a[1] += float(r["Calls"]) / steps
print("\nby category (ms/step, launches/step):")
for c, (ms, calls) in sorted(agg.items(), key=lambda kv: -kv[1][0]):
    print("  %-40s %8.3f ms %8.0f" % (c, ms, calls))
if "other" in agg:
    oth = [r for r in rows if not any(any(k in r["Name"] for k in keys) for _, keys in CATS)]
    for r in oth[:8]:
        print("     other: %8.3f ms/step  %s" % (float(r["TotalDurationNs"]) / 1e6 / steps, r["Name"][:90]))
