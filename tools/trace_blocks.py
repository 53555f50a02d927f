"""Per-block kernel durations from a rocprofv3 kernel trace: consecutive launches of the same kernel
(a benchmark loop) -> one line with the average GPU duration.  usage: trace_blocks.py s_kernel_trace.csv [min_calls]"""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
minc = int(sys.argv[2]) if len(sys.argv) > 2 else 5
skip = ("multi_colsum", "elementwise", "fillBuffer", "distribution", "copyBuffer")
blocks = []
for r in rows:
    n = r["Kernel_Name"]
    if any(s in n for s in skip):
        continue
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    if blocks and blocks[-1][0] == n:
        blocks[-1][1].append(d)
    else:
        blocks.append((n, [d]))
for n, ds in blocks:
    if len(ds) >= minc:
        ds2 = sorted(ds)[: max(1, len(ds) * 3 // 4)]       # drop the slow tail (warm-up)
        print("%8.1f us x%-3d %s" % (sum(ds2) / len(ds2), len(ds), n[:100]))
