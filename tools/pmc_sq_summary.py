"""Per-kernel SQ counter summary of a rocprofv3 counter_collection.csv: values are summed
over the per-XCC rows of a dispatch, then averaged over the dispatches of a kernel.
    python tools/pmc_sq_summary.py <csv> <out.json>
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)
(GRBM_GUI_ACTIVE is reported per XCC as well: the sum over 8 XCCs is 8x the kernel's cycles)."""
import collections
import csv
import json
import sys

per = collections.defaultdict(lambda: collections.defaultdict(float))   # (kernel, dispatch) -> counter -> sum
for r in csv.DictReader(open(sys.argv[1])):
    per[(r["Kernel_Name"], r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for (k, _d), c in per.items():
    for n, v in c.items():
        a = acc[k][n]
        a[0] += v
        a[1] += 1
out = {}
rows = []
for k, c in acc.items():
    m = {n: v[0] / v[1] for n, v in c.items()}
    n_disp = max(v[1] for v in c.values())
    cyc = m.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    e = {"dispatches": n_disp, "mean": m}
    if cyc > 0 and "SQ_VALU_MFMA_BUSY_CYCLES" in m:
        e["mfma_util"] = m["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0)
    wc = m.get("SQ_WAVE_CYCLES", 0.0)
    if wc > 0:
        for n, key in (("SQ_WAIT_ANY", "wait_any"), ("SQ_WAIT_INST_ANY", "wait_inst"),
                       ("SQ_ACTIVE_INST_ANY", "active")):
            if n in m:
                e[key] = m[n] / wc
    out[k] = e
    rows.append((cyc * n_disp, k, e))
json.dump(out, open(sys.argv[2], "w"), indent=1)
rows.sort(reverse=True, key=lambda r: r[0])
for _t, k, e in rows[:14]:
    print("%-58s mfma %5.1f%%  wait_any %.2f  wait_inst %.2f  active %.2f" % (
        k[:58], 100 * e.get("mfma_util", 0.0), e.get("wait_any", 0), e.get("wait_inst", 0),
        e.get("active", 0)))
