"""Aggregate rocprofv3 --pmc counter_collection CSVs (one pass per counter) into a
per-kernel summary: mean counter value per dispatch.  Usage:
    python tools/pmc_summary.py out.json FETCH_SIZE=<csv> WRITE_SIZE=<csv>"""
import csv, json, sys, collections
out = {}
for arg in sys.argv[2:]:
    counter, path = arg.split("=", 1)
    acc = collections.defaultdict(lambda: [0.0, 0])
    with open(path) as f:
        for r in csv.DictReader(f):
            if r.get("Counter_Name") != counter:
                continue
            a = acc[r["Kernel_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, (s, n) in acc.items():
        out.setdefault(k, {})[counter] = {"mean": s / n, "dispatches": n}
json.dump(out, open(sys.argv[1], "w"), indent=1)
for k, v in sorted(out.items(), key=lambda kv: -sum(x["mean"] * x["dispatches"] for x in kv[1].values()))[:14]:
    print(k[:70], {c: round(x["mean"], 1) for c, x in v.items()})
