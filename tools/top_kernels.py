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
