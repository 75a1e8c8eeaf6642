"""Summarise a rocprofv3 *_kernel_stats.csv: per-kernel ms per step."""
import csv, glob, sys
path = sys.argv[1]
steps = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
f = glob.glob(path)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 34]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")[:70]
    print("%-72s %5d %8.2f ms/step %8.1f us %5.1f%%" % (n, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6 / steps, float(r["AverageNs"]) / 1e3, float(r["Percentage"])))
print("total ms/step", tot / 1e6 / steps)
