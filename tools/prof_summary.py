"""Print a per-kernel table from a rocprofv3 kernel_stats CSV (helper for reading gpurun_out/)."""
import csv, sys, glob
path = sys.argv[1]
files = glob.glob(path + "/**/*kernel_stats.csv", recursive=True) if not path.endswith(".csv") else [path]
rows = list(csv.DictReader(open(files[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[: int(sys.argv[2]) if len(sys.argv) > 2 else 25]:
    n = r["Name"].replace("(anonymous namespace)::", "").replace("void ", "")
    n = n.split("(")[0][:60]
    print("%-60s calls %6s tot %9.2f ms avg %9.1f us %5.1f%%" % (n, r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / tot * 100))
print("total ms", tot / 1e6)
