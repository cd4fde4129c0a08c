"""Prints the top rows of a rocprofv3 kernel_stats.csv found under the given directory."""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
for r in rows[:n]:
    print(r["Name"][:100].ljust(100), r["Calls"].rjust(6), r["TotalDurationNs"].rjust(12), r["AverageNs"][:9].rjust(10), r["Percentage"])
