"""Summarise a rocprofv3 --pmc csv: per kernel, mean of each counter per dispatch."""
import csv, sys, collections, glob
files = glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True)
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for row in csv.DictReader(open(f)):
        name = row["Kernel_Name"].split("(")[0].replace("void ", "").replace("lmufft::", "")
        acc[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in sorted(acc.items()):
    if not (k.startswith("k_")):
        continue
    print(k[:60], {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
