"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel, mean counter values."""
import csv, glob, sys, collections
d = sys.argv[1]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        for c, v in sorted(cs.items()):
            print(f"   {c:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
