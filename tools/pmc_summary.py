"""Summarises a rocprofv3 --pmc counter_collection.csv: mean counter value per kernel launch."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            k = r["Kernel_Name"].split("(")[0][:60]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v)/len(v):14.4g}  (n={len(v)})")
