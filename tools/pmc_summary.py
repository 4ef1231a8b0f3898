"""Summarise a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter per dispatch."""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
acc = defaultdict(lambda: defaultdict(list))
for r in rows:
    name = r.get("Kernel_Name", r.get("Kernel Name", "?"))[:60]
    key = (name, r.get("Grid_Size", ""), r.get("LDS_Block_Size", ""))
    acc[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
for key, cs in acc.items():
    n = max(len(v) for v in cs.values())
    print(key, "dispatches", n)
    for c, v in sorted(cs.items()):
        print(f"    {c:32s} {sum(v)/len(v):16.1f}")
