"""Aggregate a rocprofv3 --pmc counter_collection.csv per kernel: mean of each counter over dispatches."""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for f in [a for a in sys.argv[1:] if not a.startswith("--")]:
    for r in csv.DictReader(open(f)):
        name = r["Kernel_Name"].split("(")[0]
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    if "attn" not in k and "--all" not in sys.argv:
        continue
    print(k)
    for c, v in sorted(cs.items()):
        print(f"   {c:32s} {sum(v) / len(v):18.0f}  (n={len(v)})")
