"""Summarise a rocprofv3 --pmc counter_collection CSV: mean counter value per kernel (short name)."""
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    with open(path) as f:
        for r in csv.DictReader(f):
            full = r["Kernel_Name"].split("(")[0]
            k = full if len(full) <= 70 else full[:70]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    if "nsx" not in k and "deform" not in k and "ens_" not in k:
        continue
    print(k)
    for c, v in sorted(d.items()):
        print(f"   {c:34s} mean {sum(v)/len(v):16.1f}  n={len(v)}")
