"""Summarise a rocprofv3 --pmc counter_collection csv per kernel (mean per dispatch)."""
import csv, collections, sys
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.defaultdict(set)
for r in csv.DictReader(open(sys.argv[1])):
    kn = r["Kernel_Name"].split("(")[0]
    agg[kn][r["Counter_Name"]] += float(r["Counter_Value"]); disp[kn].add(r["Dispatch_Id"])
for kn, d in agg.items():
    if "rpw" in kn:
        n = max(1, len(disp[kn]))
        print(kn, "dispatches", n, {k: round(v / n) for k, v in sorted(d.items())})
