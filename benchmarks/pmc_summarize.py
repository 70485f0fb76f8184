"""Average each hardware counter per kernel from a rocprofv3 counter-collection CSV.
usage: pmc_summarize.py <counter_collection.csv> [kernel-name substring ...]"""
import collections
import csv
import sys

agg = collections.defaultdict(lambda: collections.defaultdict(list))
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"]
        if len(sys.argv) > 2 and not any(s in name for s in sys.argv[2:]):
            continue
        agg[name[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print(k)
    for c, vals in sorted(v.items()):
        print(f"    {c:28s} mean {sum(vals) / len(vals):14.4g}   (n={len(vals)})")
    if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
        h, m = sum(v["TCC_HIT_sum"]), sum(v["TCC_MISS_sum"])
        print(f"    L2 hit rate {100 * h / max(1.0, h + m):.1f}%")
