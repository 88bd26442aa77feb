"""Aggregate a rocprofv3 --pmc counter_collection.csv by kernel: sum of each counter over all dispatches."""
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float))
n = collections.Counter()
with open(sys.argv[1]) as f:
    for row in csv.DictReader(f):
        k = row["Kernel_Name"].split("(")[0]
        agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
        n[(k, row["Counter_Name"])] += 1
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    if d.get("SQ_WAVE_CYCLES", 0) < 1e6: continue
    wc = d["SQ_WAVE_CYCLES"]
    print(k, "dispatches", max(v for (kk, c), v in n.items() if kk == k))
    for c, v in sorted(d.items()):
        print("   %-28s %.4g  (%.3f of WAVE_CYCLES)" % (c, v, v / wc))
