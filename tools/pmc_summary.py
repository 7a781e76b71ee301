"""Per-kernel means of rocprofv3 PMC counters: python tools/pmc_summary.py counter_collection.csv [name-filter]"""
import collections
import csv
import sys

rows = csv.DictReader(open(sys.argv[1]))
flt = sys.argv[2] if len(sys.argv) > 2 else ""
agg = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(set)
for r in rows:
    name = r["Kernel_Name"]
    if flt and flt not in name:
        continue
    key = (name[:34], r["Grid_Size"], r.get("Workgroup_Size", ""))
    agg[key][r["Counter_Name"]] += float(r["Counter_Value"])
    cnt[key].add(r["Dispatch_Id"])
for key, c in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_BUSY_CYCLES", kv[1].get("SQ_WAVE_CYCLES", 0))):
    n = len(cnt[key])
    print(key, "n=%d" % n, " ".join(f"{k}={v / n:.3g}" for k, v in sorted(c.items())))
