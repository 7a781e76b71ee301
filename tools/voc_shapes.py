"""Vocoder kernel time by launch shape from a rocprofv3 kernel trace CSV: python tools/voc_shapes.py TRACE.csv"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r["Kernel_Name"]
    if "k_v" in n or "voc" in n:
        key = (n[:20], r["Grid_Size_X"], r["Grid_Size_Y"], r["Workgroup_Size_X"])
        agg[key][0] += 1
        agg[key][1] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
tot = sum(v[1] for v in agg.values())
print("total vocoder kernel time (us)", round(tot))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[: int(sys.argv[2]) if len(sys.argv) > 2 else 30]:
    print(k, v[0], f"{v[1] / v[0]:.1f} us avg", f"{100 * v[1] / tot:.1f}%")
