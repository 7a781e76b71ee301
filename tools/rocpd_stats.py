"""Kernel summary (the --stats table) from a rocprofv3 rocpd .db: python tools/rocpd_stats.py IN.db OUT.csv"""
import csv
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name = "name" if "name" in cols else "kernel_name"
rows = cur.execute(f"select {name}, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) from kernels group by {name}").fetchall()
total = float(sum(r[2] for r in rows)) or 1.0
rows.sort(key=lambda r: -r[2])
with open(sys.argv[2], "w", newline="") as f:
    w = csv.writer(f)
    w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs"])
    for r in rows:
        w.writerow([r[0], r[1], r[2], f"{r[3]:.1f}", f"{100.0 * r[2] / total:.3f}", r[4], r[5]])
print(f"{len(rows)} kernels, {total / 1e6:.1f} ms of kernel time")
