"""HBM-side traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes). FETCH_SIZE is in KiB and, on gfx950, tallies 128-B read requests at 64 B: doubled here.

  python tools/pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv "kernel substring" GRID_SIZE OUT.json
"""
import csv
import json
import sys


def mean_counter(path, name, grid, counter):
    tot, n = 0.0, 0
    for r in csv.DictReader(open(path)):
        if name in r["Kernel_Name"] and r["Grid_Size"] == grid and r["Counter_Name"] == counter:
            tot += float(r["Counter_Value"])
            n += 1
    return (tot / n if n else 0.0), n


fpath, wpath, name, grid, out = sys.argv[1:6]
f_kib, nf = mean_counter(fpath, name, grid, "FETCH_SIZE")
w_kib, nw = mean_counter(wpath, name, grid, "WRITE_SIZE")
fetch = f_kib * 1024.0 * 2.0  # gfx950: FETCH_SIZE reports half of a wide coalesced read stream
write = w_kib * 1024.0
res = {"kernel": name, "grid_size": int(grid), "launches_fetch_pass": nf, "launches_write_pass": nw,
       "FETCH_SIZE_KiB_mean": round(f_kib, 1), "WRITE_SIZE_KiB_mean": round(w_kib, 1),
       "fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(write), "hbm_bytes_per_launch": int(fetch + write),
       "correction": "FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE x 1024; Infinity-Cache hits are counted "
                     "(memory-side requests), per MI355X_MICROARCH.md",
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --probe-only (two separate runs)"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
