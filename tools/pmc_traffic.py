"""HBM-side traffic per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes). FETCH_SIZE is in KiB and, on gfx950, tallies 128-B read requests at 64 B: doubled here.

  python tools/pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv "kernel substring" GRID_SIZE OUT.json [MIN_FETCH_KIB]
MIN_FETCH_KIB separates two launch shapes that share a kernel instance and a grid size (the counter CSV has neither K nor the grid's
x / y split): only launches of the FETCH pass that report at least that many KiB are averaged, and the WRITE pass is averaged over the
launches at the same positions of the dispatch sequence.
"""
import csv
import json
import sys


def values(path, name, grid, counter):
    return [float(r["Counter_Value"]) for r in csv.DictReader(open(path))
            if name in r["Kernel_Name"] and r["Grid_Size"] == grid and r["Counter_Name"] == counter]


fpath, wpath, name, grid, out = sys.argv[1:6]
min_kib = float(sys.argv[6]) if len(sys.argv) > 6 else 0.0
fv, wv = values(fpath, name, grid, "FETCH_SIZE"), values(wpath, name, grid, "WRITE_SIZE")
keep = [i for i, v in enumerate(fv) if v >= min_kib]
sel_w = [wv[i] for i in keep if i < len(wv)] if len(wv) == len(fv) else wv
f_kib, nf = (sum(fv[i] for i in keep) / len(keep) if keep else 0.0), len(keep)
w_kib, nw = (sum(sel_w) / len(sel_w) if sel_w else 0.0), len(sel_w)
fetch = f_kib * 1024.0 * 2.0  # gfx950: FETCH_SIZE reports half of a wide coalesced read stream
write = w_kib * 1024.0
res = {"kernel": name, "grid_size": int(grid), "min_fetch_kib_filter": min_kib, "launches_fetch_pass": nf, "launches_write_pass": nw,
       "FETCH_SIZE_KiB_mean": round(f_kib, 1), "WRITE_SIZE_KiB_mean": round(w_kib, 1),
       "fetch_bytes_per_launch": int(fetch), "write_bytes_per_launch": int(write), "hbm_bytes_per_launch": int(fetch + write),
       "correction": "FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE x 1024; Infinity-Cache hits are counted "
                     "(memory-side requests), per MI355X_MICROARCH.md",
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --probe-only (two separate runs)"}
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res))
