"""Memory-side traffic of the probe leg from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes): per launch of every kernel kind of a decoder block, and for the frame step as a whole.

  python tools/pmc_traffic.py FETCH_counter_collection.csv WRITE_counter_collection.csv OUT.json

FETCH_SIZE is in KiB and, on gfx950, tallies 128-B read requests at 64 B: doubled here. Infinity-Cache hits are counted (memory-side
requests). The counter CSV has neither K nor the grid's x / y split, and several kinds share a kernel instance and a grid size (the
Talker's and the Predictor's QKV GEMM; O and down projection), so a kind is a (kernel substring, range of reported FETCH KiB) pair; the
WRITE pass is matched by position in the dispatch sequence of the same kernel instance (the two runs launch the same sequence).
The probe leg (`bench.py --probe-only`): 2 x 64 utterances x 24 frames, codes only, eager frame steps; one frame step = the dispatches
between two k_sample_input launches. Prefill kernels (k_bgemm_big, k_qk_prep, k_attend<2, false>, k_prompt_rows, ...) are left out of the
frame-step sum.
"""
import csv
import json
import sys

KINDS = [  # name, kernel substring, FETCH_SIZE KiB range as reported (half of the bytes read)
    ("Talker gate/up GEMM", "k_bgemm<4, 3, 2, true,", 0, 1e12),
    ("Talker QKV GEMM", "k_bgemm<2, 2, 5, false,", 6000, 1e12),
    ("Predictor QKV GEMM", "k_bgemm<2, 2, 5, false,", 0, 6000),
    ("Talker down projection", "k_bgemm<1, 2, 8, false,", 8000, 1e12),
    ("Talker O projection", "k_bgemm<1, 2, 8, false,", 3500, 8000),   # (below 3500: the Predictor's 15 head GEMMs, same instance)
    ("Predictor gate/up GEMM", "k_bgemm<2, 3, 4, false,", 0, 1e12),
    # memory-side bytes = the weights once + the row operands once PER XCD (eight L2s): O 4.19 + 8 x 0.26 + 0.26 MB = 3 190 KiB as reported,
    # down 6.29 + 8 x 0.39 + 0.26 MB = 4 730 KiB
    ("Predictor down projection", "k_bgemm<1, 1, 8, false,", 3900, 1e12),
    ("Predictor O projection", "k_bgemm<1, 1, 8, false,", 0, 3900),
    ("Talker attention", "k_attend_gqa2", 0, 1e12),
    ("Predictor attention", "k_attend_small<2>", 0, 1e12),
]
DECODE = ("k_sample_input", "k_bgemm<", "k_attend_small", "k_attend_pair", "k_attend_gqa2", "k_pred_next")


def rows(path, counter):
    out = []
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            out.append((r["Kernel_Name"], r["Grid_Size"], float(r["Counter_Value"])))
    return out


fpath, wpath, out = sys.argv[1:4]
F, W = rows(fpath, "FETCH_SIZE"), rows(wpath, "WRITE_SIZE")
res = {"correction": "FETCH_SIZE x 1024 x 2 (gfx950 tallies 128-B requests at 64 B), WRITE_SIZE x 1024; Infinity-Cache hits are counted (memory-side requests), per MI355X_MICROARCH.md",
       "command": "rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --kernel-trace --output-format csv -- python3 bench.py --probe-only (two separate runs); python tools/pmc_traffic.py",
       "by_kernel": {}}
for kind, sub, lo, hi in KINDS:
    fi = [(i, v) for i, (n, g, v) in enumerate([x for x in F if sub in x[0]])]
    wv = [v for (n, g, v) in W if sub in n]
    keep = [i for i, v in fi if lo <= v < hi]
    if not keep:
        continue
    fk = sum(fi[i][1] for i in keep) / len(keep)
    same = len(wv) == len(fi)
    wk = (sum(wv[i] for i in keep) / len(keep)) if same else (sum(wv) / max(1, len(wv)))
    res["by_kernel"][kind] = {"kernel": sub, "fetch_kib_range": [lo, hi if hi < 1e11 else None], "launches": len(keep), "FETCH_SIZE_KiB_mean": round(fk, 1),
                              "WRITE_SIZE_KiB_mean": round(wk, 1), "write_matched_by_position": same,
                              "hbm_bytes_per_launch": int(fk * 2048.0 + wk * 1024.0)}
steps = sum(1 for (n, g, v) in F if "k_sample_input" in n)
fsum = sum(v for (n, g, v) in F if any(d in n for d in DECODE) and "k_bgemm_big" not in n)
wsum = sum(v for (n, g, v) in W if any(d in n for d in DECODE) and "k_bgemm_big" not in n)
wsteps = sum(1 for (n, g, v) in W if "k_sample_input" in n)
if steps and wsteps:
    res["frame_step"] = int(fsum / steps * 2048.0 + wsum / wsteps * 1024.0)
    res["frame_steps_counted"] = steps
    res["frame_step_fetch_bytes"] = int(fsum / steps * 2048.0)
    res["frame_step_write_bytes"] = int(wsum / wsteps * 1024.0)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res)[:2000])
