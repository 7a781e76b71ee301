"""Per-kernel timeline of ONE frame step out of a rocprofv3 --kernel-trace CSV (eager frame steps: Q3TTS_NO_GRAPH=1).
  python tools/frame_timeline.py <kernel_trace.csv> [rows] [frame_index]
Prints the Predictor pass-5 layers and the first Talker layers kernel by kernel, plus the Predictor / Talker totals."""
import csv
import sys

f = sys.argv[1]
rows_want = int(sys.argv[2]) if len(sys.argv) > 2 else 64
which = int(sys.argv[3]) if len(sys.argv) > 3 else 10
rows = [r for r in csv.DictReader(open(f)) if 'voc' not in r['Kernel_Name'] and 'vgemm' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# the frame's first kernel: rows sampler workgroups + 64 x ceil(rows / 16) projection tiles (1024 predictor columns)
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_sample') and int(r['Grid_Size_X']) == (rows_want + 64 * ((rows_want + 15) // 16)) * 256]
fr = rows[idx[which]:idx[which + 1]]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
print("kernels in frame:", len(fr), "span us", (int(fr[-1]['End_Timestamp']) - int(fr[0]['Start_Timestamp'])) / 1e3)


def show(seq):
    for r in seq:
        print(f"{r['Kernel_Name'][:34]:34s} g=({int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])},{r['Grid_Size_Y']}) {dur(r):7.2f} us")


pn = [i for i, r in enumerate(fr) if r['Kernel_Name'].startswith('k_pred_next')]
print("--- head of frame"); show(fr[:8])
print("--- pass 5"); show(fr[pn[4]:pn[4] + 7])
print("--- talker"); show(fr[pn[-1]:pn[-1] + 7]); show(fr[-2:])
for nm, part in (("predictor", fr[:pn[-1]]), ("talker", fr[pn[-1]:])):
    print(nm, len(part), "kernels,", round(sum(dur(r) for r in part), 1), "us")
