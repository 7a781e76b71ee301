# kernel trace of the probe leg (64 utterances, codes only) in graph mode; prints one frame step kernel by kernel
cd /tmp && export TMPDIR=/tmp && export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
rm -rf /tmp/frame_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/frame_trace -- python3 $GRAFT_REPO_ROOT/bench.py --probe-only predictor > /tmp/frame_trace.log 2>&1
f=$(find /tmp/frame_trace -name '*kernel_trace.csv' | head -1)
python3 $GRAFT_REPO_ROOT/tools/frame_timeline.py "$f" 64 12
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if r['Kernel_Name'].startswith('k_sample') and int(r['Grid_Size_X']) == (64 + 64 * 4) * 256]
fr = rows[idx[12]:idx[13]]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0])
for r in fr:
    k = r['Kernel_Name'].split('(')[0][:40] + ' g' + str(int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])) + 'x' + r['Grid_Size_Y']
    agg[k][0] += 1; agg[k][1] += dur(r)
span = (int(fr[-1]['End_Timestamp']) - int(fr[0]['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"frame: {len(fr)} kernels, span {span:.1f} us, sum of durations {tot:.1f} us, gaps {span - tot:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]:8.1f} us  {v[0]:4d} x {v[1] / v[0]:6.2f}  {k}")
PY
