# per-launch durations of the vocoder's kernels in one batched call (64 slots x 4 frames), in launch order
cd /tmp && export TMPDIR=/tmp && export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
rm -rf /tmp/voc_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/voc_trace -- python3 $GRAFT_REPO_ROOT/bench.py --probe-only vocoder > /tmp/voc_trace.log 2>&1
f=$(find /tmp/voc_trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
# the last call: from the last k_voc_embed to the end
idx = max(i for i, n in enumerate(names) if n.startswith('k_voc_embed'))
t0 = int(rows[idx]['Start_Timestamp'])
for r in rows[idx - 1:]:
    n = r['Kernel_Name'].split('(')[0]
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} wg {r['Workgroup_Size_X']}  {n}")
PY
