# kernel trace of one bench step; prints the prefill phase (first k_prompt_rows .. first k_sample_input) aggregated by kernel
cd /tmp && export TMPDIR=/tmp && export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
rm -rf /tmp/prefill_trace
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/prefill_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --no-single --no-cpu-baseline --no-probe > /tmp/prefill_trace.log 2>&1
f=$(find /tmp/prefill_trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, collections
rows = [r for r in csv.DictReader(open(sys.argv[1]))]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'] for r in rows]
i0 = max(i for i, n in enumerate(names) if n.startswith('k_prompt_rows'))   # the timed step's prefill (the last one)
i1 = next(i for i in range(i0, len(names)) if names[i].startswith('k_sample_input'))
fr = rows[i0:i1]
dur = lambda r: (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
agg = collections.defaultdict(lambda: [0, 0.0])
for r in fr:
    k = r['Kernel_Name'].split('(')[0][:44]
    agg[k][0] += 1; agg[k][1] += dur(r)
span = (int(fr[-1]['End_Timestamp']) - int(fr[0]['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"prefill phase: {len(fr)} kernels, span {span:.1f} us, sum of kernel durations {tot:.1f} us, gaps {span - tot:.1f} us")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print(f"{v[1]:9.1f} us  {v[0]:5d} x {v[1] / v[0]:8.2f}  {k}")
print('layer 2, launch by launch:')
big = [i for i, r in enumerate(fr) if 'bgemm' in r['Kernel_Name'] and dur(r) > 30]
for r in fr[big[8]:big[12] + 1]:
    print(f"{dur(r):9.1f} us  grid {r['Grid_Size_X']}x{r['Grid_Size_Y']} wg {r['Workgroup_Size_X']}  {r['Kernel_Name'].split('(')[0][:40]}")
PY
