# how much does the vocoder beside the decoder cost a frame step? kernel trace of one bench step in graph mode; per frame step (64 rows):
# its span (k_sample_input to the next k_sample_input) against the vocoder kernel time that ran inside that span
cd /tmp && export TMPDIR=/tmp && export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
rm -rf /tmp/ov_trace
rocprofv3 --kernel-trace --output-format csv -d /tmp/ov_trace -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-probe > /tmp/ov_trace.log 2>&1
f=$(find /tmp/ov_trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'PY'
import csv, sys, bisect, collections
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
isv = lambda n: 'voc' in n or 'vgemm' in n or 'vconv' in n
voc = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if isv(r['Kernel_Name'])]
# k_bgemm launches of the vocoder's transformer run on the vocoder stream too: tell them apart by queue
qs = {}
for r in rows:
    qs.setdefault(r.get('Queue_Id', '?'), [0, 0]); qs[r.get('Queue_Id', '?')][isv(r['Kernel_Name'])] += 1
print('queues (decoder-named, vocoder-named kernels):', qs)
vq = max(qs, key=lambda q: qs[q][1])
voc = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if r.get('Queue_Id', '?') == vq]
dq_count = collections.Counter(r.get('Queue_Id', '?') for r in rows if r['Kernel_Name'].startswith('k_sample_input'))
dq = dq_count.most_common(1)[0][0]
dec = [r for r in rows if r.get('Queue_Id', '?') == dq]
starts = [i for i, r in enumerate(dec) if r['Kernel_Name'].startswith('k_sample_input')]
vs = [v[0] for v in voc]
out = []
for a, b in zip(starts[:-1], starts[1:]):
    g = int(dec[a]['Grid_Size_X']) // int(dec[a]['Workgroup_Size_X'])
    t0, t1 = int(dec[a]['Start_Timestamp']), int(dec[b]['Start_Timestamp'])
    if t1 - t0 > 20e6: continue
    ov = 0
    for s, e in voc[max(0, bisect.bisect_left(vs, t0) - 200):]:
        if s >= t1: break
        ov += max(0, min(e, t1) - max(s, t0))
    busy = sum(int(r['End_Timestamp']) - int(r['Start_Timestamp']) for r in dec[a:b])
    out.append((g, (t1 - t0) / 1e3, ov / 1e3, busy / 1e3, b - a))
full = [o for o in out if o[0] == max(o[0] for o in out)]
print('frame steps at the full bucket:', len(full), 'kernels per step', full[0][4])
none = [o for o in full if o[2] < 50]; some = [o for o in full if o[2] >= 50]
m = lambda x: sum(x) / max(1, len(x))
print(f"no vocoder inside: {len(none)} steps, span {m([o[1] for o in none]):.0f} us, kernel time {m([o[3] for o in none]):.0f} us")
print(f"vocoder inside:    {len(some)} steps, span {m([o[1] for o in some]):.0f} us, kernel time {m([o[3] for o in some]):.0f} us, vocoder kernel time inside {m([o[2] for o in some]):.0f} us")
for lo, hi in ((50, 1000), (1000, 2000), (2000, 3000), (3000, 4000), (4000, 1e9)):
    s = [o for o in full if lo <= o[2] < hi]
    if s: print(f"  vocoder {lo}-{hi} us inside: {len(s)} steps, span {m([o[1] for o in s]):.0f} us, kernel time {m([o[3] for o in s]):.0f}")
allspan = m([o[1] for o in full]); print(f"all: {allspan:.0f} us")
# (gaps between consecutive kernels of a step cannot be attributed under the tracer: with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 the host submits the
#  graph's nodes one by one, so ~600-800 us of gaps per step are the host's even with nothing else on the GPU)
PY
