# same-box A/B of the vocoder variants beside the decoder: bench.py --no-probe, default / NOTAP / old output conv
cd $GRAFT_REPO_ROOT
for v in "" "Q3TTS_VOC_NOTAP=1" "Q3TTS_VOC_NOTAP=1 Q3TTS_VOC_OUT_OLD=1" ""; do
  env $v python bench.py --no-probe --steps 3 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', d['value'], d['frame_step_ms'], d['ms_per_step'])"
done
