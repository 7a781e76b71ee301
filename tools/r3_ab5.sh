# same-box A/B with more steps: default vs Q3TTS_VOC_POLITE=1, alternating
cd $GRAFT_REPO_ROOT
run() { env "$@" python bench.py --no-probe --no-single --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', d['value'], d['frame_step_ms'], d['ms_per_step'], d['utterance_latency_rtf']['mean'])"; }
for i in 1 2 3; do run Q3TTS_VOC_POLITE=0; run X=$i; done
