# same-box A/B: the vocoder call replayed as a hipGraph (default) vs launched kernel by kernel (Q3TTS_VOC_NO_GRAPH=1)
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q > gpurun_out/r3_ab6_tests.log 2>&1; tail -2 gpurun_out/r3_ab6_tests.log
run() { env "$@" python bench.py --no-probe --no-single --no-cpu-baseline --steps 4 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', d['value'], d['frame_step_ms'], d['ms_per_step'], d['stage_ms_last_step'])"; }
for i in 1 2 3; do run Q3TTS_VOC_NO_GRAPH=1; run X=$i; done
