# same-box A/B: "polite" vocoder (one 4-wave workgroup per CU, >= 81 KiB LDS) beside the default
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "bgemm or full_shape or 64_slots" > gpurun_out/r3_ab4_tests.log 2>&1; tail -2 gpurun_out/r3_ab4_tests.log
run() { env "$@" python bench.py --no-probe --no-single --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$*', d['value'], d['frame_step_ms'], d['ms_per_step'])"; }
run X=0
run Q3TTS_VOC_POLITE=1
run X=1
run Q3TTS_VOC_POLITE=1
python bench.py --probe-only talker 2>/dev/null | tail -c 300
