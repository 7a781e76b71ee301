# same-box A/B: the vocoder stream confined to a subset of the CUs (Q3TTS_VOC_CUMASK) beside the decoder
cd $GRAFT_REPO_ROOT
for v in ${MASKS:-"" "Q3TTS_VOC_CUMASK=55555555" "Q3TTS_VOC_CUMASK=77777777" ""}; do
  env $v python bench.py --no-probe --no-single --no-cpu-baseline --steps 2 --warmup 1 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$v', d['value'], d['frame_step_ms'], d['ms_per_step'], d.get('continuous_batching', {}).get('audio_sec_per_s'))"
done
