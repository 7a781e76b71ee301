# A/B of vocoder variants beside the decoder: prints value / frame step per variant (one process each, same box)
for v in base nofuse noring base2; do
  case $v in
    base|base2) env_v="A=1";;
    nofuse) env_v="Q3TTS_VOC_NOFUSE=1";;
    noring) env_v="Q3TTS_VOC_NORING=1";;
  esac
  env $env_v python bench.py --no-single --no-cpu-baseline --no-probe --steps 2 --warmup 1 > gpurun_out/ab_$v.json 2> gpurun_out/ab_$v.err
  python -c "
import json;d=json.load(open('gpurun_out/ab_$v.json'));print('$v', d['value'], d['frame_step_ms'], d['stage_ms_last_step'])"
done
