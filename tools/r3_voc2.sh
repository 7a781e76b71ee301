cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "vocoder or full_shape" > gpurun_out/r3_voc_tests.log 2>&1 || { tail -30 gpurun_out/r3_voc_tests.log; exit 1; }
tail -2 gpurun_out/r3_voc_tests.log
python bench.py --probe-only vocoder 2>/dev/null | grep -o '"ms_per_call": [0-9.]*'
Q3TTS_VOC_RES192_64=1 python bench.py --probe-only vocoder 2>/dev/null | grep -o '"ms_per_call": [0-9.]*'
bash tools/voc_trace.sh > gpurun_out/r3_voc_timeline.txt 2>&1
grep -v k_voc_zero gpurun_out/r3_voc_timeline.txt | tail -32 | cut -c1-110
