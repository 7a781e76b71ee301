# vocoder A/B of the round: GPU tests, then the vocoder-only leg with and without k_vconv_tap, then one call launch by launch
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/r3_voc_tests.log 2>&1 || { tail -30 gpurun_out/r3_voc_tests.log; exit 1; }
tail -2 gpurun_out/r3_voc_tests.log
python bench.py --probe-only vocoder > gpurun_out/r3_voc_tap.json 2> gpurun_out/r3_voc_tap.err && tail -c 600 gpurun_out/r3_voc_tap.json
Q3TTS_VOC_NOTAP=1 python bench.py --probe-only vocoder > gpurun_out/r3_voc_notap.json 2> gpurun_out/r3_voc_notap.err && tail -c 600 gpurun_out/r3_voc_notap.json
bash tools/voc_trace.sh > gpurun_out/r3_voc_timeline.txt 2>&1
tail -45 gpurun_out/r3_voc_timeline.txt | cut -c1-110
if [ -f tools/exp/libq3tts_vstamps.so ]; then
  Q3TTS_LIB=$GRAFT_REPO_ROOT/tools/exp/libq3tts_vstamps.so python bench.py --probe-only vocoder > gpurun_out/r3_vstamps.json 2> gpurun_out/r3_vstamps.err; grep stamps gpurun_out/r3_vstamps.err
fi
python bench.py > gpurun_out/r3_bench_voc.json 2> gpurun_out/r3_bench_voc.err; tail -c 300 gpurun_out/r3_bench_voc.err
