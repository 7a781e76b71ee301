# Collects the round's profile summaries on the GPU box into gpurun_out/prof_r04/ (copy what is to be judged into profiles/r04/).
#   bash tools/collect_profiles.sh [1|2]     part 1: rocprofv3 kernel stats + PMC passes; part 2: timelines, in-kernel timestamps, the bench line
# Every rocprofv3 call runs the program itself after "--"; counters (--pmc) in their own passes; graph replay is traceable with
# DEBUG_CLR_GRAPH_PACKET_CAPTURE=0 (profiles/README.md).
set -o pipefail
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/prof_r04; mkdir -p $OUT
PART=${1:-1}
cd /tmp && export TMPDIR=/tmp && export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
stats() {  # name, env assignment or "-", bench args...
  name=$1; shift; envv=$1; shift
  rm -rf /tmp/prof_$name
  if [ "$envv" = "-" ]; then rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $R/bench.py "$@" > $OUT/$name.log 2>&1
  else export $envv; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$name -- python3 $R/bench.py "$@" > $OUT/$name.log 2>&1; unset ${envv%%=*}; fi
  f=$(find /tmp/prof_$name -name '*kernel_stats.csv' | head -1)
  [ -n "$f" ] && cp "$f" $OUT/${name}_kernel_stats.csv && echo "$name: $(wc -l < $OUT/${name}_kernel_stats.csv) kernels"
}
if [ "$PART" = "pmc" ]; then
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --probe-only > $OUT/pmc_$c.log 2>&1
    cp "$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)" $OUT/pmc_$c.csv
  done
  python3 $R/tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
  rm -f $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv
  tail -c 300 $OUT/pmc_traffic.log
elif [ "$PART" = "1" ]; then
  stats probe - --probe-only && \
  stats vocoder_only - --probe-only vocoder && \
  stats bench_b64_graph - --steps 1 --warmup 0 --no-single --no-cpu-baseline --no-probe && \
  stats bench_b64_eager Q3TTS_NO_GRAPH=1 --steps 1 --warmup 0 --no-single --no-cpu-baseline --no-probe
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_$c
    rocprofv3 --pmc $c --kernel-trace --output-format csv -d /tmp/pmc_$c -- python3 $R/bench.py --probe-only > $OUT/pmc_$c.log 2>&1
    cp "$(find /tmp/pmc_$c -name '*counter_collection.csv' | head -1)" $OUT/pmc_$c.csv
    echo "pmc $c done"
  done
  python3 $R/tools/pmc_traffic.py $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv $OUT/pmc_traffic.json > $OUT/pmc_traffic.log 2>&1
  rm -f $OUT/pmc_FETCH_SIZE.csv $OUT/pmc_WRITE_SIZE.csv   # (tens of MB each; the summary is what is kept)
  tail -c 400 $OUT/pmc_traffic.log
else
  bash $R/tools/frame_trace.sh > $OUT/frame_step_timeline.txt 2>&1; tail -3 $OUT/frame_step_timeline.txt
  bash $R/tools/voc_trace.sh > $OUT/vocoder_call_timeline.txt 2>&1; tail -2 $OUT/vocoder_call_timeline.txt
  bash $R/tools/prefill_trace.sh > $OUT/prefill_timeline.txt 2>&1; head -3 $OUT/prefill_timeline.txt
  (cd $R && bash tools/build_stamps.sh chain > $OUT/build_stamps.log 2>&1 && bash tools/build_stamps.sh voc >> $OUT/build_stamps.log 2>&1)   # (tools/exp and the tool binaries do not travel: built on the box)
  timeout -k 10 120 $R/tools/chain_stamps 64 > $OUT/chain_stamps.txt 2>&1; tail -3 $OUT/chain_stamps.txt
  if [ -f $R/tools/exp/libq3tts_vstamps.so ]; then  # in-kernel stamps of the vocoder's ring GEMM and residual units
    (cd $R && Q3TTS_LIB=$R/tools/exp/libq3tts_vstamps.so python bench.py --probe-only vocoder 2>&1 >/dev/null | grep stamps > $OUT/vocoder_stamps.txt); tail -2 $OUT/vocoder_stamps.txt
  fi
  bash $R/tools/overlap_trace.sh > $OUT/decoder_vocoder_overlap.txt 2>&1; tail -4 $OUT/decoder_vocoder_overlap.txt
  for b in xcd_bench kernarg_bench; do (cd $R && hipcc --offload-arch=gfx950 -O3 -w -o tools/$b tools/$b.hip) && timeout -k 10 150 $R/tools/$b > $OUT/$b.txt 2>&1; done
  cd $R && python bench.py > $OUT/bench_b64_n1.json 2> $OUT/bench_b64_n1.err; tail -c 600 $OUT/bench_b64_n1.json
fi
