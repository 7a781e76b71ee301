# Same-box A/B of engine switches: alternating bench runs, one process each; prints value / frame step per run.
#   bash tools/ab.sh [-n STEPS] [-r REPEATS] [-b "extra bench args"] VARIANT [VARIANT ...]
# A VARIANT is an environment assignment ("Q3TTS_VOC_POLITE=0", several joined by commas: "A=1,B=2") or "-" for the defaults.
cd ${GRAFT_REPO_ROOT:-.}
STEPS=4; REPS=3; EXTRA=""
while getopts "n:r:b:" o; do case $o in n) STEPS=$OPTARG;; r) REPS=$OPTARG;; b) EXTRA=$OPTARG;; esac; done
shift $((OPTIND - 1))
run() {
  local v=$1 envs=""
  [ "$v" != "-" ] && envs=$(echo "$v" | tr ',' ' ')
  env $envs python bench.py --no-probe --no-single --no-cpu-baseline --steps $STEPS --warmup 1 $EXTRA 2>/dev/null | python3 -c "
import sys, json
d = json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('%-40s' % '$v', 'audio-sec/s', d['value'], 'frame_step_ms', d['frame_step_ms'], 'ms_per_step', d['ms_per_step'], 'latency_rtf_mean', d['utterance_latency_rtf']['mean'])"
}
for i in $(seq $REPS); do for v in "$@"; do run "$v"; done; done
