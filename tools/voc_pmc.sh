# SQ counters of the vocoder's kernels (one pass per counter group): bash tools/voc_pmc.sh
cd /tmp && export TMPDIR=/tmp && export DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_LDS" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  rm -rf /tmp/voc_pmc
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d /tmp/voc_pmc -- python3 $GRAFT_REPO_ROOT/bench.py --probe-only vocoder > /tmp/voc_pmc.log 2>&1 || { tail -5 /tmp/voc_pmc.log; continue; }
  f=$(find /tmp/voc_pmc -name '*counter_collection.csv' | head -1)
  echo "== $grp"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py "$f" k_v | head -12
done
