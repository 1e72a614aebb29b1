cd /tmp && export TMPDIR=/tmp
K=10; W=3
for mode in 0; do
  rm -rf /tmp/prof$mode
  DD_PIPE_TUNE=0 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof$mode -o b -- python $GRAFT_REPO_ROOT/bench.py --child --steps $K --warmup $W --no-cpu-baseline --pmc off --pipeline $mode > /tmp/prof$mode.log 2>&1
  DB=$(ls /tmp/prof$mode/*/*.db /tmp/prof$mode/*.db 2>/dev/null | head -1)
  python $GRAFT_REPO_ROOT/tools/rocprof_summary.py $DB $((K + W)) $W > $GRAFT_REPO_ROOT/gpurun_out/r6j_rocprof_kernel_stats_pipeline$mode.csv
done
