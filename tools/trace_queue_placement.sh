# which hardware queues the four ranges of one plan land on when `idle` other streams exist in the process, and what it costs:
# bash tools/trace_queue_placement.sh [B]   (GPU_MAX_HW_QUEUES as exported, default of the loader = 8)
cd $GRAFT_REPO_ROOT; B=${1:-1024}; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for idle in 0 3 5 6 9; do
  rm -rf /tmp/tq$idle
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tq$idle -o t -- python3 $GRAFT_REPO_ROOT/tools/probe_trace_ranges.py run 4 $B $idle 2>/dev/null | grep "fits/s"
  f=$(find /tmp/tq$idle -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/probe_trace_ranges.py show $f 4 | head -8
done
