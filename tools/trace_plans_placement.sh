# queue ids of two plans in flight created BEHIND one plan's ranges (tools/probe_inflight_ranges.py's order): bash tools/trace_plans_placement.sh
cd $GRAFT_REPO_ROOT; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for grid in 1x1,1x2,1x4,1x6,2x1 2x1; do
  rm -rf /tmp/tp
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tp -o t -- python3 $GRAFT_REPO_ROOT/tools/probe_inflight_ranges.py 1250 --grid=$grid 2>/dev/null | grep "fits/s"
  f=$(find /tmp/tp -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/probe_trace_ranges.py tail $f 100
done
