# kernel traces of one plan fitted as k ranges (k = 2, 3, 4) at B spectra: bash tools/trace_ranges.sh [B]
cd $GRAFT_REPO_ROOT; B=${1:-1250}; O=$GRAFT_REPO_ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
for k in 2 3 4; do
  rm -rf /tmp/tr$k
  rocprofv3 --kernel-trace --output-format csv -d /tmp/tr$k -o t -- python3 $GRAFT_REPO_ROOT/tools/probe_trace_ranges.py run $k $B 2>/dev/null | grep "fits/s"
  f=$(find /tmp/tr$k -name "*kernel_trace.csv" | head -1)
  python3 $GRAFT_REPO_ROOT/tools/probe_trace_ranges.py show $f $k
done
