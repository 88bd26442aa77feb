#!/bin/bash
# durations of the Gram variants' launches in one single-stream step, longest (= fullest) first: bash tools/gram_ns_stats.sh 1 2 4
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out
for ns in "$@"; do
  export HIPDRT_GRAM_NS=$ns
  rocprofv3 --kernel-trace --output-format csv -d $O/prof_gns$ns -- python3 bench.py --inflight 1 --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs --no-matrix-build > $O/prof_gns$ns.log 2>&1
  echo "== HIPDRT_GRAM_NS=$ns"
  python3 -c "
import csv, glob
d = []
for f in glob.glob('$O/prof_gns$ns/*/*kernel_trace.csv'):
    for r in csv.DictReader(open(f)):
        if 'gram_kernel' in r['Kernel_Name']: d.append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
print(len(d), 'launches, total ms %.1f' % (sum(d) / 1e3), '| in launch order, us:', ' '.join('%.0f' % v for v in d[:60]))
"
  rm -rf $O/prof_gns$ns
done
