# per-kernel times (rocprofv3 --kernel-trace --stats, one batch in flight) of the working library and of variant libraries:
#   bash tools/gram_ab.sh <tag> [variant.so ...]  -> gpurun_out/<tag>_kernels.txt
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out; T=${1:-gram_ab}; shift; mkdir -p $O; : > $O/${T}_kernels.txt
one() {
  rm -rf $O/prof_$T
  timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 bench.py --inflight 1 --steps ${STEPS:-2} --warmup ${WARM:-1} --no-cpu-baseline --no-other-configs --no-matrix-build > $O/prof_${T}.log 2>&1
  echo "== $1" >> $O/${T}_kernels.txt
  python3 - $O/prof_$T/*/*kernel_stats.csv >> $O/${T}_kernels.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("gram_kernel", "gram_kr", "qvec_kernel", "qp_kernel", "hyper_kernel", "lpt_order")):
        print("%-44s calls %4s  total %9.3f ms  avg %8.1f us  min %8.1f  max %8.1f" % (r["Name"].split("(")[0][-44:], r["Calls"],
              float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
  rm -rf $O/prof_$T
}
one working
for alt in "$@"; do export HIPDRT_LIB="$PWD/$alt"; one "$alt"; done
cat $O/${T}_kernels.txt
