#!/bin/bash
# A/B of two builds of the library on the same box: bash tools/ab_lib.sh <other.so> [repeats]; prints fits/s and the
# average QP launch time of each
alt="$1"; rep="${2:-2}"
run() { timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-matrix-build --no-scale-reference 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_ms'],3), {k: round(v,1) for k,v in d['phase_ms_per_step'].items()})"; }
for i in $(seq "$rep"); do
  run base
  HIPDRT_LIB="$PWD/$alt" run alt
done
