#!/bin/bash
# A/B/C.. of library builds on one box: bash tools/ab_libs.sh a.so b.so ...; fits/s, QP ms per launch, phases of each (twice)
run() { timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-matrix-build 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_ms'],3), {k: round(v,1) for k,v in d['phase_ms_per_step'].items()})"; }
for i in 1 2; do
  run base
  for alt in "$@"; do HIPDRT_LIB="$PWD/$alt" run "$alt"; done
done
