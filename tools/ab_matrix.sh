#!/bin/bash
# matrix-build A/B of library builds on one box: bash tools/ab_matrix.sh [other.so ...]; prints GB/s of the batched Z'/Z'' build
run() { timeout 300 python bench.py --no-other-configs --no-cpu-baseline --steps 1 --warmup 0 --inflight 1 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1])['matrix_build_roofline']; print('$1', round(d['achieved'],1), 'GB/s', round(d['avg_launch_ms'],4), 'ms')"; }
for i in 1 2; do
  run base
  for alt in "$@"; do HIPDRT_LIB="$PWD/$alt" run "$alt"; done
done
