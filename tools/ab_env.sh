#!/bin/bash
# A/B of one environment variable on one box: bash tools/ab_env.sh VAR v1 v2 ...; fits/s, QP ms per launch, phases of each (twice)
var="$1"; shift
run() { env "$var=$1" timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-matrix-build --no-scale-reference 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$var=$1', round(d['value'],1), round(d['roofline']['avg_launch_ms'],3), {k: round(v,1) for k,v in d['phase_ms_per_step'].items()})"; }
for i in 1 2; do for v in "$@"; do run "$v"; done; done
