#!/bin/bash
# A/B of one library under two environments on the same box: bash tools/ab_env.sh "VAR=value ..." [repeats]
# (e.g. HIPDRT_QP_KERNEL=resident for the 32-column coneqp kernel); prints fits/s, the average QP launch time and the
# phase split of each
alt="$1"; rep="${2:-2}"
run() { timeout 300 python bench.py --no-other-configs --no-cpu-baseline --no-matrix-build 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_ms'],3), {k: round(v,1) for k,v in d['phase_ms_per_step'].items()}, 'single', round(d['single_stream']['value'],1))"; }
for i in $(seq "$rep"); do
  run base
  env $alt bash -c "$(declare -f run); run alt"
done
