#!/bin/bash
# round 6: the fat four-wavefront coneqp kernel against the eight-wavefront one on the same box --
# whole fits bit for bit (tools/dump_fit.py), then fits/s and QP launch time of both (bench.py, 2 repeats)
mkdir -p gpurun_out
export TMPDIR=/tmp
HIPDRT_QP_WAVES=8 timeout 600 python tools/dump_fit.py /tmp/d_n8.npz 2>&1 | tail -1
HIPDRT_QP_WAVES=4 timeout 600 python tools/dump_fit.py /tmp/d_n4.npz 2>&1 | tail -3
echo "eight wavefronts vs fat:"; python tools/dump_fit.py --cmp /tmp/d_n8.npz /tmp/d_n4.npz
run() { timeout 400 python bench.py --no-other-configs --no-cpu-baseline --no-matrix-build --no-scale-reference 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_ms'],3), {k: round(v,1) for k,v in d['phase_ms_per_step'].items()})"; }
for i in $(seq ${1:-2}); do
  HIPDRT_QP_WAVES=8 run w8
  HIPDRT_QP_WAVES=4 run fat
done
