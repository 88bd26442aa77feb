#!/bin/bash
# round 6, step A of the fat kernel: bit-identity of whole fits (round-5 library | this tree, 8 wavefronts | this tree, fat) and timing
mkdir -p gpurun_out
export TMPDIR=/tmp
HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_r5.so timeout 600 python tools/dump_fit.py /tmp/d_r5.npz 2>&1 | tail -1
timeout 600 python tools/dump_fit.py /tmp/d_n8.npz 2>&1 | tail -1
HIPDRT_QP_WAVES=4 timeout 600 python tools/dump_fit.py /tmp/d_n4.npz 2>&1 | tail -3
echo "r5 vs new8:"; python tools/dump_fit.py --cmp /tmp/d_r5.npz /tmp/d_n8.npz
echo "new8 vs fat:"; python tools/dump_fit.py --cmp /tmp/d_n8.npz /tmp/d_n4.npz
run() { timeout 400 python bench.py --no-other-configs --no-cpu-baseline --no-matrix-build --no-scale-reference 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],1), round(d['roofline']['avg_launch_ms'],3), {k: round(v,1) for k,v in d['phase_ms_per_step'].items()})"; }
for i in 1 2; do
  HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_r5.so run r5
  run new8
  HIPDRT_QP_WAVES=4 run fat
done
