#!/bin/bash
# same-box A/B of the 64-column factorisation against the 32-column one (libhipdrt_p32.so = make VARIANT=p32 EXTRA=-DHIPDRT_QP_PANEL64=0):
# bit-for-bit comparison of whole fits, then fits/s and QP launch time of both
mkdir -p gpurun_out
HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_p32.so timeout 600 python tools/dump_fit.py /tmp/dump_p32.npz 2>&1 | tail -2
timeout 600 python tools/dump_fit.py /tmp/dump_p64.npz 2>&1 | tail -2
python tools/dump_fit.py --cmp /tmp/dump_p32.npz /tmp/dump_p64.npz
bash tools/ab_lib.sh hybrid-drt_amd/libhipdrt_p32.so ${1:-2}
