#!/bin/bash
# are whole fits of this tree's library the bits of round 5's (libhipdrt_r5.so = the sources of commit 53fdf90)?
export TMPDIR=/tmp
HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_r5.so timeout 600 python tools/dump_fit.py /tmp/d_r5.npz 2>&1 | tail -1
timeout 600 python tools/dump_fit.py /tmp/d_new.npz 2>&1 | tail -1
python tools/dump_fit.py --cmp /tmp/d_r5.npz /tmp/d_new.npz
HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_r5.so timeout 900 python -m pytest "tests/test_gpu_hybrid.py::test_randomised_joint_fits_follow_the_oracle" -x -q 2>&1 | tail -3
timeout 900 python -m pytest "tests/test_gpu_hybrid.py::test_randomised_joint_fits_follow_the_oracle" -q 2>&1 | tail -8
