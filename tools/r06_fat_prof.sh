#!/bin/bash
# round 6: in-kernel phase counters and factorisation time line of the fat kernel next to the eight-wavefront one (PROFILE build)
export TMPDIR=/tmp
export HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_prof.so
for w in ${1:-4 8}; do
  echo "======== HIPDRT_QP_WAVES=$w: phases"
  HIPDRT_QP_WAVES=$w timeout 600 python tools/probe_fit_profile.py 2>&1 | tail -25
  echo "======== HIPDRT_QP_WAVES=$w: time line"
  HIPDRT_QP_WAVES=$w timeout 600 python tools/probe_factor_timeline.py 2>&1 | tail -60
done
