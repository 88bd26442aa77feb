cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( timeout 600 python -m pytest tests/test_gpu_qp.py -m gpu -x -q 2>&1 | tail -3
  HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_tl16.so timeout 300 python tools/probe_timeline.py 1078 16
  for i in 1 2; do timeout 600 python tools/probe_single.py -1; done ) 2>&1 | grep -v "Extension modules" > $O/tl2.txt
cat $O/tl2.txt
