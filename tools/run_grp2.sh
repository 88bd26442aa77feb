cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
P=$PWD/hybrid-drt_amd/libhipdrt_prof.so
( timeout 600 python -m pytest tests/test_gpu_qp.py -m gpu -x -q 2>&1 | tail -4
  HIPDRT_LIB=$P timeout 300 python tools/probe_group.py 514 1 8
  HIPDRT_LIB=$P timeout 300 python tools/probe_group.py 1078 1 16
  timeout 600 python tools/probe_single.py -1 ) 2>&1 | grep -v "Extension modules" > $O/grp2.txt
tail -60 $O/grp2.txt
