cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_qp.py -m gpu -x -q 2>&1 | tail -2
  HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_rr4.so timeout 900 python -m pytest tests/test_gpu_qp.py -m gpu -x -q 2>&1 | tail -2
  for i in 1 2; do for L in "" rr4 rr6; do echo "lib ${L:-base (8)}"; if [ -z "$L" ]; then timeout 600 python tools/probe_single.py -1; else HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_$L.so timeout 600 python tools/probe_single.py -1; fi; done; done ) 2>&1 | grep -v "Extension modules" > $O/r03k.txt
cat $O/r03k.txt
