cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( timeout 300 python -m pytest tests/test_gpu_fit.py -m gpu -x -q -k "kernel_choice" 2>&1 | tail -2
  for i in 1 2; do
  timeout 600 python tools/probe_single.py -1
  HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_ovr2.so timeout 600 python tools/probe_single.py -1
  done ) 2>&1 | grep -v "Extension modules" > $O/r03i_ovr.txt
cat $O/r03i_ovr.txt
