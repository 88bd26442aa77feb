cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( timeout 900 python -m pytest tests/test_gpu_qp.py tests/test_resolve.py -m gpu -x -q 2>&1 | tail -2
  for i in 1 2; do timeout 600 python tools/probe_single.py -1; done ) 2>&1 | grep -v "Extension modules" > $O/r03k.txt
cat $O/r03k.txt
