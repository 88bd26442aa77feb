cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hybrid.py -m gpu -x -q -s -k "config5" 2>&1 | grep -v "Extension modules" | tail -8 > $O/r03f_c5.txt
for nfl in 1 2 4 6 8; do
  timeout 400 python bench.py --steps 8 --warmup 2 --inflight $nfl --no-other-configs --no-cpu-baseline --no-matrix-build 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('inflight $nfl', round(d['value'],1), 'fits/s', round(d['roofline']['avg_launch_ms'],3), 'ms/launch')"
done > $O/r03f_inflight.txt 2>&1
cat $O/r03f_c5.txt $O/r03f_inflight.txt
