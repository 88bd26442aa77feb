cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "Extension modules" | tail -15 > $O/r03d_pytest.txt
( timeout 600 python tools/probe_single.py 0 -1 ) 2>&1 | grep -v "Extension modules" > $O/r03d_single.txt
timeout 300 python bench.py --steps 3 --warmup 1 > $O/r03d_bench.json 2> $O/r03d_bench.err
cat $O/r03d_pytest.txt $O/r03d_single.txt $O/r03d_bench.json; tail -3 $O/r03d_bench.err
