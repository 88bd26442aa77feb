cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_hybrid.py -m gpu -x -q -s -k "config5 or hyper_step or batch_members" 2>&1 | grep -v "Extension modules" | tail -12 > $O/r03g_c5.txt
timeout 600 python tools/probe_single.py -1 2>&1 | grep -v "Extension modules" > $O/r03g_single.txt
cat $O/r03g_c5.txt $O/r03g_single.txt
