cd $GRAFT_REPO_ROOT
for q in 4 8; do echo "GPU_MAX_HW_QUEUES=$q"; GPU_MAX_HW_QUEUES=$q timeout 600 python tools/probe_subbatch.py 1250 2>&1 | grep "B ="; done
bash tools/ab_env.sh GPU_MAX_HW_QUEUES 4 8
export GPU_MAX_HW_QUEUES=8; bash tools/trace_ranges.sh 1250 2>&1 | grep -v "kernel time"
