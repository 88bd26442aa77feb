cd $GRAFT_REPO_ROOT
bash tools/ab_env.sh HIPDRT_QP_PRIORITY 0 1 > gpurun_out/r05q_ab_qp_priority.txt 2>&1
for v in 0 1; do echo "HIPDRT_QP_PRIORITY=$v" >> gpurun_out/r05q_ab_qp_priority.txt; HIPDRT_QP_PRIORITY=$v timeout 600 python tools/probe_subbatch.py 1024 1250 >> gpurun_out/r05q_ab_qp_priority.txt 2>&1; done
for v in 0 1; do for k in 2 3; do HIPDRT_QP_PRIORITY=$v timeout 300 python bench.py --config c4 --total 1250 --inflight $k --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-scale-reference 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('share 1250 prio $v inflight $k', round(d['value'],1))" >> gpurun_out/r05q_ab_qp_priority.txt; done; done
cat gpurun_out/r05q_ab_qp_priority.txt
