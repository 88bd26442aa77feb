cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "Extension modules" | tail -15 > $O/r03h_pytest.txt
timeout 600 python bench.py --steps 3 --warmup 1 > $O/r03h_bench.json 2> $O/r03h_bench.err
cat $O/r03h_pytest.txt; tail -1 $O/r03h_bench.json | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value'],1), {k:round(v['seconds'],4) for k,v in d['other_configs'].items()}); a=d['cpu_baseline']['all_cores']; print(a['value'], a['cores'], a['cores_available'], a['efficiency_vs_cores'], [(s['workers'], round(s['value'],1), round(s['cpu_seconds_over_wall'],2)) for s in a['sweep']])"
tail -3 $O/r03h_bench.err
