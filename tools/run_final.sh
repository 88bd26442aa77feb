# final measurement set of a round: tools/collect_profiles.sh + the c4 leg on one GPU + the two-rank gloo functional check (self-launched,
# no torchrun) + single fits + SQ counters + the full-size fuzz + the GPU test suite + smoke
cd $GRAFT_REPO_ROOT; O=gpurun_out; T=${1:-r06}
bash tools/collect_profiles.sh $T
timeout 900 python bench.py --config c4 --no-cpu-baseline --no-other-configs > $O/bench_${T}_c4.json 2> $O/bench_${T}_c4.err; tail -1 $O/bench_${T}_c4.json | cut -c1-300
timeout 600 python bench.py --config c4 --total 1250 --no-cpu-baseline --no-other-configs > $O/bench_${T}_c4_share1250.json 2> $O/bench_${T}_c4_share1250.err; tail -1 $O/bench_${T}_c4_share1250.json | cut -c1-200
timeout 900 python bench.py --gpus 2 --backend gloo --config c4 --total 2000 --steps 1 --warmup 0 --no-cpu-baseline --no-other-configs > $O/bench_${T}_gloo2.json 2> $O/bench_${T}_gloo2.err; tail -1 $O/bench_${T}_gloo2.json | cut -c1-400
timeout 600 python bench.py --force-dist --steps 4 --no-cpu-baseline --no-other-configs --no-matrix-build > $O/bench_${T}_force_dist.json 2> $O/bench_${T}_force_dist.err; tail -1 $O/bench_${T}_force_dist.json | cut -c1-300
timeout 600 python tools/probe_single.py 0 -1 2>&1 | grep -v "Extension modules" > $O/${T}_single.txt; cat $O/${T}_single.txt
timeout 900 python tools/probe_subbatch.py 1024 1250 2500 > $O/${T}_subbatch_sweep.txt 2>&1; cat $O/${T}_subbatch_sweep.txt
bash tools/pmc_sq_collect.sh $T > /dev/null 2>&1; tail -25 $O/${T}_pmc_sq_summary.txt
timeout 120 ./tools/cholinv16_bench.bin > $O/${T}_cholinv16_bench.txt 2>&1; cat $O/${T}_cholinv16_bench.txt
timeout 900 python tools/fuzz_parity.py --c2 --count 256 > $O/${T}_fuzz_c2.txt 2>&1; tail -6 $O/${T}_fuzz_c2.txt
timeout 2400 python -m pytest tests -q -m gpu 2>&1 | tail -8 > $O/${T}_pytest_gpu.txt; cat $O/${T}_pytest_gpu.txt; cp $O/_parity_measured.txt $O/${T}_parity_measured.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
