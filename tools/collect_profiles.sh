cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out; T=${1:-r02b}
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${T}_fetch -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-matrix-build --no-other-configs --no-scale-reference --no-single-caller > $O/pmc_${T}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${T}_write -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-matrix-build --no-other-configs --no-scale-reference --no-single-caller > $O/pmc_${T}_write.log 2>&1
python tools/pmc_traffic.py $O/pmc_${T}_fetch/*/*counter_collection.csv $O/pmc_${T}_write/*/*counter_collection.csv | grep hbm_bytes; cp profiles/qp_traffic.json $O/qp_traffic.json; cp profiles/qp_traffic.json $O/qp_traffic_${T}.json
cp $O/pmc_${T}_fetch/*/*counter_collection.csv $O/${T}_pmc_fetch_size.csv; cp $O/pmc_${T}_write/*/*counter_collection.csv $O/${T}_pmc_write_size.csv
python bench.py > $O/bench_${T}.json 2> $O/bench_${T}.err; tail -1 $O/bench_${T}.json | cut -c1-200
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T} -- python3 bench.py --no-cpu-baseline --no-other-configs --no-scale-reference > $O/prof_${T}_bench.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_${T}_if1 -- python3 bench.py --inflight 1 --steps 3 --no-cpu-baseline --no-other-configs --no-single-caller --no-scale-reference > $O/prof_${T}_if1_bench.log 2>&1
cp $O/prof_${T}/*/*kernel_stats.csv $O/${T}_kernel_stats.csv; cp $O/prof_${T}_if1/*/*kernel_stats.csv $O/${T}_kernel_stats_inflight1.csv
head -4 $O/${T}_kernel_stats_inflight1.csv | cut -c1-200
rm -rf $O/pmc_${T}_fetch $O/pmc_${T}_write $O/prof_${T} $O/prof_${T}_if1
