# the two PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs) of tools/collect_profiles.sh on their own: bash tools/pmc_traffic_only.sh <tag>
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out; T=${1:-r04}
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_${T}_fetch -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-matrix-build --no-other-configs --no-scale-reference --no-single-caller > $O/pmc_${T}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_${T}_write -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-matrix-build --no-other-configs --no-scale-reference --no-single-caller > $O/pmc_${T}_write.log 2>&1
python tools/pmc_traffic.py $O/pmc_${T}_fetch/*/*counter_collection.csv $O/pmc_${T}_write/*/*counter_collection.csv | grep -E "hbm_bytes|fetch_bytes_per_launch_x2|write_bytes"; cp profiles/qp_traffic.json $O/qp_traffic_${T}.json
cp $O/pmc_${T}_fetch/*/*counter_collection.csv $O/${T}_pmc_fetch_size.csv; cp $O/pmc_${T}_write/*/*counter_collection.csv $O/${T}_pmc_write_size.csv
rm -rf $O/pmc_${T}_fetch $O/pmc_${T}_write
