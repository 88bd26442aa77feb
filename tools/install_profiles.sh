#!/bin/bash
# Move a freshly collected profile set from gpurun_out/ into profiles/ under tag $1, retiring the set tagged $2:
# (new and old tags must differ: the old set is removed)
#   bash tools/install_profiles.sh r02i r02h     (run in the repository root, after tools/collect_profiles.sh and pmc_sq_collect.sh)
new="$1"; old="$2"
for f in kernel_stats.csv kernel_stats_inflight1.csv pmc_fetch_size.csv pmc_write_size.csv pmc_sq_summary.txt; do
  cp gpurun_out/${new}_$f profiles/ && git rm -q --cached profiles/${old}_$f 2>/dev/null; rm -f profiles/${old}_$f
done
cp gpurun_out/qp_traffic.json profiles/qp_traffic.json
tail -1 gpurun_out/bench_${new}.json > profiles/${new}_bench.json
git rm -q --cached profiles/${old}_bench.json 2>/dev/null; rm -f profiles/${old}_bench.json
git mv profiles/${old}_bench_c4_1gpu.json profiles/${new}_bench_c4_1gpu.json 2>/dev/null
git mv profiles/${old}_kernel_resources.txt profiles/${new}_kernel_resources.txt 2>/dev/null
mkdir -p /tmp/st_res
( cd hybrid-drt_amd/csrc && for f in api gram hyper matrices qp; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include -c $f.hip -o /tmp/st_res/$f.o --save-temps=obj 2>/dev/null; done )
python tools/kernel_resources.py /tmp/st_res > profiles/${new}_kernel_resources.txt
sed -i "s/${old}/${new}/g" DESIGN.md README.md
python - "$new" <<'PY'
import json, sys
d = json.load(open(f"profiles/{sys.argv[1]}_bench.json"))
r = d["roofline"]
print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), {k: round(v, 1) for k, v in d["phase_ms_per_step"].items()})
print("frac", round(r["frac"], 4), "TFLOP/s", round(r["achieved"], 2), "ms/launch", round(r["avg_launch_ms"], 3), "traffic GB", round((r["traffic"] or 0) / 1e9, 2), "GB/s", r["traffic_GBps"])
print("single", round(d["single_stream"]["value"], 1), "with transfers", round(d["with_transfers"]["value"], 1), "matrix build", round(d["matrix_build_roofline"]["frac"], 3))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"]); print(d["other_configs"])
import bench
print("stamp", bench.source_hash(), json.load(open("profiles/qp_traffic.json"))["source_hash"])
PY
