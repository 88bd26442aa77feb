#!/bin/bash
# Copy the set tools/run_final.sh <tag> left in gpurun_out/ into profiles/ (tracked) and print the figures the documents quote:
#   bash tools/install_final.sh r04          (in the repository root; nothing is removed, no document is edited)
T="$1"
for f in kernel_stats.csv kernel_stats_inflight1.csv pmc_fetch_size.csv pmc_write_size.csv pmc_sq_summary.txt fuzz_c2.txt pytest_gpu.txt \
         parity_measured.txt subbatch_sweep.txt cholinv16_bench.txt; do cp gpurun_out/${T}_$f profiles/; done
cp gpurun_out/${T}_single.txt profiles/${T}_single_fits.txt
cp gpurun_out/qp_traffic.json profiles/qp_traffic.json
tail -1 gpurun_out/bench_${T}.json > profiles/${T}_bench.json
tail -1 gpurun_out/bench_${T}_c4.json > profiles/${T}_bench_c4_1gpu.json
tail -1 gpurun_out/bench_${T}_gloo2.json > profiles/${T}_bench_gloo2.json
tail -1 gpurun_out/bench_${T}_c4_share1250.json > profiles/${T}_bench_c4_share1250.json
grep '"metric"' gpurun_out/bench_${T}_force_dist.json | tail -1 > profiles/${T}_bench_force_dist.json
mkdir -p /tmp/st_res
( cd hybrid-drt_amd/csrc && for f in api gram hyper matrices qp; do /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -I../../include -c $f.hip -o /tmp/st_res/$f.o --save-temps=obj 2>/dev/null; done )
python tools/kernel_resources.py /tmp/st_res > profiles/${T}_kernel_resources.txt
python - "$T" <<'PY'
import csv, json, sys
T = sys.argv[1]
d = json.load(open(f"profiles/{T}_bench.json"))
r = d["roofline"]
print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 1), {k: round(v, 1) for k, v in d["phase_ms_per_step"].items()})
print("frac", round(r["frac"], 4), "TFLOP/s", round(r["achieved"], 2), "ms/launch", round(r["avg_launch_ms"], 3), "traffic GB", round((r["traffic"] or 0) / 1e9, 2), "GB/s", r["traffic_GBps"])
h = d["roofline_hbm"]
print("hbm frac", round(h["frac"], 4), "traffic/algorithmic", h["traffic_over_algorithmic"], "traffic frac of peak", h["traffic_frac_of_hbm_peak"])
print("single", round(d["single_stream"]["value"], 1), "single_caller", round(d["single_caller"]["value"], 1), "with transfers", round(d["with_transfers"]["value"], 1),
      "matrix build", round(d["matrix_build_roofline"]["frac"], 3), "gram frac", round(d["roofline_gram"]["frac"], 3), "hyper ms", round(d["roofline_hyper"]["ms_per_step"], 1))
print("cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["all_cores"]["value"])
print({k: round(v["seconds"], 4) for k, v in d["other_configs"].items()})
for f in (f"{T}_bench_c4_1gpu", f"{T}_bench_c4_share1250", f"{T}_bench_gloo2"):
    e = json.load(open(f"profiles/{f}.json")); print(f, round(e["value"], 1), round(e["ms_per_step"], 1), e["n_gpus"])
for r_ in csv.DictReader(open(f"profiles/{T}_kernel_stats_inflight1.csv")):
    if "qp_kernel_resident" in r_["Name"]: print("rocprof one-plan run: qp_kernel_resident calls", r_["Calls"], "avg ms %.3f" % (float(r_["AverageNs"]) / 1e6))
import bench
print("stamp", bench.source_hash(), json.load(open("profiles/qp_traffic.json"))["source_hash"])
PY
tail -2 profiles/${T}_pytest_gpu.txt; tail -1 profiles/${T}_fuzz_c2.txt; cat profiles/${T}_single_fits.txt; cat profiles/${T}_subbatch_sweep.txt
