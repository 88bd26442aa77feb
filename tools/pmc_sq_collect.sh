# SQ activity counters of one single-stream bench step (separate --pmc passes, kernel trace only):
#   bash tools/pmc_sq_collect.sh <tag>   ->  gpurun_out/<tag>_pmc_sq_summary.txt
cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out; T=${1:-r02f}
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_ANY" \
         "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS" \
         "SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS" \
         "SQ_WAVE_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_INSTS_MFMA SQ_INSTS_VALU" \
         "SQ_WAVE_CYCLES SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
         "SQ_WAVE_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $C --output-format csv -d $O/sq_${T}_$i -- python3 bench.py --steps 1 --warmup 0 --inflight 1 --no-cpu-baseline --no-matrix-build --no-other-configs --no-scale-reference > $O/sq_${T}_$i.log 2>&1
done
python3 - "$O" "$T" <<'PY' > $O/${T}_pmc_sq_summary.txt
import csv, glob, sys, collections
O, T = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); disp = collections.Counter()
for f in sorted(glob.glob(f"{O}/sq_{T}_*/**/*counter_collection.csv", recursive=True)):
    seen = set()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0]
        # SQ_WAVE_CYCLES is in every pass: keep the first pass's
        if r["Counter_Name"] == "SQ_WAVE_CYCLES" and "sq_%s_1/" % T not in f: continue
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": disp[k] += 1
print("rocprofv3 --pmc, one pass per group, bench.py --steps 1 --warmup 0 --inflight 1 (1024 spectra, single stream); sums over all dispatches")
for k, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = d.get("SQ_WAVE_CYCLES", 0)
    if wc < 1e8: continue
    print(k, "dispatches", disp[k])
    for c, v in sorted(d.items()):
        print("   %-30s %.4g  (%.3f of SQ_WAVE_CYCLES)" % (c, v, v / wc))
PY
cat $O/${T}_pmc_sq_summary.txt | head -60
rm -rf $O/sq_${T}_*/
