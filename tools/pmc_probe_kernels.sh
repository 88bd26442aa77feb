cd /tmp; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT; O=gpurun_out/pmc_r02h; mkdir -p $O
rocprofv3 -L > $O/counters.txt 2>&1
grep -oE "(TCP|TCC)_[A-Z0-9_]+" $O/counters.txt | sort -u | tr '\n' ' ' | cut -c1-3000
for K in super resident; do
  for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
    T=$(echo $C | tr ' ' '_')
    HIPDRT_QP_KERNEL=$K rocprofv3 --pmc $C --output-format csv -d $O/${K}_$T -- python3 tools/probe_qp.py 256 > $O/${K}_$T.log 2>&1
  done
done
python3 - <<'PY'
import glob,csv,collections,os
O="gpurun_out/pmc_r02h"
for d in sorted(glob.glob(O+"/*_*")):
    if not os.path.isdir(d): continue
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        agg=collections.defaultdict(float); n=collections.defaultdict(int)
        for r in csv.DictReader(open(f)):
            if 'qp_kernel' in r['Kernel_Name']:
                agg[(r['Kernel_Name'][:40], r['Counter_Name'])]+=float(r['Counter_Value']); n[(r['Kernel_Name'][:40], r['Counter_Name'])]+=1
        for k,v in agg.items(): print(os.path.basename(d), k, 'sum', v, 'dispatch-rows', n[k])
PY
rm -rf $O/*/  # keep only the logs
