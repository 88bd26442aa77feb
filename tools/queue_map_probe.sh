# stream index -> hardware queue id, for 4 and 8 hardware queues: bash tools/queue_map_probe.sh
cd $GRAFT_REPO_ROOT; R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for q in 4 8; do
  export GPU_MAX_HW_QUEUES=$q
  rm -rf /tmp/qm; rocprofv3 --kernel-trace --output-format csv -d /tmp/qm -o t -- $R/tools/queue_map_probe.bin 20 5 > /dev/null 2>&1
  f=$(find /tmp/qm -name "*kernel_trace.csv" | head -1)
  python3 - $f $q <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "marker" in r["Kernel_Name"]]
m = {}
for r in rows:
    g = int(r.get("Grid_Size", r.get("Grid_Size_X", 0))) // 64
    m[g] = r["Queue_Id"]
print(f"GPU_MAX_HW_QUEUES={sys.argv[2]}: stream i -> queue:", " ".join(f"{g - 1}:{m[g]}" for g in sorted(m) if g <= 99))
print("   re-created after destroying streams 0..4:", " ".join(f"{g - 100}:{m[g]}" for g in sorted(m) if 100 <= g < 200), " null stream:", m.get(200))
PY
done
