P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(sys.argv[1], round(d["value_resident"],1), round(d["value_streamed"],1), round(d["single_stream"]["value"],1), d["c3"]["steps"] if d.get("c3") else d["steps"])'
F="--no-cpu-baseline --no-other-configs --no-matrix-build --no-scale-reference --no-single-caller"
python bench.py $F --steps 8 2>/dev/null | python -c "$P" auto8
python bench.py $F --steps 4 2>/dev/null | python -c "$P" auto4
python bench.py $F --config c3 --steps 8 2>/dev/null | python -c "$P" c3_8
python bench.py $F --config c3 --steps 12 2>/dev/null | python -c "$P" c3_12
GPU_MAX_HW_QUEUES=8 python bench.py $F --steps 8 2>/dev/null | python -c "$P" auto8_q8
python -c "
import torch, runpy, sys
torch.cuda.set_device(0); torch.cuda.synchronize()
sys.argv=['bench.py']+'$F --steps 8'.split()
runpy.run_path('bench.py', run_name='__main__')" 2>/dev/null | python -c "$P" auto8_torch
