#!/bin/bash
# bench.py at several numbers of batches in flight (same box, back to back); prints "<inflight> <fits/s>"
for f in "$@"; do
  timeout 300 python bench.py --inflight "$f" --steps 8 --no-other-configs --no-cpu-baseline --no-matrix-build 2>/dev/null \
    | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print($f, round(d['value'],1))"
done
