# Gram phase (ms per 1024-spectrum step, one plan, one range) of the working library against an older build: bash tools/r06_gram_phase_ab.sh old.so
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in working "$1"; do
    if [ "$lib" = working ]; then unset HIPDRT_LIB; else export HIPDRT_LIB="$PWD/$lib"; fi
    timeout 300 python bench.py --config c3 --inflight 1 --steps 6 --warmup 1 --no-cpu-baseline --no-other-configs --no-matrix-build --no-single-caller --no-scale-reference 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$lib', 'gram %.2f ms  qp %.2f  hyper %.2f  fits/s %.1f  gram frac %.4f' % (d['phase_ms_per_step']['gram'], d['phase_ms_per_step']['qp'], d['phase_ms_per_step']['hyper'], d['single_stream']['value'], d['roofline_gram']['frac']))"
  done
done
