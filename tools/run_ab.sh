# same-box A/B: bit comparison against the round-2 library, tests, bench of the working library against the given ones,
# phase probe of the PROFILE build (if present)
cd $GRAFT_REPO_ROOT; O=gpurun_out; T=${1:-ab}; shift; mkdir -p $O
timeout 300 python tools/dump_fit.py $O/dump_new.npz 2>&1 | grep -v "Extension modules" | tail -5 > $O/${T}_dump.txt
HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_r02.so timeout 300 python tools/dump_fit.py $O/dump_r02.npz 2>&1 | tail -2 >> $O/${T}_dump.txt
python tools/dump_fit.py --cmp $O/dump_new.npz $O/dump_r02.npz >> $O/${T}_dump.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_qp.py tests/test_gpu_hybrid.py tests/test_gpu_fit.py -m gpu -x -q 2>&1 | tail -3 >> $O/${T}_dump.txt
bash tools/ab_libs.sh "$@" > $O/${T}_ab.txt 2>&1
if [ -f hybrid-drt_amd/libhipdrt_prof.so ]; then HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_prof.so timeout 300 python tools/probe_qp.py 256 2>&1 | tail -12 > $O/${T}_probe.txt; fi
cat $O/${T}_dump.txt $O/${T}_ab.txt $O/${T}_probe.txt
