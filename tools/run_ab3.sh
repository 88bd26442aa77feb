# same-box A/B of the working library against variant libraries + bit comparison + QP tests
cd $GRAFT_REPO_ROOT; O=gpurun_out; T=${1:-ab}; shift; mkdir -p $O
timeout 300 python tools/dump_fit.py $O/dump_new.npz 2>&1 | grep -v "Extension modules" | tail -2 > $O/${T}_dump.txt
HIPDRT_LIB=$PWD/$1 timeout 300 python tools/dump_fit.py $O/dump_old.npz 2>&1 | tail -1 >> $O/${T}_dump.txt
python tools/dump_fit.py --cmp $O/dump_new.npz $O/dump_old.npz >> $O/${T}_dump.txt 2>&1
timeout 900 python -m pytest tests/test_gpu_qp.py tests/test_gpu_fit.py -m gpu -x -q 2>&1 | tail -2 >> $O/${T}_dump.txt
bash tools/ab_libs.sh "$@" > $O/${T}_ab.txt 2>&1
cat $O/${T}_dump.txt $O/${T}_ab.txt
