cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
python -m pytest tests -m gpu -x -q 2>&1 | tail -5 > $O/r03a_pytest.txt
python tools/dump_fit.py $O/dump_new.npz > $O/r03a_dump.txt 2>&1
HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_r02.so python tools/dump_fit.py $O/dump_r02.npz >> $O/r03a_dump.txt 2>&1
HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_nopre.so python tools/dump_fit.py $O/dump_nopre.npz >> $O/r03a_dump.txt 2>&1
python tools/dump_fit.py --cmp $O/dump_new.npz $O/dump_r02.npz >> $O/r03a_dump.txt 2>&1
python tools/dump_fit.py --cmp $O/dump_new.npz $O/dump_nopre.npz >> $O/r03a_dump.txt 2>&1
bash tools/ab_libs.sh hybrid-drt_amd/libhipdrt_r02.so hybrid-drt_amd/libhipdrt_nopre.so > $O/r03a_ab.txt 2>&1
cat $O/r03a_pytest.txt $O/r03a_dump.txt $O/r03a_ab.txt
