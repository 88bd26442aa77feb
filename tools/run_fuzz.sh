cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
tools/cholinv16_bench.bin > $O/r03_cholinv16_bench.txt 2>&1; cat $O/r03_cholinv16_bench.txt
echo "host cpus: $(nproc)" > $O/r03_fuzz_c2.txt
timeout 1700 python tools/fuzz_parity.py --c2 --count 256 2>&1 | grep -v "Extension modules" >> $O/r03_fuzz_c2.txt
cat $O/r03_fuzz_c2.txt
