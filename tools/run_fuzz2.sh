cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( echo "randomised differential tests beyond the suite's seeds (tools/fuzz_parity.py 1000 60; --eis 1000 60), round-3 library:"
  timeout 1500 python tools/fuzz_parity.py 1000 60 2>&1 | grep -v "Extension modules" | tail -12
  timeout 900 python tools/fuzz_parity.py --eis 1000 60 2>&1 | grep -v "Extension modules" | tail -8 ) > $O/r03_fuzz_random.txt
cat $O/r03_fuzz_random.txt
