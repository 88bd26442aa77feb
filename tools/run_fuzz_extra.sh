# beyond the suite's seeds, on the round's final library: bash tools/run_fuzz_extra.sh <tag>
cd $GRAFT_REPO_ROOT; O=gpurun_out; T=${1:-r05}
{ echo "== randomised differential tests, seeds 3000..3059 (prepared path, option mixes, joint fits)"; timeout 1500 python tools/fuzz_parity.py 3000 60 2>&1 | tail -12
  echo "== EIS differential test, seeds 3000..3039"; timeout 900 python tools/fuzz_parity.py 3000 40 --eis 2>&1 | tail -6
  echo "== full size, spectra 1024..1279 of the bench's batch"; timeout 900 python tools/fuzz_parity.py --c2 --first 1024 --count 256 2>&1 | tail -14; } > $O/${T}_fuzz_random.txt 2>&1
timeout 900 python tools/fuzz_group_qp.py 40 2027 > $O/${T}_fuzz_group_qp.txt 2>&1
tail -25 $O/${T}_fuzz_random.txt; tail -6 $O/${T}_fuzz_group_qp.txt
