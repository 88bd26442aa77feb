# beyond the suite's seeds, on the round's final library: bash tools/run_fuzz_extra.sh <tag>
cd $GRAFT_REPO_ROOT; O=gpurun_out; T=${1:-r06}; S=${2:-4000}
{ echo "== randomised differential tests, seeds $S..+59 (prepared path, option mixes, joint fits)"; timeout 1500 python tools/fuzz_parity.py $S 60 2>&1 | tail -12
  echo "== EIS differential test, seeds $S..+39"; timeout 900 python tools/fuzz_parity.py $S 40 --eis 2>&1 | tail -6
  echo "== full size, spectra 2048..2303 of the bench's batch"; timeout 900 python tools/fuzz_parity.py --c2 --first 2048 --count 256 2>&1 | tail -14; } > $O/${T}_fuzz_random.txt 2>&1
timeout 900 python tools/fuzz_group_qp.py 40 2028 > $O/${T}_fuzz_group_qp.txt 2>&1
tail -25 $O/${T}_fuzz_random.txt; tail -6 $O/${T}_fuzz_group_qp.txt
