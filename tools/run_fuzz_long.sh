# a long differential run on the round's final library (about ten minutes of box time): bash tools/run_fuzz_long.sh <tag>
cd $GRAFT_REPO_ROOT; O=gpurun_out; T=${1:-r06}
{ echo "== full size, spectra ${F:-2304}..+2047 of the bench's workload (2048 spectra against the CPU checker)"; timeout 2400 python tools/fuzz_parity.py --c2 --first ${F:-2304} --count 2048 2>&1 | tail -14
  echo "== randomised differential tests, seeds ${S:-5000}..+199 (prepared path, option mixes, joint fits)"; timeout 3000 python tools/fuzz_parity.py ${S:-5000} 200 2>&1 | tail -4
  echo "== EIS differential test, seeds ${S:-5000}..+199"; timeout 2400 python tools/fuzz_parity.py ${S:-5000} 200 --eis 2>&1 | tail -3
  echo "== group kernel against the CPU and the batch kernel, 200 launches"; timeout 1800 python tools/fuzz_group_qp.py 200 ${G:-2029} 2>&1 | tail -2; } > $O/${T}_fuzz_long${SUF}.txt 2>&1
cat $O/${T}_fuzz_long${SUF}.txt
