# bench-only A/B of library builds on one box
cd $GRAFT_REPO_ROOT; O=gpurun_out; T=${1:-ab}; shift; mkdir -p $O
bash tools/ab_libs.sh "$@" > $O/${T}_ab.txt 2>&1
cat $O/${T}_ab.txt
