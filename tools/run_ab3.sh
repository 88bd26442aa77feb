# same-box A/B of the working library against variant libraries + GPU tests + the resolve probe
cd $GRAFT_REPO_ROOT; O=gpurun_out; T=${1:-ab}; shift; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | grep -v "Extension modules" | tail -8 > $O/${T}_pytest.txt
bash tools/ab_libs.sh "$@" > $O/${T}_ab.txt 2>&1
timeout 600 python tools/probe_single.py 0 -1 2>&1 | grep -v "Extension modules" > $O/${T}_single.txt
timeout 900 python tools/probe_resolve_c2grid.py 2>&1 | grep -v "Extension modules" > $O/${T}_resolve.txt
cat $O/${T}_pytest.txt $O/${T}_ab.txt $O/${T}_single.txt $O/${T}_resolve.txt
