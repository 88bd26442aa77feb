cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
P=$PWD/hybrid-drt_amd/libhipdrt_prof.so
( HIPDRT_LIB=$P timeout 300 python tools/probe_group.py 514 1 8
  HIPDRT_LIB=$P timeout 300 python tools/probe_group.py 1078 1 16 ) 2>&1 | grep -v "Extension modules" > $O/grp2.txt
tail -60 $O/grp2.txt
