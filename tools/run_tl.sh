cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
( echo "== n = 514, 8 members, block column 8 of 17 (k cycles relative to the member's own wavefront-1 column top)"
  HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_tl.so timeout 300 python tools/probe_timeline.py 514 8
  echo "== n = 1078, 16 members, block column 16 of 34"
  HIPDRT_LIB=$PWD/hybrid-drt_amd/libhipdrt_tl16.so timeout 300 python tools/probe_timeline.py 1078 16 ) 2>&1 | grep -v "Extension modules" > $O/r03_group_timeline.txt
cat $O/r03_group_timeline.txt
