#!/bin/bash
# bash tools/build_variant_from.sh <commit> <name>: hybrid-drt_amd/libhipdrt_<name>.so = today's library with the coneqp kernel
# sources (qp*.hpp, qp.hip) of <commit> -- same C-ABI as the working tree, for same-box A/B runs (tools/ab_libs.sh)

c="$1"; name="$2"; d=/tmp/variant_$name
rm -rf $d; mkdir -p $d/hybrid-drt_amd $d/include
cp -r hybrid-drt_amd/csrc $d/hybrid-drt_amd/; cp include/hipdrt.h include/hipdrt_debug.h $d/include/
rm -f $d/hybrid-drt_amd/csrc/*.o
for f in qp_resident.hpp qp_common.hpp qp.hip qp_super.hpp; do
  git show $c:hybrid-drt_amd/csrc/$f > $d/hybrid-drt_amd/csrc/$f 2>/dev/null || rm -f $d/hybrid-drt_amd/csrc/$f
done
(cd $d/hybrid-drt_amd/csrc && make 2>&1 | grep -E "error" -A3; true)
cp $d/hybrid-drt_amd/libhipdrt.so hybrid-drt_amd/libhipdrt_$name.so
ls -la hybrid-drt_amd/libhipdrt_$name.so
