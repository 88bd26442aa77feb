"""Register / scratch / LDS figures of every kernel from the code-object metadata hipcc writes with --save-temps
(`.vgpr_count`, `.agpr_count`, `.sgpr_count`, `.vgpr_spill_count`, `.private_segment_fixed_size`,
`.group_segment_fixed_size`): python tools/kernel_resources.py <dir with *-gfx950.s> > profiles/<tag>_kernel_resources.txt
rocprofv3's kernel trace reports VGPR_Count rounded to the allocation granule and the arch/acc split differently; these are
the compiler's own numbers."""
import glob
import re
import subprocess
import sys

keys = ("vgpr_count", "agpr_count", "sgpr_count", "vgpr_spill_count", "private_segment_fixed_size",
        "group_segment_fixed_size", "max_flat_workgroup_size")
print("%-78s %5s %5s %5s %6s %8s %8s %5s" % ("kernel", "vgpr", "agpr", "sgpr", "spill", "scratchB", "staticLDS", "wg"))
for path in sorted(glob.glob(sys.argv[1] + "/*-hip-amdgcn-amd-amdhsa-gfx950.s")):
    text = open(path).read()
    meta = text[text.rfind("amdhsa.kernels:"):]
    for blk in meta.split("  - .agpr_count:")[1:]:
        blk = ".agpr_count:" + blk
        vals = {k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1)) for k in keys}
        name = re.search(r"\.name:\s+(\S+)", blk).group(1)
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        print("%-78s %5d %5d %5d %6d %8d %8d %5d" % ((name[:78],) + tuple(vals[k] for k in keys)))
