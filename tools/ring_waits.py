"""Compiler-inserted `s_waitcnt vmcnt(..)` between FP64 MFMAs of the coneqp kernels, by source line.

The operand rings issue their loads by hand (qp_common.hpp: gload16) and count them with their own vm_wait<N>; the compiler
does not model those loads, but it does keep its own books for every load IT emitted (tile initialisation, prefetched source
tiles).  If the first use of such a value sits behind a branch inside a ring, its wait-count pass places a vmcnt(0) there, and
because one path around the loop has no wait the pending state survives the back edge: every ring step then drains all
hand-issued loads (measured: 9.27 -> 11.7 ms per launch).  Waits listed here that point INTO a ring (factor_rows /
factor_lookahead / ring_* lines) are that bug; the ones at the panel solves behind barrier (A) are ordinary.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -gline-tables-only -Iinclude -c hybrid-drt_amd/csrc/qp.hip -o /tmp/x/qp.o --save-temps=obj
    python tools/ring_waits.py /tmp/x/qp-hip-amdgcn-amd-amdhsa-gfx950.s
"""
import re
import sys

txt = open(sys.argv[1]).read()
for m in re.finditer(r'^(_ZN6hipdrt\d+qp_kernel_\w+):', txt, re.M):
    name = m.group(1)
    body = txt[m.start():txt.index('.Lfunc_end', m.start())].split('\n')
    where = None
    for i, line in enumerate(body):
        loc = re.search(r'; \./(\w+\.hpp):(\d+)', line)
        if loc and '.loc' in line:
            where = loc.group(1) + ':' + loc.group(2)
        if 's_waitcnt' in line and 'vmcnt' in line:
            before = any('mfma_f64' in c for c in body[max(0, i - 6):i])
            after = any('mfma_f64' in c for c in body[i + 1:i + 7])
            if before and after:
                print(name[10:48], line.strip().split(';')[0].strip(), where)
