"""Where a kernel's scratch (spill) instructions sit, by source line: compile with -gline-tables-only --save-temps=obj and run
python tools/spill_sites.py <file.s> <mangled kernel name>.  The coneqp kernel's hand-issued asm loads must not have their
destination registers spilled while in flight, so its operand-ring loops have to be free of scratch traffic."""
import collections
import re
import sys

txt = open(sys.argv[1]).read()
start = txt.index(sys.argv[2] + ':')
body = txt[start:txt.index('.Lfunc_end', start)]
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', txt):
    files[int(m.group(1))] = (m.group(3) or m.group(2))
cur, cnt = None, collections.Counter()
for line in body.split('\n'):
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', line)
    if m:
        cur = (files.get(int(m.group(1)), '?').split('/')[-1], int(m.group(2)))
    elif 'scratch_' in line:
        cnt[(cur, 'store' if 'store' in line else 'load')] += 1
for k, v in sorted(cnt.items(), key=lambda kv: (kv[0][0][0], kv[0][0][1])):
    print(k[0][0], k[0][1], k[1], v)
