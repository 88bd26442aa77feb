"""Rough VGPR pressure profile of one kernel from hipcc's assembly (-gline-tables-only --save-temps=obj): a backward liveness
pass over the LINEAR instruction stream (branches ignored, so loop-carried values are under-counted) and, per source line,
the largest number of simultaneously live VGPRs seen.  python tools/vgpr_pressure.py <file.s> <mangled kernel> [top]"""
import re
import sys

txt = open(sys.argv[1]).read()
name = sys.argv[2]
top = int(sys.argv[3]) if len(sys.argv) > 3 else 25
start = txt.index(name + ':')
body = txt[start:txt.index('.Lfunc_end', start)].split('\n')
files = {}
for m in re.finditer(r'\.file\s+(\d+)\s+"([^"]*)"(?:\s+"([^"]*)")?', txt):
    files[int(m.group(1))] = (m.group(3) or m.group(2)).split('/')[-1]
NODEF = ('global_store', 'scratch_store', 'ds_write', 'buffer_store', 'flat_store', 'v_cmp', 'v_cmpx', 's_', 'ds_add', 'global_atomic',
         'v_readlane', 'v_readfirstlane', 'ds_gws', 'v_nop')
def regs(tok):
    out = []
    for m in re.finditer(r'\bv\[(\d+):(\d+)\]|\bv(\d+)\b', tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out
ins, cur = [], None
for l in body:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m:
        cur = (files.get(int(m.group(1)), '?'), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(('.', ';')) or t.endswith(':'):
        continue
    t = t.split(';')[0]
    parts = t.split(None, 1)
    mn = parts[0]
    ops = [o.strip() for o in parts[1].split(',')] if len(parts) > 1 else []
    if mn.startswith(NODEF):
        d, u = [], sum((regs(o) for o in ops), [])
    else:
        d = regs(ops[0]) if ops else []
        u = sum((regs(o) for o in ops[1:]), [])
        if 'mfma' in mn or 'fmac' in mn or 'writelane' in mn or 'mac' in mn:
            u += d
    ins.append((cur, d, u))
live, best = set(), {}
for cur, d, u in reversed(ins):
    live -= set(d)
    live |= set(u)
    if cur is not None and len(live) > best.get(cur, 0):
        best[cur] = len(live)
for (f, ln), v in sorted(best.items(), key=lambda kv: -kv[1])[:top]:
    print(f"{v:4d} live VGPRs at {f}:{ln}")
