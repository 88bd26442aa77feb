"""gfx9 hazard "VALU writes an SGPR (v_readlane / v_readfirstlane: SGPR spill reloads, uniform_ptr) -> VMEM reads that SGPR
needs 5 wait states".  hipcc's hazard recognizer covers its own memory instructions, NOT the hand-issued loads of the coneqp
kernels (inline asm: global_load_dwordx4 v, v, s[base:base+1]) -- with the base reloaded from a VGPR lane right in front of such a
load the load takes a stale base (observed: memory faults on address 0 / on pointers with a wrong upper half in code regions
that were new, i.e. where register allocation put spill reloads next to the loads; profiles/r03p_ab_sweep_lookahead.txt).
This lists every hand-issued load whose scalar base was written by a VALU instruction fewer than 5 wait states earlier:

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -Iinclude -S --cuda-device-only hybrid-drt_amd/csrc/qp.hip -o /tmp/qp.s
    python tools/sgpr_hazard.py /tmp/qp.s        (no output = clean)
"""
import re
import sys

def _merge(a, b):
    """union of two 'recently written' states, the younger write of an SGPR wins"""
    d = dict(a)
    for s_, age in b:
        if s_ not in d or age < d[s_]:
            d[s_] = age
    return sorted(d.items())


def scan(txt):
    """-> list of (kernel, instruction, sgpr, wait states) for every hand-issued load that reads a VALU-written SGPR too early.
    Control flow is followed: the state at a label is the union of the fall-through state and of the states at every branch
    to it (backward branches included: the scan of a function is repeated until the label states stop changing), so a spill
    reload at the bottom of a loop body, or at the end of the block in front of a loop header, is seen by a load at the top."""
    found = set()
    for m in re.finditer(r'^(_ZN6hipdrt\w+):', txt, re.M):
        name = m.group(1)
        end = txt.find('.Lfunc_end', m.start())
        lines = [l.split(';')[0].strip() for l in txt[m.start():end].split('\n')]
        lines = [l for l in lines if l and not l.startswith('.set') and not l.startswith('#') and
                 (not l.startswith('.') or l.endswith(':'))]
        label_in = {}        # label -> state carried in by branches
        for _ in range(4):
            changed = False
            recent, live = [], True          # (sgpr number, wait states since the write); live = reachable by fall-through
            for l in lines:
                if l.endswith(':'):
                    lab = l[:-1]
                    recent = _merge(recent if live else [], label_in.get(lab, []))
                    live = True
                    continue
                op = l.split()[0]
                w = re.match(r'v_read(?:first)?lane_b32\s+s(\d+)', l)
                if op.startswith('global_load') or op.startswith('global_store'):
                    sb = re.search(r's\[(\d+):(\d+)\]', l)
                    if sb:
                        lo, hi = int(sb.group(1)), int(sb.group(2))
                        for s_, age in recent:
                            if lo <= s_ <= hi and age < 5:
                                found.add((name, l, s_, age))
                nops = re.match(r's_nop\s+(\d+)', l)
                step = int(nops.group(1)) + 1 if nops else 1
                recent = [(s_, age + step) for s_, age in recent if age + step < 8]
                if w:
                    recent.append((int(w.group(1)), 0))
                elif op.startswith('s_') and not op.startswith(('s_cmp', 's_cbranch', 's_branch', 's_nop', 's_waitcnt', 's_barrier',
                                                                  's_sleep', 's_setreg', 's_bitcmp', 's_endpgm', 's_setprio')):
                    # a scalar instruction that overwrites the register ends the hazard: the value a later load reads is SALU-written
                    dm = re.match(r'\S+\s+(?:s(\d+)|s\[(\d+):(\d+)\])\s*,', l)
                    if dm:
                        lo_ = int(dm.group(1) if dm.group(1) is not None else dm.group(2))
                        hi_ = int(dm.group(1) if dm.group(1) is not None else dm.group(3))
                        recent = [(s_, age) for s_, age in recent if not (lo_ <= s_ <= hi_)]
                br = re.match(r's_c?branch\w*\s+(\S+)', l)
                if br:
                    lab = br.group(1)
                    merged = _merge(label_in.get(lab, []), recent)
                    if merged != label_in.get(lab, []):
                        label_in[lab] = merged
                        changed = True
                    if op == 's_branch':
                        live = False
                elif op in ('s_endpgm', 's_setpc_b64'):
                    live = False
            if not changed:
                break
    return sorted(found)


if __name__ == '__main__':
    found = scan(open(sys.argv[1]).read())
    for name, l, s_, age in found:
        print(name[:60], '|', l[:80], '| s%d written %d wait state(s) earlier' % (s_, age))
    print('%d hazard(s)' % len(found))
