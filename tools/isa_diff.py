"""Compares the device assembly of one kernel between two `hipcc -save-temps` outputs (labels and comments normalised):
python tools/isa_diff.py a.s b.s <mangled-name-prefix-a> [<prefix-b>] -- 'identical' means a source change did not reach that kernel."""
import re, sys, difflib
def body(path, key):
    s = open(path).read()
    m = re.search(r'^(' + re.escape(key) + r'[^:\n]*):', s, re.M)
    if not m: raise SystemExit(f"{key} not found in {path}")
    name = m.group(1)
    a = s.index('\n', m.end()) + 1
    b = s.index('s_endpgm', a)
    txt = s[a:b].replace(name, 'K')
    txt = re.sub(r';.*', '', txt)
    txt = re.sub(r'\.LBB\d+_', '.L', txt)
    return [l.strip() for l in txt.split('\n') if l.strip()]
if __name__ == "__main__":
    ka = sys.argv[3]; kb = sys.argv[4] if len(sys.argv) > 4 else ka
    a, b = body(sys.argv[1], ka), body(sys.argv[2], kb)
    if a == b: print("identical", len(a), "lines")
    else:
        d = list(difflib.unified_diff(a, b, lineterm='', n=0))
        print("DIFFERENT", len(a), len(b), "diff lines", len(d)); print('\n'.join(d[:60]))
