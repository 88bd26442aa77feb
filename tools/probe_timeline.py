"""Diagnostic: per-member, per-wavefront time stamps of one block column of the group kernel's factorisation (library built
with -DHIPDRT_GRP_TIMELINE=<block column>: make VARIANT=tl EXTRA=-DHIPDRT_GRP_TIMELINE=8; HIPDRT_LIB=.../libhipdrt_tl.so).
python tools/probe_timeline.py [n] [G]   -- k cycles relative to the member's own wavefront-1 column top"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import _ffi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 514
G = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rng = np.random.default_rng(n)
A = rng.standard_normal((n + 50, n)) / np.sqrt(n)
P = A.T @ A + 1e-3 * np.eye(n); q = -A.T @ (A @ np.maximum(rng.standard_normal(n), 0))
ctx = _ffi.get_context(0)
ctx.debug_qp_group(G)
ctx.qp_batch(P[None], q[None], np.zeros(n))
buf = (C.c_ulonglong * (32 * 8 * 10 + 32))()
lib = C.CDLL(os.environ["HIPDRT_LIB"])
assert lib.hipdrt_debug_group_timeline(buf) == 0
t = np.array(buf[:2560], dtype=np.uint64).reshape(32, 8, 10).astype(np.int64); own = np.array(buf[2560:])
names = {0: ["top", "block staged", "first 16x16 inverted", "", "", "chain done", "(A)", "(A2)", "at (A2)", ""],
         1: ["top", "totals", "", "new range done / flag seen", "published / fetched", "at (A)", "(A)", "end", "", ""]}
rows = ["top", "rowdone", "", "", "old range done", "rank-k done", "(A)", "end", "", "ring starts"]
for g in range(G):
    base = t[g, 1, 0]
    if base == 0: continue
    print(f"member {g}{' (OWNER of the look-ahead rows)' if own[g] else ''}")
    for w in range(8):
        nm = names.get(w, rows)
        print(f"   wavefront {w}: " + ", ".join(f"{nm[e]} {(t[g, w, e] - base) / 1000:.1f}" for e in range(len(nm)) if t[g, w, e] and nm[e]))
ctx.debug_qp_group(-1)
