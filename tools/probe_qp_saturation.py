"""How the time of ONE interior-point iteration of a resident workgroup depends on how many problems are in flight: B copies of
one C2-size QP (shared P, so the upload is small; identical iteration counts, so no ragged end), hipdrt_qp_batch wall time per
iteration and per round of 512 workgroup slots (256 CUs x 2).  python tools/probe_qp_saturation.py [B ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from hipdrt import synth, _ffi
from oracle import drt_oracle as orc

c2 = synth.config_c2()
d = orc.OracleDRT(fixed_basis_tau=c2["tau"])
m = d.prepare(c2["freq"])
z = synth.zarc2_spectrum(c2["freq"], 0)
cs = (z.real.max() - z.real.min()) / 14
rv = np.concatenate([z.real, z.imag]) / cs
hyp = orc.get_default_hypers()
l2 = orc.calculate_qp_l2_matrix(hyp, np.ones(3), m["pen"], [np.ones(514)] * 3, 2)
w = np.full(512, 50.0)
wa = w[:, None] * m["rzm"]
P = wa.T @ wa + l2
q = -wa.T @ (w * rv)
h = np.zeros(514)
ctx = _ffi.get_context(0)
ctx.debug_qp_group(0)                       # the batch kernel at every B
for B in [int(a) for a in sys.argv[1:]] or [1, 64, 128, 256, 384, 512, 768, 1024, 2048, 4096]:
    qs = np.tile(q, (B, 1))
    best = 1e9
    for rep in range(3):
        t = time.perf_counter(); res = ctx.qp_batch(P, qs, h); dt = time.perf_counter() - t
        best = min(best, dt)
    it = int(res["iterations"][0]) + 1      # + the start point's factorisation
    rounds = -(-B // 512)
    print(f"B = {B:5d}: {best * 1e3:8.2f} ms for {it} factorisations per problem = {best * 1e3 / it:6.3f} ms each per round of all problems, "
          f"{best * 1e3 / it / rounds:6.3f} ms per factorisation and slot round, {B * it / best / 1e3:8.1f} factorisations per ms", flush=True)
