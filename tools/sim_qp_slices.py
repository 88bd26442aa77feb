"""Would decoupling the spectra of a batch from the pace of its slowest QP pay?  (CPU only; the checker's iteration counts.)

Every outer iteration of a range is ONE coneqp launch: all its spectra wait for the launch's slowest problem.  The alternative
considered in round 6: cap a launch at K interior-point iterations per problem, let unfinished problems carry their iterates
(they live in global memory anyway) into the next launch and let finished ones go on to their hyper-parameter update -- every
spectrum at its own pace, per-spectrum bits unchanged.  This script takes the interior-point counts of every QP of N full-size
fits from the CPU checker and plays both schedules: a round lasts max(slowest problem, all problems / workgroup slots) x one
factorisation time, plus a fixed part per round (hyper-parameter kernels, Gram launch, read-back) and a Gram part per updated spectrum.

    python tools/sim_qp_slices.py [N=256]       (8 worker processes, about a minute)
"""
import os, sys, time
for v in ("OMP_NUM_THREADS", "OPENBLAS_NUM_THREADS", "MKL_NUM_THREADS"):
    os.environ[v] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multiprocessing as mp
import numpy as np
from hipdrt import synth
from oracle import drt_oracle as orc

c2 = synth.config_c2()
N = int(sys.argv[1]) if len(sys.argv) > 1 else 256
z = synth.zarc2_batch(c2["freq"], N)
od = orc.OracleDRT(fixed_basis_tau=c2["tau"])
od.prepare(c2["freq"])


def one(i):
    od.fit_eis(c2["freq"], z[i], structure='fast', keep_history=True)
    return [l["iterations"] for l in od.qp_log]


F_MS, SLOTS, ROUND_MS, GRAM_MS = 0.44, 256, 0.9, 0.00306     # factorisation under load (profiles/r06_qp_saturation.txt), one workgroup per CU, per round, per spectrum


def play(logs, B, K):
    seqs = [logs[i % len(logs)] for i in range(B)]
    pos, rem = [0] * B, [s[0] + 1 for s in seqs]               # factorisations left in the current QP: start point + iterations
    active, t, rounds, busy = set(range(B)), 0.0, 0, 0.0
    while active:
        members = list(active)
        work = np.array([min(rem[b], K if K else 10 ** 9) for b in members], float)
        t_round = max(work.max(), work.sum() / SLOTS) * F_MS
        updated = 0
        for b, w in zip(members, work):
            rem[b] -= int(w)
            if rem[b] <= 0:
                updated += 1
                pos[b] += 1
                if pos[b] >= len(seqs[b]):
                    active.discard(b)
                else:
                    rem[b] = seqs[b][pos[b]] + 1
        t += t_round + ROUND_MS + updated * GRAM_MS
        busy += work.sum() * F_MS
        rounds += 1
    return t, rounds, busy / (t * SLOTS)


if __name__ == "__main__":
    t0 = time.time()
    with mp.get_context("fork").Pool(8) as pool:
        logs = pool.map(one, range(N), chunksize=4)
    allc = np.concatenate([np.array(l) for l in logs])
    print(f"{N} fits from the CPU checker in {time.time() - t0:.0f} s: {np.mean([len(l) for l in logs]):.1f} QPs per spectrum; interior-point "
          f"iterations per QP: mean {allc.mean():.2f}, median {int(np.median(allc))}, 90th pct {int(np.quantile(allc, .9))}, "
          f"99th pct {int(np.quantile(allc, .99))}, max {allc.max()}")
    for B in (312, 1024, 2500):
        for K in (0, 8, 6, 5, 4, 3):
            t, r, u = play(logs, B, K)
            print(f"range of {B:5d} spectra, {'no cap' if not K else f'cap {K:2d}'}: {t:7.1f} ms in {r:3d} rounds, {B / t * 1000:7.1f} fits/s, "
                  f"workgroup slots busy {u:.2f}")
