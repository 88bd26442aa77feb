"""One plan, one caller, k ranges: a few fits for `rocprofv3 --kernel-trace` (tools/trace_ranges.sh); the analysis half reads the
trace and prints, for the LAST fit, how the launch sequences of the ranges overlap.
    python tools/probe_trace_ranges.py run <k> <B> [idle]   (under rocprofv3; `idle` contexts = idle streams are created first)
    python tools/probe_trace_ranges.py show <kernel_trace.csv> <k>"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

if sys.argv[1] == "run":
    from hipdrt import synth
    from hipdrt.models import DRT
    k, B = int(sys.argv[2]), int(sys.argv[3])
    c2 = synth.config_c2()
    z = synth.zarc2_batch(c2["freq"], B)
    idle = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    from hipdrt import _ffi
    dummies = [_ffi.Context(0) for _ in range(idle)]
    drt = DRT(fixed_basis_tau=c2["tau"], context=_ffi.Context(0)) if idle else DRT(fixed_basis_tau=c2["tau"])
    plan = drt.stage_batch(c2["freq"], z)
    plan.set_subbatches(k)
    drt.fit_staged(); drt.fit_staged()
    t0 = time.perf_counter(); drt.fit_staged(); dt = time.perf_counter() - t0
    print(f"k = {k}, B = {B}, idle streams first {idle}: {B / dt:.1f} fits/s, last fit {dt * 1e3:.1f} ms", flush=True)
elif sys.argv[1] == "tail":
    # python tools/probe_trace_ranges.py tail <kernel_trace.csv> <n>: queues and overlap of the LAST n coneqp launches
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    qp = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0))
                for r in rows if "qp_kernel_resident" in r["Kernel_Name"])[-int(sys.argv[3]):]
    pts = sorted([(e[0], 1) for e in qp] + [(e[1], -1) for e in qp])
    cur, prev, hist = 0, qp[0][0], {}
    for t, d in pts:
        hist[cur] = hist.get(cur, 0) + (t - prev)
        cur += d; prev = t
    print(f"last {len(qp)} coneqp launches: time with n running, ms:", {n: round(v / 1e6, 1) for n, v in sorted(hist.items())})
    by_q = {}
    for e in qp:
        by_q.setdefault(e[2], []).append(e)
    for q, es in sorted(by_q.items()):
        print(f"  queue {q}: {len(es)} launches, grids {sorted(set(e[3] for e in es), reverse=True)[:3]}")
else:
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    k = int(sys.argv[3])
    ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", r.get("Queue_ID", "0")),
           int(r.get("Grid_Size", r.get("Grid_Size_X", 0)) or 0)) for r in rows]
    ev.sort()
    # the last fit: everything after the last gap of more than 2 ms between consecutive kernel starts... take the final third
    qp = [e for e in ev if "qp_kernel_resident" in e[2]]
    n_fit = len(qp) // 3
    last = qp[-n_fit:]
    t_lo, t_hi = last[0][0], max(e[1] for e in last)
    win = [e for e in ev if e[0] >= t_lo - 2_000_000 and e[1] <= t_hi + 2_000_000]
    span = (t_hi - t_lo) / 1e6
    print(f"last fit: {len(last)} coneqp launches on {len(set(e[3] for e in last))} queues, {span:.1f} ms from the first coneqp start to the last end")
    # time with 0 / 1 / 2 / ... coneqp kernels running
    pts = sorted([(e[0], 1) for e in last] + [(e[1], -1) for e in last])
    cur, prev, hist = 0, t_lo, {}
    for t, d in pts:
        hist[cur] = hist.get(cur, 0) + (t - prev)
        cur += d; prev = t
    print("time with n coneqp launches running, ms:", {n: round(v / 1e6, 1) for n, v in sorted(hist.items())})
    by_q = {}
    for e in last:
        by_q.setdefault(e[3], []).append(e)
    for q, es in by_q.items():
        d = [(e[1] - e[0]) / 1e6 for e in es]
        print(f"queue {q}: {len(es)} launches, sum {sum(d):.1f} ms, mean {np.mean(d):.2f}, first five {[round(x, 2) for x in d[:5]]}, grid of the first {es[0][4]}")
    names = {}
    for e in win:
        key = e[2].split("(")[0][-40:]
        names[key] = names.get(key, 0) + (e[1] - e[0]) / 1e6
    print("kernel time inside the window, ms:", {k_: round(v, 1) for k_, v in sorted(names.items(), key=lambda kv: -kv[1])[:8]})
