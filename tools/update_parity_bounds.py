"""Rewrite tests/parity_bounds.json from the record of a GPU test run.

    python tools/update_parity_bounds.py gpurun_out/_parity_measured.txt [--keep-larger]

Every `conftest.parity()` call of the GPU suite appends "label | measured | bound" to that record; this tool turns the
measured deviations into the committed table `label -> [measured, bound]`:

    bound = 20 x measured, rounded UP to 1 / 2 / 5 x 10^k, at least FLOOR (1e-13: a few ulps of the metric), and

* never above 1e-7 for a label that ends in a coefficient quantity (".x", ":x", "hist_x", ...) -- the documented tolerance
   -- unless the label is listed in EXCEPTIONS with its reason;
* labels written with an explicit bound at the call (parity_close) are not in the table and are left alone.

The policy: a regression from the measured 1e-11...1e-9 to 1e-6 must FAIL; a bound sits 10-30 x above what
the hardware measures so that a re-ordered reduction does not.  Results are bit-reproducible from box to box (fixed reduction
orders, 256 CUs everywhere), so the measured column does not move unless the library does.
"""
import json
import math
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TABLE = os.path.join(ROOT, "tests", "parity_bounds.json")
FLOOR = 1e-13
FACTOR = 20.0
COEFF = re.compile(r"[:.](x|x_\d|x2|hist_x|hx|hw|x_dop|x_res|x_drt|step_x|obs_x|fit_x)$")
# label pattern -> (cap, reason): quantities whose bound may exceed the documented 1e-7
EXCEPTIONS = [
    (re.compile(r"^c3\.first_max_iter_spectrum\."), 5e-6,
     "a fit that stops at max_iter = 50 without converging: every implementation amplifies its rounding (DESIGN section 2)"),
    (re.compile(r"test_resolve_c2grid"), 5e-6,
     "the coupled QP's inputs are seven fits, two of which stop at max_iter in both implementations"),
    (re.compile(r"random"), 2e-6,
     "draws with QPs that stop at coneqp's start point: a direct solve, cond * eps in any implementation"),
]


def round_up_125(v):
    if v <= 0:
        return FLOOR
    k = math.floor(math.log10(v))
    for m in (1, 2, 5, 10):
        cand = float(f"{m}e{k}")
        if cand >= v * (1 - 1e-12):
            return cand
    return float(f"1e{k + 1}")


def main():
    rec = sys.argv[1]
    keep_larger = "--keep-larger" in sys.argv
    try:
        with open(TABLE) as f:
            table = json.load(f)
    except OSError:
        table = {}
    n_new = n_tight = 0
    for line in open(rec):
        if line.startswith("#") or "|" not in line:
            continue
        label, meas, bound, _ = [t.strip() for t in line.split("|")]
        if ":" not in label and not label.startswith(("c2.", "c3.", "c4.")):
            continue                                   # parity_close labels carry their bound at the call
        meas = float(meas)
        new = max(round_up_125(FACTOR * meas), FLOOR)
        cap, reason, why = 1e-7 if COEFF.search(label) else None, None, None
        for pat, c, r in EXCEPTIONS:
            if pat.search(label):
                cap, reason = c, r
        if cap is not None and new > cap:
            if meas > cap:
                print(f"!! {label}: measured {meas:.2e} exceeds the cap {cap:.0e}")
            new = cap
        if reason is not None and new > 1e-7:
            why = reason                               # the stated reason for a bound above the documented 1e-7
        old = table.get(label)
        if old is not None and keep_larger and old[1] > new:
            new = old[1]
        if old is None:
            n_new += 1
        elif new < old[1]:
            n_tight += 1
        table[label] = [float(f"{meas:.3e}"), new] + ([why] if why else [])
    with open(TABLE, "w") as f:
        f.write("{\n" + ",\n".join(f" {json.dumps(k)}: {json.dumps(table[k])}" for k in sorted(table)) + "\n}\n")
    print(f"{len(table)} labels in {TABLE} ({n_new} new, {n_tight} tightened)")


if __name__ == "__main__":
    main()
