"""Fold two rocprofv3 counter passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md's HBM section prescribes) into profiles/qp_traffic.json: HBM bytes per launch of the dominant
kernel.  FETCH_SIZE / WRITE_SIZE are in KB; on gfx950 FETCH_SIZE counts half of the bytes moved by 16-byte-per-lane
streaming reads (the guide's correction), and every global load in the QP kernel is 16 B/lane, so fetch is doubled.

    python tools/pmc_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> [kernel substring]
"""
import csv
import json
import os
import sys


def fold(path, counter, kernel):
    """sum and count over the kernel's FULL-BATCH launches only (the largest grid in the file: one workgroup per spectrum of
    the 1024-spectrum batch) -- the launches bench.py's `roofline.avg_launch_ms` averages over; a run that also holds launches
    of sub-batch ranges (the one-caller leg) must not dilute the per-launch figure"""
    per = {}
    with open(path) as f:
        for row in csv.DictReader(f):
            if kernel in row["Kernel_Name"] and row["Counter_Name"] == counter:
                d = per.setdefault(row["Dispatch_Id"], [int(row["Grid_Size"]), 0.0])
                d[1] += float(row["Counter_Value"])
    if not per:
        return 0.0, 0, 0
    full = max(g for g, _ in per.values())
    vals = [v for g, v in per.values() if g == full]
    return sum(vals), len(vals), len(per) - len(vals)


def main():
    fetch_csv, write_csv = sys.argv[1], sys.argv[2]
    kernel = sys.argv[3] if len(sys.argv) > 3 else "qp_kernel_resident"
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import source_hash
    fetch_kb, nf, skipped = fold(fetch_csv, "FETCH_SIZE", kernel)
    write_kb, nw, _ = fold(write_csv, "WRITE_SIZE", kernel)
    if nf == 0 or nw == 0:
        raise SystemExit("kernel %r not found in the counter files" % kernel)
    fetch_raw = fetch_kb * 1024.0 / nf
    write_b = write_kb * 1024.0 / nw
    out = {
        "kernel": kernel,
        "source_hash": source_hash(),      # bench.py reports this figure only while the library's sources are these
        "launches": nf,
        "smaller_launches_not_counted": skipped,
        "fetch_size_kb_sum": fetch_kb,
        "write_size_kb_sum": write_kb,
        "fetch_bytes_per_launch_raw": fetch_raw,
        "fetch_bytes_per_launch_x2": 2.0 * fetch_raw,
        "write_bytes_per_launch": write_b,
        "hbm_bytes_per_launch": 2.0 * fetch_raw + write_b,
        "note": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over `bench.py --steps 1 "
                "--warmup 0 --inflight 1 --no-cpu-baseline --no-single-caller` (every launch = the 1024-spectrum batch of ONE plan, ONE range: "
                "the launches `roofline.avg_launch_ms` averages over; only full-batch launches are counted.  CORRECTION: the figures of "
                "rounds 4-5 (26.1 ... 26.4 GB) were averages that included the one-caller leg's half-size launches -- per full-batch "
                "launch they were 33.1 GB (profiles/r05_pmc_*.csv by grid size)); FETCH_SIZE doubled per "
                "MI355X_MICROARCH.md (gfx950 reports half of 16-B/lane streaming reads; the kernel's loads are all "
                "16 B/lane); WRITE_SIZE taken as is; KB -> bytes x1024.",
    }
    dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "qp_traffic.json")
    json.dump(out, open(dst, "w"), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
