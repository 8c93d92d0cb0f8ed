#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a per-kernel stats CSV.

    rocprofv3 --kernel-trace --stats -d <dir> -o <name> -- python bench.py ...
    python tools/rocpd_stats.py <dir>/<name>_results.db profiles/<tag>_kernel_stats.csv [windows | @B] [family.json precision]

Columns: kernel, calls, total_us, avg_us, min_us, max_us, pct, us_per_window (if `windows` given).
`@B` derives the window count from the trace itself: argmax_rows_kernel runs exactly once per forward, so
windows = (its call count) x B windows per forward - no hand-counted warm-up / replay bookkeeping.
`family.json`: the traced time of the implicit-GEMM conv / linear family per window (the kernels bench.py's roofline object is
about), tagged with the sha of the kernel sources, so that bench.py can quote the TRACED steady-state figure next to its
event-bracketed one (`roofline.traced`) and the fraction can be re-derived from profiles/ alone.
"""
import csv
import sqlite3
import sys


def main(db, out, windows=None):
    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration)/1e3, avg(duration)/1e3, min(duration)/1e3, "
                     "max(duration)/1e3 from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows)
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        hdr = ["kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"]
        if windows:
            hdr.append("us_per_window")
        w.writerow(hdr)
        for name, calls, tot, avg, mn, mx in rows:
            row = [name, calls, round(tot, 1), round(avg, 2), round(mn, 2), round(mx, 2), round(100 * tot / total, 2)]
            if windows:
                row.append(round(tot / windows, 1))
            w.writerow(row)
        w.writerow(["TOTAL", sum(r[1] for r in rows), round(total, 1), "", "", "", 100.0] +
                   ([round(total / windows, 1)] if windows else []))
    print(f"{out}: {len(rows)} kernels, {total / 1e3:.2f} ms of GPU time")


# kernels of the implicit-GEMM conv / linear family (bench.py: `roofline`), by the names they carry in a trace
FAMILY = ("igemm_kernel", "igemm2_kernel", "igemm4_kernel", "igemm5_kernel", "conv3x3_c64", "linear_k256_kernel", "rowchain_kernel",
          "rowchain_x3_kernel", "splitk_epilogue_kernel")


def family_summary(db, out_json, windows, per_forward, precision):
    import json
    import os

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from tools.pmc_traffic import source_sha16

    c = sqlite3.connect(db)
    rows = c.execute("select name, count(*), sum(duration)/1e3 from kernels group by name").fetchall()
    fam = {n: (cnt, us) for n, cnt, us in rows if any(k in n for k in FAMILY) and "sample" not in n}
    tot_us = sum(us for _, us in fam.values())
    all_us = sum(us for _, _, us in rows)
    blob = {"lib_sha16": source_sha16(), "precision": precision, "windows_per_forward": per_forward, "windows_in_trace": windows,
            "igemm_family_ms_per_window": round(tot_us / windows / 1e3, 4), "all_kernels_ms_per_window": round(all_us / windows / 1e3, 4),
            "launches_per_window": round(sum(cnt for cnt, _ in fam.values()) / windows, 2),
            "source": "rocprofv3 --kernel-trace of bench.py --lanes 1 (HIP-graph replay, one forward at a time); tools/rocpd_stats.py",
            "kernels_us_per_window": {n[:120]: round(us / windows, 1) for n, (cnt, us) in sorted(fam.items(), key=lambda kv: -kv[1][1])}}
    with open(out_json, "w") as f:
        json.dump(blob, f, indent=1)
    print(f"{out_json}: conv / linear family {blob['igemm_family_ms_per_window']} ms per window of {blob['all_kernels_ms_per_window']}")


def windows_from_trace(db, per_forward):
    c = sqlite3.connect(db)
    n = c.execute("select count(*) from kernels where name like '%argmax_rows_kernel%'").fetchone()[0]
    print(f"{n} forwards in the trace x {per_forward} windows per forward")
    return float(n * per_forward) if n else None


if __name__ == "__main__":
    win = None
    if len(sys.argv) > 3:
        win = windows_from_trace(sys.argv[1], int(sys.argv[3][1:])) if sys.argv[3].startswith("@") else float(sys.argv[3])
    main(sys.argv[1], sys.argv[2], win)
    if len(sys.argv) > 5 and win:
        family_summary(sys.argv[1], sys.argv[4], win, int(sys.argv[3][1:]) if sys.argv[3].startswith("@") else 0, sys.argv[5])
