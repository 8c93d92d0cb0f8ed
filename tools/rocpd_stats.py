#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a per-kernel stats CSV.

    rocprofv3 --kernel-trace --stats -d <dir> -o <name> -- python bench.py ...
    python tools/rocpd_stats.py <dir>/<name>_results.db profiles/<tag>_kernel_stats.csv [windows | @B]

Columns: kernel, calls, total_us, avg_us, min_us, max_us, pct, us_per_window (if `windows` given).
`@B` derives the window count from the trace itself: argmax_rows_kernel runs exactly once per forward, so
windows = (its call count) x B windows per forward - no hand-counted warm-up / replay bookkeeping.
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
