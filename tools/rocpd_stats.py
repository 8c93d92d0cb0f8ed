#!/usr/bin/env python
"""Summarise a rocprofv3 (ROCm 7.2, rocpd sqlite output) kernel trace into a per-kernel stats CSV.

    rocprofv3 --kernel-trace --stats -d <dir> -o <name> -- python bench.py ...
    python tools/rocpd_stats.py <dir>/<name>_results.db profiles/<tag>_kernel_stats.csv [windows]

Columns: kernel, calls, total_us, avg_us, min_us, max_us, pct, us_per_window (if `windows` given).
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


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else None)
