#!/usr/bin/env python
"""Idle time between kernels in a rocprofv3 kernel trace (rocpd sqlite): for the steady-state forwards of bench.py (the
span between consecutive argmax_rows_kernel launches) report the wall time, the time at least one kernel was running, the
idle remainder and how it is distributed over the gaps, per stream and overall.

    python tools/rocpd_gaps.py <dir>/<name>_results.db [out.json]
"""
import json
import sqlite3
import sys


def main(db, out=None):
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)").fetchall()]
    scol = "stream_id" if "stream_id" in cols else ("queue_id" if "queue_id" in cols else None)
    q = f"select name, start, end, {scol or '0'} from kernels order by start"
    rows = c.execute(q).fetchall()
    marks = [r[1] for r in rows if "argmax_rows_kernel" in r[0]]
    if len(marks) < 4:
        print("columns:", cols, "kernels:", len(rows), "forwards:", len(marks))
        return
    res = []
    for t0, t1 in zip(marks[-4:-1], marks[-3:]):
        ks = [r for r in rows if r[1] >= t0 and r[1] < t1]
        ev = sorted((r[1], r[2]) for r in ks)
        busy, gaps, cur_s, cur_e = 0, [], ev[0][0], ev[0][1]
        for s, e in ev[1:]:
            if s > cur_e:
                busy += cur_e - cur_s
                gaps.append(s - cur_e)
                cur_s, cur_e = s, e
            else:
                cur_e = max(cur_e, e)
        busy += cur_e - cur_s
        per_stream = {}
        for r in ks:
            per_stream.setdefault(r[3], [0, 0])
            per_stream[r[3]][0] += 1
            per_stream[r[3]][1] += r[2] - r[1]
        g = sorted(gaps)
        res.append({"wall_ms": (t1 - t0) / 1e6, "kernels": len(ks), "busy_any_stream_ms": busy / 1e6,
                    "idle_ms": (t1 - t0 - busy) / 1e6, "gaps": len(g), "gap_median_us": g[len(g) // 2] / 1e3 if g else 0,
                    "gap_p90_us": g[int(len(g) * 0.9)] / 1e3 if g else 0, "gap_max_us": g[-1] / 1e3 if g else 0,
                    "gaps_over_20us": sum(1 for x in g if x > 20000), "idle_in_gaps_over_20us_ms": sum(x for x in g if x > 20000) / 1e6,
                    "per_stream": {str(k): {"kernels": v[0], "kernel_ms": v[1] / 1e6} for k, v in per_stream.items()}})
    for r in res:
        print(json.dumps(r))
    if out:
        json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:3])
