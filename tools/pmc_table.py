#!/usr/bin/env python
"""Per-kernel table of hardware counters from rocprofv3 --pmc passes (rocpd sqlite output, `--kernel-trace` in each pass):

    python tools/pmc_table.py out.json pass1_results.db [pass2_results.db ...]

Every counter found in a pass is summed per kernel name; durations and launch counts come from the first pass.  For FETCH_SIZE /
WRITE_SIZE (KiB) the table adds the HBM-side bytes per launch, the read side doubled on gfx950 as MI355X_MICROARCH.md
prescribes for 16-byte-per-lane streams (tools/pmc_traffic.py), and the rate they stand for."""
import json
import sqlite3
import sys


def counters(db):
    c = sqlite3.connect(db)
    out = {}
    for name, counter, v in c.execute("select kernel_name, counter_name, sum(value) from counters_collection group by kernel_name, counter_name"):
        out.setdefault(name, {})[counter] = v
    dur = {n: (k, t) for n, k, t in c.execute("select name, count(*), sum(duration) from kernels group by name").fetchall()}
    return out, dur


def main(out, *dbs):
    merged, dur = {}, None
    for db in dbs:
        cnt, d = counters(db)
        dur = dur or d
        for k, v in cnt.items():
            merged.setdefault(k, {}).update(v)
    rows = []
    for name, (calls, t_ns) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        c = merged.get(name, {})
        r = {"kernel": name[:150], "launches": calls, "ms": round(t_ns / 1e6, 3)}
        r.update({k: v for k, v in sorted(c.items())})
        if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
            rd, wr = 2.0 * c.get("FETCH_SIZE", 0.0) * 1024, c.get("WRITE_SIZE", 0.0) * 1024
            r["hbm_read_gb_x2"], r["hbm_write_gb"] = round(rd / 1e9, 3), round(wr / 1e9, 3)
            r["hbm_gb_per_launch"] = round((rd + wr) / calls / 1e9, 4)
            r["hbm_tbs"] = round((rd + wr) / t_ns / 1e3, 3)
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c and c["TCC_HIT_sum"] + c["TCC_MISS_sum"] > 0:
            r["l2_hit_rate"] = round(c["TCC_HIT_sum"] / (c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        rows.append(r)
    json.dump({"note": "per kernel name, summed over its launches in the profiled command; FETCH_SIZE / WRITE_SIZE in KiB as "
                       "reported; hbm_read_gb_x2 = 2 x FETCH_SIZE (gfx950 correction for wide coalesced reads)", "kernels": rows},
              open(out, "w"), indent=1)
    for r in rows[:40]:
        print(f'{r["ms"]:8.2f} ms x{r["launches"]:4d}  {r.get("hbm_gb_per_launch", 0):8.3f} GB/launch  {r.get("hbm_tbs", 0):6.2f} TB/s  '
              f'L2 hit {r.get("l2_hit_rate", "-")}  {r["kernel"][:80]}')


if __name__ == "__main__":
    main(sys.argv[1], *sys.argv[2:])
