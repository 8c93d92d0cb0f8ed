#!/usr/bin/env python
"""Per-kernel matrix-pipe utilisation and effective shader clock from two rocprofv3 --pmc passes of the same command
(rocpd sqlite output; `--kernel-trace` in both so that the dispatch durations are in the same file as the counter):

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES -d out/mfma -o r -- python bench.py --steps 1 --no-graph --lanes 1 ...
    rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE         -d out/gui  -o r -- (the same)
    python tools/pmc_mfma.py out/mfma/..._results.db out/gui/..._results.db profiles/<tag>_mfma_utilisation.json

Per kernel name (summed over its launches): time, GRBM_GUI_ACTIVE (summed over the 8 XCDs by the tool) -> effective clock =
active cycles / 8 / time; SQ_VALU_MFMA_BUSY_CYCLES -> the fraction of the chip's 1024 SIMD-cycles (256 CUs x 4) at that clock
during which a matrix pipe was busy.  (A 32x32x16 16-bit MFMA keeps its SIMD's pipe busy for 32 cycles, a 16x16x32 one for 16:
MI355X_MICROARCH.md.)"""
import json
import sqlite3
import sys


def table(db, counter):
    c = sqlite3.connect(db)
    cnt = dict(c.execute("select kernel_name, sum(value) from counters_collection where counter_name = ? group by kernel_name",
                         (counter,)).fetchall())
    dur = {n: (k, t) for n, k, t in c.execute("select name, count(*), sum(duration) from kernels group by name").fetchall()}
    return cnt, dur


def main(mfma_db, gui_db, out):
    busy, dur_m = table(mfma_db, "SQ_VALU_MFMA_BUSY_CYCLES")
    act, dur_g = table(gui_db, "GRBM_GUI_ACTIVE")
    rows = []
    for name, (calls, t_ns) in sorted(dur_m.items(), key=lambda kv: -kv[1][1]):
        if name not in busy or name not in act or name not in dur_g:
            continue
        clock = act[name] / 8.0 / dur_g[name][1]            # GHz: cycles per ns, one XCD's active cycles
        simd_cycles = 1024.0 * clock * t_ns                  # available SIMD-cycles during this kernel's launches of pass 1
        rows.append({"kernel": name[:140], "launches": calls, "ms": round(t_ns / 1e6, 3), "effective_clock_ghz": round(clock, 3),
                     "mfma_busy_fraction": round(busy[name] / simd_cycles, 4) if simd_cycles else None})
    tot_t = sum(r["ms"] for r in rows)
    res = {"note": "per kernel, summed over its launches of one eager forward pass (one forward at a time, no HIP graph); "
                   "mfma_busy_fraction = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x effective clock x time)",
           "kernel_ms_total": round(tot_t, 2),
           "time_weighted_mfma_busy_fraction": round(sum(r["ms"] * (r["mfma_busy_fraction"] or 0) for r in rows) / tot_t, 4),
           "time_weighted_clock_ghz": round(sum(r["ms"] * r["effective_clock_ghz"] for r in rows) / tot_t, 3),
           "kernels": rows[:40]}
    json.dump(res, open(out, "w"), indent=1)
    print(out, "time-weighted MFMA busy", res["time_weighted_mfma_busy_fraction"], "clock", res["time_weighted_clock_ghz"], "GHz")
    for r in rows[:16]:
        print(f'{r["ms"]:8.2f} ms  {r["effective_clock_ghz"]:.2f} GHz  mfma {r["mfma_busy_fraction"]:.3f}  {r["kernel"][:90]}')


if __name__ == "__main__":
    main(*sys.argv[1:4])
