#!/usr/bin/env python
"""Aggregate HBM traffic of the implicit-GEMM kernel from two rocprofv3 --pmc passes (rocpd sqlite output):

  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d out/fetch -o r -- python bench.py --steps 2 --warmup 1 --no-graph ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d out/write -o r -- python bench.py --steps 2 --warmup 1 --no-graph ...
  python tools/pmc_traffic.py out/fetch/r_results.db out/write/r_results.db profiles/r3_igemm_traffic_pmc.json [precision B]

`precision`, `B` (windows per forward) of the profiled command and a sha256 over the kernel sources it ran are stored in the
file: bench.py quotes the measurement as `roofline.traffic` only for the matching configuration AND the same kernel sources.
Besides the conv / linear family the file carries the WHOLE forward: HBM bytes of every kernel, per forward and per window
(forwards counted from the trace: argmax_rows_kernel runs once per forward).

Corrections per /opt/skills/guides/MI355X_MICROARCH.md §HBM: FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE
counts 128-byte requests as 64 B for wide (16 B/lane) coalesced reads, so the read side is doubled.
"""
import json
import sqlite3
import sys


FAMILY = ("%igemm%_kernel%", "%conv3x3_c64_kernel%", "%conv3x3_c64_x3_kernel%", "%conv3x3_c64_ring_kernel%", "%linear_k256_kernel%", "%splitk_epilogue_kernel%", "%rowchain%_kernel%")   # the conv / linear kernels of csrc/igemm*.hip


def source_sha16():
    """identity of the kernels a measurement ran: the source sha compiled into the built libpgt_hip.so (pgtformer_amd.build.
    binary_sha16: the BINARY's stamp); without a stamped library, the sha of the sources in the tree"""
    import os
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from pgtformer_amd import build
    return build.binary_sha16() or build.source_sha16()


def per_kernel(db, counter, likes):
    c = sqlite3.connect(db)
    rows = []
    for like in likes:
        rows += c.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? "
                          "and kernel_name like ? group by kernel_name", (counter, like)).fetchall()
    tot = sum(r[1] for r in rows)
    n = sum(r[2] for r in rows)
    return tot, n


def main(fetch_db, write_db, out, precision=None, windows_per_forward=None):
    f_kib, nf = per_kernel(fetch_db, "FETCH_SIZE", FAMILY)
    w_kib, nw = per_kernel(write_db, "WRITE_SIZE", FAMILY)
    res = {"kernel": "conv/linear family: igemm*_kernel (igemm, igemm2, igemm4, igemm5), conv3x3_c64_kernel / conv3x3_c64_x3_kernel / conv3x3_c64_ring_kernel (igemm6, igemm6x3, igemm8), linear_k256_kernel (igemm7), rowchain kernels (fused LayerNorm -> Linear / proj -> Mlp chains), split-K reduce", "launches": nf,
           "fetch_bytes_per_launch_raw": f_kib * 1024 / max(nf, 1),
           "fetch_bytes_per_launch_corrected_x2": 2 * f_kib * 1024 / max(nf, 1),
           "write_bytes_per_launch": w_kib * 1024 / max(nw, 1),
           "hbm_bytes_per_launch": (2 * f_kib * 1024) / max(nf, 1) + w_kib * 1024 / max(nw, 1),
           "note": "read side doubled per MI355X_MICROARCH.md (gfx950 FETCH_SIZE under-count for 16 B/lane streams); "
                   "Infinity-Cache hits are included in these L2 fabric counters"}
    fa_kib, na = per_kernel(fetch_db, "FETCH_SIZE", ("%",))
    wa_kib, _ = per_kernel(write_db, "WRITE_SIZE", ("%",))
    _, fwd = per_kernel(fetch_db, "FETCH_SIZE", ("%argmax_rows_kernel%",))
    fwd = max(fwd, 1)
    res["whole_forward"] = {"forwards_in_trace": fwd, "kernel_launches_per_forward": na / fwd,
                            "hbm_read_bytes_per_forward_corrected_x2": 2 * fa_kib * 1024 / fwd,
                            "hbm_write_bytes_per_forward": wa_kib * 1024 / fwd,
                            "hbm_bytes_per_forward": (2 * fa_kib + wa_kib) * 1024 / fwd}
    if precision is not None:
        res["precision"] = precision
        res["windows_per_forward"] = int(windows_per_forward)
        res["whole_forward"]["hbm_gb_per_window"] = res["whole_forward"]["hbm_bytes_per_forward"] / int(windows_per_forward) / 1e9
    res["lib_sha16"] = source_sha16()      # of the kernel sources this measurement ran (bench.py refuses another build's file)
    json.dump(res, open(out, "w"), indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main(*sys.argv[1:6])
