"""The tools that turn rocprofv3 output (rocpd sqlite) into the numbers DESIGN.md / bench.py quote - tools/rocpd_stats.py,
tools/pmc_traffic.py, tools/pmc_table.py, tools/pmc_mfma.py - on a synthetic database with known contents: unit conversion (KiB),
the gfx950 read-side correction (x2), forwards counted from the trace, the family filter, and the staleness key (sha256 of the
kernel sources) that bench.py checks before it quotes a PMC file."""
import csv
import json
import os
import sqlite3
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)

IG = "void (anonymous namespace)::igemm4_kernel<2, 4, false, false, false, half_t>((anonymous namespace)::ConvP)"
AF = "void (anonymous namespace)::affine_act_kernel<half_t, false>(half_t const*, int)"
AM = "(anonymous namespace)::argmax_rows_kernel(float const*, int, int, int, int*)"


def _db(path, counter, values, durations):
    """values: {kernel: [per-launch counter value]}, durations: {kernel: [ns]}"""
    c = sqlite3.connect(path)
    c.execute("create table kernels (name text, duration integer)")
    c.execute("create table counters_collection (kernel_name text, counter_name text, value real)")
    for k, ds in durations.items():
        c.executemany("insert into kernels values (?, ?)", [(k, d) for d in ds])
    for k, vs in values.items():
        c.executemany("insert into counters_collection values (?, ?, ?)", [(k, counter, v) for v in vs])
    c.commit()
    c.close()


def test_traffic_table_and_stats_on_a_synthetic_trace(tmp_path):
    from tools import pmc_table, pmc_traffic, rocpd_stats
    dur = {IG: [1000000] * 4, AF: [200000] * 2, AM: [1000] * 2}           # two forwards: 2 convs + 1 apply + 1 arg-max each
    f, w = str(tmp_path / "f.db"), str(tmp_path / "w.db")
    _db(f, "FETCH_SIZE", {IG: [1024.0] * 4, AF: [512.0] * 2, AM: [0.0] * 2}, dur)   # KiB per launch
    _db(w, "WRITE_SIZE", {IG: [512.0] * 4, AF: [1024.0] * 2, AM: [0.0] * 2}, dur)
    out = str(tmp_path / "t.json")
    pmc_traffic.main(f, w, out, "x3f16", 32)
    t = json.load(open(out))
    assert t["launches"] == 4                                               # the conv / linear family only
    assert t["fetch_bytes_per_launch_raw"] == 1024 * 1024 and t["fetch_bytes_per_launch_corrected_x2"] == 2 * 1024 * 1024
    assert t["hbm_bytes_per_launch"] == 2 * 1024 * 1024 + 512 * 1024
    wf = t["whole_forward"]
    assert wf["forwards_in_trace"] == 2 and wf["kernel_launches_per_forward"] == 4
    per_fwd = (2 * (4 * 1024 + 2 * 512) + (4 * 512 + 2 * 1024)) * 1024 / 2
    assert wf["hbm_bytes_per_forward"] == per_fwd and abs(wf["hbm_gb_per_window"] - per_fwd / 32 / 1e9) < 1e-12
    assert t["precision"] == "x3f16" and t["windows_per_forward"] == 32 and t["lib_sha16"] == pmc_traffic.source_sha16()

    tab = str(tmp_path / "tab.json")
    pmc_table.main(tab, f, w)
    rows = {r["kernel"]: r for r in json.load(open(tab))["kernels"]}
    assert rows[IG]["launches"] == 4 and abs(rows[IG]["hbm_gb_per_launch"] - (2 * 1024 + 512) * 1024 / 1e9) < 1e-4
    assert abs(rows[IG]["hbm_tbs"] - (2 * 1024 + 512) * 1024 * 4 / 4e6 / 1e3) < 1e-3   # bytes / ns -> TB/s

    stats = str(tmp_path / "s.csv")
    rocpd_stats.main(f, stats, rocpd_stats.windows_from_trace(f, 32))
    rd = list(csv.reader(open(stats)))
    assert rd[0][:3] == ["kernel", "calls", "total_us"] and rd[1][0] == IG and int(rd[1][1]) == 4 and float(rd[1][2]) == 4000.0
    assert rd[-1][0] == "TOTAL" and float(rd[-1][2]) == 4402.0 and abs(float(rd[-1][-1]) - 4402.0 / 64) < 0.06
    # the traced conv / linear family per window (what bench.py quotes as roofline.traced): 4 launches of 1 ms over 64 windows
    fam = str(tmp_path / "fam.json")
    rocpd_stats.family_summary(f, fam, 64.0, 32, "x3f16")
    fj = json.load(open(fam))
    assert fj["igemm_family_ms_per_window"] == round(4.0 / 64, 4) and fj["all_kernels_ms_per_window"] == round(4.402 / 64, 4)
    assert fj["lib_sha16"] == pmc_traffic.source_sha16() and fj["precision"] == "x3f16" and fj["windows_per_forward"] == 32
    assert list(fj["kernels_us_per_window"]) == [IG[:120]]


def test_mfma_utilisation_on_a_synthetic_trace(tmp_path):
    from tools import pmc_mfma
    dur = {IG: [1000000] * 2}                                               # 2 ms at 2 cycles / ns = 2.0 GHz
    a, b = str(tmp_path / "a.db"), str(tmp_path / "b.db")
    _db(a, "SQ_VALU_MFMA_BUSY_CYCLES", {IG: [1024.0 * 2.0e6 * 0.5 / 2] * 2}, dur)   # half of the SIMD-cycles of each launch
    _db(b, "GRBM_GUI_ACTIVE", {IG: [8 * 2.0e6] * 2}, dur)                   # summed over the 8 XCDs
    out = str(tmp_path / "m.json")
    pmc_mfma.main(a, b, out)
    m = json.load(open(out))
    r = m["kernels"][0]
    assert abs(r["effective_clock_ghz"] - 2.0) < 1e-9 and abs(r["mfma_busy_fraction"] - 0.25) < 1e-4
    assert abs(m["time_weighted_clock_ghz"] - 2.0) < 1e-9


def test_binary_stamp_and_source_sha(tmp_path):
    """the key bench.py compares before quoting a measurement file is the stamp of the BINARY: pgtformer_amd/build.py compiles the
    sha256 over csrc / include into pgt_version() (\"... src:<sha16>\") and rebuilds by content hash, so a library built from these
    sources carries exactly build.source_sha16(); the sha changes with any source and with nothing else"""
    import ctypes
    from pgtformer_amd import build
    from tools import pmc_traffic
    lib = build.build(verbose=False)
    sha = build.source_sha16()
    assert len(sha) == 16 and build.binary_sha16(lib) == sha == pmc_traffic.source_sha16()
    import torch  # noqa: F401  (one shared HIP runtime before the .so is loaded)
    h = ctypes.CDLL(lib)
    h.pgt_version.restype = ctypes.c_char_p
    assert h.pgt_version().decode().endswith("src:" + sha)
    # the source hash follows the sources only
    csrc, inc = build.CSRC, build.INCLUDE
    try:
        build.CSRC, build.INCLUDE = str(tmp_path / "csrc"), str(tmp_path / "include")
        import shutil
        shutil.copytree(inc, build.INCLUDE)
        os.makedirs(build.CSRC)
        for fn in ("common.h", "misc.hip"):
            shutil.copy(os.path.join(csrc, fn), os.path.join(build.CSRC, fn))
        a = build.source_sha16()
        (tmp_path / "README.md").write_text("x")
        assert build.source_sha16() == a != sha
        with open(os.path.join(build.CSRC, "misc.hip"), "a") as f:
            f.write("\n// edit\n")
        assert build.source_sha16() != a
    finally:
        build.CSRC, build.INCLUDE = csrc, inc

