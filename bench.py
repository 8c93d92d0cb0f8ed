#!/usr/bin/env python
"""Headline benchmark: restored 512x512 frames/s of the PGTFormer forward path on MI355X.

One "step" = one forward of B = `--windows-per-forward` consecutive sliding windows = B restored frames
(reference driver semantics, inference.py:12-19, 38-74: every output frame is the middle frame of a 3-frame
window; the reference itself only accepts one window per call).  The B windows of a step cover B+2
consecutive frames: everything per-frame (BiSeNet, the encoder up to its first temporal attention) is
computed once per frame and gathered to window order (results equal B separate forwards), and after the
decoder's last temporal operation (the 256x256 fusion block's temporal mix) only the middle frame of every
window - the one the driver keeps - is computed (`--full-tail`: all three, as the reference computes and discards).
Workload = BASELINE.json configs[1]: pgtformer-base, synthetic degraded 512x512 clip, 3-frame window,
16-bit MFMA arithmetic with fp32 accumulation (default precision "x3f16": decoder / fusion in IEEE half, the
code-prediction branch on split-half operands so that the codes equal the fp32 reference's - the mode that holds the
1e-3 dB PSNR contract, tests/test_gpu_model.py::test_psnr_contract_at_the_operating_point), random-init weights of the
exact architecture (no checkpoint / network here).

`value`: the uint8 clip is resident in HBM when the timed region starts; the region covers the halo exchange and
every forward (HIP-graph replay, uint8 in -> uint8 restored frames out).  `value_from_pinned_host` is the same job with
the clip in PINNED HOST memory (configs[1]: "u8 on host"): H2D and D2H copies inside the timed region, double-buffered
on a copy stream (driver.restore_clip_host); `--resident` skips that second measurement.

N>1: one process per GPU (torchrun), the clip is sharded by output-frame range, ranks exchange the
1-frame halos with ONE all_gather (RCCL over xGMI) inside the timed region; weak scaling (each rank
restores `steps` x B frames).  value = frames restored by all ranks / max-over-ranks time.

Prints ONE JSON line (rank 0) with the `roofline` (dominant kernel: the MFMA implicit-GEMM conv,
measured live with events on the launch stream in a separate instrumented eager pass) and
`cpu_baseline` (the CPU oracle = a port of the reference's fp32 eager path, timed on this host).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

PEAK_TFLOPS = {"x3f16": 2500.0, "bf16x3": 2500.0, "bf16": 2500.0, "mixed": 2500.0, "fp32": 157.3}   # MI355X_MICROARCH.md dense MFMA peaks (bf16 = f16)
PEAK_HBM_GBS = 8000.0


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--precision", default="x3f16", choices=["x3f16", "bf16x3", "bf16", "mixed", "fp32"],
                    help="x3f16 (default): the mode that holds the 1e-3 dB PSNR contract (split-half code branch, IEEE-half decoder)")
    ap.add_argument("--resident", action="store_true", help="clip resident in HBM (no H2D/D2H in the timed region)")
    ap.add_argument("--no-overlap", action="store_true", help="stack 3 frames per window (no per-frame reuse)")
    ap.add_argument("--full-tail", action="store_true",
                    help="push all 3 frames of every window through the decoder's per-frame tail, as the reference does before "
                         "discarding two of them (default: middle frames only after the last temporal operation)")
    ap.add_argument("--windows-per-forward", type=int, default=32,
                    help="independent 3-frame windows batched into one forward (reference semantics: B separate calls)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--lanes", type=int, default=2,
                    help="forwards in flight: HIP graphs of one step each on separate streams (1 = one step at a time)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = min(32, physical cores))")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary figures of the line: value_full_tail (the reference's discarded decoder work) and micro "
                         "(BASELINE.json configs 4 / 5 points)")
    ap.add_argument("--dump-clip", default="",
                    help="--clip-frames mode, rank 0: write the sha256 of the restored clip (uint8 frames in order) to this file")
    ap.add_argument("--plumbing-only", action="store_true",
                    help="launcher / rendezvous check without a model (runs on a host without a GPU under PGT_DIST_BACKEND=gloo): "
                         "every rank joins the process group, the timing collectives run, rank 0 prints the world it saw")
    ap.add_argument("--clip-frames", type=int, default=0,
                    help="BASELINE.json configs[2]: ONE synthetic clip of this many frames (256 in the config), sharded by "
                         "output-frame range over the --gpus ranks with one all_gather of boundary frames, restored frames gathered "
                         "to rank 0; time = first H2D -> last D2H; a step = one pass over the clip (strong scaling)")
    return ap.parse_args()


def _lib_sha16():
    """identity of the kernel build: the source sha256 compiled INTO the loaded libpgt_hip.so (pgt_version() ends in
    src:<sha16>; pgtformer_amd/build.py) - the stamp of the binary that runs, whatever sources lie next to it"""
    from pgtformer_amd import hip
    v = hip.lib().pgt_version().decode()
    return v.rsplit("src:", 1)[1] if "src:" in v else "unstamped"


def live_roofline(runner, frames, precision, nwin):
    """Instrumented eager pass of ONE step (the runner's own forward: same frames, same window index): every kernel
    launch of pgtformer_amd.ops bracketed by events on its launch stream (one stream: a bracketed launch runs alone).
    Returns the roofline object of the dominant family (the implicit-GEMM convs / linears) with `kernels`: one entry per
    kernel family - launches, ms per step, the roofline that bounds it, achieved rate and fraction of the gfx950 peak."""
    from pgtformer_amd import ops
    from pgtformer_amd.archs import pgtformer_arch
    side, pgtformer_arch.SIDE_STREAM = pgtformer_arch.SIDE_STREAM, False
    runner.static_in.copy_(frames)
    runner._forward(runner.static_in)      # warm
    torch.cuda.synchronize()
    recs = []
    ops.PROFILE = recs
    try:
        runner._forward(runner.static_in)
        torch.cuda.synchronize()
    finally:
        ops.PROFILE = None
        pgtformer_arch.SIDE_STREAM = side
    for r in recs:
        r["ms"] = r["events"][0].elapsed_time(r["events"][1])
    peak = PEAK_TFLOPS[precision]

    def family(name, sel, bound, note=None):
        rs = [r for r in recs if sel(r)]
        if not rs:
            return None
        ms, fl, by = sum(r["ms"] for r in rs), sum(r["flops"] for r in rs), sum(r["bytes"] for r in rs)
        pk = (157.3 if name.endswith("fp32") else 2500.0) if bound == "mfma" else PEAK_HBM_GBS
        ach = fl / (ms * 1e-3) / 1e12 if bound == "mfma" else by / (ms * 1e-3) / 1e9
        e = {"name": name, "launches": len(rs), "ms_per_step": round(ms, 3), "bound": bound, "achieved": round(ach, 2),
             "peak": pk, "unit": "TFLOP/s" if bound == "mfma" else "GB/s", "frac": round(ach / pk, 4),
             "algorithmic_gb_per_step": round(by / 1e9, 3)}
        if bound == "mfma":
            e["hbm_gbs"] = round(by / (ms * 1e-3) / 1e9, 1)
        if note:
            e["note"] = note
        return e

    conv = lambda r: r["kernel"] == "igemm"   # noqa: E731
    kernels = [
        family("igemm 16-bit (bf16 / f16 operands: igemm_kernel, igemm4/5, conv3x3_c64)", lambda r: conv(r) and not r["x3"] and r["dt"] != "float32", "mfma"),
        family("igemm split-half", lambda r: conv(r) and r["x3"], "mfma",
               "algorithmic FLOPs (every reference product once); the launches execute 3 f16 MFMAs per product"),
        family("fused token-row chains (rowchain.hip: LN -> q|k|v, proj -> LN -> Mlp, LN -> Mlp; also counted in the two igemm rows above)",
               lambda r: conv(r) and r.get("chain"), "mfma",
               "algorithmic FLOPs and bytes of the fused launches (rows in, rows out, weights): the LayerNorm / GELU / residual passes they absorb have none"),
        family("igemm exact fp32", lambda r: conv(r) and not r["x3"] and r["dt"] == "float32", "mfma",
               "3/8-input-channel first convs, fused-upsample and 19-channel BiSeNet heads on v_mfma_f32_32x32x2_f32"),
        family("mha_mfma (code transformer, L = 3072 per window)", lambda r: r["kernel"] == "mha", "mfma",
               "algorithmic FLOPs; split-half operands: 3 MFMAs per product"),
        family("window_attn_mfma", lambda r: r["kernel"] == "window_attention", "hbm",
               "reads the qkv rows once, writes the output rows once"),
        family("layernorm", lambda r: r["kernel"] == "layernorm", "hbm"),
        family("groupnorm apply + SiLU / AdaIN apply (affine_act)", lambda r: r["kernel"] == "norm_apply_act", "hbm"),
        family("groupnorm statistics pass", lambda r: r["kernel"] == "groupnorm_stats", "hbm",
               "only the GroupNorms whose statistics do not come out of the producing conv's epilogue"),
        family("fp32 <-> split-half / half conversions", lambda r: r["kernel"] == "x3_convert", "hbm"),
        family("gathers / copies / pad zeroing", lambda r: r["kernel"] == "copy_gather", "hbm"),
        family("weight-rounding compensation (sampled channel means + per-frame bias)", lambda r: r["kernel"] == "mean_field", "hbm",
               "small launches: 4096 sampled pixels per frame, a (K x Cout) matrix-vector product per frame"),
    ]
    kernels = [k for k in kernels if k]
    ig = [r for r in recs if conv(r)]
    t_ms, flops, byts, n = sum(r["ms"] for r in ig), sum(r["flops"] for r in ig), sum(r["bytes"] for r in ig), len(ig)
    achieved = flops / (t_ms * 1e-3) / 1e12
    top = sorted(ig, key=lambda r: -r["ms"])[:5]
    if os.environ.get("PGT_DUMP_SHAPES"):   # per-shape table of the instrumented pass (tuning aid)
        agg = {}
        for r in ig:
            a = agg.setdefault((r["shape"], r.get("cfg"), r["dt"] + ("x3" if r["x3"] else "")), [0, 0.0, 0.0, 0.0])
            a[0] += 1
            a[1] += r["ms"] * 1e3
            a[2] += r["flops"]
            a[3] += r["bytes"]
        with open(os.environ["PGT_DUMP_SHAPES"], "w") as f:
            f.write("shape(N,H,W,Cin,Cout,k,stride,ups) cfg(kernel,bm,bn) dtype launches total_us avg_us TFLOP/s GB/s\n")
            for (shape, cfg, dt), (cnt, us, fl, by) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
                f.write(f"{shape} {cfg} {dt} {cnt} {us:.1f} {us / cnt:.1f} {fl / us / 1e6:.1f} {by / us / 1e3:.0f}\n")
    if os.environ.get("PGT_DUMP_OPS"):      # every non-conv launch of the instrumented pass: kernel, ms, algorithmic GB/s
        with open(os.environ["PGT_DUMP_OPS"], "w") as f:
            f.write("kernel us MB GB/s\n")
            for r in recs:
                if not conv(r):
                    f.write(f"{r['kernel']} {r['ms'] * 1e3:.1f} {r['bytes'] / 1e6:.1f} {r['bytes'] / (r['ms'] * 1e-3) / 1e9:.0f}\n")
    # HBM traffic of the igemm family from separate rocprofv3 --pmc passes (tools/pmc_traffic.py): only a measurement taken
    # with THIS library build (sha256 of libpgt_hip.so) in this configuration is quoted - a stale file is refused
    traffic, tsrc, wf_pmc = None, None, None
    sha = _lib_sha16()
    for name in sorted(os.listdir(os.path.join(REPO, "profiles")), reverse=True):
        if not name.endswith("igemm_traffic_pmc.json"):
            continue
        tj = json.load(open(os.path.join(REPO, "profiles", name)))
        if tj.get("windows_per_forward") == nwin and tj.get("precision") == precision:
            if tj.get("lib_sha16") == sha:
                traffic, tsrc = round(tj["hbm_bytes_per_launch"] / 1e9, 4), f"profiles/{name} (GB per launch, same library build)"
                wf_pmc = tj.get("whole_forward", {}).get("hbm_gb_per_window")
                break
            tsrc = f"profiles/{name} is stale (library {tj.get('lib_sha16')} != {sha}): not quoted"
    x3 = [r for r in ig if r.get("x3")]
    executed = flops + 2.0 * sum(r["flops"] for r in x3)        # split-half launches issue 3 f16 MFMA products per product
    all_ms, all_by, all_fl = sum(r["ms"] for r in recs), sum(r["bytes"] for r in recs), sum(r["flops"] for r in recs)
    return {"bound": "mfma", "kernel": "igemm family (implicit-GEMM conv/linear: igemm_kernel, igemm4/5, conv3x3_c64, linear_k256, fused token-row chains)",
            "achieved": round(achieved, 2),
            "peak": peak, "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "traffic": traffic, "traffic_source": tsrc,
            "algorithmic_gb_per_launch": round(byts / n / 1e9, 4),
            "windows_per_forward": nwin, "launches_per_forward": n,
            "algorithmic_gflop_per_launch": round(flops / n / 1e9, 3), "avg_launch_us": round(t_ms * 1e3 / n, 2),
            "algorithmic_gb_per_window": round(byts / nwin / 1e9, 3), "igemm_ms_per_window": round(t_ms / nwin, 3),
            "split_bf16_launches": len(x3),
            "executed_mfma_tflops": round(executed / (t_ms * 1e-3) / 1e12, 2),
            "executed_mfma_frac": round(executed / (t_ms * 1e-3) / 1e12 / peak, 4),
            "split_bf16_note": "algorithmic FLOPs count every product once; split-half launches execute 3 MFMAs per product",
            "split_bf16_algorithmic_tflops": round(sum(r["flops"] for r in x3) / max(1e-9, sum(r["ms"] for r in x3) * 1e-3) / 1e12, 2) if x3 else None,
            "kernels": kernels,
            "whole_forward": {"kernel_ms_per_step": round(all_ms, 2), "launches": len(recs),
                              "algorithmic_gb_per_window": round(all_by / nwin / 1e9, 3),
                              "algorithmic_tflop_per_window": round(all_fl / nwin / 1e12, 3),
                              "hbm_gbs_over_the_step": round(all_by / (all_ms * 1e-3) / 1e9, 1),
                              "pmc_hbm_gb_per_window": None if wf_pmc is None else round(wf_pmc, 3)},
            "lib_sha16": sha,
            "slowest_launches": [{"shape_NHWCinCoutKSU": list(r["shape"]), "dtype": r["dt"] + ("x3" if r["x3"] else ""),
                                  "us": round(r["ms"] * 1e3, 1), "tflops": round(r["flops"] / (r["ms"] * 1e-3) / 1e12, 1)}
                                 for r in top]}


def traced_family(precision, nwin, algorithmic_tflop_per_window):
    """The rocprofv3-traced steady-state time of the conv / linear family (tools/rocpd_stats.py family_summary, committed under
    profiles/ by the measurement pass) - quoted only if it was taken with THIS build of the kernels (sha of the sources)."""
    sha = _lib_sha16()
    for name in sorted(os.listdir(os.path.join(REPO, "profiles")), reverse=True):
        if not name.endswith("traced_family.json"):
            continue
        tj = json.load(open(os.path.join(REPO, "profiles", name)))
        if tj.get("windows_per_forward") == nwin and tj.get("precision") == precision and tj.get("lib_sha16") == sha:
            ms = tj["igemm_family_ms_per_window"]
            ach = algorithmic_tflop_per_window / (ms * 1e-3)
            return {"source": f"profiles/{name} (rocprofv3 --kernel-trace of bench.py --lanes 1, HIP-graph replay, same library build)",
                    "igemm_ms_per_window": ms, "achieved": round(ach, 2), "frac": round(ach / PEAK_TFLOPS[precision], 4),
                    "all_kernels_ms_per_window": tj.get("all_kernels_ms_per_window")}
    return None


def micro_points():
    """BASELINE.json configs[3] / configs[4] points, event-timed (tools/bench_micro.py has the full sweeps; the rocprofv3 sweep of
    config 5 is profiles/r5_config5_sweep.csv): nearest-code search at 262144 tokens, window attention 3x8x8 / C = 512 / fp16."""
    from pgtformer_amd import ops
    from tools.bench_micro import timeit
    out = []
    heads, c, win = 8, 512, (3, 8, 8)
    n = win[0] * win[1] * win[2]
    bias = (0.02 * torch.randn((heads, n, n), device="cuda")).float()
    for (d, h, w) in [(3, 64, 64), (3, 128, 128), (6, 128, 128), (3, 256, 256)]:
        nw = (d // win[0]) * (h // win[1]) * (w // win[2])
        qkv = torch.randn((d * h * w, 3 * c), device="cuda").to(torch.float16)
        us = timeit(lambda: ops.window_attention3d(qkv, bias, 1, d, h, w, c, heads, win, (0, 0, 0)), 20)
        byts = d * h * w * c * 2 * 4
        out.append({"config": 5, "bench": "window_attention3d 3x8x8 C=512 fp16", "nW": nw, "us": round(us, 1),
                    "hbm_frac": round(byts / us / 1e3 / PEAK_HBM_GBS, 3)})
    t, h, w, c2 = 3, 128, 128, 256
    qkv = torch.randn((t * h * w, 3 * c2), device="cuda").to(torch.float16)
    b2 = (0.02 * torch.randn((heads, 48, 48), device="cuda")).float()
    us = timeit(lambda: ops.window_attention(qkv, b2, 1, t, h, w, c2, heads, (4, 4), (2, 2)), 20)
    out.append({"config": 5, "bench": "window_attention 3x4x4 C=256 fp16 (shipping shape, shifted)", "nW": 1024, "us": round(us, 1),
                "hbm_frac": round(t * h * w * c2 * 2 * 4 / us / 1e3 / PEAK_HBM_GBS, 3)})
    book = torch.randn((1024, 512), device="cuda")
    enorm = book.pow(2).sum(1).contiguous()
    book_t = book.to(torch.bfloat16).contiguous()
    x = torch.randn((262144, 512), device="cuda").to(torch.bfloat16)
    xn = ops.row_sumsq(x)
    us = timeit(lambda: ops.rq_nearest(x, book_t, xn, enorm), 10)
    out.append({"config": 4, "bench": "rq_nearest (arg-min inside the distance GEMM) bf16", "Ntok": 262144, "us": round(us, 1),
                "mfma_frac": round(2.0 * 262144 * 1024 * 512 / us / 1e6 / 2500.0, 3)})
    return out


def _cpu_model_name():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def _physical_cores():
    try:
        import psutil
        return psutil.cpu_count(logical=False) or os.cpu_count()
    except Exception:
        return os.cpu_count()


def cpu_baseline(cfg, sd, window_u8, budget_s=40.0, threads=0):
    """The oracle (a port of the reference's fp32 eager CPU path) timed on this host as SURVEY 8(d) prescribes: threads =
    physical cores, 2 warm-up windows, then the median of up to 5 timed windows - bounded by `budget_s` of CPU work
    (slow hosts get fewer timed windows; the count is reported)."""
    from oracle import pgt_oracle as O      # reported CPU baseline only (never on the product path)
    # measured on the GPU box's 2 x 64-core EPYC 9575F (tools/cpu_threads_probe.py, profiles/r2_cpu_threads.jsonl): 32 torch
    # threads run the eager fp32 oracle 2.7x faster than all 128 physical cores (9.0 s vs 24.3 s per window): the many
    # small ops of the graph do not scale across two sockets.  The baseline uses the fastest setting found.
    cores = threads or min(32, _physical_cores())
    torch.set_num_threads(cores)
    x = torch.from_numpy(window_u8.astype(np.float32) / np.float32(255.0)).permute(0, 3, 1, 2).contiguous()
    times, warm = [], []
    for i in range(7):
        t0 = time.time()
        O.pgtformer_forward(sd, cfg, x, w=1.0)
        dt = time.time() - t0
        (warm if i < 2 else times).append(dt)
        if sum(times) > budget_s:
            break
    med = float(np.median(times))
    return {"value": round(1.0 / med, 4), "unit": "frames/s", "cores": cores, "kind": "port",
            "sample": f"{len(times)} timed windows (3x512x512 in -> 1 restored frame each) after {len(warm)} warm-ups, median; "
                      "fp32 eager torch-CPU oracle",
            "seconds_per_window": round(med, 2), "warmup_seconds": [round(t, 2) for t in warm],
            "cpu_model": _cpu_model_name(), "host_logical_cpus": os.cpu_count()}


def clip_mode(args, model, dev, rank, world):
    """configs[2]: "pgtformer-base, 8xMI355X, 256-frame synthetic clip sharded by window with xGMI boundary all-gather"
    (SURVEY 8d config 3 / 8e).  Every rank holds only its own frame range (pinned host, uint8) - frame_range shard, the
    synthetic generator is per-frame - and per pass: H2D of its frames, ONE all_gather of first / last frames (the halos),
    its forwards, gather of the restored frames to rank 0, D2H of the whole restored clip on rank 0."""
    from pgtformer_amd import parallel
    from pgtformer_amd.driver import WindowRunner, restore_clip
    from pgtformer_amd.synth import make_clip

    F, B = args.clip_frames, args.windows_per_forward
    s0, e0 = parallel.frame_range(F, rank, world)
    lq_u8, _ = make_clip(e0 - s0, 512, seed=1234, start=s0)
    mine = torch.from_numpy(lq_u8).pin_memory()
    out_host = torch.empty((F, 512, 512, 3), dtype=torch.uint8).pin_memory() if rank == 0 else None
    runner = WindowRunner(model, 1.0, not args.no_graph, 512, 512, batch=min(B, max(1, (e0 - s0 + args.lanes - 1) // args.lanes)), overlap=not args.no_overlap,
                          full_tail=args.full_tail, lanes=args.lanes)

    def one_pass():
        allf = restore_clip(runner, mine, rank, world, gather=True, n_total=F)   # H2D, halo all_gather, forwards, gather to rank 0
        if rank == 0:
            out_host.copy_(allf, non_blocking=True)                       # D2H of the restored clip

    dist_on = torch.distributed.is_available() and torch.distributed.is_initialized()     # (also a 1-rank world under PGT_FORCE_COLLECTIVE=1)

    def fence():
        torch.cuda.synchronize()
        if dist_on:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    for _ in range(max(1, args.warmup)):
        one_pass()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        one_pass()
    fence()
    dt = time.perf_counter() - t0
    if dist_on:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
        dt = float(tmax.item())
    if rank == 0 and args.dump_clip:
        import hashlib
        with open(args.dump_clip, "w") as f:
            json.dump({"frames": F, "sha256": hashlib.sha256(out_host.numpy().tobytes()).hexdigest(), "world": world,
                       "collectives": "RCCL" if (torch.distributed.is_available() and torch.distributed.is_initialized()
                                                 and torch.distributed.get_backend() == "nccl") else "none"}, f)
    return {"metric": "restored 512x512 frames/sec", "value": round(F * args.steps / dt, 3), "unit": "frames/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": DTYPE_NAMES[args.precision], "data": "synthetic",
            "config": {"workload": f"pgtformer-base, ONE {F}-frame synthetic degraded 512x512 clip sharded by output-frame range over "
                                   f"{world} GPU(s) ({e0 - s0} frames on rank 0), 1 all_gather of boundary frames, restored frames "
                                   "gathered to rank 0 (BASELINE.json configs[2]); a step = one pass over the clip, timed from the "
                                   "first H2D to the last D2H",
                       "precision": args.precision, "clip_frames": F, "frames_per_rank": e0 - s0, "windows_per_forward": runner.batch,
                       "hip_graph": not args.no_graph, "steps_in_flight": runner.lanes,
                       "clip_location": "pinned host memory on every rank (H2D / D2H inside the timed region)",
                       "parallelism": f"frame-range shard x{world}, 1 all_gather of boundary frames + 1 gather of restored frames"}}


DTYPE_NAMES = {"x3f16": "f16 / bf16 (16-bit MFMA, fp32 accumulate: IEEE-half decoder, code branch on split-half operands)",
               "bf16x3": "bf16 (bf16 MFMA, fp32 accumulate; code branch on split-half operands)", "bf16": "bf16",
               "mixed": "bf16 (decoder) / f32 (code branch)", "fp32": "f32"}


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args):
    """`python bench.py --gpus N` outside a launcher: re-run this command line as N ranks (one per GPU) under
    `python -m torch.distributed.run` on 127.0.0.1 and hand back its exit code; rank 0's JSON line goes to this stdout."""
    import subprocess
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    return subprocess.call(cmd, env=env)


def plumbing_only(args, world, rank, backend):
    """what every rank does around the timed region, without the model: barrier, max-over-ranks all_reduce, one JSON line"""
    import torch.distributed as dist
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    dist.barrier()
    t = torch.tensor([float(rank + 1)], device=dev, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": "launcher plumbing (no model)", "n_gpus": world, "ranks": dist.get_world_size(),
                          "backend": dist.get_backend(), "max_over_ranks": float(t.item()), "gpus_arg": args.gpus}), flush=True)
    dist.barrier()
    dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and not (args.gpus == 1 and world == 1):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    backend = os.environ.get("PGT_DIST_BACKEND", "nccl")      # "gloo": several ranks on ONE GPU (a test rig for the N > 1 control flow)
    if args.plumbing_only:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
        return plumbing_only(args, world, rank, backend)
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())
    # under torchrun (RANK set) the process group exists even in a 1-rank world when PGT_FORCE_COLLECTIVE=1: the collectives of
    # the path then run through RCCL on ONE GPU (tests/test_gpu_model.py::test_configs2_clip_through_rccl_at_one_gpu)
    dist_on = world > 1 or ("RANK" in os.environ and os.environ.get("PGT_FORCE_COLLECTIVE") == "1")
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local_rank)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)

    from pgtformer_amd import PGTFormer, default_config, parallel
    from pgtformer_amd.driver import WindowRunner, restore_clip_host
    from pgtformer_amd.manifest import pgtformer_manifest
    from pgtformer_amd.synth import make_clip
    from pgtformer_amd.weightgen import generate_state_dict

    cfg = default_config()
    sd = generate_state_dict(pgtformer_manifest(cfg), cfg, seed=0)
    model = PGTFormer(**cfg)
    model.load_state_dict(sd, strict=True)
    model.prepare(dev, args.precision)

    if args.clip_frames:
        res = clip_mode(args, model, dev, rank, world)
        if rank == 0:
            print(json.dumps(res), flush=True)
        if dist_on:
            torch.distributed.barrier()
            torch.distributed.destroy_process_group()
        return

    # this rank's slice of the synthetic clip.  One step = one forward of B windows, so a rank restores steps*B frames
    # (weak scaling: the per-rank clip is fixed as ranks are added).
    B = args.windows_per_forward
    n_local = args.steps * B
    lq_u8, _ = make_clip(min(n_local, 8), 512, seed=1234 + rank)
    reps = (n_local + lq_u8.shape[0] - 1) // lq_u8.shape[0]
    clip = torch.from_numpy(np.concatenate([lq_u8] * reps, 0)[:n_local])
    padded_host = torch.empty((n_local + 2, 512, 512, 3), dtype=torch.uint8).pin_memory()
    padded_host[1:n_local + 1].copy_(clip)
    out_host = torch.empty((n_local, 512, 512, 3), dtype=torch.uint8).pin_memory()
    runner = WindowRunner(model, 1.0, not args.no_graph, 512, 512, batch=B, overlap=not args.no_overlap, full_tail=args.full_tail,
                          lanes=args.lanes)

    n_warm = max(B, min(args.warmup * B, n_local))
    warm_host = torch.empty((n_warm + 2, 512, 512, 3), dtype=torch.uint8).pin_memory()   # own buffer: the halo rows
    warm_host[1:n_warm + 1].copy_(clip[:n_warm])                                          # are written in place

    def one_pass_host(n):
        restore_clip_host(runner, padded_host if n == n_local else warm_host, out_host[:n], rank, world, n_total=n * world)

    local_dev = clip.to(dev)
    out_dev = torch.empty_like(local_dev)

    def one_pass_resident(n):
        padded = parallel.padded_local_clip(local_dev[:n], rank, world, n_total=n * world)
        runner.run_clip(padded, out_dev[:n])

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
            torch.cuda.synchronize()

    def timed(one_pass):
        one_pass(n_warm)
        fence()
        t0 = time.perf_counter()
        one_pass(n_local)
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(tmax, op=torch.distributed.ReduceOp.MAX)
            dt = float(tmax.item())
        return dt

    # `value`: the clip is resident in HBM when the timed region starts (halo exchange + every forward inside it).  The
    # PCIe-inclusive rate of the pinned-host pipeline (H2D / D2H double-buffered on a copy stream) is reported next to it.
    dt = timed(one_pass_resident)
    host_rate = None
    if not args.resident:
        host_rate = round(n_local * world / timed(one_pass_host), 3)

    res = {"metric": "restored 512x512 frames/sec", "value": round(n_local * world / dt, 3), "unit": "frames/s",
           "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
           "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "rccl_ranks": (torch.distributed.get_world_size() if dist_on and backend == "nccl" else (1 if not dist_on else 0)),
           "dist_backend": (torch.distributed.get_backend() if dist_on else "none"),
           "dtype": DTYPE_NAMES[args.precision],
           "data": "synthetic",
           "config": {"workload": "pgtformer-base, 3-frame 512x512 window -> 1 restored frame, synthetic degraded "
                                  "VFHQ-shape clip, random-init weights (BASELINE.json configs[1])",
                      "precision": args.precision, "exact_weight_stages": list(__import__("pgtformer_amd.ops", fromlist=["x"]).EXACT_W_STAGES),
                      "compensated_stages": ("all" if __import__("pgtformer_amd.ops", fromlist=["x"]).WCOMP_STAGES is None
                                             else list(__import__("pgtformer_amd.ops", fromlist=["x"]).WCOMP_STAGES)),
                      "frames_per_step": B, "frames_per_rank": n_local, "hip_graph": not args.no_graph,
                      "windows_per_forward": B, "per_frame_reuse": not args.no_overlap,
                      "steps_in_flight": runner.lanes,
                      "decoder_tail": ("all 3 frames of every window" if args.full_tail else
                                       "middle frame only after the last temporal operation (the driver keeps [0][1], inference.py:15; "
                                       "identical restored frames; --full-tail for the reference's discarded work)"),
                      "clip_location": "HBM (uint8 frames resident when the timed region starts; restored uint8 frames left in HBM)",
                      "value_definition": "measurement contract (4): whole-job throughput with the inputs resident in HBM when the timed "
                                          "region starts (the contract rules the PCIe-inclusive rate out as `value`); the rate of the "
                                          "same job from pinned host memory (configs[1]: u8 on host; H2D / D2H double-buffered inside "
                                          "the timed region) is value_from_pinned_host; value_full_tail: with the two frames per "
                                          "window the reference decodes and discards",
                      "parallelism": f"frame-range shard x{world}, 1 all_gather of boundary frames"}}
    if world > 1 and os.environ.get("PGT_BENCH_CONFIGS2", "1") != "0":
        # BASELINE.json configs[2] as named, in the same run (so that a multi-GPU scaling run records it without extra flags):
        # ONE 256-frame clip sharded by output-frame range, halo all_gather, restored frames gathered to rank 0.  The headline
        # fields above are already final; a failure of this extra pass is recorded, not raised.
        import copy
        a2 = copy.copy(args)
        a2.clip_frames, a2.steps, a2.warmup = 256, 2, 1
        del runner
        torch.cuda.empty_cache()
        try:
            c2 = clip_mode(a2, model, dev, rank, world)
            res["configs2_clip256"] = {"value": c2["value"], "unit": c2["unit"], "ms_per_pass": c2["ms_per_step"], "scaling": "strong",
                                       "workload": c2["config"]["workload"]}
        except Exception as e:      # noqa: BLE001
            res["configs2_clip256"] = {"error": repr(e)[:300]}
    if host_rate is not None:
        res["value_from_pinned_host"] = host_rate      # same job with H2D / D2H of the uint8 frames inside the timed region
    if rank == 0:
        if not args.no_roofline and world == 1:
            nin = runner.static_in.shape[0]
            res["roofline"] = live_roofline(runner, local_dev[:nin] if runner.overlap else torch.cat(
                [local_dev[i:i + 3] for i in range(B)], 0), args.precision, B)
            tr = traced_family(args.precision, B, res["roofline"]["algorithmic_gflop_per_launch"] * res["roofline"]["launches_per_forward"] / B / 1e3)
            if tr is not None:      # the same family in the rocprofv3 trace of the steady state (launches NOT isolated)
                res["roofline"]["traced"] = tr
        if world == 1 and not args.no_extras and not args.full_tail:
            # the reference's forward decodes all three frames of a window and the driver drops two (archs/pgtformer_arch.py:684-712,
            # inference.py:15): the same job with that discarded work computed too
            del runner
            torch.cuda.empty_cache()
            try:
                r2 = WindowRunner(model, 1.0, not args.no_graph, 512, 512, batch=B, overlap=not args.no_overlap, full_tail=True, lanes=args.lanes)
                n_ft = min(n_local, 4 * B)
                pad_ft = parallel.padded_local_clip(local_dev[:n_ft], rank, world, n_total=n_ft)
                r2.run_clip(pad_ft, out_dev[:n_ft])
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                r2.run_clip(pad_ft, out_dev[:n_ft])
                torch.cuda.synchronize()
                res["value_full_tail"] = round(n_ft / (time.perf_counter() - t0), 3)
                del r2
                torch.cuda.empty_cache()
                res["micro"] = micro_points()
            except Exception as e:      # noqa: BLE001  (secondary figures: recorded, not raised)
                res["extras_error"] = repr(e)[:300]
        if not args.no_cpu_baseline and world == 1:
            res["cpu_baseline"] = cpu_baseline(cfg, sd, lq_u8[:3] if lq_u8.shape[0] >= 3 else np.repeat(lq_u8[:1], 3, 0),
                                               threads=args.cpu_threads)
        print(json.dumps(res), flush=True)
    if dist_on:
        torch.distributed.barrier()
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
