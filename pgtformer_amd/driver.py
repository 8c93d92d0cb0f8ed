"""Clip driver with the semantics of the reference's inference.py (:6-19 u8<->float conversion,
:21-80 sliding 3-frame window with first/last-frame replication, raw rgb24 frame protocol), built for
GPUs: frames cross PCIe as uint8, the /255 normalisation and the clamp*255-truncate are device kernels,
the whole window forward is replayed from a HIP graph, and a clip is sharded by output-frame range over
the ranks of one node (pgtformer_amd.parallel).

CLI:  python -m pgtformer_amd.driver -i in.{rgb|mp4} -o out.{rgb|mp4} [--size 512] [--precision bf16]
      (mp4 needs an `ffmpeg` binary on PATH; .rgb is raw rgb24, W*H*3 bytes per frame)
"""
import argparse
import os
import shutil
import subprocess

import numpy as np
import torch

from . import parallel
from .config import DEFAULT_PRECISION


class WindowRunner:
    """Runs the model on uint8 clips, `batch` 3-frame windows per forward; optional HIP-graph replay (static shapes).

    overlap=True (default): a forward takes the batch + 2 CONSECUTIVE frames its `batch` sliding windows cover and the
    model computes everything per-frame (BiSeNet, the encoder up to its first temporal attention) once per frame
    (PGTFormer.forward_nhwc(win=...)); overlap=False stacks the windows' 3 frames each (batch*3 frames, the reference's
    per-window recomputation).  The model computes the outer two frames of a window only as far as the middle frame depends
    on them (up to the decoder's last temporal operation): the driver keeps `[0][1]` (reference inference.py:15)."""

    def __init__(self, model, w=1.0, use_graph=True, height=512, width=512, batch=1, overlap=True, full_tail=False, lanes=1,
                 check_range=None):
        self.model, self.w = model, w
        # range telemetry of the IEEE-half modes: the first real batch also runs once eagerly with every operator counting
        # the outputs that sit at the half saturation limit (PGTFormer.check_range); a saturating layer raises instead of
        # producing silently clamped frames.  check_range=None: on for the precisions that store halves, PGT_RANGE_CHECK=0 off.
        if check_range is None:
            check_range = os.environ.get("PGT_RANGE_CHECK", "1") != "0" and getattr(model, "precision", "") in ("x3f16", "bf16x3")
        self._range_pending = bool(check_range) and model.dev.type == "cuda"
        self._range_bad = None         # the saturating layers of a failed check: every later launch keeps raising
        # PGT_RANGE_CHECK_EVERY=N: re-check on every N-th launch of lane 0 (default 0: the first batch only - a clip whose later
        # frames drive the decoder out of range is then not noticed; the check costs one eager forward)
        self._range_every = int(os.environ.get("PGT_RANGE_CHECK_EVERY", "0"))
        self._launches = 0
        self.full_tail = full_tail     # True: all 3 frames of every window through the decoder's per-frame tail (discarded)
        self.dev = model.dev
        self.t = model.t
        self.batch = batch
        self.overlap = overlap
        # lanes > 1: that many forwards in flight - one HIP graph, one pair of static buffers and one stream per lane;
        # run_clip deals the batches of a clip round-robin.  The kernels of two forwards fill each other's tails and pair
        # HBM-bound passes with MFMA-bound convs (measured +6 % at 2 lanes, +1 % more at 3).
        self.lanes = max(1, int(lanes)) if (use_graph and self.dev.type == "cuda") else 1
        n_in = batch + self.t - 1 if overlap else batch * self.t
        self.static_ins = [torch.zeros((n_in, height, width, 3), dtype=torch.uint8, device=self.dev) for _ in range(self.lanes)]
        self.static_outs = [torch.zeros((batch, height, width, 3), dtype=torch.uint8, device=self.dev) for _ in range(self.lanes)]
        self.static_in, self.static_out = self.static_ins[0], self.static_outs[0]
        self.win = None
        if overlap:
            self.win = (torch.arange(batch, dtype=torch.int32)[:, None] + torch.arange(self.t, dtype=torch.int32)[None, :]
                        ).reshape(-1).to(self.dev)
        from . import ops as _ops
        self._lane_ids = [_ops.new_lane_id() for _ in range(self.lanes)]      # arrival-counter sets no other runner / model shares
        self.graphs, self.static_ress = [None] * self.lanes, [None] * self.lanes
        self.graph = None
        self._pipe = None
        self._lane_streams = None
        # Kernel selection is DETERMINISTIC by default (the library's static per-shape heuristic): two processes produce
        # bit-identical frames, as the reference does.  PGT_AUTOTUNE=1 opts in to timing-based per-shape selection during the
        # first eager passes (bf16 launches only: the variants change fp32 summation orders, so runs then differ at the
        # decoder's rounding level); PGT_AUTOTUNE_CACHE=<file> reloads / stores the tuned table across processes
        cache = os.environ.get("PGT_AUTOTUNE_CACHE")
        tune = os.environ.get("PGT_AUTOTUNE", "0") == "1" and self.dev.type == "cuda"
        self._tune_fresh = False
        if tune:
            from . import ops
            if not (cache and os.path.exists(cache) and ops.load_autotune(cache)):
                ops.enable_autotune()
                self._tune_fresh = True
        if use_graph:
            for lane in range(self.lanes):
                self._capture(lane)
            self.graph, self.static_res = self.graphs[0], self.static_ress[0]
        elif tune:
            self._forward(self.static_in)
            torch.cuda.synchronize(self.dev)
        if tune and cache and self._tune_fresh:       # no file yet, or a stale one (other library version / key layout): (re)written
            ops.save_autotune(cache)

    def _forward(self, frames_u8, lane=0):
        from . import ops
        kw = {"win": self.win} if self.overlap else {}
        if self.full_tail:
            kw["full_tail"] = True
        keep, ops.LANE = ops.LANE, self._lane_ids[lane]    # concurrent forwards (lanes, other runners): own arrival counters (ops.frame_bias)
        try:
            return self.model.restore_middle_u8(frames_u8, w=self.w, out=self.static_outs[lane], **kw)
        finally:
            ops.LANE = keep

    def _capture(self, lane=0):
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for _ in range(2 if lane == 0 else 1):  # warm-up: allocator pools, lazy module loading, kernel autotune
                self._forward(self.static_ins[lane], lane)
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            res = self._forward(self.static_ins[lane], lane)
        self.graphs[lane], self.static_ress[lane] = g, res

    def _launch(self, lane=0):
        """one forward on the frames currently in the lane's static input -> (batch,H,W,3) uint8 (overwritten by the next
        launch of that lane); runs on the current stream."""
        if self._range_bad is not None:
            self._raise_range()
        self._launches += lane == 0
        if self._range_pending or (self._range_every > 0 and lane == 0 and self._launches % self._range_every == 0 and self._launches > 1):
            kw = {"win": self.win} if self.overlap else {}
            bad = self.model.check_range(self.static_ins[lane], w=self.w, full_tail=self.full_tail, **kw)
            # the eager pass's activations (one more set on top of the graph pools) are free for re-use by the allocator
            torch.cuda.synchronize(self.dev)
            # (returning the eager pass's blocks to the driver here is opt-in.  The graph replay that once died right after such a
            #  release had another cause - ROCm's hipGraphLaunch on multi-root graphs, archs/pgtformer_arch.py: no fork under capture)
            if os.environ.get("PGT_EMPTY_CACHE_AFTER_CHECK", "0") == "1":
                torch.cuda.empty_cache()
            if bad:
                self._range_bad = bad          # stays set: a caller that catches the error and calls again is refused again
                self._raise_range()
            self._range_pending = False        # only after a clean pass
        if self.graphs[lane] is None:
            res = self._forward(self.static_ins[lane], lane)
        else:
            self.graphs[lane].replay()
            res = self.static_ress[lane]
        return res.reshape(self.batch, *res.shape[-3:])

    def _raise_range(self):
        from .hip import PgtError
        raise PgtError("activations leave the IEEE-half range in precision %r (stores saturate at +-65504): %s ... - "
                       "prepare the model with precision='bf16x3' (bf16 decoder, no range limit) or 'fp32'"
                       % (self.model.precision, self._range_bad[:4]))

    def run(self, frames_u8):
        """frames_u8: uint8 device tensor, (batch+2,H,W,3) consecutive frames (overlap) or (batch*3,H,W,3) windows back to
        back -> restored middle frames (batch,H,W,3) uint8 ((H,W,3) when batch == 1).  Overwritten by the next call."""
        self.static_in.copy_(frames_u8, non_blocking=True)
        res = self._launch()
        return res[0] if self.batch == 1 else res

    def _fill_static(self, padded, j, dst):
        """copy the input frames of windows j .. j+batch-1 of the padded clip (device or pinned host) into dst (the
        static input or a staging buffer of its shape).  Windows past the clip end (ragged tail batch) read whatever
        dst held before; their outputs are discarded."""
        n_pad, t, b = padded.shape[0], self.t, self.batch
        if self.overlap:
            k = min(b + t - 1, n_pad - j)
            dst[:k].copy_(padded[j:j + k], non_blocking=True)
        else:
            k = min(b, n_pad - (t - 1) - j)
            for i in range(k):
                dst[i * t:(i + 1) * t].copy_(padded[j + i:j + i + t], non_blocking=True)

    def _streams(self):
        if self._lane_streams is None:
            self._lane_streams = [torch.cuda.Stream(device=self.dev) for _ in range(self.lanes)]
        return self._lane_streams

    def run_clip(self, padded, out):
        """padded: (n+2,H,W,3) u8 = [prev halo, n frames, next halo]; fills out (n,H,W,3) with the restored
        frames, `batch` windows per forward (the windows of a ragged tail batch past the clip end are discarded).
        A host-resident (pinned) clip is streamed: see _run_clip_pipelined."""
        n, b = out.shape[0], self.batch
        if padded.device.type == "cpu" and self.dev.type == "cuda":
            return self._run_clip_pipelined(padded, out)
        if self.lanes == 1:
            for j in range(0, n, b):
                k = min(b, n - j)
                self._fill_static(padded, j, self.static_in)
                out[j:j + k].copy_(self._launch()[:k])
            return out
        ms = torch.cuda.current_stream(self.dev)
        ls = self._streams()
        for s in ls:
            s.wait_stream(ms)
        for i, j in enumerate(range(0, n, b)):
            k, lane = min(b, n - j), i % self.lanes
            with torch.cuda.stream(ls[lane]):
                self._fill_static(padded, j, self.static_ins[lane])
                out[j:j + k].copy_(self._launch(lane)[:k])
        for s in ls:
            ms.wait_stream(s)
        return out

    def _run_clip_pipelined(self, padded_host, out_host):
        """Host-resident (pinned) clip: uint8 frames cross PCIe in batch-sized chunks on a copy stream, staged on both
        sides, overlapped with the forwards of the other batches (the reference syncs every frame: .cuda() ... .cpu(),
        inference.py:13-17).  Batch i runs on lane i % lanes; with one lane the two staging slots alternate."""
        n, b = out_host.shape[0], self.batch
        dev = self.dev
        L = self.lanes
        S = L + 1                          # staging slots (slot i % S serves batch i): the copies run one batch ahead of the lanes
        if self._pipe is None:
            self._pipe = {"cs": torch.cuda.Stream(device=dev),
                          "in": [torch.empty_like(self.static_in) for _ in range(S)],
                          "out": [torch.empty_like(self.static_out) for _ in range(S)]}
        cs, stin, stout = self._pipe["cs"], self._pipe["in"], self._pipe["out"]
        ms = torch.cuda.current_stream(dev)
        ls = self._streams() if L > 1 else [ms]
        ev_in = [torch.cuda.Event() for _ in range(S)]        # H2D of a batch landed in stin[i]
        ev_used = [torch.cuda.Event() for _ in range(S)]      # the compute stream consumed stin[i]
        ev_out = [torch.cuda.Event() for _ in range(S)]       # results of a batch are in stout[i]
        ev_sent = [torch.cuda.Event() for _ in range(S)]      # D2H of stout[i] finished
        starts = list(range(0, n, b))

        def h2d(i):
            with torch.cuda.stream(cs):
                if i >= S:
                    cs.wait_event(ev_used[i % S])
                self._fill_static(padded_host, starts[i], stin[i % S])
                ev_in[i % S].record(cs)

        cs.wait_stream(ms)
        if L > 1:
            for s in ls:
                s.wait_stream(ms)
        for i in range(min(S - 1, len(starts))):
            h2d(i)
        for i, j in enumerate(starts):
            if i + S - 1 < len(starts):
                h2d(i + S - 1)
            k, lane, slot = min(b, n - j), i % L, i % S
            st = ls[lane]
            with torch.cuda.stream(st):
                st.wait_event(ev_in[slot])
                self.static_ins[lane].copy_(stin[slot], non_blocking=True)
                ev_used[slot].record(st)
                res = self._launch(lane)
                if i >= S:
                    st.wait_event(ev_sent[slot])
                stout[slot].copy_(res, non_blocking=True)
                ev_out[slot].record(st)
            with torch.cuda.stream(cs):
                cs.wait_event(ev_out[slot])
                out_host[j:j + k].copy_(stout[slot][:k], non_blocking=True)
                ev_sent[slot].record(cs)
        if L > 1:
            for s in ls:
                ms.wait_stream(s)
        ms.wait_stream(cs)
        return out_host


def restore_clip(runner, frames_u8, rank=0, world=1, group=None, gather=True, n_total=None):
    """frames_u8: this rank's OWN output-range frames (n_local,H,W,3) uint8 (host or device), ranges as
    parallel.frame_range(n_total, rank, world).  Returns restored frames: all of them on rank 0 if `gather`, else this rank's
    range.  n_total: the clip's frame count when the caller knows it (otherwise the ranks' counts are summed first)."""
    local = frames_u8.to(runner.dev, non_blocking=True)
    n_local = local.shape[0]
    padded = parallel.padded_local_clip(local, rank, world, group, n_total)   # one all_gather of boundary frames
    out = runner.run_clip(padded, torch.empty_like(local)) if n_local else torch.empty_like(local)
    if (world > 1 or parallel._force_collective()) and gather:
        if n_total is None:
            cnt = torch.tensor([n_local], device=runner.dev)
            torch.distributed.all_reduce(cnt, group=group)
            n_total = int(cnt.item())
        return parallel.gather_outputs(out, n_total, rank, world, 0, group)
    return out


def restore_clip_host(runner, padded_host, out_host, rank=0, world=1, group=None, n_total=None):
    """Streaming form for host-resident clips.  padded_host: pinned uint8 (n_local+2,H,W,3) whose rows 1..n_local hold this
    rank's own frames (rows 0 and -1 are filled here with the halo frames: one all_gather of boundary frames on the
    device, replicate padding at the clip ends); out_host: pinned uint8 (n_local,H,W,3).  H2D / forward / D2H are
    pipelined per batch (WindowRunner._run_clip_pipelined).  Returns out_host with its last D2H copies possibly still in
    flight on the runner's copy stream: synchronise (torch.cuda.synchronize(), or an event on that stream) before reading."""
    n_local = padded_host.shape[0] - 2
    dev = runner.dev
    if n_local <= 0:
        # a rank without frames (clip shorter than the world) still joins the collective - with an EMPTY tensor, so that its
        # neighbours skip it (parallel.exchange_halo) instead of taking uninitialised pinned memory as their halos
        parallel.exchange_halo(torch.empty((0,) + tuple(padded_host.shape[1:]), dtype=torch.uint8, device=dev), rank, world, group)
        return out_host
    edge = torch.stack([padded_host[1], padded_host[n_local]]).to(dev, non_blocking=True)    # first / last own frame
    prev_halo, next_halo = parallel.exchange_halo(edge, rank, world, group, n_total)     # (n_total known: no device -> host read)
    padded_host[0].copy_(prev_halo, non_blocking=True)
    padded_host[n_local + 1].copy_(next_halo, non_blocking=True)
    torch.cuda.current_stream(dev).synchronize()     # the two halo frames are on the host before the pipeline reads them
    return runner.run_clip(padded_host, out_host)


def restore_stream(runner, chunks, sink, segment=256):
    """Bounded-memory form for clips of any length: `chunks` yields uint8 (k,H,W,3) arrays (iter_frames), `sink` receives
    the restored frames in order as uint8 numpy arrays.  The clip is cut into segments of `segment` frames; a segment's
    halos are the last frame of the previous segment and the first frame after it (replicated at the clip ends, the reference
    driver's first / last frame duplication, inference.py:38-74), so the result equals one pass over the whole clip.  Two
    pinned staging buffers of segment + 2 and segment frames are all the host memory held.  Returns the frame count."""
    it = iter(chunks)
    buf, nbuf, prev, total, eof = [], 0, None, 0, False
    pin_in = pin_out = None
    cuda = runner.dev.type == "cuda"
    while True:
        while not eof and nbuf < segment + 1:
            try:
                c = next(it)
            except StopIteration:
                eof = True
                break
            if c.shape[0]:
                buf.append(np.asarray(c))
                nbuf += c.shape[0]
        if nbuf == 0:
            break
        frames = buf[0] if len(buf) == 1 else np.concatenate(buf, 0)
        take = nbuf if eof else segment          # not at the end: >= segment + 1 frames are held, the extra one is the halo
        seg, rest = frames[:take], frames[take:]
        if pin_in is None or pin_in.shape[0] < take + 2:
            shape = tuple(frames.shape[1:])
            pin_in = torch.empty((max(take, segment) + 2,) + shape, dtype=torch.uint8)
            pin_out = torch.empty((max(take, segment),) + shape, dtype=torch.uint8)
            if cuda:
                pin_in, pin_out = pin_in.pin_memory(), pin_out.pin_memory()
        pin_in[0].copy_(torch.from_numpy(np.ascontiguousarray(prev if prev is not None else seg[0])))
        pin_in[1:take + 1].copy_(torch.from_numpy(np.ascontiguousarray(seg)))
        pin_in[take + 1].copy_(torch.from_numpy(np.ascontiguousarray(rest[0] if rest.shape[0] else seg[-1])))
        runner.run_clip(pin_in[:take + 2], pin_out[:take])
        if cuda:
            torch.cuda.synchronize(runner.dev)
        sink(pin_out[:take].numpy())
        prev = np.array(seg[-1])
        buf, nbuf = ([rest], rest.shape[0]) if rest.shape[0] else ([], 0)
        total += take
        if eof and nbuf == 0:
            break
    return total


# ---- frame I/O (raw rgb24 files, or ffmpeg pipes with the reference's arguments) -----------------
def probe_video(path):
    """(width, height, fps) of a video file through ffprobe (the reference probes with cv2.VideoCapture,
    inference.py:148-152)."""
    fp = shutil.which("ffprobe")
    if fp is None:
        raise RuntimeError("probing %s needs ffprobe on PATH" % path)
    out = subprocess.run([fp, "-v", "error", "-select_streams", "v:0", "-show_entries", "stream=width,height,r_frame_rate",
                          "-of", "csv=p=0", path], stdout=subprocess.PIPE, check=True).stdout.decode().strip()
    w, h, rate = out.split(",")[:3]
    num, _, den = rate.partition("/")
    return int(w), int(h), float(num) / float(den or 1)


def iter_frames(path, width, height, chunk=32):
    """Yield uint8 (k,H,W,3) chunks of a clip without buffering it: raw .rgb files are memory-mapped, anything else is
    decoded by an ffmpeg pipe with the reference's arguments (inference.py:23-27), W*H*3 bytes per frame."""
    fbytes = width * height * 3
    if path.endswith(".rgb"):
        size = os.path.getsize(path)
        if size % fbytes:
            raise ValueError(f"{path}: {size} bytes is not a whole number of {width}x{height} rgb24 frames")
        raw = np.memmap(path, np.uint8, "r").reshape(-1, height, width, 3)
        for i in range(0, raw.shape[0], chunk):
            yield np.ascontiguousarray(raw[i:i + chunk])
        return
    ff = shutil.which("ffmpeg")
    if ff is None:
        raise RuntimeError("decoding %s needs an ffmpeg binary on PATH (or pass a raw .rgb file)" % path)
    pw, ph, _ = probe_video(path)
    if (pw, ph) != (width, height):
        raise ValueError(f"{path} is {pw}x{ph}; the model takes {width}x{height} frames (resize the clip first)")
    cmd = [ff, "-i", path, "-f", "image2pipe", "-pix_fmt", "rgb24", "-vcodec", "rawvideo", "-"]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, bufsize=fbytes * 4)
    done, tail = False, b""
    try:
        while True:
            buf = tail + proc.stdout.read(fbytes * chunk)
            if len(buf) < fbytes:
                tail = buf
                break
            k = len(buf) // fbytes
            tail = buf[k * fbytes:]
            yield np.frombuffer(buf[:k * fbytes], np.uint8).reshape(k, height, width, 3)
        done = True
    finally:
        proc.stdout.close()
        rc = proc.wait()
    # a decoder that failed or a stream cut inside a frame must not pass for a short clip
    if done and rc != 0:
        raise RuntimeError(f"ffmpeg exited with status {rc} while decoding {path}")
    if done and tail:
        raise RuntimeError(f"{path}: the decoded stream ends {len(tail)} bytes into a {width}x{height} rgb24 frame")


def read_frames(path, width, height):
    chunks = list(iter_frames(path, width, height))
    if not chunks:
        return np.zeros((0, height, width, 3), np.uint8)
    return np.concatenate(chunks, 0)


class FrameWriter:
    """Streaming sink: raw .rgb file or an ffmpeg encoder pipe with the reference's arguments (libx265, crf 18, hvc1;
    inference.py:29-35)."""

    def __init__(self, path, width, height, fps=30):
        self.proc, self.f = None, None
        if path.endswith(".rgb"):
            self.f = open(path, "wb")
            return
        ff = shutil.which("ffmpeg")
        if ff is None:
            raise RuntimeError("encoding %s needs an ffmpeg binary on PATH (or write a raw .rgb file)" % path)
        cmd = [ff, "-y", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{width}x{height}", "-r", str(fps), "-i", "-", "-an",
               "-vcodec", "libx265", "-crf", "18", "-tag:v", "hvc1", path]
        self.proc = subprocess.Popen(cmd, stdin=subprocess.PIPE, stderr=subprocess.DEVNULL)
        self.f = self.proc.stdin

    def write(self, frames):
        self.f.write(np.ascontiguousarray(frames).tobytes())

    def close(self):
        self.f.close()
        if self.proc is not None and self.proc.wait() != 0:
            raise RuntimeError("ffmpeg encoder failed")


def write_frames(path, frames, fps=30):
    frames = np.ascontiguousarray(frames)
    wr = FrameWriter(path, frames.shape[2], frames.shape[1], fps)
    wr.write(frames)
    wr.close()


def load_architecture(precision=DEFAULT_PRECISION, weights=None, device="cuda", seed=0, synthetic=False):
    """Counterpart of inference.py:109-121.  `weights`: a directory with config.json + model.safetensors (the layout of
    `PGTFormer.from_pretrained`), a hub id, or a .safetensors / .pth (`params_ema` | `params` | flat state dict)
    checkpoint of the reference model.  Deterministic synthetic weights only on explicit request (`synthetic=True`)."""
    from . import PGTFormer, default_config
    from .manifest import pgtformer_manifest
    from .weightgen import generate_state_dict

    if weights is None:
        if not synthetic:
            raise ValueError("no checkpoint given: pass weights=<dir | hub id | .safetensors | .pth>, or synthetic=True for "
                             "random-init weights (benchmarks / tests only - the output is not a restoration)")
        cfg = default_config()
        model = PGTFormer(**cfg)
        model.load_state_dict(generate_state_dict(pgtformer_manifest(cfg), cfg, seed=seed), strict=True)
        return model.prepare(device, precision)
    if os.path.isdir(weights) or not weights.endswith((".safetensors", ".pth", ".pt", ".ckpt")):
        return PGTFormer.from_pretrained(weights, device=device, precision=precision)     # a directory or a hub id ("org/name-v1.5")
    cfg = default_config()
    model = PGTFormer(**cfg)
    if weights.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(weights)
    else:
        sd = torch.load(weights, map_location="cpu", weights_only=True)
        for key in ("params_ema", "params"):
            if isinstance(sd, dict) and key in sd:
                sd = sd[key]
                break
    model.load_state_dict(sd, strict=True)
    return model.prepare(device, precision)


def main(argv=None):
    ap = argparse.ArgumentParser(description="PGTFormer blind video face restoration (MI355X)")
    ap.add_argument("-i", "--input_video", default="assets/inputdemovideo.mp4")
    ap.add_argument("-o", "--output_video", default="exp/output_demo.mp4")
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--fps", type=float, default=None, help="output frame rate (default: probed from the input, else 30)")
    ap.add_argument("--precision", default=DEFAULT_PRECISION, choices=["x3f16", "bf16x3", "bf16", "mixed", "fp32"])
    ap.add_argument("--weights", default=None, help="checkpoint: directory (config.json + model.safetensors), hub id, "
                                                    ".safetensors or .pth")
    ap.add_argument("--synthetic", action="store_true", help="random-init weights (smoke tests only)")
    ap.add_argument("--batch", type=int, default=32, help="sliding windows per forward")
    ap.add_argument("--lanes", type=int, default=2, help="forwards in flight (HIP graphs on separate streams)")
    ap.add_argument("--segment", type=int, default=256, help="frames held in host memory at a time (multiple of --batch)")
    args = ap.parse_args(argv)
    if args.weights is None and not args.synthetic:
        ap.error("--weights is required (the reference downloads kepeng/pgtformer-base; no network here). "
                 "Use --synthetic only to exercise the pipeline with random weights.")
    fps = args.fps
    if fps is None:
        try:
            fps = probe_video(args.input_video)[2] if not args.input_video.endswith(".rgb") else 30
        except Exception:
            fps = 30
    model = load_architecture(args.precision, args.weights, synthetic=args.synthetic)
    runner = WindowRunner(model, 1.0, True, args.size, args.size, batch=args.batch, lanes=args.lanes)
    # decode -> restore -> encode in segments: bounded host memory for clips of any length
    os.makedirs(os.path.dirname(os.path.abspath(args.output_video)), exist_ok=True)
    wr = FrameWriter(args.output_video, args.size, args.size, fps)
    n = restore_stream(runner, iter_frames(args.input_video, args.size, args.size), wr.write, segment=args.segment)
    wr.close()
    print(f"{n} frames restored -> {args.output_video}")


if __name__ == "__main__":
    main()
