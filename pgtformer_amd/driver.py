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


class WindowRunner:
    """Runs the model on uint8 windows, `batch` independent 3-frame windows per forward; optional HIP-graph
    replay (static shapes)."""

    def __init__(self, model, w=1.0, use_graph=True, height=512, width=512, batch=1):
        self.model, self.w = model, w
        self.dev = model.dev
        self.t = model.t
        self.batch = batch
        self.static_in = torch.zeros((batch * self.t, height, width, 3), dtype=torch.uint8, device=self.dev)
        self.graph = None
        self.static_out = None
        # per-shape kernel selection during the first eager passes (bf16 launches only); PGT_AUTOTUNE=0 keeps the
        # library's static heuristic, PGT_AUTOTUNE_CACHE=<file> reloads / stores the tuned table across processes
        cache = os.environ.get("PGT_AUTOTUNE_CACHE")
        tune = os.environ.get("PGT_AUTOTUNE", "1") != "0" and self.dev.type == "cuda"
        if tune:
            from . import ops
            if cache and os.path.exists(cache):
                ops.load_autotune(cache)
            else:
                ops.enable_autotune()
        if use_graph:
            self._capture()
        elif tune:
            self.model.restore_middle_u8(self.static_in, w=self.w)
            torch.cuda.synchronize(self.dev)
        if tune and cache and not os.path.exists(cache):
            ops.save_autotune(cache)

    def _capture(self):
        s = torch.cuda.Stream(device=self.dev)
        s.wait_stream(torch.cuda.current_stream(self.dev))
        with torch.cuda.stream(s):
            for _ in range(2):  # warm-up: allocator pools, lazy module loading
                self.model.restore_middle_u8(self.static_in, w=self.w)
        torch.cuda.current_stream(self.dev).wait_stream(s)
        torch.cuda.synchronize(self.dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.static_out = self.model.restore_middle_u8(self.static_in, w=self.w)

    def run(self, windows_u8):
        """windows_u8: (batch*3,H,W,3) uint8 device tensor (batch windows back to back) -> restored middle
        frames (batch,H,W,3) uint8 ((H,W,3) when batch == 1).  Overwritten by the next call when graphs are on."""
        if self.graph is None:
            return self.model.restore_middle_u8(windows_u8, w=self.w)
        self.static_in.copy_(windows_u8, non_blocking=True)
        self.graph.replay()
        return self.static_out

    def run_clip(self, padded, out):
        """padded: (n+2,H,W,3) u8 = [prev halo, n frames, next halo]; fills out (n,H,W,3) with the restored
        frames, `batch` windows per forward (the tail batch is padded with repeats of the last window)."""
        n, b, t = out.shape[0], self.batch, self.t
        offs = torch.arange(t, device=padded.device)
        for j in range(0, n, b):
            idx = torch.arange(j, j + b, device=padded.device).clamp_(max=n - 1)
            wins = padded[(idx[:, None] + offs[None, :]).reshape(-1)]        # (b*3,H,W,3) gather of u8 frames
            res = self.run(wins)
            k = min(b, n - j)
            out[j:j + k].copy_(res.reshape(b, *res.shape[-3:])[:k])
        return out


def restore_clip(runner, frames_u8, rank=0, world=1, group=None, gather=True):
    """frames_u8: this rank's OWN output-range frames (n_local,H,W,3) uint8 (host or device).
    Returns restored frames: all of them on rank 0 if `gather`, else this rank's range."""
    local = frames_u8.to(runner.dev, non_blocking=True)
    n_local = local.shape[0]
    padded = parallel.padded_local_clip(local, rank, world, group)   # one all_gather of boundary frames
    out = runner.run_clip(padded, torch.empty_like(local))
    if world > 1 and gather:
        n_total = torch.tensor([n_local], device=runner.dev)
        torch.distributed.all_reduce(n_total, group=group)
        return parallel.gather_outputs(out, int(n_total.item()), rank, world, 0, group)
    return out


# ---- frame I/O (raw rgb24 files, or ffmpeg pipes with the reference's arguments) -----------------
def read_frames(path, width, height):
    if path.endswith(".rgb"):
        raw = np.fromfile(path, np.uint8)
        return raw.reshape(-1, height, width, 3)
    ff = shutil.which("ffmpeg")
    if ff is None:
        raise RuntimeError("decoding %s needs an ffmpeg binary on PATH (or pass a raw .rgb file)" % path)
    cmd = [ff, "-i", path, "-f", "image2pipe", "-pix_fmt", "rgb24", "-vcodec", "rawvideo", "-"]
    raw = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, check=True).stdout
    return np.frombuffer(raw, np.uint8).reshape(-1, height, width, 3)


def write_frames(path, frames, fps=30):
    frames = np.ascontiguousarray(frames)
    if path.endswith(".rgb"):
        frames.tofile(path)
        return
    ff = shutil.which("ffmpeg")
    if ff is None:
        raise RuntimeError("encoding %s needs an ffmpeg binary on PATH (or write a raw .rgb file)" % path)
    h, w = frames.shape[1:3]
    cmd = [ff, "-y", "-f", "rawvideo", "-pix_fmt", "rgb24", "-s", f"{w}x{h}", "-r", str(fps), "-i", "-", "-an",
           "-vcodec", "libx265", "-crf", "18", "-tag:v", "hvc1", path]
    subprocess.run(cmd, input=frames.tobytes(), stderr=subprocess.DEVNULL, check=True)


def load_architecture(precision="bf16", weights=None, device="cuda", seed=0):
    """Counterpart of inference.py:109-121.  `weights`: a .safetensors / .pth (`params_ema` | `params` |
    flat state dict) checkpoint of the reference model; None -> deterministic synthetic weights."""
    from . import PGTFormer, default_config
    from .manifest import pgtformer_manifest
    from .weightgen import generate_state_dict

    cfg = default_config()
    model = PGTFormer(**cfg)
    if weights is None:
        sd = generate_state_dict(pgtformer_manifest(cfg), cfg, seed=seed)
    elif weights.endswith(".safetensors"):
        from safetensors.torch import load_file
        sd = load_file(weights)
    else:
        sd = torch.load(weights, map_location="cpu")
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
    ap.add_argument("--fps", type=int, default=30)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "mixed", "fp32"])
    ap.add_argument("--weights", default=None)
    ap.add_argument("--batch", type=int, default=16, help="independent windows per forward (<= 20: 2 GiB tensor limit)")
    args = ap.parse_args(argv)
    frames = read_frames(args.input_video, args.size, args.size)
    model = load_architecture(args.precision, args.weights)
    runner = WindowRunner(model, 1.0, True, args.size, args.size, batch=args.batch)
    out = restore_clip(runner, torch.from_numpy(np.ascontiguousarray(frames)))
    write_frames(args.output_video, out.cpu().numpy(), args.fps)


if __name__ == "__main__":
    main()
