"""Synthetic degraded clips with the geometry of the reference's inputs.

GT frame = smooth low-frequency field (8 random sinusoids per channel + bicubic-free 64x64 noise
field upsampled bilinearly), translated by a per-frame offset; LQ = GT -> 4x area down-sample to
128^2 -> + N(0, 5/255) -> bilinear up-sample to 512^2 with align_corners=True -> clamp [0,1] -> u8.
This mirrors the LQ synthesis recipe the reference's (training-only) dataloader describes
(reference: data/vfhq_full_dataset.py:838-845, 868-872) without importing it. Pure numpy so the clip
is bit-identical in the build container and on the GPU box.
"""
import numpy as np


def _bilinear_up_align_corners(img, out_h, out_w):
    """img: (h,w,c) float32 -> (out_h,out_w,c), align_corners=True bilinear."""
    h, w, _ = img.shape
    ys = np.arange(out_h, dtype=np.float64) * ((h - 1) / (out_h - 1))
    xs = np.arange(out_w, dtype=np.float64) * ((w - 1) / (out_w - 1))
    y0 = np.minimum(np.floor(ys).astype(np.int64), h - 2)
    x0 = np.minimum(np.floor(xs).astype(np.int64), w - 2)
    fy = (ys - y0).astype(np.float32)[:, None, None]
    fx = (xs - x0).astype(np.float32)[None, :, None]
    a = img[y0][:, x0]
    b = img[y0][:, x0 + 1]
    c = img[y0 + 1][:, x0]
    d = img[y0 + 1][:, x0 + 1]
    return (a * (1 - fy) * (1 - fx) + b * (1 - fy) * fx + c * fy * (1 - fx) + d * fy * fx).astype(np.float32)


def _gt_field(rng, size, n_frames, margin=64):
    big = size + 2 * margin
    yy, xx = np.meshgrid(np.arange(big, dtype=np.float32), np.arange(big, dtype=np.float32), indexing="ij")
    field = np.zeros((big, big, 3), np.float32)
    for ch in range(3):
        acc = np.zeros((big, big), np.float32)
        for _ in range(8):
            fx, fy = rng.uniform(-0.02, 0.02, 2)
            ph = rng.uniform(0, 2 * np.pi)
            amp = rng.uniform(0.05, 0.15)
            acc += amp * np.sin(2 * np.pi * (fx * xx + fy * yy) + ph).astype(np.float32)
        field[:, :, ch] = 0.5 + acc
    noise = rng.uniform(-0.15, 0.15, (big // 8, big // 8, 3)).astype(np.float32)
    field += _bilinear_up_align_corners(noise, big, big)
    return np.clip(field, 0.0, 1.0), margin


def make_clip(n_frames, size=512, seed=1234, start=0):
    """Return (lq_u8 (N,size,size,3) uint8, gt (N,size,size,3) float32 in [0,1]): frames start .. start+N-1 of the clip
    `seed` (a frame depends on the seed and its own index only, so ranks generate their own frame ranges)."""
    rng = np.random.default_rng(seed)
    field, margin = _gt_field(rng, size, n_frames)
    lq = np.empty((n_frames, size, size, 3), np.uint8)
    gt = np.empty((n_frames, size, size, 3), np.float32)
    for j in range(n_frames):
        i = start + j
        # +-2 px inter-frame translation, periodic so long clips stay inside the margin
        dx = int(round(2 * ((i % 32) - 16) * (1 if (i // 32) % 2 == 0 else -1)))
        dy = int(round(2 * (((i * 7) % 32) - 16)))
        y0, x0 = margin + dy, margin + dx
        g = field[y0:y0 + size, x0:x0 + size]
        gt[j] = g
        small = g.reshape(size // 4, 4, size // 4, 4, 3).mean(axis=(1, 3)).astype(np.float32)
        frng = np.random.default_rng(seed * 100003 + i)
        small = small + frng.normal(0.0, 5.0 / 255.0, small.shape).astype(np.float32)
        up = np.clip(_bilinear_up_align_corners(small, size, size), 0.0, 1.0)
        lq[j] = np.floor(up * 255.0 + 0.5).astype(np.uint8)
    return lq, gt


def window_from_clip(lq_u8, i):
    """3-frame window (3,H,W,3) u8 for output frame i, replicate-padded at the clip ends
    (window policy of the reference driver: inference.py:38-74)."""
    n = lq_u8.shape[0]
    idx = [max(i - 1, 0), i, min(i + 1, n - 1)]
    return lq_u8[idx]
