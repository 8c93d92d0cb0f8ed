"""Clip-level parallelism: frame-range sharding across the GPUs of one node.

Each output frame i depends only on input frames {i-1, i, i+1} (replicate at the clip ends;
reference: inference.py:38-74) and the model keeps no state between windows, so ranks own contiguous
output-frame ranges and run model replicas.  The only exchange is the 1-frame halo at range
boundaries: ONE all-gather of every rank's first and last frame (2 x H*W*3 bytes per rank; RCCL over
xGMI on GPUs, gloo on CPU tests).  There is no all-reduce anywhere on the path.
"""
import torch
import torch.distributed as dist


def _staged(t, group):
    """gloo moves host tensors only: on a single-GPU test rig (several ranks on one device, backend gloo) device tensors are
    staged through host memory for the collective; on RCCL (`nccl`) they go as they are."""
    return t.is_cuda and dist.get_backend(group) == "gloo"


def frame_range(n_frames, rank, world):
    """Contiguous, balanced output-frame range [start, end) of `rank` (ranges differ by at most 1)."""
    base, rem = divmod(n_frames, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def needed_inputs(n_frames, rank, world):
    """Input-frame indices (clamped) rank needs: own range plus one halo frame on each side."""
    s, e = frame_range(n_frames, rank, world)
    if s == e:
        return []
    return list(range(max(s - 1, 0), min(e + 1, n_frames)))


def exchange_halo(local_frames, rank, world, group=None):
    """local_frames: (n_local, H, W, 3) uint8 - the frames of this rank's own output range only (n_local may be 0 when
    the clip has fewer frames than ranks: such ranks still take part in the collective).  Returns (prev_halo, next_halo):
    the last frame of the nearest non-empty previous rank and the first frame of the nearest non-empty next rank,
    replicate-padded at the clip ends (the reference driver's first/last-frame duplication, inference.py:38-74);
    (None, None) on an empty rank.  ONE all_gather per rank: [first frame | last frame | n_local as 8 bytes]."""
    assert local_frames.dim() == 4 and local_frames.dtype == torch.uint8
    n_local = local_frames.shape[0]
    if world == 1:
        return (local_frames[0], local_frames[-1]) if n_local else (None, None)
    shape = tuple(local_frames.shape[1:])
    fbytes = shape[0] * shape[1] * shape[2]
    mine = torch.zeros(2 * fbytes + 8, dtype=torch.uint8, device=local_frames.device)
    if n_local:
        mine[:fbytes] = local_frames[0].reshape(-1)
        mine[fbytes:2 * fbytes] = local_frames[-1].reshape(-1)
    mine[2 * fbytes:] = torch.tensor([n_local], dtype=torch.int64).view(torch.uint8).to(local_frames.device)
    if _staged(mine, group):
        host = [torch.empty_like(mine, device="cpu") for _ in range(world)]
        dist.all_gather(host, mine.cpu(), group=group)
        gathered = [h.to(mine.device) for h in host]
    else:
        gathered = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(gathered, mine, group=group)
    if n_local == 0:
        return None, None
    counts = torch.stack([g[2 * fbytes:] for g in gathered]).cpu().view(torch.int64).reshape(-1).tolist()
    prev_halo, next_halo = local_frames[0], local_frames[-1]
    for r in range(rank - 1, -1, -1):
        if counts[r]:
            prev_halo = gathered[r][fbytes:2 * fbytes].reshape(shape)
            break
    for r in range(rank + 1, world):
        if counts[r]:
            next_halo = gathered[r][:fbytes].reshape(shape)
            break
    return prev_halo, next_halo


def padded_local_clip(local_frames, rank, world, group=None):
    """(n_local+2, H, W, 3): [prev halo, own frames..., next halo]; window for local output j is
    rows j, j+1, j+2."""
    prev_halo, next_halo = exchange_halo(local_frames, rank, world, group)
    if prev_halo is None:
        return local_frames      # empty rank: nothing to restore
    return torch.cat([prev_halo.unsqueeze(0), local_frames, next_halo.unsqueeze(0)], 0)


def gather_outputs(local_out, n_frames, rank, world, dst=0, group=None):
    """Collect restored frames (n_local,H,W,3) uint8 on `dst` in clip order (ranges may differ by one)."""
    if world == 1:
        return local_out
    sizes = [frame_range(n_frames, r, world) for r in range(world)]
    maxn = max(e - s for s, e in sizes)
    pad = torch.zeros((maxn,) + tuple(local_out.shape[1:]), dtype=local_out.dtype, device=local_out.device)
    pad[:local_out.shape[0]] = local_out
    if _staged(pad, group):
        host = [torch.empty_like(pad, device="cpu") for _ in range(world)] if rank == dst else None
        dist.gather(pad.cpu(), host, dst=dst, group=group)
        bufs = [h.to(pad.device) for h in host] if rank == dst else None
    else:
        bufs = [torch.empty_like(pad) for _ in range(world)] if rank == dst else None
        dist.gather(pad, bufs, dst=dst, group=group)
    if rank != dst:
        return None
    return torch.cat([bufs[r][: sizes[r][1] - sizes[r][0]] for r in range(world)], 0)
