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


_HALO_BUFS = {}      # (frame shape, device, world) -> (send buffer, [receive buffers]): the exchange allocates once, not per clip


def _force_collective():
    """PGT_FORCE_COLLECTIVE=1 with an initialised process group: the collectives run even in a 1-rank world (the -m gpu test that
    puts RCCL and the configs[2] workload into the single-GPU test run; bench.py under torchrun --nproc-per-node 1)"""
    import os
    return os.environ.get("PGT_FORCE_COLLECTIVE") == "1" and dist.is_available() and dist.is_initialized()


def exchange_halo(local_frames, rank, world, group=None, n_total=None):
    """local_frames: (n_local, H, W, 3) uint8 - the frames of this rank's own output range only (n_local may be 0 when
    the clip has fewer frames than ranks: such ranks still take part in the collective).  Returns (prev_halo, next_halo):
    the last frame of the nearest non-empty previous rank and the first frame of the nearest non-empty next rank,
    replicate-padded at the clip ends (the reference driver's first/last-frame duplication, inference.py:38-74);
    (None, None) on an empty rank.  ONE all_gather per rank: [first frame | last frame | n_local as 8 bytes].
    n_total: the clip's frame count when the caller knows it - which ranks are empty then follows from frame_range on the HOST
    and nothing is read back from the device (without it the gathered counts make one device -> host round trip)."""
    assert local_frames.dim() == 4 and local_frames.dtype == torch.uint8
    n_local = local_frames.shape[0]
    if world == 1 and not _force_collective():
        return (local_frames[0], local_frames[-1]) if n_local else (None, None)
    shape = tuple(local_frames.shape[1:])
    fbytes = shape[0] * shape[1] * shape[2]
    dev = local_frames.device
    staged = _staged(local_frames, group) if local_frames.is_cuda else False
    key = (shape, str(dev), world, staged)
    if key not in _HALO_BUFS:
        recv_dev = "cpu" if staged else dev
        _HALO_BUFS[key] = (torch.zeros(2 * fbytes + 8, dtype=torch.uint8, device=dev),
                           [torch.empty(2 * fbytes + 8, dtype=torch.uint8, device=recv_dev) for _ in range(world)])
    mine, recv = _HALO_BUFS[key]
    if n_local:
        mine[:fbytes].copy_(local_frames[0].reshape(-1))
        mine[fbytes:2 * fbytes].copy_(local_frames[-1].reshape(-1))
    mine[2 * fbytes:].copy_(torch.tensor([n_local], dtype=torch.int64).view(torch.uint8), non_blocking=True)
    if staged:
        dist.all_gather(recv, mine.cpu(), group=group)
        gathered = [h.to(dev) for h in recv]
    else:
        dist.all_gather(recv, mine, group=group)
        gathered = recv
    if n_local == 0:
        return None, None
    if n_total is not None:
        counts = [e - s for s, e in (frame_range(n_total, r, world) for r in range(world))]
    else:
        counts = torch.stack([g[2 * fbytes:] for g in gathered]).cpu().view(torch.int64).reshape(-1).tolist()
    prev_halo, next_halo = local_frames[0], local_frames[-1]
    for r in range(rank - 1, -1, -1):
        if counts[r]:
            prev_halo = gathered[r][fbytes:2 * fbytes].reshape(shape).clone()      # (the receive buffers are reused by the next call)
            break
    for r in range(rank + 1, world):
        if counts[r]:
            next_halo = gathered[r][:fbytes].reshape(shape).clone()
            break
    return prev_halo, next_halo


def padded_local_clip(local_frames, rank, world, group=None, n_total=None):
    """(n_local+2, H, W, 3): [prev halo, own frames..., next halo]; window for local output j is
    rows j, j+1, j+2."""
    prev_halo, next_halo = exchange_halo(local_frames, rank, world, group, n_total)
    if prev_halo is None:
        return local_frames      # empty rank: nothing to restore
    return torch.cat([prev_halo.unsqueeze(0), local_frames, next_halo.unsqueeze(0)], 0)


def gather_outputs(local_out, n_frames, rank, world, dst=0, group=None):
    """Collect restored frames (n_local,H,W,3) uint8 on `dst` in clip order (ranges may differ by one)."""
    if world == 1 and not _force_collective():
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
