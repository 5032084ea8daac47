"""Multi-GPU split of a batch of independent blocks.

Blocks are independent, so there is no data-path collective: rank g takes a contiguous index range
chosen by a prefix sum of the block sizes (byte-balanced), works on it in its own HBM, and only the
int32 size vector is gathered (RCCL all_gather over xGMI when the process group is NCCL, gloo on
CPU tests).  Pure host logic + torch.distributed; no kernels here."""
from __future__ import annotations

from typing import List, Sequence, Tuple

import numpy as np


def byte_balanced_ranges(lengths: Sequence[int], world_size: int) -> List[Tuple[int, int]]:
    """Contiguous [lo, hi) per rank such that every rank gets ~ total_bytes / world_size.
    Deterministic; ranges cover [0, n) without overlap; empty ranges are allowed."""
    lens = np.asarray(lengths, dtype=np.int64).clip(min=0)
    n = int(lens.size)
    if world_size <= 0:
        raise ValueError("world_size must be positive")
    csum = np.concatenate(([0], np.cumsum(lens)))
    total = int(csum[-1])
    bounds = [0]
    for r in range(1, world_size):
        target = total * r / world_size
        # first index whose cumulative start reaches the target (ties -> earlier), never backwards
        idx = int(np.searchsorted(csum, target, side="left"))
        if idx > 0 and idx <= n and abs(csum[idx - 1] - target) <= abs(csum[min(idx, n)] - target):
            idx -= 1
        idx = min(max(idx, bounds[-1]), n)
        bounds.append(idx)
    bounds.append(n)
    return [(bounds[i], bounds[i + 1]) for i in range(world_size)]


def gather_sizes(local_sizes, ranges: Sequence[Tuple[int, int]], group=None):
    """all_gather the per-rank int32 size slices into the full size vector (torch tensors).
    `local_sizes` is this rank's slice (on the device the backend needs: cuda for nccl)."""
    import torch
    import torch.distributed as dist

    world = dist.get_world_size(group)
    counts = [hi - lo for lo, hi in ranges]
    maxc = max(counts) if counts else 0
    if dist.get_backend(group) == "gloo":
        local_sizes = local_sizes.cpu()            # (two ranks sharing one GPU run their size exchange over gloo)
    pad = torch.zeros(maxc, dtype=torch.int32, device=local_sizes.device)
    pad[:local_sizes.numel()] = local_sizes
    parts = [torch.empty_like(pad) for _ in range(world)]
    dist.all_gather(parts, pad, group=group)
    return torch.cat([p[:c] for p, c in zip(parts, counts)]) if counts else pad


def sharded_pickle_roundtrip(backend, lens_all, rank: int, world: int, group=None):
    """ONE batch over `world` ranks (BASELINE.json configs[3]): rank r takes the byte-balanced range r of the
    batch's messages, pickles and unpickles it in its own memory, and the only thing exchanged is the int32 vector of
    envelope sizes (all_gather).  `backend.pickle_unpickle(data, off, lens)` does the per-rank work and returns
    (envelope sizes as an int32 torch tensor on the backend's device, pickle seconds, unpickle seconds, round trip ok).
    Returns (ranges, gathered sizes for the WHOLE batch, this rank's timings dict)."""
    import numpy as np
    from . import corpus
    ranges = byte_balanced_ranges(lens_all, world)
    lo, hi = ranges[rank]
    data, off, lens = corpus.config4_share(np.asarray(lens_all), lo, hi)
    sizes, t_p, t_u, ok = backend.pickle_unpickle(data, off, lens)
    full = gather_sizes(sizes, ranges, group)
    mine = {"messages": int(hi - lo), "bytes": int(lens.astype(np.int64).sum()), "pickle_s": t_p, "unpickle_s": t_u,
            "roundtrip_ok": bool(ok), "longest_bytes": int(lens.max()) if lens.size else 0, "longest_s": 0.0}
    if lens.size and hasattr(backend, "time_alone"):
        # The range's longest messages on their own: with one wavefront per message the slowest of them is the floor of the
        # range's time.  Content alternates with the global index (random bytes go fast, text does not), so the longest of
        # either parity is timed.
        idx = np.arange(lo, hi)
        for parity in (0, 1):
            sel = np.flatnonzero((idx & 1) == parity)
            if sel.size:
                i = int(sel[np.argmax(lens[sel])])
                mine["longest_s"] = max(mine["longest_s"], float(backend.time_alone(data[int(off[i]):int(off[i]) + int(lens[i])])))
    return ranges, full, mine
