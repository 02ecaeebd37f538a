"""Window-sharded MSM over the GPUs of one node (one process per GPU, torch.distributed).

The path's only exchange step: every rank accumulates a contiguous range of Pippenger
windows (or negabase digit positions) over ALL points and reduces each of them to its window sum
S_w (one XYZZ point, 128 B).  One all-gather of
those byte records over RCCL/xGMI (backend "nccl"; "gloo" on CPU for the tests) gives every
rank all windows, and each rank runs the same host Horner tail.  Elliptic-curve addition is
not an RCCL reduction operator, hence all-gather + local combine rather than (all-)reduce;
the payload is 2 KiB for 16 windows, so the step is latency-bound and the xGMI link bandwidth
is immaterial.

The reference has no distributed code; this module is new (it replaces the role Rayon's
chunking plays inside best_multiexp with window ownership across GPUs).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import numpy as np


def window_range(num_windows: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced split: rank r owns [W*r//G, W*(r+1)//G)."""
    return num_windows * rank // world, num_windows * (rank + 1) // world


def all_gather_records(local: np.ndarray, num_windows: int, rec_bytes: int, world: int, rank: int,
                       device=None, group=None) -> np.ndarray:
    """All-gather the per-window byte records; returns the (num_windows * rec_bytes,) array in
    window order on every rank.  `local` holds this rank's windows (window_range order)."""
    import torch
    import torch.distributed as dist

    w0, w1 = window_range(num_windows, world, rank)
    local = np.ascontiguousarray(local, np.uint8).reshape(-1)
    assert local.size == (w1 - w0) * rec_bytes, (local.size, w0, w1, rec_bytes)
    if world == 1:
        return local.copy()
    max_w = max(window_range(num_windows, world, r)[1] - window_range(num_windows, world, r)[0] for r in range(world))
    buf = np.zeros(max_w * rec_bytes, np.uint8)
    buf[: local.size] = local
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * max_w * rec_bytes, dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    allb = out.cpu().numpy().reshape(world, max_w * rec_bytes)
    parts: List[np.ndarray] = []
    for r in range(world):
        a, b = window_range(num_windows, world, r)
        parts.append(allb[r, : (b - a) * rec_bytes])
    return np.concatenate(parts) if parts else np.zeros(0, np.uint8)


def sharded_msm(ctx, curve, d_scalars: int, d_points: int, n: int, world: int, rank: int, device=None, group=None) -> np.ndarray:
    """One n-point MSM over `world` ranks; every rank holds all scalars and points (replicated in
    its own HBM) and returns the same Jacobian result."""
    W, rec = ctx.msm_plan(curve, n)
    w0, w1 = window_range(W, world, rank)
    local = ctx.msm_partial_device(curve, d_scalars, d_points, n, w0, w1)
    allrec = all_gather_records(local, W, rec, world, rank, device, group)
    return ctx.msm_combine(curve, n, allrec)


def sharded_lhs_msm(ctx, curve, d_scalars: int, d_points_affine: int, n: int, base: int, world: int, rank: int,
                    device=None, group=None):
    """compute_lhs_witness MSM core sharded by negabase digit position."""
    d, rec = ctx.lhs_plan(curve, base)
    p0, p1 = window_range(d, world, rank)
    local = ctx.lhs_partial_device(curve, d_scalars, d_points_affine, n, base, p0, p1)
    allrec = all_gather_records(local, d, rec, world, rank, device, group)
    return ctx.lhs_combine(curve, base, allrec)


def point_range(n: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous balanced split of the pairs: rank r owns [n*r//G, n*(r+1)//G)."""
    return n * rank // world, n * (rank + 1) // world


def sharded_msm_by_points(ctx, curve, d_scalars: int, d_points: int, n: int, world: int, rank: int, device=None,
                          group=None) -> np.ndarray:
    """The alternative partition: rank r runs the whole pipeline (all windows) over
    its own n/G pairs -- `d_scalars` / `d_points` address the FULL arrays here, a rank with only its
    slice resident passes slice pointers and (n, world=1 ranges) itself -- and the ranks all-gather
    one 96-byte Jacobian partial each, which every rank sums on the host.  This is the structure of
    halo2's best_multiexp itself (chunks of pairs per thread, partial results folded).  Nothing is
    replicated (the digit pass and the point conversion shard too), so a rank's device time at
    8 GPUs is ~10 % lower than with window sharding (profiles/r01/o_shard_rehearsal_one_gpu.txt);
    window sharding stays the default because the north star prescribes it."""
    a, b = point_range(n, world, rank)
    local = ctx.msm_device(curve, d_scalars + a * 32, d_points + a * 64, b - a) if b > a else np.zeros(12, np.uint64)
    from .api import jacobian_sum
    if world == 1:
        return local
    allj = all_gather_fixed(np.ascontiguousarray(local, np.uint64).view(np.uint8), world, device, group)
    return jacobian_sum(curve, allj.view(np.uint64).reshape(world, 12))


def all_gather_fixed(local_bytes: np.ndarray, world: int, device=None, group=None) -> np.ndarray:
    """All-gather of one fixed-size byte record per rank; returns (world, len) uint8."""
    import torch
    import torch.distributed as dist

    t = torch.from_numpy(np.ascontiguousarray(local_bytes, np.uint8).reshape(-1).copy())
    if device is not None:
        t = t.to(device)
    out = torch.empty(world * t.numel(), dtype=torch.uint8, device=t.device)
    dist.all_gather_into_tensor(out, t, group=group)
    return out.cpu().numpy().reshape(world, -1)
