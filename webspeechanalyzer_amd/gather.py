"""Collecting the per-GPU feature matrices on rank 0.

Clips are independent units, so ranks process disjoint shards with no exchange during compute; the
only collective is this gather at the end: row counts first (tiny), then ONE gather of the padded rows,
each row = its 8 int32 of metadata (bit-cast to 4 float64 slots) followed by the 53 float64 features.
On GPUs the backend is "nccl" (= RCCL: each peer sends over its own xGMI link to the root); the same code runs under "gloo" on CPU tensors,
which is how the N > 1 path is tested without GPUs."""
import torch
import torch.distributed as dist


def gather_rows(meta, feat, n_rows, clip_base, dst=0, group=None):
    """meta [cap, 8] int32, feat [cap, 53] float64 (device of the process group's backend), the first
    n_rows valid.  clip_base = global index of this rank's first clip (added to meta[:, 0]).
    Returns (meta_all [N, 8], feat_all [N, 53]) on rank `dst` in (rank, clip, si) order, else (None, None)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = feat.device
    cnt = torch.tensor([n_rows], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c) for c in torch.cat(cnts).tolist()]                       # one device -> host sync
    m = max(max(counts), 1)
    packed = torch.zeros((m, 4 + 53), dtype=torch.float64, device=dev)
    own_meta = meta[:n_rows].clone()
    own_meta[:, 0] += int(clip_base)
    packed[:n_rows, :4] = own_meta.contiguous().view(torch.float64)          # bit-cast, no conversion
    packed[:n_rows, 4:] = feat[:n_rows]
    bufs = [torch.empty_like(packed) for _ in range(world)] if rank == dst else None
    dist.gather(packed, bufs, dst=dst, group=group)
    if rank != dst:
        return None, None
    rows = torch.cat([bufs[r][:counts[r]] for r in range(world)], dim=0)
    return rows[:, :4].contiguous().view(torch.int32), rows[:, 4:].contiguous()


def shard_range(n_clips, rank, world):
    """contiguous block partition of clips over ranks (first `n_clips % world` ranks get one more)."""
    q, r = divmod(n_clips, world)
    a = rank * q + min(rank, r)
    return a, a + q + (1 if rank < r else 0)
