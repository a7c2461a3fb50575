"""Collecting the per-GPU feature matrices on rank 0.

Clips are independent units, so ranks process disjoint shards with no exchange during compute; the only
exchange is this collection at the end of a step, as §8e lays it out: the row counts go to the root (one
small gather), then every peer sends exactly its rows — the [rows, 53] float64 feature matrix and the
[rows, 8] int32 metadata in their own dtypes — and the root receives each peer's rows straight into
its slice of the output tables (`batch_isend_irecv`: grouped send / recv, one xGMI link per peer under
RCCL).  Nothing is padded to the largest count, a rank without rows sends nothing, and only the root
reads the counts on the host (it has to size its receives); the senders never synchronise.
At the shard size of BASELINE config 4 (12 500 clips x 10 s per GPU, ~75 k rows) a peer sends ~34 MB per
step (75 k x (424 + 32) B): ~0.24 GB into the root from 7 peers, 0.2 ms per link at 153 GB/s.
On GPUs the backend is "nccl" (= RCCL); the same code runs under "gloo" on CPU tensors, which is how
the N > 1 path is tested without GPUs."""
import torch
import torch.distributed as dist


def gather_rows(meta, feat, n_rows, clip_base, dst=0, group=None, out=None):
    """meta [cap, 8] int32, feat [cap, 53] float64 (device of the process group's backend), the first
    n_rows valid.  clip_base = global index of this rank's first clip (added to meta[:, 0]).
    out = optional (meta_all [>= N, 8] int32, feat_all [>= N, 53] float64) buffers on the root to receive into.
    Returns (meta_all [N, 8], feat_all [N, 53]) on rank `dst` in (rank, clip, si) order, else (None, None)."""
    # `dst` and the peers of the sends / receives below are ranks of the DEFAULT group (what dist.gather and P2POp take): a sub-group whose
    # members are not ranks 0 .. N-1 would address the wrong peers, so only the default group is served
    if group is not None and group is not dist.group.WORLD:
        raise ValueError("gather_rows serves the default process group only")
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    dev = feat.device
    n_rows = int(n_rows)
    cnt = torch.tensor([n_rows], dtype=torch.int64, device=dev)
    cnts = [torch.zeros_like(cnt) for _ in range(world)] if rank == dst else None
    dist.gather(cnt, cnts, dst=dst, group=group)
    if rank != dst:
        if n_rows:
            m = meta[:n_rows].clone()
            m[:, 0] += int(clip_base)
            ops = [dist.P2POp(dist.isend, m, dst, group), dist.P2POp(dist.isend, feat[:n_rows].contiguous(), dst, group)]
            for w in dist.batch_isend_irecv(ops):
                w.wait()
        return None, None
    counts = [int(c) for c in torch.cat(cnts).tolist()]          # the root sizes its receives: its one host read per step
    total = sum(counts)
    # receive buffers that are too small: the peers have their sends posted by now (they never see the counts), so the exchange is
    # completed into fresh tables FIRST and the error raised afterwards — raising here would leave every peer blocked in its send.
    # (A caller that reads its own buffers afterwards must not get fresh ones in silence: it is an error, not a fallback.)
    too_small = None
    if out is not None and (out[0].shape[0] < total or out[1].shape[0] < total):
        too_small = f"gather_rows: the receive buffers hold {min(out[0].shape[0], out[1].shape[0])} rows, {total} arrive"
        out = None
    if out is not None:
        meta_all, feat_all = out[0][:total], out[1][:total]
    else:
        meta_all = torch.empty((total, 8), dtype=torch.int32, device=dev)
        feat_all = torch.empty((total, 53), dtype=torch.float64, device=dev)
    ops, off = [], 0
    for r in range(world):
        c = counts[r]
        if r == dst:
            if c:
                meta_all[off:off + c] = meta[:c]
                meta_all[off:off + c, 0] += int(clip_base)
                feat_all[off:off + c] = feat[:c]
        elif c:
            ops.append(dist.P2POp(dist.irecv, meta_all[off:off + c], r, group))
            ops.append(dist.P2POp(dist.irecv, feat_all[off:off + c], r, group))
        off += c
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    if too_small:
        raise ValueError(too_small)
    return meta_all, feat_all


def shard_range(n_clips, rank, world):
    """contiguous block partition of clips over ranks (first `n_clips % world` ranks get one more)."""
    q, r = divmod(n_clips, world)
    a = rank * q + min(rank, r)
    return a, a + q + (1 if rank < r else 0)
