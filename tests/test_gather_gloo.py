"""The N > 1 path on CPU: two processes, gloo backend, 127.0.0.1 rendezvous.  Each rank runs the
oracle on its shard of clips (standing in for the GPU back end, which needs a device) and the
product's gather code collects the feature matrices on rank 0, which must equal the single-process
result in (clip, si) order."""
import os
import socket
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rows_for(clips, base):
    from oracle import pyoracle
    fe = pyoracle.FrontEnd(pyoracle.fe_cfg())
    meta, feat = [], []
    for ci, x in enumerate(clips):
        out = pyoracle.run_backend(fe.run(x), pyoracle.default_cfg(level=5))
        for cb in out["callbacks"]:
            meta.append([ci, cb[0], 0, 0, 0, 0, 0, 0])
            feat.append(np.nan_to_num(np.asarray(cb[3], dtype=np.float64), nan=-1.0, posinf=-2.0))
    m = np.array(meta, dtype=np.int32).reshape(-1, 8)
    f = np.array(feat, dtype=np.float64).reshape(-1, 53)
    return m, f


def _worker(rank, world, port, pcm, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from webspeechanalyzer_amd.gather import gather_rows, shard_range
    a, b = shard_range(len(pcm), rank, world)
    m, f = _rows_for(pcm[a:b], a)
    cap = len(m) + 5
    meta = torch.zeros((cap, 8), dtype=torch.int32); meta[:len(m)] = torch.from_numpy(m)
    feat = torch.zeros((cap, 53), dtype=torch.float64); feat[:len(f)] = torch.from_numpy(f)
    ma, fa = gather_rows(meta, feat, len(m), a)
    # unequal counts (exact-size sends, nothing padded): every rank in turn contributes no rows at all, and the root
    # receives into buffers it was handed
    extra = []
    for empty in range(world):
        n = 0 if rank == empty else len(m)
        out = (torch.full((200, 8), -7, dtype=torch.int32), torch.full((200, 53), -7.0, dtype=torch.float64)) if rank == 0 else None
        mb, fb = gather_rows(meta, feat, n, a, out=out)
        if rank == 0:
            assert mb.data_ptr() == out[0].data_ptr() and fb.data_ptr() == out[1].data_ptr()
            assert (out[0][len(mb):] == -7).all()
            extra.append((mb.numpy().copy(), fb.numpy().copy()))
    if rank == 0:
        q.put((ma.numpy(), fa.numpy(), extra))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gather_equals_single_process():
    sys.path.insert(0, ROOT)
    from webspeechanalyzer_amd.synth import synth_clips
    from webspeechanalyzer_amd.gather import shard_range
    assert [shard_range(5, r, 2) for r in range(2)] == [(0, 3), (3, 5)]
    pcm = synth_clips(5, 64000, seed=9).numpy()
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, pcm, q)) for r in range(2)]
    for p in procs:
        p.start()
    ma, fa, extra = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    m1, f1 = _rows_for(pcm, 0)
    assert len(m1) > 5
    assert np.array_equal(ma[:, :2], m1[:, :2])
    assert np.array_equal(fa, f1)
    # a rank without rows: what arrives is exactly the other rank's rows
    split = int((m1[:, 0] < 3).sum())                      # rank 0 holds clips 0..2
    assert 0 < split < len(m1)
    (mb0, fb0), (mb1, fb1) = extra
    assert np.array_equal(mb0[:, :2], m1[split:, :2]) and np.array_equal(fb0, f1[split:])      # rank 0 empty
    assert np.array_equal(mb1[:, :2], m1[:split, :2]) and np.array_equal(fb1, f1[:split])      # rank 1 empty


def _worker3(rank, world, port, q):
    """three ranks, the MIDDLE one without rows; then receive buffers that are too small on the root: the exchange still completes on
    every rank (no peer is left blocked in its send) and only the root raises."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from webspeechanalyzer_amd.gather import gather_rows
    n = [4, 0, 3][rank]
    meta = torch.zeros((8, 8), dtype=torch.int32); feat = torch.zeros((8, 53), dtype=torch.float64)
    for i in range(n):
        meta[i, 0] = i; meta[i, 1] = 100 * rank + i; feat[i] = rank + i / 64.0
    base = [0, 4, 4][rank]
    ma, fa = gather_rows(meta, feat, n, base)
    raised = None
    try:
        gather_rows(meta, feat, n, base, out=(torch.zeros((5, 8), dtype=torch.int32), torch.zeros((5, 53), dtype=torch.float64)) if rank == 0 else None)
    except ValueError as e:
        raised = str(e)
    # the group is still usable afterwards
    mb, fb = gather_rows(meta, feat, n, base)
    if rank == 0:
        q.put((ma.numpy(), fa.numpy(), raised, mb.numpy(), fb.numpy()))
    else:
        assert raised is None and ma is None and mb is None
    dist.barrier()
    dist.destroy_process_group()


def test_three_rank_gather_with_an_empty_middle_rank_and_short_buffers():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker3, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    ma, fa, raised, mb, fb = q.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ma.shape == (7, 8) and fa.shape == (7, 53)
    assert ma[:, 0].tolist() == [0, 1, 2, 3, 4, 5, 6]                     # clip_base added: rank 2's clips follow rank 0's
    assert ma[:, 1].tolist() == [0, 1, 2, 3, 200, 201, 202]
    assert np.array_equal(fa[:, 0], np.array([0, 1 / 64, 2 / 64, 3 / 64, 2, 2 + 1 / 64, 2 + 2 / 64]))
    assert raised is not None and "hold 5 rows, 7 arrive" in raised
    assert np.array_equal(ma, mb) and np.array_equal(fa, fb)
