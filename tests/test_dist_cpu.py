"""world_size-2 gloo test of the sharded-search host logic (CPU): partitioning, the two
all-gathers, own-slice merge.  Local search and merge are the CPU oracle here; on the GPU box
the same class runs with libmirx (tests/test_dist_gpu.py)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import search as OS


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _np_merge(scores, ids, metric):
    w, q, k = scores.shape
    s = scores.permute(1, 0, 2).reshape(q, w * k).numpy()
    i = ids.permute(1, 0, 2).reshape(q, w * k).numpy()
    out_s = np.empty((q, k))
    out_i = np.empty((q, k), dtype=np.int64)
    for r in range(q):
        key = sorted(range(w * k), key=lambda j: (i[r, j] < 0, -s[r, j], i[r, j]))[:k]
        out_s[r], out_i[r] = s[r, key], i[r, key]
    return torch.from_numpy(out_s), None, torch.from_numpy(out_i)


def _worker(rank, world, port, n, d, ql, k, ret, chunks=1):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mirx.dist import ShardedSearcher, shard_bounds
        g = torch.nn.functional.normalize(torch.randn(n, d, generator=torch.Generator().manual_seed(1)), dim=1)
        q = torch.nn.functional.normalize(torch.randn(world * ql, d, generator=torch.Generator().manual_seed(2)), dim=1)
        lo, hi = shard_bounds(n, world, rank)

        def local(qa, kk):
            s, i = OS.topk(qa.numpy(), g[lo:hi].numpy(), kk, ids=np.arange(lo, hi))
            return torch.from_numpy(s), torch.from_numpy(i)

        ss = ShardedSearcher(local, "COSINE", merge=_np_merge)
        s, _, i = ss.search(q[rank * ql:(rank + 1) * ql], k, chunks=chunks)
        ret[rank] = (s.numpy(), i.numpy())
    finally:
        dist.destroy_process_group()


def test_shard_bounds_cover_exactly():
    from mirx.dist import shard_bounds
    for n, w in ((10, 3), (1_000_000, 8), (5, 8), (0, 2)):
        spans = [shard_bounds(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
        assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def test_sharded_search_world2_equals_global_oracle():
    n, d, ql, k, world = 2001, 32, 7, 10, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, d, ql, k, ret), nprocs=world, join=True)
    g = torch.nn.functional.normalize(torch.randn(n, d, generator=torch.Generator().manual_seed(1)), dim=1)
    q = torch.nn.functional.normalize(torch.randn(world * ql, d, generator=torch.Generator().manual_seed(2)), dim=1)
    o_s, o_i = OS.topk(q.numpy(), g.numpy(), k)
    for r in range(world):
        s, i = ret[r]
        np.testing.assert_array_equal(i, o_i[r * ql:(r + 1) * ql])
        np.testing.assert_array_equal(s, o_s[r * ql:(r + 1) * ql])


@pytest.mark.parametrize("chunks", [2, 4])
def test_sharded_search_in_pieces_with_async_gathers(chunks):
    """chunks > 1: piece c of every rank's queries is searched while the candidate all-gather of piece c - 1 is in
    flight; the result must not depend on the cut."""
    n, d, ql, k, world = 1500, 16, 8, 5, 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), n, d, ql, k, ret, chunks), nprocs=world, join=True)
    g = torch.nn.functional.normalize(torch.randn(n, d, generator=torch.Generator().manual_seed(1)), dim=1)
    q = torch.nn.functional.normalize(torch.randn(world * ql, d, generator=torch.Generator().manual_seed(2)), dim=1)
    o_s, o_i = OS.topk(q.numpy(), g.numpy(), k)
    for r in range(world):
        s, i = ret[r]
        np.testing.assert_array_equal(i, o_i[r * ql:(r + 1) * ql])
        np.testing.assert_array_equal(s, o_s[r * ql:(r + 1) * ql])
