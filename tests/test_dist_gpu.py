"""world_size-2 run of the sharded search with the PRODUCT local search and merge (libmirx) on
the GPU.  Both ranks share cuda:0 (one-GPU box), so the two all-gathers go through gloo on host
copies; the nccl path itself is exercised by bench.py --gpus N on a multi-GPU node."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import search as OS

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class _HostGatherSearcher:
    """ShardedSearcher with its collectives routed through host tensors (gloo)."""

    def __new__(cls, *a, **k):
        from mirx.dist import ShardedSearcher

        class _S(ShardedSearcher):
            def gather_queries(self, q_local):
                out = torch.empty((self.world_size * q_local.shape[0], q_local.shape[1]))
                dist.all_gather_into_tensor(out, q_local.cpu().contiguous())
                return out.to(q_local.device)

        return _S(*a, **k)


def _worker(rank, world, port, n, d, ql, k, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from mirx.dist import shard_bounds
        from mirx.index import FlatIndex, topk_merge
        dev = torch.device("cuda:0")
        g = torch.nn.functional.normalize(torch.randn(n, d, generator=torch.Generator().manual_seed(1)), dim=1)
        q = torch.nn.functional.normalize(torch.randn(world * ql, d, generator=torch.Generator().manual_seed(2)), dim=1)
        lo, hi = shard_bounds(n, world, rank)
        ix = FlatIndex(d, "COSINE", 0)
        ix.add(g[lo:hi], np.arange(lo, hi))

        def local(qa, kk):
            s, i = ix.search(qa.to(dev), kk, return_f64=True)
            return s.cpu(), i.cpu()                      # candidates travel through gloo on the host

        def merge(s, i, metric):
            return topk_merge(s.to(dev), i.to(dev), metric)

        ss = _HostGatherSearcher(local, "COSINE", merge=merge)
        s64, s32, ids = ss.search(q[rank * ql:(rank + 1) * ql].to(dev), k)
        ret[rank] = (s64.cpu().numpy(), ids.cpu().numpy(), ix.last_stats())
    finally:
        dist.destroy_process_group()


def test_two_rank_sharded_search_on_gpu():
    n, d, ql, k, world = 90001, 256, 33, 10, 2
    ret = mp.Manager().dict()
    mp.spawn(_worker, args=(world, _free_port(), n, d, ql, k, ret), nprocs=world, join=True)
    g = torch.nn.functional.normalize(torch.randn(n, d, generator=torch.Generator().manual_seed(1)), dim=1)
    q = torch.nn.functional.normalize(torch.randn(world * ql, d, generator=torch.Generator().manual_seed(2)), dim=1)
    o_s, o_i = OS.topk(q.numpy(), g.numpy(), k)
    for r in range(world):
        s, i, st = ret[r]
        np.testing.assert_array_equal(i, o_i[r * ql:(r + 1) * ql])
        np.testing.assert_array_equal(s, o_s[r * ql:(r + 1) * ql])
        assert st["tier1_answered"] > 0                  # each 45k-row shard went through the MFMA tier


def test_plain_bench_command_runs_two_ranks_in_rehearsal_mode():
    """`python bench.py --gpus 2` as the driver types it (no torchrun): the script starts its own two ranks.  On this
    one-GPU box both share cuda:0 and the collectives run over gloo (MIRX_BENCH_REHEARSE=1) -- the control flow, the
    per-stage times and the single JSON line are what is checked; RCCL itself needs two GPUs (never run here)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MIRX_BENCH_CHILD")}
    env["MIRX_BENCH_REHEARSE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--gallery", "80000", "--queries", "256", "--embed-batch", "256", "--no-extras", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    assert d["config"]["queries_per_gpu_per_step"] == 256
    assert set(d["config"]["pipeline_ms_per_step_max_over_ranks"]) == {"embed", "gather_q", "search", "gather_cand", "merge"}
    assert d["config"]["search_stats_last_step"]["nq"] == 512        # every rank searches all ranks' queries
