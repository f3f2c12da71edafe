"""Gallery sharding across the GPUs of one node.

One process per GPU (torch.distributed, backend "nccl" == RCCL over xGMI).  Rank r owns the
contiguous gallery rows [r*N/W, (r+1)*N/W) with GLOBAL ids.  A search is:

  1. every rank embeds its own queries (plain data-parallel replicas, no collective),
  2. all-gather of the query embeddings  [Q_local, D] fp32  ->  [W*Q_local, D],
  3. local exact top-k of ALL queries against the local shard (libmirx),
  4. ONE all-gather of the packed per-shard candidates (fp64 ranking score, int64 id) [Q, k]
     (optionally in pieces, each launched asynchronously behind its local search: `chunks`),
  5. k-way merge of this rank's own queries (score desc, id asc) -- exact, because the top-k
     of a union is contained in the union of the per-shard top-k lists.

The reference has no sharded search (its only collective is dist.all_gather of validation
embeddings, train.py:604-609); the merge rule is the oracle's, so the result is identical to
a single-GPU search over the whole gallery.

`local_search` and `merge` are injectable so the host logic (partitioning, gather layout,
own-slice selection) runs under gloo on CPU in the tests; the defaults are the HIP paths.
"""
import torch
import torch.distributed as dist


def shard_bounds(n_total, world_size, rank):
    """Contiguous row range of `rank`; the first n_total % world_size ranks get one extra row."""
    base, rem = divmod(int(n_total), int(world_size))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _default_merge(scores, ids, metric):
    from .index import topk_merge
    return topk_merge(scores, ids, metric)


class ShardedSearcher:
    def __init__(self, local_search, metric="COSINE", group=None, merge=None):
        """local_search(q[Qtot,D], k) -> (fp64 ranking scores [Qtot,k], ids [Qtot,k]) on q's device."""
        self.local_search = local_search
        self.metric = metric
        self.group = group
        self.merge = merge or _default_merge

    @property
    def world_size(self):
        return dist.get_world_size(self.group) if dist.is_initialized() else 1

    @property
    def rank(self):
        return dist.get_rank(self.group) if dist.is_initialized() else 0

    def gather_queries(self, q_local):
        w = self.world_size
        if w == 1:
            return q_local
        out = q_local.new_empty((w * q_local.shape[0], q_local.shape[1]))
        dist.all_gather_into_tensor(out, q_local.contiguous(), group=self.group)
        return out

    def search(self, q_local, k, chunks=1, events=None):
        """Top-k over the WHOLE gallery for this rank's queries.

        -> (fp64 ranking scores [Q_local,k], reported fp32 values [Q_local,k], ids [Q_local,k])
        Every rank must call with the same Q_local, k and chunks.

        chunks > 1: every rank's queries are cut into `chunks` equal pieces; piece c of ALL ranks is searched locally and its
        candidate all-gather is launched asynchronously, so that it overlaps the local search of piece c + 1 (the collective
        runs on the backend's own stream; the search kernels stay on the current stream).
        events: optional dict that receives torch.cuda.Event pairs per stage ("gather_q", "search", "gather_cand", "merge")
        for the caller's stage timing (CUDA tensors only)."""
        w, r = self.world_size, self.rank
        ql = q_local.shape[0]

        def mark(name, begin):
            if events is not None and q_local.is_cuda:
                ev = torch.cuda.Event(enable_timing=True)
                ev.record()
                events.setdefault(name, []).append((begin, ev))

        mark("gather_q", True)
        q_all = self.gather_queries(q_local)
        mark("gather_q", False)
        if w == 1:
            mark("search", True)
            s_loc, i_loc = self.local_search(q_all, k)
            mark("search", False)
            mark("merge", True)
            out = self.merge(s_loc[None], i_loc[None], self.metric)
            mark("merge", False)
            return out
        if chunks < 1 or ql % chunks:
            raise ValueError("chunks must divide the per-rank query count")
        qc = ql // chunks
        blocks = q_all.view(w, ql, q_all.shape[1])
        pending = []
        for c in range(chunks):
            qa = q_all if chunks == 1 else blocks[:, c * qc:(c + 1) * qc].reshape(w * qc, q_all.shape[1])
            mark("search", True)
            s_loc, i_loc = self.local_search(qa, k)
            mark("search", False)
            # one packed all-gather per piece: [Q, k] x (score, id) as 2 x int64 words
            packed = torch.stack([s_loc.contiguous().view(torch.int64), i_loc.contiguous()], 0)
            flat = packed.new_empty(w * packed.numel())
            mark("gather_cand", True)
            work = dist.all_gather_into_tensor(flat, packed.view(-1), group=self.group, async_op=chunks > 1)
            mark("gather_cand", False)
            pending.append((work, flat, tuple(packed.shape)))
        outs = []
        for work, flat, shape in pending:
            if work is not None:
                work.wait()
            gathered = flat.view((w,) + shape)
            mine = gathered[:, :, r * qc:(r + 1) * qc, :].contiguous()        # [W, 2, qc, k]
            mark("merge", True)
            outs.append(self.merge(mine[:, 0].contiguous().view(torch.float64), mine[:, 1].contiguous(), self.metric))
            mark("merge", False)
        if len(outs) == 1:
            return outs[0]
        return tuple(None if outs[0][j] is None else torch.cat([o[j] for o in outs], 0) for j in range(3))
