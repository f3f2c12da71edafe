"""BASELINE.json full size (1M x 1024 gallery): size-independent properties of the HIP search
path, plus the oracle itself on a handful of queries.  GPU only; ~1 minute."""
import numpy as np
import pytest
import torch

from oracle import search as OS

pytestmark = pytest.mark.gpu

N, D, K = 1_000_000, 1024, 10


@pytest.fixture(scope="module")
def big():
    from mirx.index import FlatIndex
    dev = torch.device("cuda:0")
    ix = FlatIndex(D, "COSINE", 0)
    ix.reserve(N)
    chunk = 1 << 17
    for c, s in enumerate(range(0, N, chunk)):
        g = torch.Generator(device=dev).manual_seed(1234 + c)
        m = min(chunk, N - s)
        ix.add(torch.nn.functional.normalize(torch.randn(m, D, generator=g, device=dev), dim=1))
    q = torch.nn.functional.normalize(
        torch.randn(512, D, generator=torch.Generator(device=dev).manual_seed(4321), device=dev), dim=1)
    return ix, q


def test_tier1_equals_exact_tier_and_is_deterministic(big):
    from mirx import _lib as L
    ix, q = big
    s1, i1 = ix.search(q, K, return_f64=True)
    st = ix.last_stats()
    assert st["tier1_answered"] == q.shape[0], st            # the MFMA tier answered everything
    s2, i2 = ix.search(q, K, return_f64=True)
    assert torch.equal(i1, i2) and torch.equal(s1, s2)       # deterministic (no atomics-order effects)
    ix.set_option(L.OPT_TIERS, L.TIER_EXACT_ONLY)
    try:
        s3, i3 = ix.search(q[:64], K, return_f64=True)
    finally:
        ix.set_option(L.OPT_TIERS, L.TIER_AUTO)
    assert torch.equal(i1[:64], i3) and torch.equal(s1[:64], s3)
    # sortedness of every result list: score desc, ties -> id asc
    ds = s1[:, 1:] - s1[:, :-1]
    assert torch.all(ds <= 0)
    tie = ds == 0
    assert torch.all(i1[:, 1:][tie] > i1[:, :-1][tie])


def test_self_retrieval_and_exclusion(big):
    ix, _ = big
    rows = torch.as_tensor((np.arange(300) * 3331 + 17) % N, device="cuda")
    g, ids = ix.rows(0, 1)                                     # id convention: row number
    assert int(ids[0]) == 0
    qs = torch.stack([ix.rows(int(r), 1)[0][0] for r in rows[:300]])
    s, i = ix.search(qs, K, return_f64=True)
    assert torch.equal(i[:, 0], rows)                          # every row finds itself first
    self_dot = torch.as_tensor(np.diag(OS.scores(qs.cpu().numpy()[:32], qs.cpu().numpy()[:32])))
    assert torch.equal(s[:32, 0].cpu(), self_dot)              # ... with the bit-exact fp64 <g,g>
    s_ex, i_ex = ix.search(qs, K, exclude_ids=rows, return_f64=True)
    assert not torch.any(i_ex == rows[:, None])
    assert torch.equal(i_ex[:, : K - 1], i[:, 1:]) and torch.equal(s_ex[:, : K - 1], s[:, 1:])


def test_two_shards_merge_to_the_whole(big):
    from mirx.index import FlatIndex, topk_merge
    ix, q = big
    qs = q[:128]
    s_all, i_all = ix.search(qs, K, return_f64=True)
    parts_s, parts_i = [], []
    half = N // 2 + 12345
    for lo, hi in ((0, half), (half, N)):
        sh = FlatIndex(D, "COSINE", 0)
        sh.reserve(hi - lo)
        step = 1 << 17
        for s0 in range(lo, hi, step):
            rows, ids = ix.rows(s0, min(step, hi - s0))
            sh.add(rows, ids)
        s, i = sh.search(qs, K, return_f64=True)
        parts_s.append(s)
        parts_i.append(i)
        del sh
    m64, _, mid = topk_merge(torch.stack(parts_s), torch.stack(parts_i), "COSINE")
    assert torch.equal(mid, i_all) and torch.equal(m64, s_all)


def test_oracle_on_a_few_queries_at_full_size(big):
    ix, q = big
    host = np.empty((N, D), dtype=np.float32)
    step = 1 << 17
    for s0 in range(0, N, step):
        rows, _ = ix.rows(s0, min(step, N - s0))
        host[s0:s0 + rows.shape[0]] = rows.cpu().numpy()
    qs = q[:8]
    s, i = ix.search(qs, K, return_f64=True)
    o_s, o_i = OS.topk(qs.cpu().numpy(), host, K)
    np.testing.assert_array_equal(i.cpu().numpy(), o_i)
    np.testing.assert_array_equal(s.cpu().numpy(), o_s)


@pytest.mark.gpu
def test_eight_million_rows_planted_neighbours():
    """8M x 1024 rows (48 GB resident: byte offsets past 2^32 everywhere): planted near-duplicates of 256
    gallery rows must come back as (itself, its twin) with fp64-exact scores, from the MFMA tier."""
    from mirx.index import FlatIndex
    dev = torch.device("cuda:0")
    n, d, nq = 8_000_000, 1024, 256
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 70 * 2**30:
        pytest.skip("needs ~50 GB of free HBM")
    g = torch.Generator(device=dev).manual_seed(99)
    ix = FlatIndex(d, "COSINE", 0)
    ix.reserve(n)
    qrows = torch.arange(nq, device=dev) * 31_013 + 17            # spread over the whole id range (< 8M)
    twins = n - 1 - torch.arange(nq, device=dev) * 7              # their near-duplicates live at the far end
    keep = {}
    chunk = 1 << 19
    for s in range(0, n, chunk):
        m = min(chunk, n - s)
        x = torch.nn.functional.normalize(torch.randn(m, d, generator=g, device=dev), dim=1)
        for j, (a, b) in enumerate(zip(qrows.tolist(), twins.tolist())):
            if s <= a < s + m:
                keep[j] = x[a - s].clone()
        for j, b in enumerate(twins.tolist()):                     # twins are in the last chunk(s): after their originals
            if s <= b < s + m and j in keep:
                noise = torch.randn(d, generator=g, device=dev)
                x[b - s] = torch.nn.functional.normalize(keep[j] + 0.05 * noise / noise.norm(), dim=0)
        ix.add(x)
    assert len(ix) == n and len(keep) == nq
    q = torch.stack([keep[j] for j in range(nq)])
    sc, ids = ix.search(q, 2, return_f64=True)
    st = ix.last_stats()
    assert st["tier1_answered"] == nq and st["exact_answered"] == 0
    assert torch.equal(ids[:, 0], qrows) and torch.equal(ids[:, 1], twins)
    rows, _ = ix.rows(int(twins[5]), 1)
    want = float((q[5].double() * rows[0].double()).sum())
    assert abs(float(sc[5, 1]) - want) < 1e-12
    assert torch.all(sc[:, 0] > 0.999999) and torch.all(sc[:, 1] > 0.99)
