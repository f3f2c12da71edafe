"""Parity of the HIP search path (through the C ABI) against the CPU oracle.  GPU only.

Bar: ids identical; fp64 ranking scores bit-identical to oracle/search_ref.c (same lane-tree
order); reported fp32 values equal to the rounded oracle value."""
import numpy as np
import pytest
import torch

from oracle import search as OS

pytestmark = pytest.mark.gpu


def _unit(n, d, seed, device="cpu"):
    g = torch.Generator().manual_seed(seed)
    return torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1)


def _check(index, q, k, metric, exclude=None, ids=None, gallery=None):
    sc64, got_ids = index.search(q, k, exclude_ids=exclude, return_f64=True)
    sc32, got_ids2 = index.search(q, k, exclude_ids=exclude)
    o_s, o_i = OS.topk(q.numpy(), gallery.numpy(), k, metric=metric,
                       exclude=None if exclude is None else np.asarray(exclude), ids=ids)
    np.testing.assert_array_equal(got_ids.cpu().numpy(), o_i)
    np.testing.assert_array_equal(got_ids2.cpu().numpy(), o_i)
    np.testing.assert_array_equal(sc64.cpu().numpy(), o_s)               # bit-identical fp64
    np.testing.assert_array_equal(sc32.cpu().numpy(), OS.reported_value(o_s, metric).astype(np.float32))


@pytest.mark.parametrize("metric", ["COSINE", "L2"])
@pytest.mark.parametrize("n,d,nq", [(5000, 64, 37), (3001, 1024, 19), (777, 256, 5), (40, 36, 3)])
def test_exact_tier_small_gallery(metric, n, d, nq):
    from mirx.index import FlatIndex, metric_code
    g = _unit(n, d, 1234)
    q = _unit(nq, d, 4321)
    ix = FlatIndex(d, metric, 0)
    ix.add(g[: n // 2])
    ix.add(g[n // 2:].cuda())                     # second append from device memory
    assert len(ix) == n
    dp = (d + 3) // 4 * 4
    gp = torch.nn.functional.pad(g, (0, dp - d))
    qp = torch.nn.functional.pad(q, (0, dp - d))
    sc64, ids = ix.search(q, 10, return_f64=True)
    o_s, o_i = OS.topk(qp.numpy(), gp.numpy(), 10, metric=metric_code(metric))
    np.testing.assert_array_equal(ids.cpu().numpy(), o_i)
    np.testing.assert_array_equal(sc64.cpu().numpy(), o_s)
    st = ix.last_stats()
    assert st["exact_answered"] == nq and st["tier1_answered"] == 0


def test_exclude_custom_ids_ragged_and_empty():
    from mirx.index import FlatIndex
    d = 64
    g = _unit(300, d, 7)
    ids = (np.arange(300)[::-1] * 3 + 11).astype(np.int64)           # arbitrary, not monotone
    ix = FlatIndex(d, "COSINE", 0)
    ix.add(g, ids)
    # self-retrieval with the query's own id excluded (test.py:1081 fill_diagonal_)
    _check(ix, g[:50], 10, 0, exclude=ids[:50], ids=ids, gallery=g)
    _check(ix, g[:50], 10, 0, exclude=np.full(50, -1), ids=ids, gallery=g)
    # k larger than the gallery: tail is (-1, -inf)
    small = FlatIndex(d, "L2", 0)
    small.add(g[:6])
    s, i = small.search(g[:4], 10)
    assert torch.all(i[:, 6:] == -1) and torch.all(torch.isinf(s[:, 6:])) and torch.all(i[:, :6] >= 0)
    o_s, o_i = OS.topk(g[:4].numpy(), g[:6].numpy(), 10, metric=1)
    np.testing.assert_array_equal(i.cpu().numpy(), o_i)
    empty = FlatIndex(d, "COSINE", 0)
    s, i = empty.search(g[:3], 5)
    assert torch.all(i == -1) and torch.all(torch.isinf(s))
    # zero queries
    s, i = ix.search(torch.zeros(0, d), 5)
    assert s.shape == (0, 5)


def test_ties_resolve_to_lowest_id():
    from mirx.index import FlatIndex
    d = 128
    base = _unit(50, d, 3)
    g = torch.cat([base, base, base[:10]], 0)            # duplicates: exact score ties
    ix = FlatIndex(d, "COSINE", 0)
    ix.add(g)
    _check(ix, base[:20], 7, 0, gallery=g)
    ix2 = FlatIndex(d, "L2", 0)
    ix2.add(g)
    _check(ix2, base[:20], 7, 1, gallery=g)


@pytest.mark.parametrize("metric", ["COSINE", "L2"])
@pytest.mark.parametrize("n,d,nq", [(100000, 1024, 96), (40000, 256, 300), (65537, 512, 64)])
def test_tier1_mfma_path_matches_oracle(metric, n, d, nq):
    from mirx.index import FlatIndex, metric_code
    g = _unit(n, d, 1234)
    q = _unit(nq, d, 4321)
    ix = FlatIndex(d, metric, 0)
    ix.add(g)
    _check(ix, q, 10, metric_code(metric), gallery=g)
    st = ix.last_stats()
    assert st["tier1_answered"] + st["exact_answered"] == nq
    assert st["tier1_answered"] >= nq * 0.9, st          # the MFMA tier must do the work
    # self-retrieval over rows spread across every tile position
    rows = (np.arange(256) * 977) % n
    ex = torch.as_tensor(rows)
    _check(ix, g[rows], 10, metric_code(metric), exclude=rows, gallery=g)
    s, i = ix.search(g[rows], 1)
    np.testing.assert_array_equal(i[:, 0].cpu().numpy(), rows)


def test_guard_fallbacks_give_identical_answers():
    """Force the threshold too high (incomplete) and too low (overflow): both must fall through
    to the exact scan and still return the oracle's answer."""
    from mirx import _lib as L
    from mirx.index import FlatIndex
    n, d, nq = 50000, 256, 40
    g, q = _unit(n, d, 11), _unit(nq, d, 12)
    ix = FlatIndex(d, "COSINE", 0)
    ix.add(g)
    o_s, o_i = OS.topk(q.numpy(), g.numpy(), 10)
    for tau, expect in ((0.9, "incomplete"), (-1.0, "overflowed")):
        ix.set_option(L.OPT_FORCE_TAU, int(np.float32(tau).view(np.uint32)))
        s, i = ix.search(q, 10, return_f64=True)
        np.testing.assert_array_equal(i.cpu().numpy(), o_i)
        np.testing.assert_array_equal(s.cpu().numpy(), o_s)
        st = ix.last_stats()
        assert st[expect] == nq and st["exact_answered"] == nq, st
    ix.set_option(L.OPT_FORCE_TAU, L.FORCE_TAU_OFF)
    s, i = ix.search(q, 10, return_f64=True)
    np.testing.assert_array_equal(i.cpu().numpy(), o_i)
    assert ix.last_stats()["tier1_answered"] > 0


@pytest.mark.parametrize("n,d", [(300, 64), (1000, 32), (3000, 128), (2049, 64)])
@pytest.mark.parametrize("metric", ["COSINE", "L2"])
def test_rank_all_matches_oracle(n, d, metric):
    from mirx.index import FlatIndex, metric_code
    g = _unit(n, d, 21)
    ix = FlatIndex(d, metric, 0)
    ix.add(g)
    ex = np.arange(n)
    ranks, sc = ix.rank_all(g, exclude_ids=ex, with_scores=True)
    o_r, o_s = OS.rank_all(g.numpy(), g.numpy(), metric=metric_code(metric), exclude=ex, with_scores=True)
    np.testing.assert_array_equal(ranks.cpu().numpy(), o_r)
    assert np.all(ranks[:, -1].cpu().numpy() == ex)                    # excluded row last
    np.testing.assert_array_equal(sc.cpu().numpy()[:, :-1],
                                  OS.reported_value(o_s, metric_code(metric)).astype(np.float32)[:, :-1])


def test_topk_merge_equals_global_search():
    from mirx.index import FlatIndex, topk_merge
    n, d, nq, k = 9000, 64, 33, 10
    g, q = _unit(n, d, 31), _unit(nq, d, 32)
    whole = FlatIndex(d, "COSINE", 0)
    whole.add(g)
    s_all, i_all = whole.search(q, k, return_f64=True)
    parts_s, parts_i = [], []
    for lo, hi in ((0, 2000), (2000, 2005), (2005, 9000)):
        ix = FlatIndex(d, "COSINE", 0)
        ix.add(g[lo:hi], np.arange(lo, hi))
        s, i = ix.search(q, k, return_f64=True)
        parts_s.append(s)
        parts_i.append(i)
    m64, m32, mid = topk_merge(torch.stack(parts_s), torch.stack(parts_i), "COSINE")
    assert torch.equal(mid, i_all) and torch.equal(m64, s_all)


def test_l2_normalize_kernel():
    from mirx.index import l2_normalize_
    x = torch.randn(513, 1024, generator=torch.Generator().manual_seed(5)) * 3.0
    x[7] = 0.0                                                   # eps path: stays zero
    ref = OS.l2_normalize(x.numpy())
    y = l2_normalize_(x.clone().cuda()).cpu()
    assert np.max(np.abs(y.numpy() - ref)) <= 6e-8               # <= 1 ulp of values <= 1
    np.testing.assert_allclose(y.numpy(), torch.nn.functional.normalize(x, dim=1).numpy(), atol=2e-7)
    z = torch.randn(10, 37)
    np.testing.assert_allclose(l2_normalize_(z.clone().cuda()).cpu().numpy(),
                               torch.nn.functional.normalize(z, dim=1).numpy(), atol=2e-7)


@pytest.mark.parametrize("metric", ["COSINE", "L2"])
def test_clustered_gallery_second_chance_pass(metric):
    """Tight clusters make the sampled threshold too high for the guard; the second filter pass
    (tau2 = kth - 2 eps) must answer those queries in the MFMA tier -- and identically to the oracle."""
    from mirx.index import FlatIndex, metric_code
    n, d, nq, ncls = 60000, 256, 96, 40
    g = torch.Generator().manual_seed(3)
    centers = torch.nn.functional.normalize(torch.randn(ncls, d, generator=g), dim=1)
    lab = torch.randint(0, ncls, (n,), generator=g)
    x = torch.nn.functional.normalize(centers[lab] + 0.3 / d ** 0.5 * torch.randn(n, d, generator=g), dim=1)
    q = torch.nn.functional.normalize(x[torch.randint(0, n, (nq,), generator=g)] + 0.01 * torch.randn(nq, d, generator=g), dim=1)
    ix = FlatIndex(d, metric, 0)
    ix.add(x)
    _check(ix, q, 10, metric_code(metric), gallery=x)
    st = ix.last_stats()
    assert st["incomplete"] > 0, st                       # the first pass did reject queries ...
    assert st["tier1_answered"] >= nq * 0.9, st           # ... and the second pass answered them


@pytest.mark.parametrize("nq", [40, 100, 300])
def test_adjacent_passing_rows_claim_distinct_candidate_slots(nq):
    """the per-region fill counters of the filter GEMM are shared by the lanes that hold the same query
    (2 lanes in k_gemm, 4 in k_gemm16).  A gallery of 200-row runs of identical vectors and a forced mid-range threshold
    make MANY adjacent rows of one query pass in the same accumulator tile: every one of them must get its own slot --
    stats.candidates equals the exact number of passing rows (no overflow, no loss), and the answer is the oracle's
    (ties -> lowest ids)."""
    from mirx import _lib as L
    from mirx.index import FlatIndex
    d, run, nbase = 256, 200, 200
    n = run * nbase
    base = _unit(nbase, d, 21)
    g = base.repeat_interleave(run, dim=0)                    # rows 200 i .. 200 i + 199 are copies of base[i]
    qi = torch.arange(nq) % nbase
    q = base[qi]
    ix = FlatIndex(d, "COSINE", 0)
    ix.add(g)
    dots = (base @ base.t()).fill_diagonal_(-1)
    assert float(dots.max()) < 0.45                            # every other base is far below the threshold
    ix.set_option(L.OPT_FORCE_TAU, int(np.float32(0.5).view(np.uint32)))
    s, i = ix.search(q, 10, return_f64=True)
    st = ix.last_stats()
    assert st["candidates"] == nq * run, st                     # exactly the copies of the query's base, each once
    assert st["overflowed"] == 0 and st["incomplete"] == 0 and st["tier1_answered"] == nq, st
    o_s, o_i = OS.topk(q.numpy(), g.numpy(), 10)
    np.testing.assert_array_equal(i.cpu().numpy(), o_i)
    np.testing.assert_array_equal(s.cpu().numpy(), o_s)
    assert np.array_equal(o_i[:, 0], (qi * run).numpy())       # the lowest id of the run


def test_a_whole_accumulator_tile_of_passing_scores_takes_the_spill_path():
    """k_gemm16 parks the lanes that hold a passing score in a 64-entry ring per wave; when more of them turn up inside ONE
    gallery tile the ring moves to the wave's global spill area (round 4).  256 identical queries against 200-row runs of
    identical rows and a forced mid-range threshold make every score of a 128 x 64 wave tile pass at once (1024 ring
    entries): nothing may be lost -- stats.candidates is exact and the answer is the oracle's."""
    from mirx import _lib as L
    from mirx.index import FlatIndex
    d, run, nbase, nq = 256, 200, 200, 256                     # 40 000 rows: the MFMA tier (small galleries are scanned exactly)
    base = _unit(nbase, d, 23)
    g = base.repeat_interleave(run, dim=0)
    q = base[torch.zeros(nq, dtype=torch.long)].clone()
    q[nq // 2:] = base[7]                                      # second half of the queries: another run, another tile
    ix = FlatIndex(d, "COSINE", 0)
    ix.add(g)
    dots = (base @ base.t()).fill_diagonal_(-1)
    assert float(dots.max()) < 0.45
    ix.set_option(L.OPT_FORCE_TAU, int(np.float32(0.5).view(np.uint32)))
    s, i = ix.search(q, 10, return_f64=True)
    st = ix.last_stats()
    assert st["candidates"] == nq * run, st
    assert st["incomplete"] == 0 and st["tier1_answered"] + st["exact_answered"] == nq, st
    o_s, o_i = OS.topk(q.numpy(), g.numpy(), 10)
    np.testing.assert_array_equal(i.cpu().numpy(), o_i)
    np.testing.assert_array_equal(s.cpu().numpy(), o_s)



def test_more_than_8192_queries_multi_pass():
    """8192 + 300 queries in one call = two internal passes (8192, then 300 at a different query tile);
    ids, fp64 scores and the aggregated stats must be those of one search (W = 8 ranks hit this path with 32 768 queries)."""
    from mirx.index import FlatIndex
    n, d, nq = 40000, 256, 8192 + 300
    g, q = _unit(n, d, 31), _unit(nq, d, 32)
    ix = FlatIndex(d, "COSINE", 0)
    ix.add(g)
    s, i = ix.search(q, 10, return_f64=True)
    st = ix.last_stats()
    o_s, o_i = OS.topk(q.numpy(), g.numpy(), 10)
    np.testing.assert_array_equal(i.cpu().numpy(), o_i)
    np.testing.assert_array_equal(s.cpu().numpy(), o_s)
    assert st["nq"] == nq and st["tier1_answered"] + st["exact_answered"] == nq, st
    ex = np.arange(nq) % n                                      # with exclusions, reported fp32 values
    s32, i2 = ix.search(q, 10, exclude_ids=ex)
    o_s2, o_i2 = OS.topk(q.numpy(), g.numpy(), 10, exclude=ex)
    np.testing.assert_array_equal(i2.cpu().numpy(), o_i2)
    np.testing.assert_array_equal(s32.cpu().numpy(), OS.reported_value(o_s2, 0).astype(np.float32))


def test_search_begin_end_equals_search():
    """mirx_index_search_begin / _end: the first pass is enqueued without blocking the host, other work may be queued in
    between, the result equals the blocking call -- including when the follow-up passes are needed (forced thresholds)."""
    from mirx import _lib as L
    from mirx.index import FlatIndex
    n, d, nq = 50000, 256, 600
    g, q = _unit(n, d, 41), _unit(nq, d, 42)
    ix = FlatIndex(d, "COSINE", 0)
    ix.add(g)
    o_s, o_i = OS.topk(q.numpy(), g.numpy(), 10)
    for tau in (None, 0.9, -1.0):                               # normal, every query incomplete, every query overflowed
        if tau is not None:
            ix.set_option(L.OPT_FORCE_TAU, int(np.float32(tau).view(np.uint32)))
        h = ix.search_begin(q, 10)
        filler = torch.randn(2048, 2048, device="cuda") @ torch.randn(2048, 2048, device="cuda")   # queued behind the first pass
        with pytest.raises(L.MirxError):
            ix.search(q, 10)                                    # the index is busy until search_end
        s, i = ix.search_end(h, return_f64=True)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(i.cpu().numpy(), o_i)
        np.testing.assert_array_equal(s.cpu().numpy(), o_s)
        s32, i32 = ix.search(q, 10)
        np.testing.assert_array_equal(i32.cpu().numpy(), o_i)
        assert filler.shape == (2048, 2048)
    ix.set_option(L.OPT_FORCE_TAU, L.FORCE_TAU_OFF)
    assert ix._lib.mirx_index_search_end(ix._h) == 0           # nothing pending: a no-op
