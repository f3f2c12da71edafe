"""Near-tie audit shared by the parity tests.

The reference ranks with fp32 `-torch.cdist` / `e @ e.T` and an unstable sort; the product (and the oracle) rank by
fp64 scores, ties -> lowest id.  The two rankings may therefore differ -- but ONLY inside fp32 near-ties.  These
helpers make that statement checkable instead of bounding a count:

  * `reference_ranking(z, metric)` rebuilds the reference's own fp32 ranking of a golden set from the patch the
    generator stored against a plain-numpy fp64 ranking (tests/golden/make_golden.py: base_ranking);
  * `audit(P, R, emb, metric)` returns the queries whose ranking differs and asserts, for every position where they
    differ, that the fp64 scores of the two ids involved are closer than `gap` (1e-6): a genuinely wrong ranking fails.
"""
import numpy as np


def base_ranking(emb, metric):
    e = np.asarray(emb, dtype=np.float64)
    if metric == "cosine":
        s = e @ e.T
    else:
        s = -np.sqrt(np.maximum(((e[:, None, :] - e[None, :, :]) ** 2).sum(-1), 0.0))
    np.fill_diagonal(s, -np.inf)
    return np.argsort(-s, axis=1, kind="stable"), s


def reference_ranking(z, metric):
    """[nq, n] row per query: the ranking the reference's fp32 path produced for golden set `z`."""
    r, _ = base_ranking(z["embeds"], metric)
    patch = z[f"{metric}_refpatch"].astype(np.int64)
    r[patch[:, 0], patch[:, 1]] = patch[:, 2]
    assert np.array_equal(np.sort(r, axis=1), np.sort(base_ranking(z["embeds"], metric)[0], axis=1))   # still permutations
    if f"{metric}_ranks_ref_fp32" in z.files:                       # the small sets also store it whole
        assert np.array_equal(r, z[f"{metric}_ranks_ref_fp32"].astype(np.int64).T)
    return r


def audit(product_ranks, reference_ranks, emb, metric, gap=1e-6):
    """-> sorted array of the queries whose rankings differ.  Asserts that every differing position holds two ids whose
    fp64 scores are within `gap` of each other (the reference swapped an fp32 near-tie; nothing else is tolerated)."""
    p = np.asarray(product_ranks)
    r = np.asarray(reference_ranks)
    assert p.shape == r.shape
    _, s = base_ranking(emb, metric)
    qq, pp = np.nonzero(p != r)
    if qq.size:
        d = np.abs(s[qq, p[qq, pp]] - s[qq, r[qq, pp]])
        worst = int(np.argmax(d))
        assert d[worst] < gap, (f"query {qq[worst]} position {pp[worst]}: product id {p[qq[worst], pp[worst]]} vs "
                                f"reference id {r[qq[worst], pp[worst]]}, fp64 score gap {d[worst]:.3e} is not a near-tie")
    return np.unique(qq)
