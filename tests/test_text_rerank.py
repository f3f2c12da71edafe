"""Text-similarity re-ranking (reference test.py:599-647): mirx.fusion.text_rerank_* against the loop
restatement in oracle/fusion.py.  The reference path needs the ConceptCLIP checkpoint (absent offline), so
this piece is pinned by the restatement only ("parity unpinned" for it in DESIGN.md)."""
import numpy as np
import pytest
import torch

from oracle import fusion as ofu


def _case(n, d, e, classes, seed):
    rng = np.random.default_rng(seed)
    unit = lambda a: (a / np.linalg.norm(a, axis=1, keepdims=True)).astype(np.float32)       # noqa: E731
    labels = rng.integers(0, classes, size=n)
    centers = rng.standard_normal((classes, d))
    emb = unit(centers[labels] + 0.9 * rng.standard_normal((n, d)))
    cimg = unit(rng.standard_normal((n, e)))
    text = unit(rng.standard_normal((classes, e)))
    return emb, cimg, text, labels


@pytest.mark.parametrize("n,rerank_k,w", [(60, 10, 0.7), (45, 100, 0.3), (30, 0, 0.5), (50, 5, 1.0)])
def test_text_rerank_scores_cpu(n, rerank_k, w):
    from mirx.fusion import text_rerank_scores
    emb, cimg, text, labels = _case(n, 32, 24, 4, n)
    want = ofu.text_rerank_dists(emb, cimg, text, labels, rerank_k, w)
    got = text_rerank_scores(torch.from_numpy(emb), cimg, text, labels, rerank_k, w).numpy()
    assert np.array_equal(np.isinf(got), np.isinf(want))
    fin = ~np.isinf(want)
    assert np.abs(got[fin] - want[fin]).max() < 1e-12
    if w == 1.0 or rerank_k == 0:
        s = emb.astype(np.float64) @ emb.astype(np.float64).T
        assert np.abs(got[fin] - s[fin]).max() < 1e-12


@pytest.mark.gpu
def test_text_rerank_evaluate_gpu_matches_oracle():
    from mirx.fusion import text_rerank_evaluate
    emb, cimg, text, labels = _case(400, 64, 48, 5, 11)
    dev = torch.device("cuda:0")
    got = text_rerank_evaluate(torch.from_numpy(emb).to(dev), torch.from_numpy(cimg).to(dev), torch.from_numpy(text).to(dev),
                               torch.from_numpy(labels).to(dev), rerank_k=20, text_weight=0.7, kappas=(1, 5, 10))
    acc, m_ap, pr = ofu.text_rerank_metrics(ofu.text_rerank_dists(emb, cimg, text, labels, 20, 0.7), labels, (1, 5, 10))
    assert np.abs(got["accuracy"] - acc).max() < 1e-3          # percent
    assert abs(got["mAP"] - m_ap) < 1e-5
    assert np.abs(np.asarray(got["pr"]) - np.asarray(pr)).max() < 1e-5
