"""Device metric tail (mirx_rank_metrics) against the oracle metrics and the golden fixtures.

The oracle (oracle/metrics.py) follows the reference's loops (test.py:58-146, 941-985;
fusion_eval/metrics.py:41-94); it is pinned against the reference's own outputs in
tests/test_oracle_metrics.py.  Tolerance: AP / precision are fp64 sums of exact ratios in a different
order -> 1e-12; counts are integers -> exact."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import metrics as om

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _ranking(n, d, seed, dev):
    from mirx.evaluate import rank_self
    g = torch.Generator().manual_seed(seed)
    e = torch.nn.functional.normalize(torch.randn(n, d, generator=g), dim=1).to(dev)
    ranks, _ = rank_self(e, "cdist")
    return e, ranks


@pytest.mark.parametrize("n,ncls", [(1, 1), (2, 2), (63, 3), (64, 3), (65, 2), (300, 3), (1000, 7), (3000, 3)])
def test_compute_map_device_matches_oracle(n, ncls):
    from mirx import metrics as mm
    dev = torch.device("cuda:0")
    _, ranks = _ranking(n, 32, n, dev)
    labels = (np.arange(n) * 7 + 3) % ncls
    kappas = [1, 5, 10]
    want = om.compute_map(ranks.cpu().numpy().T, labels, kappas)
    got = mm.compute_map(ranks.t(), labels, kappas)
    assert abs(got[0] - want[0]) < 1e-12
    np.testing.assert_allclose(got[1], want[1], rtol=0, atol=1e-12)
    np.testing.assert_allclose(got[2], want[2], rtol=0, atol=1e-12)
    np.testing.assert_allclose(got[3], want[3], rtol=0, atol=1e-12)
    # and the host (numpy) implementation of the same function
    host = mm.compute_map(ranks.cpu().numpy().T, labels, kappas)
    assert abs(got[0] - host[0]) < 1e-12


def test_compute_map_device_known_answer():
    """SURVEY 8c known answer: compute_map(ranks=[[1,0,0],[2,2,1],[0,1,2]], gnd=[0,0,1], kappas=[1,2])."""
    from mirx import metrics as mm
    ka = json.load(open(os.path.join(GOLD, "known_answers.json")))
    ranks = torch.tensor([[1, 0, 0], [2, 2, 1], [0, 1, 2]], dtype=torch.int64, device="cuda:0")
    m, aps, pr, prs = mm.compute_map(ranks, [0, 0, 1], [1, 2])
    assert abs(m - 0.5833333333333334) < 1e-15
    np.testing.assert_allclose(aps, [0.7916666666666666, 0.7916666666666666, 0.16666666666666666], atol=1e-15)
    np.testing.assert_allclose(pr, [2 / 3, 1 / 3], atol=1e-15)
    np.testing.assert_allclose(prs, [[1, .5], [1, .5], [0, 0]], atol=1e-15)
    assert ka is not None


@pytest.mark.parametrize("n,c,thr", [(120, 14, 0.4), (500, 14, 0.5), (257, 5, 0.3)])
def test_multilabel_map_device_matches_oracle(n, c, thr):
    from mirx import metrics as mm
    dev = torch.device("cuda:0")
    e, ranks = _ranking(n, 24, 100 + n, dev)
    rng = np.random.default_rng(n)
    lab = (rng.random((n, c)) < 0.25).astype(np.float32)
    lab[0] = 0                                                   # an image with no finding
    # the reference ranks with argsort(-dists, axis=0): self first (distance 0); rank_self puts it
    # last -- the function takes whatever ranking it is given, so compare on the same one
    r_np = ranks.cpu().numpy().T
    want = om.compute_map_multilabel(r_np.T, lab, thr)
    got = mm.compute_map_multilabel(None, lab, thr, ranks=ranks.t())
    host = mm.compute_map_multilabel(None, lab, thr, ranks=r_np)
    assert abs(got - want) < 1e-12 and abs(host - want) < 1e-12


@pytest.mark.parametrize("n", [6, 200, 1001])
def test_fusion_metrics_device_matches_oracle(n):
    from mirx import metrics as mm
    dev = torch.device("cuda:0")
    e, ranks = _ranking(n, 16, 7 * n, dev)
    labels = [["a", "b", "c", "lonely"][i % 3 if i else 3] for i in range(n)]     # one class of size 1
    paths = [f"p{i}" for i in range(n)]
    want = om.fusion_metrics_from_ranks(ranks.cpu().numpy(), labels, paths, (1, 5, 10))
    got = mm.evaluate_retrieval_metrics_from_similarity(None, labels, paths, (1, 5, 10), ranks=ranks)
    assert set(got) == set(want)
    for k in want:
        assert abs(got[k] - want[k]) < 1e-10, k


def test_self_in_the_middle_is_dropped_not_just_ignored():
    """drop_self moves later ranks up (fusion_eval/metrics.py:58-60) also when the query is not last."""
    from mirx import metrics as mm
    n = 130
    rng = np.random.default_rng(5)
    ranks = np.stack([rng.permutation(n) for _ in range(n)])      # self anywhere in the list
    labels = [str(i % 4) for i in range(n)]
    paths = [f"p{i}" for i in range(n)]
    want = om.fusion_metrics_from_ranks(ranks, labels, paths, (1, 3, 70))
    got = mm.evaluate_retrieval_metrics_from_similarity(None, labels, paths, (1, 3, 70),
                                                        ranks=torch.from_numpy(ranks).to("cuda:0"))
    for k in want:
        assert abs(got[k] - want[k]) < 1e-10, k


def test_rank_metrics_rejects_bad_arguments():
    from mirx import metrics as mm
    from mirx._lib import MirxError
    r = torch.zeros((2, 4), dtype=torch.int64, device="cuda:0")
    with pytest.raises(MirxError):
        mm.rank_metrics_device(r, [0, 0, 0, 0], [0, 0], kappas=list(range(1, 10)))   # > 8 kappas
    with pytest.raises(MirxError):
        mm.rank_metrics_device(r, [0, 0, 0, 0], [0, 0], kappas=[0])
    with pytest.raises(ValueError):
        mm.rank_metrics_device(r, [0, 0, 0, 0], [0], kappas=[1])
