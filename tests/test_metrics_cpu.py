"""Product metric functions (mirx.metrics, vectorised) against the reference's golden outputs
and against the loop-level oracle restatement.  CPU only."""
import json
import os

import numpy as np
import pytest
import torch

from mirx import metrics as M
from oracle import metrics as OM
from oracle import search as OS

SETS = ["covidx300_d64", "mod3_1000_d32", "rand_257_d16"]


@pytest.fixture(scope="module")
def known(golden_dir):
    with open(os.path.join(golden_dir, "known_answers.json")) as fh:
        return json.load(fh)


def test_known_answers(known):
    t = known["compute_map_tiny"]
    mAP, aps, pr, prs = M.compute_map(np.array(t["ranks"]), np.array(t["gnd"]), t["kappas"])
    assert mAP == pytest.approx(t["mAP"], abs=1e-12)
    np.testing.assert_allclose(aps, t["aps"], atol=1e-12)
    np.testing.assert_allclose(pr, t["pr"], atol=1e-12)
    np.testing.assert_allclose(prs, t["prs"], atol=1e-12)
    for c in known["compute_ap"]:
        assert M.compute_ap(np.array(c["ranks"]), c["nres"]) == pytest.approx(c["ap"], abs=1e-12)
    for c in known["precision_at_k"]:
        assert M.precision_at_k(c["rel"], c["k"]) == pytest.approx(c["out"], abs=1e-15)
    for c in known["recall_at_k"]:
        assert M.recall_at_k(c["rel"], c["tp"], c["k"]) == pytest.approx(c["out"], abs=1e-15)
    for c in known["jaccard_score"]:
        assert M.jaccard_score(c["a"], c["b"]) == pytest.approx(c["out"], abs=1e-15)
    for c in known["majority_vote"]:
        assert M.majority_vote(np.array(c["labels"])) == c["out"]
    assert M.majority_vote([]) is None
    fm = M.evaluate_retrieval_metrics(np.random.default_rng(0).standard_normal((6, 4)).astype(np.float32),
                                      ["a", "b"] * 3, [f"p{i}" for i in range(6)], (1, 2))
    for k, v in known["fusion_6x4"].items():
        assert fm[k] == pytest.approx(v, abs=1e-9)


@pytest.mark.parametrize("name", SETS)
@pytest.mark.parametrize("metric", ["cdist", "cosine"])
def test_against_reference_outputs_on_reference_ranking(golden_dir, name, metric):
    """Fed the reference's own ranking, the functions must reproduce its outputs (1e-10)."""
    z = np.load(os.path.join(golden_dir, f"tail_{name}.npz"))
    if f"{metric}_ranks_ref_fp32" not in z.files:
        pytest.skip("no stored reference ranking for this set")
    labels = z["labels"]
    ref = z[f"{metric}_ranks_ref_fp32"].astype(np.int64)
    mAP, aps, pr, prs = M.compute_map(ref, labels, [1, 5, 10])
    assert mAP == pytest.approx(float(z[f"{metric}_mAP"]), abs=1e-10)
    np.testing.assert_allclose(aps, z[f"{metric}_aps"], atol=1e-10)
    np.testing.assert_allclose(pr, z[f"{metric}_pr"], atol=1e-10)
    np.testing.assert_allclose(prs, z[f"{metric}_prs"], atol=1e-10)
    acc = M.retrieval_accuracy(None, torch.as_tensor(labels), (1, 5, 10), topk_ids=ref.T[:, :10])
    np.testing.assert_allclose(torch.stack(acc).numpy(), z[f"{metric}_acc"], atol=1e-5)
    cls = M.compute_classification_metrics(torch.as_tensor(labels), None, [1, 5, 10, 15, 20], ranks=ref)
    for k in (1, 5, 10, 15, 20):
        np.testing.assert_allclose(list(cls[k].values()), z[f"{metric}_cls_k{k}"], atol=1e-9)
        assert list(cls[k].keys()) == ["precision_macro", "recall_macro", "f1_macro", "precision_weighted",
                                       "recall_weighted", "f1_weighted", "accuracy"]


@pytest.mark.parametrize("name", SETS)
def test_vectorised_equals_loop_oracle(golden_dir, name):
    z = np.load(os.path.join(golden_dir, f"tail_{name}.npz"))
    emb, labels = z["embeds"], z["labels"]
    ranks = OS.rank_all(emb, emb, metric=OS.METRIC_NEG_L2, exclude=np.arange(len(emb)))
    a = M.compute_map(ranks.T, labels, [1, 5, 10])
    b = OM.compute_map(ranks.T, labels, [1, 5, 10])
    assert a[0] == pytest.approx(b[0], abs=1e-12)
    for x, y in zip(a[1:], b[1:]):
        np.testing.assert_allclose(x, y, atol=1e-12)
    cls = M.compute_classification_metrics(labels, None, [1, 5, 10], ranks=ranks.T)
    ocl = OM.compute_classification_metrics(labels, ranks, (1, 5, 10))
    for k in (1, 5, 10):
        np.testing.assert_allclose(list(cls[k].values()), ocl[k], atol=1e-9)
    # score-matrix entry points behave like the reference call signature
    d = -torch.cdist(torch.as_tensor(emb), torch.as_tensor(emb))
    d.fill_diagonal_(float("-inf"))
    acc = torch.stack(M.retrieval_accuracy(d, torch.as_tensor(labels), (1, 5, 10))).numpy()
    np.testing.assert_allclose(acc, z["cdist_acc"], atol=0.5)


def test_multilabel_nih_fusion(golden_dir):
    z = np.load(os.path.join(golden_dir, "multilabel_120.npz"))
    emb = z["embeds"]
    ranks = OS.rank_all(emb, emb, metric=OS.METRIC_IP, exclude=np.arange(len(emb)))
    assert M.compute_map_multilabel(None, z["labels"], 0.5, ranks=ranks.T) == pytest.approx(float(z["map_t05"]), abs=1e-5)
    d = torch.as_tensor(emb) @ torch.as_tensor(emb).t()
    d.fill_diagonal_(float("-inf"))
    assert M.compute_map_multilabel(d, torch.as_tensor(z["labels"]), 0.4) == pytest.approx(float(z["map_t04"]), abs=1e-5)
    with open(os.path.join(golden_dir, "nih_results_40.json")) as fh:
        j = json.load(fh)
    out = M.evaluate_results(j["items"], j["threshold"], j["ks"])
    for k, v in j["metrics"].items():
        assert out[k] == pytest.approx(v, abs=1e-9), k
    f = np.load(os.path.join(golden_dir, "fusion_fuse.npz"))
    np.testing.assert_allclose(M.l2_normalize(f["a"]), f["l2"], atol=1e-7)
    np.testing.assert_allclose(M.concat_fusion(f["a"], f["c"]), f["concat"], atol=1e-7)
    np.testing.assert_allclose(M.weighted_sum_fusion(f["a"], f["b"], 0.3).embeddings, f["wsum03"], atol=1e-7)
    r = M.weighted_sum_fusion(f["a"], f["c"], 0.5)
    assert r.embeddings is None and r.skipped_reason == str(f["wsum_mismatch_reason"])
    with pytest.raises(ValueError):
        M.evaluate_retrieval_metrics_from_similarity(np.zeros((3, 4)), ["a"] * 3, ["p"] * 3)
