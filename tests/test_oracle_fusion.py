"""oracle/fusion.py against the goldens the reference's own fusion_eval functions produced
(tests/golden/make_golden_fusion.py)."""
import json
import os

import numpy as np
import pytest

from oracle import fusion as of

GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def gold():
    z = np.load(os.path.join(GOLD, "fusion_experiments.npz"), allow_pickle=True)
    rep = json.load(open(os.path.join(GOLD, "fusion_experiments.json")))
    return z, rep


@pytest.mark.parametrize("case", ["d24_d16", "d16_d16"])
def test_similarity_normalisation_and_confidence_fusion_match_reference(gold, case):
    z, _ = gold
    c, d = z[f"{case}_conv"], z[f"{case}_dino"]
    # the reference normalises twice on this path (evaluate.py:40-41 then metrics.py:14)
    cs, ds = of.similarity_matrix(of.om.l2_normalize_np(c)), of.similarity_matrix(of.om.l2_normalize_np(d))
    np.testing.assert_array_equal(of.normalize_similarity(cs, "zscore"), z[f"{case}_conv_sim_zscore"])
    np.testing.assert_array_equal(of.normalize_similarity(cs, "minmax"), z[f"{case}_conv_sim_minmax"])
    conf = of.confidence_fusion(cs, ds)
    np.testing.assert_array_equal(conf["similarity"], z[f"{case}_conf_similarity"])
    assert [conf["conv_selected_queries"], conf["dino_selected_queries"]] == z[f"{case}_conf_counts"].tolist()
    np.testing.assert_allclose([conf["alpha_mean"], conf["alpha_std"]], z[f"{case}_conf_alpha_stats"], rtol=0, atol=1e-12)


@pytest.mark.parametrize("case", ["d24_d16", "d16_d16"])
@pytest.mark.parametrize("mode", ["none", "zscore", "minmax"])
def test_experiments_match_reference(gold, case, mode):
    z, rep = gold
    got = of.run_experiments(z[f"{case}_conv"], z[f"{case}_dino"], z["labels"].tolist(), z["paths"].tolist(),
                             alpha_values=(0.2, 0.5, 0.8), k_values=(1, 5, 10), score_normalization=mode)
    want = rep[f"{case}/{mode}"]
    assert [g["experiment_name"] for g in got] == [w["experiment_name"] for w in want]
    for g, w in zip(got, want):
        assert g["skipped"] == w["skipped"] and g["skipped_reason"] == w["skipped_reason"], g["experiment_name"]
        assert set(g["metrics"]) == set(w["metrics"]), g["experiment_name"]
        for k, v in w["metrics"].items():
            assert abs(g["metrics"][k] - v) < 1e-9, (g["experiment_name"], k)


def test_file_sources_and_alignment_match_reference():
    ref = json.load(open(os.path.join(GOLD, "fusion_align.json")))
    src = os.path.join(GOLD, "fusion_sources")
    conv = of.read_embedding_file(os.path.join(src, "conv.npz"))
    dino = of.read_embedding_file(os.path.join(src, "dino.json"))
    for got, want in ((conv, ref["conv_records"]), (dino, ref["dino_records"])):
        assert [g[0] for g in got] == [w["image_path"] for w in want]
        assert [g[1] for g in got] == [w["label"] for w in want]
        np.testing.assert_array_equal(np.stack([g[2] for g in got]).astype(np.float64),
                                      np.asarray([w["embedding"] for w in want]))
    al = of.align_records(conv, dino)
    w = ref["aligned"]
    assert al["image_paths"] == w["image_paths"] and al["labels"] == w["labels"] and al["coverage"] == w["coverage"]
    np.testing.assert_array_equal(al["conv_embeddings"].astype(np.float64), np.asarray(w["conv_embeddings"]))
    np.testing.assert_array_equal(al["dino_embeddings"].astype(np.float64), np.asarray(w["dino_embeddings"]))
