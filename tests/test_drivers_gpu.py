"""Round-2 evaluation drivers on the GPU against golden outputs of the reference's own functions
(tests/golden/make_golden_r2.py): evaluate_multilabels (test.py:987-1062), evaluate_map
(nih_multilabel_training.py:66-99), eval_medsiglip.evaluate (eval_medsiglip.py:189-260), and the NIH gallery flow
(nih_zilliz_utils.py + query_nih_zilliz.py) against the search oracle."""
import json
import os
import types

import numpy as np
import pytest
import torch

from oracle import search as OS

pytestmark = pytest.mark.gpu


class _Lookup(torch.nn.Module):
    def __init__(self, table, key=None):
        super().__init__()
        self.register_buffer("table", table)
        self.key = key

    def forward(self, idx):
        e = self.table[idx]
        return {self.key: e} if self.key else e


def test_evaluate_multilabels_matches_reference(golden_dir, tmp_path, capsys):
    from mirx.evaluate import evaluate_multilabels
    z = np.load(os.path.join(golden_dir, "multilabel_drivers_150.npz"))
    emb, lab = torch.as_tensor(z["embeds"]), torch.as_tensor(z["labels"])
    n = emb.shape[0]
    loader = [(torch.arange(i, min(i + 40, n)), lab[i:i + 40]) for i in range(0, n, 40)]
    dev = torch.device("cuda:0")
    out = evaluate_multilabels(_Lookup(emb).to(dev), loader, dev, types.SimpleNamespace(save_dir=str(tmp_path)))
    assert out["mAP"][0.25] == pytest.approx(float(z["map_t025"]), abs=1e-9)
    assert out["mAP"][0.5] == pytest.approx(float(z["map_t05"]), abs=1e-9)
    for k, (p, r) in zip(z["table_k"].tolist(), z["table_pr"].tolist()):        # the reference prints two decimals
        assert round(out["precision"][k], 2) == pytest.approx(p, abs=1e-9), k
        assert round(out["recall"][k], 2) == pytest.approx(r, abs=1e-9), k
    saved = np.load(tmp_path / "evaluation_results.npz")
    np.testing.assert_array_equal(saved["embeds"], z["embeds"])
    np.testing.assert_array_equal(saved["labels"], z["labels"])
    text = capsys.readouterr().out
    assert "--- VinDr-CXR Retrieval Results ---" in text and ">> mAP (Jaccard > 0.25):" in text and "Precision@K" in text


def test_evaluate_map_matches_reference(golden_dir):
    from mirx.nih import evaluate_map
    z = np.load(os.path.join(golden_dir, "multilabel_drivers_150.npz"))
    emb, lab = torch.as_tensor(z["embeds"]), torch.as_tensor(z["labels"])
    n = emb.shape[0]
    loader = [(torch.arange(i, min(i + 40, n)), lab[i:i + 40]) for i in range(0, n, 40)]
    dev = torch.device("cuda:0")
    got = evaluate_map(_Lookup(emb, "embedding").to(dev), loader, dev, 0.4)
    assert got == pytest.approx(float(z["evaluate_map_t04"]), abs=1e-9)
    assert evaluate_map(_Lookup(emb, "embedding").to(dev), loader, dev, 0.25) == pytest.approx(float(z["evaluate_map_t025"]), abs=1e-9)
    # the host statement (CPU tensors) gives the same number
    assert evaluate_map(_Lookup(emb, "embedding"), loader, torch.device("cpu"), 0.4) == pytest.approx(got, abs=1e-9)


def test_medsiglip_evaluate_matches_reference(golden_dir, capsys):
    """eval_medsiglip.evaluate with the same stand-in dual encoder and tokenizer the golden run used: text features,
    zero-shot metrics and the retrieval tail."""
    from mirx.medsiglip_eval import COVIDX_LABEL_TO_TEXT, evaluate, get_text_features
    z = np.load(os.path.join(golden_dir, "medsiglip_eval_120.npz"))
    dev = torch.device("cuda:0")
    img_feat, txt_table = torch.as_tensor(z["img_feat"]).to(dev), torch.as_tensor(z["txt_table"]).to(dev)
    cls = z["labels"]
    assert [COVIDX_LABEL_TO_TEXT[i] for i in sorted(COVIDX_LABEL_TO_TEXT)] == z["prompts"].tolist()

    class _Tok(dict):
        def to(self, device):
            return _Tok({k: v.to(device) for k, v in self.items()})

    class _Tokenizer:
        def __call__(self, prompts, max_length, padding, truncation, return_attention_mask, return_tensors):
            assert padding == "max_length" and truncation and return_attention_mask and return_tensors == "pt"
            ids = torch.zeros((len(prompts), max_length), dtype=torch.long)
            mask = torch.zeros_like(ids)
            for i, p in enumerate(prompts):
                ln = min(max_length, 3 + len(p) % 7)
                ids[i, :ln] = torch.arange(1, ln + 1) + i
                mask[i, :ln] = 1
            return _Tok(input_ids=ids, attention_mask=mask)

    class _Dual(torch.nn.Module):
        def __init__(self):
            super().__init__()
            self.logit_scale = torch.nn.Parameter(torch.tensor(2.3))

        def get_text_features(self, input_ids, attention_mask):
            return txt_table[:input_ids.shape[0]] * (1.0 + attention_mask.sum(1, keepdim=True).float() / 10.0)

        def get_image_features(self, pixel_values):
            return img_feat[pixel_values]

    proc = types.SimpleNamespace(tokenizer=_Tokenizer())
    loader = [{"pixel_values": torch.arange(i, min(i + 16, 120)), "labels": torch.as_tensor(cls[i:i + 16])}
              for i in range(0, 120, 16)]
    model = _Dual().to(dev)
    tf = get_text_features(model, proc, dev, z["prompts"].tolist(), 16)
    np.testing.assert_allclose(tf.cpu().numpy(), z["text_features"], atol=1e-6)
    out = evaluate(model, proc, loader, dev, types.SimpleNamespace(max_text_length=16, eval_batch_size=16))
    assert out["zero_shot"]["accuracy"] == pytest.approx(float(z["zs_accuracy"]), abs=1e-9)
    assert out["zero_shot"]["precision_macro"] == pytest.approx(float(z["zs_precision"]), abs=1e-9)
    assert out["zero_shot"]["recall_macro"] == pytest.approx(float(z["zs_recall"]), abs=1e-9)
    assert out["zero_shot"]["f1_macro"] == pytest.approx(float(z["zs_f1"]), abs=1e-9)
    np.testing.assert_allclose(out["acc"], z["r_at_k"], atol=1e-5)
    # retrieval tail: well separated classes, no fp32 near-tie in this set -> the reference's numbers to 1e-9
    assert out["mAP"] == pytest.approx(float(z["mAP"]), abs=1e-9)
    np.testing.assert_allclose(out["aps"], z["aps"], atol=1e-9)
    np.testing.assert_allclose(out["pr"], z["pr"], atol=1e-9)
    fields = z["cls_fields"].tolist()
    for k, row in zip(z["cls_keys"].tolist(), z["cls_vals"]):
        got = [out["classification"][k][f] for f in fields]
        np.testing.assert_allclose(got, row, atol=1e-9)
    text = capsys.readouterr().out
    assert ">> Zero-shot Classification Metrics:" in text and ">> Retrieval Classification Metrics (Majority Voting):" in text


def test_nih_gallery_flow(tmp_path):
    """create_nih_collection -> insert_rows -> query_gallery (one batched exact search) -> evaluate_results; the hits are
    the oracle's exact ranking and carry the stored NIH fields (nih_zilliz_utils.py:136-280, query_nih_zilliz.py:49-71)."""
    from mirx.metrics import evaluate_results
    from mirx.nih import (EMBEDDING_DIM, create_nih_collection, get_nih_collection, insert_rows, query_gallery,
                          search_collection)
    rng = np.random.default_rng(3)
    n = 600
    lab = (rng.random((n, 14)) < 0.15).astype(np.float32)
    emb = lab @ rng.standard_normal((14, EMBEDDING_DIM)).astype(np.float32) + 0.5 * rng.standard_normal((n, EMBEDDING_DIM)).astype(np.float32)
    emb /= np.linalg.norm(emb, axis=1, keepdims=True)
    rows = [{"image_path": f"/nih/g_{i}.npy", "image_name": f"g_{i}.npy", "label_names": [f"L{j}" for j in np.flatnonzero(lab[i])],
             "multi_hot": lab[i].tolist(), "embedding": emb[i]} for i in range(n)]
    col = create_nih_collection("nih_dinov2_gallery", drop_old=True)
    assert col.schema == ["id", "image_path", "image_name", "label_text", "label_vector_json", "embedding"]
    insert_rows(col, rows[:250])
    insert_rows(col, rows[250:])
    assert get_nih_collection("nih_dinov2_gallery") is col and col.num_entities == n
    with pytest.raises(ValueError):
        get_nih_collection("nope")
    q = [rows[i] for i in (5, 77, 301)]
    items = query_gallery(col, q, top_k=12)
    o_s, o_i = OS.topk(emb[[5, 77, 301]], emb, 12)
    for it, r, ids, sc in zip(items, q, o_i, o_s):
        assert it["query_image_path"] == r["image_path"] and it["query_label_vector"] == r["multi_hot"]
        assert [h["id"] for h in it["results"]] == ids.tolist()
        assert [h["score"] for h in it["results"]] == pytest.approx(sc.tolist(), abs=1e-6)
        assert it["results"][0]["image_name"] == f"g_{ids[0]}.npy" and it["results"][3]["label_vector"] == lab[ids[3]].tolist()
        assert it["results"][0]["label_text"] == "|".join(rows[ids[0]]["label_names"])
    full = query_gallery(col, q[:1], top_k=0)                       # 0 = the whole gallery ranking
    assert len(full[0]["results"]) == n
    assert [h["id"] for h in full[0]["results"]] == OS.rank_all(emb[[5]], emb)[0].tolist()
    one = search_collection(col, emb[77].tolist(), top_k=12)
    assert one == items[1]["results"]
    m = evaluate_results(items, 0.4, [1, 5, 10])
    assert set(m) == {"mAP", "num_queries", "num_valid_ap_queries", "P@1", "R@1", "P@5", "R@5", "P@10", "R@10"}
    json.dumps(items)                                              # what query_nih_zilliz.py writes to disk
