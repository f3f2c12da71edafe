"""evaluate() drop-in and the Milvus-shaped retriever on the GPU against golden outputs of the
reference's evaluate() (tests/golden/evaluate_covidx300_d64.npz) and the oracle."""
import os
import types

import numpy as np
import pytest
import torch

from oracle import metrics as OM
from oracle import search as OS

pytestmark = pytest.mark.gpu


class _Lookup(torch.nn.Module):
    def __init__(self, table):
        super().__init__()
        self.register_buffer("table", table)

    def forward(self, idx):
        return self.table[idx]


def test_evaluate_matches_reference_npz(golden_dir, tmp_path, capsys):
    from mirx.evaluate import evaluate
    z = np.load(os.path.join(golden_dir, "evaluate_covidx300_d64.npz"))
    emb, labels = torch.as_tensor(z["embeds"]), torch.as_tensor(z["labels"])
    loader = [(torch.arange(i, min(i + 64, 300)), labels[i:i + 64]) for i in range(0, 300, 64)]
    args = types.SimpleNamespace(save_dir=str(tmp_path), resume="ckpt/model_x.pth")
    res = evaluate(_Lookup(emb).cuda(), loader, torch.device("cuda:0"), args)
    out = np.load(tmp_path / "model_x.npz")
    assert set(z.files) == set(out.files)
    np.testing.assert_array_equal(out["embeds"], z["embeds"])
    np.testing.assert_array_equal(out["labels"], z["labels"])
    np.testing.assert_array_equal(out["kappas"], z["kappas"])
    np.testing.assert_array_equal(out["classification_k_values"], z["classification_k_values"])
    # the ranking is the fp64 oracle's, bit for bit ...
    ranks = OS.rank_all(z["embeds"], z["embeds"], metric=OS.METRIC_NEG_L2, exclude=np.arange(300))
    np.testing.assert_array_equal(res["ranks"].cpu().numpy(), ranks)
    mAP, aps, pr, prs = OM.compute_map(ranks.T, z["labels"], [1, 5, 10])
    assert float(out["mAP"]) == pytest.approx(mAP, abs=1e-12)
    np.testing.assert_allclose(out["pr"], pr, atol=1e-12)
    # ... and differs from the reference's own fp32 cdist/argsort ranking only inside audited near-ties (fp64 score gap
    # < 1e-6, tests/_audit.py); with those swapped back the product's metric tail gives the reference's numbers to 1e-12
    from _audit import audit, reference_ranking
    from mirx.metrics import compute_classification_metrics, compute_map, retrieval_accuracy
    tail = np.load(os.path.join(golden_dir, "tail_covidx300_d64.npz"))
    np.testing.assert_array_equal(tail["embeds"], z["embeds"])
    ref = reference_ranking(tail, "cdist")
    flipped = audit(ranks, ref, z["embeds"], "cdist")
    assert len(flipped) <= 3
    r_map, r_aps, r_pr, _ = compute_map(torch.as_tensor(ref).cuda().t(), z["labels"], [1, 5, 10])     # device tail
    assert r_map == pytest.approx(float(z["mAP"]), abs=1e-12)
    np.testing.assert_allclose(r_pr, z["pr"], atol=1e-12)
    same = np.setdiff1d(np.arange(300), flipped)
    np.testing.assert_allclose(aps[same], r_aps[same], atol=1e-12)            # un-flipped queries: far inside 1e-5
    assert float(out["mAP"]) == pytest.approx(float(z["mAP"]) + float(np.sum(aps[flipped] - r_aps[flipped])) / 300, abs=1e-12)
    r_acc = torch.stack(retrieval_accuracy(None, z["labels"], topk=[1, 5, 10], topk_ids=ref[:, :10])).numpy()
    np.testing.assert_allclose(r_acc, z["acc"], atol=1e-5)
    np.testing.assert_allclose(out["acc"], z["acc"], atol=100.0 * len(flipped) / 300 + 1e-4)
    r_cls = compute_classification_metrics(z["labels"], None, [1, 5, 10, 15, 20], ranks=ref[:, :20].T)
    for k in (1, 5, 10, 15, 20):
        np.testing.assert_allclose(np.array(list(r_cls[k].values())), z[f"classification_k{k}"], atol=1e-9)
        changed = np.any(ranks[:, :k] != ref[:, :k], axis=1).sum()             # queries whose top-k changed at all
        if changed == 0:
            np.testing.assert_allclose(out[f"classification_k{k}"], z[f"classification_k{k}"], atol=1e-9)
    # dists: +L2 with +inf diagonal, same as the reference's saved matrix up to fp32 rounding
    assert np.all(np.isposinf(np.diag(out["dists"])))
    off = ~np.eye(300, dtype=bool)
    np.testing.assert_allclose(out["dists"][off], z["dists"][off], atol=2e-5)
    text = capsys.readouterr().out
    assert ">> R@K[1, 5, 10]:" in text and ">> mAP:" in text and ">> Top-20 Retrieved Images:" in text


def test_retriever_protocol(tmp_path):
    from PIL import Image
    from mirx.retriever import MilvusManager, MilvusRetriever, default_transform, search_collection
    mgr = MilvusManager(dataset="covid")
    assert mgr.connect() is True
    with pytest.raises(ValueError):
        mgr.create_collection("nope")
    with pytest.raises(ValueError):
        mgr.load_collection("densenet121")                     # does not exist yet
    col = mgr.create_collection("densenet121", drop_old=True)
    assert col.name == "covid_image_retrieval_densenet121"
    mgr.create_index("densenet121", metric_type="COSINE")
    g = torch.nn.functional.normalize(torch.randn(500, 1024, generator=torch.Generator().manual_seed(1)), dim=1)
    paths = [f"/data/img_{i}.png" for i in range(500)]
    labels = [["normal", "pneumonia", "COVID-19"][i % 3] for i in range(500)]
    for s in range(0, 500, 100):                                 # ingest_embeddings.py:399-411 batches of 100
        col.insert([paths[s:s + 100], labels[s:s + 100], g[s:s + 100].numpy().tolist()])
    col.flush()
    assert mgr.get_collection_info("densenet121")["num_entities"] == 500

    class _Fake(torch.nn.Module):                                # embeds an image to a fixed gallery row
        def __init__(self):
            super().__init__()
            self.p = torch.nn.Parameter(torch.zeros(1))

        def forward(self, x):
            idx = (x.flatten(1).abs().sum(1) * 0).long() + 7
            return g.to(x.device)[idx] * 3.0                     # un-normalised on purpose

    img = Image.fromarray((np.random.default_rng(0).random((300, 260, 3)) * 255).astype(np.uint8))
    p = tmp_path / "q.png"
    img.save(p)
    tf = default_transform(224)
    assert tf(img).shape == (3, 224, 224)
    r = MilvusRetriever(mgr, "densenet121", _Fake().cuda(), tf)
    r.load_collection()
    results, qemb = r.search(str(p), top_k=5, metric_type="COSINE")
    assert qemb.shape == (1, 1024) and qemb.is_cuda
    assert [set(d) for d in results][0] == {"id", "image_path", "label", "distance", "similarity"}
    assert results[0]["id"] == 7 and results[0]["image_path"] == "/data/img_7.png" and results[0]["label"] == labels[7]
    assert results[0]["distance"] == pytest.approx(1.0, abs=1e-6) and results[0]["similarity"] == results[0]["distance"]
    o_s, o_i = OS.topk(g[7:8].numpy(), g.numpy(), 5)
    assert [d["id"] for d in results] == o_i[0].tolist()
    assert [d["distance"] for d in results] == pytest.approx(o_s[0].tolist(), abs=1e-6)
    batch = r.batch_search([str(p), img], top_k=3)
    assert len(batch) == 2 and [d["id"] for d in batch[0]] == o_i[0][:3].tolist()
    hits = search_collection(col, g[3].tolist(), top_k=4)
    assert hits[0]["id"] == 3 and "score" in hits[0] and hits[0]["image_path"] == "/data/img_3.png"
    with pytest.raises(ValueError):
        col.insert([["a"], ["b"], [[0.0] * 7]])
    mgr.disconnect()


def test_single_query_search_hands_8bit_pixels_to_a_model_that_normalises_them(tmp_path):
    """milvus_retrieval.py:53-66 embeds ONE image per call.  With default_transform and a DenseNet121 (which applies ToTensor +
    Normalize inside its stem kernel) MilvusRetriever.search sends the 8-bit pixels: the query embedding must be the float
    path's, bit for bit; a transform with other constants, or another model, keeps the float path."""
    from PIL import Image
    from mirx.model import DenseNet121
    from mirx.retriever import MilvusManager, MilvusRetriever, SIGLIP_MEAN, SIGLIP_STD, default_transform
    torch.manual_seed(0)
    m = DenseNet121().eval().cuda()
    mgr = MilvusManager(dataset="covid")
    mgr.connect()
    mgr.create_collection("densenet121", drop_old=True)
    col = mgr.collections["densenet121"]
    g = torch.nn.functional.normalize(torch.randn(300, 1024, generator=torch.Generator().manual_seed(5)), dim=1)
    col.insert([[f"/d/{i}.png" for i in range(300)], ["normal"] * 300, g])
    img = Image.fromarray((np.random.default_rng(1).random((280, 320, 3)) * 255).astype(np.uint8))
    tf = default_transform(224)
    r = MilvusRetriever(mgr, "densenet121", m, tf)
    assert r._query_tensor(img).dtype == torch.uint8
    res, qemb = r.search(img, top_k=5)
    want = r.embed(tf(img).unsqueeze(0))
    assert torch.equal(qemb, want)
    res2, _ = MilvusRetriever(mgr, "densenet121", m, lambda im: tf(im)).search(img, top_k=5)     # a plain callable: float path
    assert [d["id"] for d in res] == [d["id"] for d in res2]
    other = default_transform(224, SIGLIP_MEAN, SIGLIP_STD)
    assert MilvusRetriever(mgr, "densenet121", m, other)._query_tensor(img).dtype == torch.float32
    mgr.disconnect()


def test_search_by_embeddings_reference_signature():
    """retrieval_analysis/milvus_adapter.py:218-275: (queries, query_embeddings, top_k, search_params, reranker,
    exclude_self, metadata_fields, batch_size) -> list[SearchResult]; self dropped by image_path after fetching
    top_k + 1; the reranker hook runs before the cut; ids equal the oracle's exact ranking."""
    from mirx.adapter import IdentityReranker, QueryRecord, RetrievedItem, SearchResult
    from mirx.retriever import MilvusManager, MilvusRetriever
    mgr = MilvusManager(dataset="isic")
    assert mgr.connect()
    col = mgr.create_collection("convnextv2", drop_old=True)
    n, d = 3000, 1024
    g = torch.nn.functional.normalize(torch.randn(n, d, generator=torch.Generator().manual_seed(5)), dim=1)
    paths = [f"/isic/im_{i}.jpg" for i in range(n)]
    labels = [f"c{i % 7}" for i in range(n)]
    col.insert([paths, labels, g])
    r = MilvusRetriever(mgr, "convnextv2", None, None)
    qi = [5, 17, 2999, 1234, 42]
    queries = [QueryRecord(image_path=paths[i], label=labels[i]) for i in qi]
    emb = g[qi].numpy().tolist()
    res = r.search_by_embeddings(queries, emb, top_k=6, search_params={"params": {"nprobe": 10}}, batch_size=2)
    assert len(res) == len(qi) and all(isinstance(x, SearchResult) for x in res)
    o_s, o_i = OS.topk(g[qi].numpy(), g.numpy(), 6, exclude=np.array(qi))
    for x, q, ids, sc, e in zip(res, queries, o_i, o_s, emb):
        assert x.query is q and x.query_source == "convnextv2" and x.query_embedding == e
        assert all(isinstance(it, RetrievedItem) for it in x.retrieved)
        assert [it.id for it in x.retrieved] == ids.tolist()
        assert [it.image_path for it in x.retrieved] == [paths[i] for i in ids]
        assert [it.label for it in x.retrieved] == [labels[i] for i in ids]
        assert [it.score for it in x.retrieved] == pytest.approx(sc.tolist(), abs=1e-6)
        assert all(it.distance == it.score and it.raw["id"] == it.id for it in x.retrieved)
    # exclude_self=False keeps the query's own row first
    keep = r.search_by_embeddings(queries[:1], emb[:1], top_k=3, exclude_self=False)[0]
    assert keep.retrieved[0].id == qi[0] and len(keep.retrieved) == 3

    class _Reverse:                                   # the hook sees the self-filtered top_k + 1 list, cut comes after
        calls = []

        def rerank(self, query, results):
            self.calls.append((query.image_path, len(results)))
            return list(results)[::-1]

    rr = _Reverse()
    rev = r.search_by_embeddings(queries[:2], emb[:2], top_k=6, reranker=rr)
    assert rr.calls == [(paths[5], 6), (paths[17], 6)]
    assert [it.id for it in rev[0].retrieved] == o_i[0].tolist()[::-1]
    same = r.search_by_embeddings(queries[:2], emb[:2], top_k=6, reranker=IdentityReranker())
    assert [it.id for it in same[1].retrieved] == o_i[1].tolist()
    assert r.search_by_embeddings([], [], top_k=3) == []
    with pytest.raises(ValueError):
        r.search_by_embeddings(queries, emb[:2], top_k=3)
    ad = r.adapter()
    assert ad.fetch_record_by_image_path(paths[9])["label"] == labels[9]
    np.testing.assert_allclose(ad.fetch_record_by_image_path(paths[9])["embedding"], g[9].numpy(), atol=0)
    got = ad.fetch_records_by_image_paths([paths[3], paths[2500], "/nope"], include_embedding=True)
    assert set(got) == {paths[3], paths[2500]} and np.array_equal(got[paths[2500]]["embedding"], g[2500].numpy())
    assert ad.fetch_record_by_image_path("/nope") is None and len(ad.list_image_paths()) == n
    one = ad.search_by_embedding(queries[3], emb[3], top_k=4)
    assert [it.id for it in one.retrieved] == o_i[3][:4].tolist()
    # a metric that differs from the collection's is rejected like Milvus does
    with pytest.raises(ValueError):
        MilvusRetriever(mgr, "convnextv2", torch.nn.Identity(), lambda im: torch.zeros(3, 4, 4)).search(
            __import__("PIL.Image", fromlist=["x"]).new("RGB", (8, 8)), top_k=3, metric_type="L2")
