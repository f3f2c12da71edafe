#!/usr/bin/env python3
"""Generate golden vectors for the retrieval tail from the reference's OWN functions.

Run in the build container only (the reference tree is not present on the GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

What it does
  * imports the reference's pure metric functions (test.py:38-223, 941-985;
    evaluate_nih_zilliz.py:12-64; fusion_eval/metrics.py:41-94; fusion_eval/fuse.py:11-52)
    from /root/reference.  `test.py` pulls in torchvision / timm / cv2 at module import
    time although none of the metric functions use them, so inert placeholder modules
    are registered for those names first (SURVEY.md section 8c recipe).  Nothing of the
    reference is copied: only INPUTS and the reference's OUTPUTS are written.
  * feeds seeded synthetic inputs through them and stores inputs + outputs as
    tests/golden/*.npz / *.json.

The fixtures are data, not code; this script is the committed recipe that made them.
"""
import importlib
import json
import os
import sys
import types

import numpy as np

REF = os.environ.get("MIRX_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    m.__path__ = []  # behave like a package for "import a.b"
    sys.modules[name] = m
    return m


def import_reference():
    import torch  # noqa: F401
    import transformers  # noqa: F401  (must be imported before the placeholders exist)

    class _Anything:
        def __init__(self, *a, **k):
            pass

        def __call__(self, *a, **k):
            return self

        def __getattr__(self, k):
            return _Anything()

    for name in ("torchvision", "torchvision.models", "torchvision.transforms",
                 "torchvision.transforms.functional", "timm", "timm.data", "cv2"):
        if name not in sys.modules:
            m = _placeholder(name)
            def _ga(k, _A=_Anything):
                if k.startswith("__"):
                    raise AttributeError(k)
                return _A()
            m.__getattr__ = _ga  # type: ignore[attr-defined]
    sys.modules["torchvision"].models = sys.modules["torchvision.models"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]
    sys.modules["timm"].data = sys.modules["timm.data"]
    sys.path.insert(0, REF)
    ref_test = importlib.import_module("test")
    ref_nih = importlib.import_module("evaluate_nih_zilliz")
    ref_fmetrics = importlib.import_module("fusion_eval.metrics")
    ref_fuse = importlib.import_module("fusion_eval.fuse")
    return ref_test, ref_nih, ref_fmetrics, ref_fuse


def covidx_labels():
    """Labels of the reference's shipped test.txt (col 3), mapped as read_data.py:281-285."""
    mapping = {"normal": 0, "pneumonia": 1, "COVID-19": 2}
    labs = []
    with open(os.path.join(REF, "test.txt")) as fh:
        for line in fh:
            parts = line.split()
            if len(parts) >= 3:
                labs.append(mapping[parts[2]])
    return np.asarray(labs, dtype=np.int64)


def base_ranking(emb, metric):
    """A plain-numpy fp64 ranking (stable argsort, row per query, self last) that the reference's fp32 ranking is stored
    RELATIVE to: tests rebuild the reference ranking as this + the stored patch (tests/_audit.py)."""
    e = np.asarray(emb, dtype=np.float64)
    if metric == "cosine":
        s = e @ e.T
    else:
        s = -np.sqrt(np.maximum(((e[:, None, :] - e[None, :, :]) ** 2).sum(-1), 0.0))
    np.fill_diagonal(s, -np.inf)
    return np.argsort(-s, axis=1, kind="stable")


def class_clustered_embeddings(labels, dim, seed, spread):
    """Unit-norm embeddings with class structure so that metrics are not trivially chance."""
    import torch
    g = torch.Generator().manual_seed(seed)
    ncls = int(labels.max()) + 1
    centers = torch.randn(ncls, dim, generator=g)
    x = centers[torch.as_tensor(labels)] + spread * torch.randn(len(labels), dim, generator=g)
    return torch.nn.functional.normalize(x, dim=1)


def main():
    import torch
    ref_test, ref_nih, ref_fm, ref_fuse = import_reference()
    torch.manual_seed(0)
    out = {}

    # ---- known answers quoted in SURVEY.md 8c (tiny, hand-checkable) --------------------
    ranks = np.array([[1, 0, 0], [2, 2, 1], [0, 1, 2]])
    gnd = np.array([0, 0, 1])
    mAP, aps, pr, prs = ref_test.compute_map(ranks.copy(), gnd, [1, 2])
    small = {
        "compute_map_tiny": {"ranks": ranks.tolist(), "gnd": gnd.tolist(), "kappas": [1, 2],
                             "mAP": float(mAP), "aps": aps.tolist(), "pr": pr.tolist(),
                             "prs": prs.tolist()},
        "compute_ap": [
            {"ranks": [0, 2], "nres": 2, "ap": float(ref_test.compute_ap(np.array([0, 2]), 2))},
            {"ranks": [2], "nres": 1, "ap": float(ref_test.compute_ap(np.array([2]), 1))},
            {"ranks": [0, 1, 2], "nres": 3, "ap": float(ref_test.compute_ap(np.array([0, 1, 2]), 3))},
            {"ranks": [], "nres": 4, "ap": float(ref_test.compute_ap(np.array([], dtype=np.int64), 4))},
            {"ranks": [5, 9, 11], "nres": 7, "ap": float(ref_test.compute_ap(np.array([5, 9, 11]), 7))},
        ],
        "precision_at_k": [
            {"rel": [1, 0, 1, 1], "k": 2, "out": ref_nih.precision_at_k([1, 0, 1, 1], 2)},
            {"rel": [1, 0, 1, 1], "k": 10, "out": ref_nih.precision_at_k([1, 0, 1, 1], 10)},
            {"rel": [], "k": 3, "out": ref_nih.precision_at_k([], 3)},
        ],
        "recall_at_k": [
            {"rel": [1, 0, 1, 1], "tp": 3, "k": 2, "out": ref_nih.recall_at_k([1, 0, 1, 1], 3, 2)},
            {"rel": [1, 0, 1, 1], "tp": 0, "k": 2, "out": ref_nih.recall_at_k([1, 0, 1, 1], 0, 2)},
            {"rel": [0, 0, 1], "tp": 1, "k": 9, "out": ref_nih.recall_at_k([0, 0, 1], 1, 9)},
        ],
        "jaccard_score": [
            {"a": [1, 0, 1], "b": [1, 1, 0], "out": ref_nih.jaccard_score([1, 0, 1], [1, 1, 0])},
            {"a": [0, 0, 0], "b": [0, 0, 0], "out": ref_nih.jaccard_score([0, 0, 0], [0, 0, 0])},
            {"a": [1, 1, 1, 0], "b": [1, 1, 1, 0], "out": ref_nih.jaccard_score([1, 1, 1, 0], [1, 1, 1, 0])},
        ],
        "majority_vote": [
            {"labels": [2, 1, 1, 2], "out": int(ref_test.majority_vote(np.array([2, 1, 1, 2])))},
            {"labels": [0], "out": int(ref_test.majority_vote(np.array([0])))},
            {"labels": [1, 2, 0, 2, 1], "out": int(ref_test.majority_vote(np.array([1, 2, 0, 2, 1])))},
        ],
    }
    e = torch.nn.functional.normalize(torch.randn(12, 8), dim=1)
    lab = torch.tensor([0, 1, 2] * 4)
    d = -torch.cdist(e, e)
    d.fill_diagonal_(float("-inf"))
    acc = [float(a) for a in ref_test.retrieval_accuracy(d, lab, topk=[1, 5, 10])]
    rk = torch.argsort(d, dim=0, descending=True).numpy()
    m12, _, pr12, _ = ref_test.compute_map(rk, lab.numpy(), [1, 5, 10])
    small["seed0_12x8"] = {"acc": acc, "mAP": float(m12), "pr": pr12.tolist()}
    fm = ref_fm.evaluate_retrieval_metrics(
        np.random.default_rng(0).standard_normal((6, 4)).astype(np.float32),
        ["a", "b"] * 3, [f"p{i}" for i in range(6)], (1, 2))
    small["fusion_6x4"] = {k: float(v) for k, v in fm.items()}
    with open(os.path.join(OUT, "known_answers.json"), "w") as fh:
        json.dump(small, fh, indent=1, sort_keys=True)

    # ---- labelled parity sets through the reference metric tail -------------------------
    lab300 = covidx_labels()
    assert lab300.shape == (300,) and np.bincount(lab300).tolist() == [100, 100, 100]
    sets = {
        "covidx300_d64": (class_clustered_embeddings(lab300, 64, 1234, 7.0), lab300),
        "mod3_1000_d32": (class_clustered_embeddings(np.arange(1000) % 3, 32, 99, 5.0),
                          np.arange(1000) % 3),
        "rand_257_d16": (torch.nn.functional.normalize(
            torch.randn(257, 16, generator=torch.Generator().manual_seed(5)), dim=1),
            np.random.default_rng(5).integers(0, 5, 257)),
    }
    for name, (emb, labels) in sets.items():
        labels_t = torch.as_tensor(labels, dtype=torch.int64)
        rec = {"embeds": emb.numpy(), "labels": labels_t.numpy()}
        for metric in ("cdist", "cosine"):
            dists = (emb @ emb.t()) if metric == "cosine" else -torch.cdist(emb, emb)
            dists.fill_diagonal_(float("-inf"))
            kappas = [1, 5, 10]
            acc = torch.stack(ref_test.retrieval_accuracy(dists, labels_t, topk=kappas)).numpy()
            ranks = torch.argsort(dists, dim=0, descending=True).numpy()
            mAP, aps, pr, prs = ref_test.compute_map(ranks, labels_t.numpy(), kappas)
            cls = ref_test.compute_classification_metrics(labels_t, dists, [1, 5, 10, 15, 20])
            rec[f"{metric}_acc"] = acc
            rec[f"{metric}_mAP"] = np.float64(mAP)
            rec[f"{metric}_aps"] = aps
            rec[f"{metric}_pr"] = pr
            rec[f"{metric}_prs"] = prs
            if len(labels) <= 300:  # reference fp32 ranking, stored whole
                rec[f"{metric}_ranks_ref_fp32"] = ranks.astype(np.int16)
            # ... and for every set as a patch against base_ranking(): rows (query, position, id the reference has there)
            base = base_ranking(emb.numpy(), metric)
            qq, pp = np.nonzero(ranks.T != base)
            rec[f"{metric}_refpatch"] = np.stack([qq, pp, ranks.T[qq, pp]], axis=1).astype(np.int32)
            for k, v in cls.items():
                rec[f"{metric}_cls_k{k}"] = np.array(list(v.values()), dtype=np.float64)
        # fusion_eval metrics (string labels, self excluded by path)
        fm = ref_fm.evaluate_retrieval_metrics(
            emb.numpy(), [str(x) for x in labels], [f"img_{i}.png" for i in range(len(labels))],
            (1, 5, 10))
        rec["fusion_keys"] = np.array(sorted(fm.keys()))
        rec["fusion_vals"] = np.array([fm[k] for k in sorted(fm.keys())], dtype=np.float64)
        np.savez_compressed(os.path.join(OUT, f"tail_{name}.npz"), **rec)

    # ---- evaluate() end to end with a stand-in model (test.py:1065-1126) ----------------
    import tempfile

    class _Lookup(torch.nn.Module):
        def __init__(self, table):
            super().__init__()
            self.table = table

        def forward(self, idx):
            return self.table[idx]

    emb, labels = sets["covidx300_d64"]
    loader = [(torch.arange(i, min(i + 64, 300)), torch.as_tensor(labels[i:i + 64]))
              for i in range(0, 300, 64)]
    with tempfile.TemporaryDirectory() as td:
        args = types.SimpleNamespace(save_dir=td, resume="ckpt/model_x.pth")
        ref_test.evaluate(_Lookup(emb), loader, torch.device("cpu"), args)
        z = np.load(os.path.join(td, "model_x.npz"))
        np.savez_compressed(os.path.join(OUT, "evaluate_covidx300_d64.npz"),
                            **{k: z[k] for k in z.files})

    # ---- multilabel AP (test.py:941-985) and NIH result-list metrics --------------------
    rng = np.random.default_rng(7)
    ml_labels = (rng.random((120, 14)) < 0.18).astype(np.float32)
    ml_emb = torch.nn.functional.normalize(
        torch.as_tensor(ml_labels @ rng.standard_normal((14, 24)).astype(np.float32)
                        + 0.7 * rng.standard_normal((120, 24)).astype(np.float32)), dim=1)
    ml_d = ml_emb @ ml_emb.t()
    ml_d.fill_diagonal_(float("-inf"))
    ml_map = ref_test.compute_map_multilabel(ml_d, torch.as_tensor(ml_labels), 0.5)
    ml_map04 = ref_test.compute_map_multilabel(ml_d, torch.as_tensor(ml_labels), 0.4)
    items = []
    sim = (ml_emb @ ml_emb.t()).numpy()
    for qi in range(0, 120, 3):
        order = np.argsort(-sim[qi])
        order = order[order != qi][:25]
        items.append({"query_label_vector": ml_labels[qi].tolist(),
                      "results": [{"score": float(sim[qi, j]),
                                   "label_vector": ml_labels[j].tolist()} for j in order]})
    nih = ref_nih.evaluate_results(items, 0.4, [1, 5, 10, 20])
    np.savez_compressed(os.path.join(OUT, "multilabel_120.npz"), embeds=ml_emb.numpy(),
                        labels=ml_labels, map_t05=np.float64(ml_map), map_t04=np.float64(ml_map04))
    with open(os.path.join(OUT, "nih_results_40.json"), "w") as fh:
        json.dump({"items": items, "threshold": 0.4, "ks": [1, 5, 10, 20], "metrics": nih}, fh)

    # ---- late fusion (fusion_eval/fuse.py) ----------------------------------------------
    a = rng.standard_normal((50, 12)).astype(np.float32)
    b = rng.standard_normal((50, 12)).astype(np.float32)
    c = rng.standard_normal((50, 20)).astype(np.float32)
    np.savez_compressed(
        os.path.join(OUT, "fusion_fuse.npz"), a=a, b=b, c=c,
        l2=ref_fuse.l2_normalize(a), concat=ref_fuse.concat_fusion(a, c),
        wsum03=ref_fuse.weighted_sum_fusion(a, b, 0.3).embeddings,
        wsum_mismatch_reason=np.array(ref_fuse.weighted_sum_fusion(a, c, 0.5).skipped_reason))
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
