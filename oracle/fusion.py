"""CPU restatement of the reference's late-fusion evaluation (test infrastructure only).

Follows fusion_eval/evaluate.py:30-214 (run_late_fusion_experiments, normalize_similarity_matrix,
confidence_based_fusion, top12_margin), fusion_eval/metrics.py:12-22 (similarity, ranking) and
fusion_eval/align.py:96-230 (file sources, alignment) of /root/reference, in plain numpy with the
reference's dtypes (fp32 similarities).  Pinned by tests/golden/fusion_experiments.{npz,json} and
fusion_align.json, which the reference's own functions produced (tests/golden/make_golden_fusion.py).
"""
import json

import numpy as np

from . import metrics as om


# -- fusion_eval/metrics.py:12-15 --------------------------------------------------------------
def similarity_matrix(embeddings):
    x = om.l2_normalize_np(np.asarray(embeddings).astype(np.float32))
    return x @ x.T


# -- fusion_eval/metrics.py:18-22 (argsort there is unstable; ties -> lowest index here) ------
def rank_rows(similarity):
    s = np.array(similarity, copy=True)
    np.fill_diagonal(s, -np.inf)
    return np.argsort(-s, axis=1, kind="stable")


# -- fusion_eval/evaluate.py:150-177 -----------------------------------------------------------
def normalize_similarity(similarity, mode="none"):
    s = np.asarray(similarity).astype(np.float32, copy=True)
    if mode == "none":
        return s
    diag = np.diag(s).copy()
    if mode == "zscore":
        mu = np.mean(s, axis=1, keepdims=True)
        sd = np.maximum(np.std(s, axis=1, keepdims=True), 1e-12)
        out = (s - mu) / sd
    elif mode == "minmax":
        lo = np.min(s, axis=1, keepdims=True)
        hi = np.max(s, axis=1, keepdims=True)
        out = (s - lo) / np.maximum(hi - lo, 1e-12)
    else:
        raise ValueError(f"Unsupported score normalization mode: {mode}. Use one of: none, zscore, minmax")
    np.fill_diagonal(out, diag)
    return out


# -- fusion_eval/evaluate.py:206-214 -----------------------------------------------------------
def top12_margin(similarity):
    if similarity.shape[1] < 2:
        raise ValueError("Need at least two gallery scores per query for confidence margin")
    srt = np.sort(similarity, axis=1)
    return srt[:, -1] - srt[:, -2]


# -- fusion_eval/evaluate.py:180-203 -----------------------------------------------------------
def confidence_fusion(conv_similarity, dino_similarity):
    if conv_similarity.shape != dino_similarity.shape:
        raise ValueError("Conv and DINO similarity matrices must have the same shape")
    c = conv_similarity.astype(np.float32, copy=True)
    d = dino_similarity.astype(np.float32, copy=True)
    np.fill_diagonal(c, -np.inf)
    np.fill_diagonal(d, -np.inf)
    mc, md = top12_margin(c), top12_margin(d)
    alpha = mc / (mc + md + 1e-8)
    fused = alpha[:, None] * c + (1.0 - alpha[:, None]) * d
    return {"similarity": fused, "conv_selected_queries": int(np.sum(alpha >= 0.5)),
            "dino_selected_queries": int(np.sum(alpha < 0.5)), "alpha": alpha,
            "alpha_mean": float(np.mean(alpha)), "alpha_std": float(np.std(alpha))}


def metrics_from_similarity(similarity, labels, paths, k_values):
    return om.fusion_metrics_from_ranks(rank_rows(similarity), labels, paths, k_values)


# -- fusion_eval/evaluate.py:30-147 ------------------------------------------------------------
def run_experiments(conv, dino, labels, paths, alpha_values=(0.2, 0.4, 0.5, 0.6, 0.8), k_values=(1, 5, 10),
                    score_normalization="none"):
    """-> list of dicts {experiment_name, num_samples, metrics, skipped, skipped_reason} in the
    reference's order: three baselines, score fusion per alpha, confidence fusion, weighted sum per alpha."""
    n = len(paths)
    out = []

    def add(name, metrics, skipped=False, reason=None):
        out.append({"experiment_name": name, "num_samples": n, "metrics": metrics, "skipped": skipped,
                    "skipped_reason": reason})

    cb, db = om.l2_normalize_np(conv), om.l2_normalize_np(dino)
    for name, emb in (("convnext_baseline", cb), ("dino_baseline", db),
                      ("concat_fusion", om.concat_fusion(conv, dino))):
        add(name, metrics_from_similarity(similarity_matrix(emb), labels, paths, k_values))
    cs = normalize_similarity(similarity_matrix(cb), score_normalization)
    ds = normalize_similarity(similarity_matrix(db), score_normalization)
    for alpha in alpha_values:
        add(f"score_fusion_alpha_{alpha:.1f}", metrics_from_similarity(alpha * cs + (1.0 - alpha) * ds, labels, paths,
                                                                       k_values))
    conf = confidence_fusion(cs, ds)
    m = metrics_from_similarity(conf["similarity"], labels, paths, k_values)
    m["conv_selected_queries"] = float(conf["conv_selected_queries"])
    m["dino_selected_queries"] = float(conf["dino_selected_queries"])
    add("confidence_fusion_top12_margin", m)
    for alpha in alpha_values:
        if conv.shape[1] != dino.shape[1]:
            add(f"weighted_sum_alpha_{alpha:.1f}", {}, True,
                f"weighted_sum_skipped_dimension_mismatch: conv_dim={conv.shape[1]}, dino_dim={dino.shape[1]}")
            continue
        fused = om.weighted_sum_fusion(conv, dino, alpha)
        add(f"weighted_sum_alpha_{alpha:.1f}", metrics_from_similarity(similarity_matrix(fused), labels, paths, k_values))
    return out


# -- fusion_eval/align.py:96-140 ---------------------------------------------------------------
def read_embedding_file(path):
    """-> list of (image_path, label, embedding fp32) in file order."""
    path = str(path)
    if path.lower().endswith(".json"):
        with open(path, "r", encoding="utf-8") as fh:
            data = json.load(fh)
        rows = data.get("records", data) if isinstance(data, dict) else data
        return [(r["image_path"], r.get("label"), np.asarray(r["embedding"], dtype=np.float32)) for r in rows]
    if path.lower().endswith(".npz"):
        z = np.load(path, allow_pickle=True)
        paths = z["image_paths"].tolist()
        labels = z["labels"].tolist() if "labels" in z else [None] * len(paths)
        return [(p, l, np.asarray(e, dtype=np.float32)) for p, l, e in zip(paths, labels, z["embeddings"])]
    raise ValueError(f"Unsupported embedding file format: {path}")


# -- fusion_eval/align.py:155-230 --------------------------------------------------------------
def align_records(conv_records, dino_records, strict_label_check=True):
    def index(records, name):
        d = {}
        for p, l, e in records:
            if p in d:
                raise ValueError(f"Duplicate image_path found in {name}: {p}")
            d[p] = (l, e)
        return d

    c, d = index(conv_records, "ConvNeXt"), index(dino_records, "DINO")
    coverage = {"present_in_conv_only": sorted(set(c) - set(d)), "present_in_dino_only": sorted(set(d) - set(c)),
                "present_in_both": sorted(set(c) & set(d))}
    paths, labels, ce, de = [], [], [], []
    for p in coverage["present_in_both"]:
        if strict_label_check and c[p][0] != d[p][0]:
            raise ValueError(f"Label mismatch for image_path={p}: conv={c[p][0]!r}, dino={d[p][0]!r}")
        paths.append(p)
        labels.append(c[p][0] or d[p][0] or "unknown")
        ce.append(c[p][1])
        de.append(d[p][1])
    if not paths:
        raise ValueError("No aligned samples found across the requested sources")
    return {"image_paths": paths, "labels": labels, "conv_embeddings": np.stack(ce).astype(np.float32),
            "dino_embeddings": np.stack(de).astype(np.float32), "coverage": coverage}


# -- test.py:599-623 (loops kept as the reference writes them; fp64 scores, ties -> lowest id) -------------
def text_rerank_dists(embeds, concept_image_embeds, text_embeds, labels, rerank_k, text_weight):
    e = np.asarray(embeds, dtype=np.float64)
    img_sim = e @ e.T
    img_text_sim = np.asarray(concept_image_embeds, dtype=np.float64) @ np.asarray(text_embeds, dtype=np.float64).T
    dists = img_sim.copy()
    alpha = float(text_weight)
    beta = 1.0 - alpha
    n = len(labels)
    for i in range(n):
        top = np.argsort(-img_sim[i], kind="stable")[: min(int(rerank_k), n)]
        for j in top:
            if i != j:
                dists[i, j] = alpha * img_sim[i, j] + beta * img_text_sim[j, labels[i]]
    np.fill_diagonal(dists, -np.inf)
    return dists


# -- test.py:625-638: R@K from rows (topk dim 1), mAP / mP@K from the ranking of COLUMNS (argsort dim 0) ------
def text_rerank_metrics(dists, labels, kappas=(1, 5, 10)):
    labels = np.asarray(labels)
    order = np.argsort(-dists, axis=1, kind="stable")[:, : max(kappas)]
    correct = labels[order] == labels[:, None]
    acc = np.array([100.0 * np.count_nonzero(correct[:, :k].any(axis=1)) / len(labels) for k in kappas])
    ranks = np.argsort(-dists, axis=0, kind="stable")
    m_ap, aps, pr, prs = om.compute_map(ranks, labels, list(kappas))
    return acc, m_ap, pr
