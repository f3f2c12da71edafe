"""Loop-level CPU restatement of the reference's retrieval metrics (test infrastructure only).

Each function names the reference lines it follows (paths relative to /root/reference).
They are written for obviousness, not speed; the product versions in the package
(`metrics.py`) are vectorised and are tested against these and against the golden
vectors the reference's own functions produced (tests/golden/).
"""
from collections import Counter

import numpy as np


# -- test.py:38-54 ---------------------------------------------------------------------------
def retrieval_accuracy(topk_ids, labels, ks=(1,)):
    """R@k in percent.  `topk_ids[i]` = ranked gallery ids of query i (self already excluded).

    The reference takes the score matrix and calls topk itself; the ranking is the search
    oracle's job here, so this takes ids.  Returned as float32 like the reference's tensors.
    """
    labels = np.asarray(labels)
    topk_ids = np.asarray(topk_ids)
    nq = len(labels)
    out = []
    for k in ks:
        hit = 0
        for i in range(nq):
            if np.any(labels[topk_ids[i, :k]] == labels[i]):
                hit += 1
        out.append(np.float32(np.float32(hit) * np.float32(100.0 / nq)))
    return out


# -- test.py:58-92 ---------------------------------------------------------------------------
def compute_ap(pos_ranks, npos):
    """Trapezoidal AP from the 0-based ranks of the positives."""
    ap = 0
    step = 1.0 / npos
    for j, r in enumerate(pos_ranks):
        r = int(r)
        p_before = 1.0 if r == 0 else float(j) / r
        p_after = float(j + 1) / (r + 1)
        ap += (p_before + p_after) * step / 2.0
    return ap


# -- test.py:95-146 --------------------------------------------------------------------------
def compute_map(ranks, gnd, kappas=()):
    """mAP / mean precision@kappa with the reference's quirks kept:

    * `ranks` is [db, nq]: column i is query i's ranking (test.py:107,129);
    * the positives of query i are ALL items with its label, the query itself included
      (test.py:119) -- it sits at the last rank because its score was -inf;
    * precision@kappa divides by min(max 1-based positive rank, kappa) (test.py:139).
    """
    ranks = np.asarray(ranks)
    gnd = np.asarray(gnd)
    nq = len(gnd)
    aps = np.zeros(nq)
    prs = np.zeros((nq, len(kappas)))
    pr = np.zeros(len(kappas))
    total = 0.0
    nempty = 0
    for i in range(nq):
        positives = set(np.flatnonzero(gnd == gnd[i]).tolist())
        if not positives:
            aps[i] = np.nan
            prs[i, :] = np.nan
            nempty += 1
            continue
        pos = np.array([r for r in range(ranks.shape[0]) if int(ranks[r, i]) in positives],
                       dtype=np.int64)
        ap = compute_ap(pos, len(positives))
        total += ap
        aps[i] = ap
        pos1 = pos + 1
        for j, kap in enumerate(kappas):
            kq = min(int(pos1.max()), kap)
            prs[i, j] = np.count_nonzero(pos1 <= kq) / kq
        pr = pr + prs[i, :]
    return total / (nq - nempty), aps, pr / (nq - nempty), prs


# -- test.py:149-162 -------------------------------------------------------------------------
def majority_vote(retrieved_labels):
    """Most common label; on a count tie the label seen first (best ranked) wins."""
    if len(retrieved_labels) == 0:
        return None
    return Counter(list(retrieved_labels)).most_common(1)[0][0]


# -- test.py:165-223 -------------------------------------------------------------------------
def compute_classification_metrics(labels, ranks_qmajor, k_values=(1, 5, 10, 15, 20)):
    """Majority-vote classification from the top-k of each query's ranking.

    `ranks_qmajor[i]` = ranked ids of query i.  Returns {k: [precision_macro, recall_macro,
    f1_macro, precision_weighted, recall_weighted, f1_weighted, accuracy]} in percent, the
    value order of the reference's result dict (test.py:209-217).  sklearn's definitions
    (zero_division=0) are restated directly.
    """
    labels = np.asarray(labels)
    res = {}
    for k in k_values:
        pred = np.array([majority_vote(labels[np.asarray(ranks_qmajor[i][:k])])
                         for i in range(len(labels))])
        present = np.unique(np.concatenate([labels, pred]))
        P, R, F, W = [], [], [], []
        for c in present:
            tp = np.count_nonzero((pred == c) & (labels == c))
            fp = np.count_nonzero((pred == c) & (labels != c))
            fn = np.count_nonzero((pred != c) & (labels == c))
            p = tp / (tp + fp) if tp + fp else 0.0
            r = tp / (tp + fn) if tp + fn else 0.0
            f = 2 * p * r / (p + r) if p + r else 0.0
            P.append(p); R.append(r); F.append(f); W.append(np.count_nonzero(labels == c))
        P, R, F, W = map(np.asarray, (P, R, F, W))
        acc = np.count_nonzero(pred == labels) / len(labels)
        res[k] = [100.0 * P.mean(), 100.0 * R.mean(), 100.0 * F.mean(),
                  100.0 * (P * W).sum() / W.sum(), 100.0 * (R * W).sum() / W.sum(),
                  100.0 * (F * W).sum() / W.sum(), 100.0 * acc]
    return res


# -- evaluate_nih_zilliz.py:12-31 ------------------------------------------------------------
def jaccard_score(a, b):
    a = np.asarray(a, dtype=np.float32)
    b = np.asarray(b, dtype=np.float32)
    inter = float((a * b).sum())
    union = float(np.clip(a + b, 0.0, 1.0).sum())
    return inter / (union + 1e-8)


def precision_at_k(rel, k):
    if not len(rel):
        return 0.0
    k = min(k, len(rel))
    return float(np.mean(rel[:k]))


def recall_at_k(rel, total_pos, k):
    if total_pos <= 0:
        return 0.0
    k = min(k, len(rel))
    return float(np.sum(rel[:k]) / total_pos)


def average_precision_ranked(rel, scores):
    """sklearn.average_precision_score(rel, scores) for binary rel: step-wise AP over
    DISTINCT score thresholds (tied scores form one threshold)."""
    rel = np.asarray(rel, dtype=np.float64)
    scores = np.asarray(scores, dtype=np.float64)
    order = np.argsort(-scores, kind="stable")
    rel, scores = rel[order], scores[order]
    npos = rel.sum()
    ap, tp, prev_recall = 0.0, 0.0, 0.0
    i = 0
    n = len(rel)
    while i < n:
        j = i
        while j < n and scores[j] == scores[i]:
            tp += rel[j]
            j += 1
        recall = tp / npos
        precision = tp / j
        ap += (recall - prev_recall) * precision
        prev_recall = recall
        i = j
    return ap


# -- evaluate_nih_zilliz.py:34-64 ------------------------------------------------------------
def evaluate_results(items, jaccard_threshold, ks):
    aps = []
    pk = {k: [] for k in ks}
    rk = {k: [] for k in ks}
    for it in items:
        hits = it["results"]
        rel = [1.0 if jaccard_score(it["query_label_vector"], h["label_vector"]) > jaccard_threshold
               else 0.0 for h in hits]
        tp = int(sum(rel))
        if tp > 0:
            aps.append(average_precision_ranked(rel, [h["score"] for h in hits]))
        for k in ks:
            pk[k].append(precision_at_k(rel, k))
            rk[k].append(recall_at_k(rel, tp, k))
    out = {"mAP": float(np.mean(aps) * 100.0) if aps else 0.0,
           "num_queries": float(len(items)), "num_valid_ap_queries": float(len(aps))}
    for k in ks:
        out[f"P@{k}"] = float(np.mean(pk[k]) * 100.0) if pk[k] else 0.0
        out[f"R@{k}"] = float(np.mean(rk[k]) * 100.0) if rk[k] else 0.0
    return out


# -- fusion_eval/metrics.py:41-94 ------------------------------------------------------------
def fusion_metrics_from_ranks(ranks_qmajor, labels, image_paths, k_values=(1, 5, 10)):
    """`ranks_qmajor[i]` = full ranking of query i with the self-match anywhere in it (the
    reference removes it by image path, fusion_eval/metrics.py:66)."""
    labels = np.asarray(labels)
    paths = np.asarray(image_paths)
    ks = sorted(set(int(k) for k in k_values))
    aps = []
    pk = {k: [] for k in ks}
    rk = {k: [] for k in ks}
    for qi in range(len(labels)):
        order = np.asarray(ranks_qmajor[qi])
        order = order[paths[order] != paths[qi]]
        rel = labels[order] == labels[qi]
        nrel = int(np.sum(labels == labels[qi]) - 1)
        if nrel <= 0:
            aps.append(0.0)
            for k in ks:
                pk[k].append(0.0)
                rk[k].append(0.0)
            continue
        hits = 0
        acc = 0.0
        for pos, r in enumerate(rel):
            if r:
                hits += 1
                acc += hits / (pos + 1)
        aps.append(acc / nrel if hits else 0.0)
        for k in ks:
            h = int(np.sum(rel[:k]))
            pk[k].append(h / k)
            rk[k].append(1.0 if h > 0 else 0.0)
    out = {"num_samples": float(len(labels)), "mAP": float(np.mean(aps) * 100.0)}
    for k in ks:
        out[f"mP@{k}"] = float(np.mean(pk[k]) * 100.0)
        out[f"R@{k}"] = float(np.mean(rk[k]) * 100.0)
    return out


# -- test.py:941-985 -------------------------------------------------------------------------
def compute_map_multilabel(ranks_qmajor, labels, threshold=0.5):
    """Jaccard-thresholded AP over the full ranking; the query itself is never relevant."""
    labels = np.asarray(labels, dtype=np.float32)
    n = labels.shape[0]
    inter = labels @ labels.T
    rows = labels.sum(axis=1).reshape(-1, 1)
    jac = inter / (rows + rows.T - inter + 1e-8)
    aps = []
    for i in range(n):
        rel = (jac[i] > threshold).astype(float)
        rel[i] = 0
        if rel.sum() > 0:
            cnt, ap = 0, 0.0
            for rank, j in enumerate(ranks_qmajor[i]):
                if rel[j] > 0:
                    cnt += 1
                    ap += cnt / (rank + 1)
            aps.append(ap / rel.sum())
    return np.mean(aps) if aps else 0


# -- fusion_eval/fuse.py:11-52 ---------------------------------------------------------------
def l2_normalize_np(x, eps=1e-12):
    nrm = np.maximum(np.linalg.norm(x, axis=1, keepdims=True), eps)
    return x / nrm


def concat_fusion(a, b):
    return l2_normalize_np(np.concatenate([l2_normalize_np(a), l2_normalize_np(b)], axis=1))


def weighted_sum_fusion(a, b, alpha):
    if a.shape[1] != b.shape[1]:
        return None
    return l2_normalize_np(alpha * l2_normalize_np(a) + (1.0 - alpha) * l2_normalize_np(b))
