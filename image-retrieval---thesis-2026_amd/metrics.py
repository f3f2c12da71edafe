"""Retrieval metrics with the reference's names, argument order and quirks, vectorised.

Mirrors (paths into /root/reference):
  retrieval_accuracy, compute_ap, compute_map, majority_vote,
  compute_classification_metrics, compute_map_multilabel        test.py:38-223, 941-985
  jaccard_score, precision_at_k, recall_at_k, evaluate_results    evaluate_nih_zilliz.py:12-64
  compute_similarity_matrix, rank_indices, evaluate_retrieval_metrics[_from_similarity]
                                                                  fusion_eval/metrics.py:12-94
  l2_normalize, concat_fusion, weighted_sum_fusion                fusion_eval/fuse.py:11-52
The heavy parts -- scores, top-k, full ranking -- come from libmirx (index.FlatIndex) and enter
here as id arrays.  Functions that take a score matrix in the reference also accept precomputed
rankings through keyword arguments.  When the ranking is a CUDA tensor (FlatIndex.rank_all), AP and
precision@k are computed on the device by mirx_rank_metrics (`rank_metrics_device`) and only the
per-query results come back; numpy inputs keep the host implementation.
"""
from dataclasses import dataclass
from typing import Optional

import numpy as np

try:  # torch only for accepting tensors; numpy arrays work without it
    import torch
except Exception:  # pragma: no cover
    torch = None


def _np(x):
    if torch is not None and isinstance(x, torch.Tensor):
        return x.detach().cpu().numpy()
    return np.asarray(x)


def _is_cuda(x):
    return torch is not None and isinstance(x, torch.Tensor) and x.is_cuda


def rank_metrics_device(ranks, gallery_labels, query_labels, kappas=(), query_ids=None, drop_self=False,
                        jaccard_threshold=None, standard_ap=False):
    """One device pass over ranked lists (include/mirx.h: mirx_rank_metrics).

    ranks: CUDA int64 [nq, n] (row per query, best first; a transposed view of a contiguous
    [n, nq] tensor is copied).  labels: class ids, or multi-hot bit masks when `jaccard_threshold`
    is given.  -> dict of CUDA tensors: ap [nq] f64 (NaN = no relevant id), cnt [nq, len(kappas)]
    i64, nrel [nq] i64, maxpos [nq] i64."""
    import ctypes
    from . import _lib
    lib = _lib.load()
    if ranks.dtype != torch.int64 or ranks.dim() != 2:
        raise ValueError("rank_metrics_device: ranks must be int64 [nq, n]")
    if ranks.stride(1) != 1 or ranks.stride(0) < ranks.shape[1]:
        ranks = ranks.contiguous()
    dev = ranks.device
    nq, n = ranks.shape
    gl = torch.as_tensor(gallery_labels).to(device=dev, dtype=torch.int64).contiguous()
    ql = torch.as_tensor(query_labels).to(device=dev, dtype=torch.int64).contiguous()
    if ql.numel() != nq:
        raise ValueError("rank_metrics_device: one query label per ranked list")
    qi = None if query_ids is None else torch.as_tensor(query_ids).to(device=dev, dtype=torch.int64).contiguous()
    ks = [int(k) for k in kappas]
    karr = (ctypes.c_int32 * max(1, len(ks)))(*ks)
    ap = torch.empty(nq, dtype=torch.float64, device=dev)
    cnt = torch.empty((nq, len(ks)), dtype=torch.int64, device=dev)
    nrel = torch.empty(nq, dtype=torch.int64, device=dev)
    maxpos = torch.empty(nq, dtype=torch.int64, device=dev)
    vp = lambda t: ctypes.c_void_p(t.data_ptr()) if t is not None and t.numel() else None  # noqa: E731
    with torch.cuda.device(dev):
        st = ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.mirx_rank_metrics(vp(ranks), nq, n, ranks.stride(0) if nq else n, vp(gl), gl.numel(), vp(ql),
                                         vp(qi), 1 if drop_self else 0, 0 if jaccard_threshold is None else 1,
                                         float(jaccard_threshold or 0.0), 1 if standard_ap else 0, karr, len(ks),
                                         vp(ap), vp(cnt), vp(nrel), vp(maxpos), st), "mirx_rank_metrics")
    return {"ap": ap, "cnt": cnt, "nrel": nrel, "maxpos": maxpos}


def _compute_map_device(ranks, gnd, kappas):
    gnd_t = torch.as_tensor(_np(gnd)).to(ranks.device)
    res = rank_metrics_device(ranks.t(), gnd_t, gnd_t, kappas)
    aps = res["ap"].cpu().numpy()
    nq, nk = aps.shape[0], len(kappas)
    nrel = res["nrel"].cpu().numpy().astype(np.float64)
    maxpos = res["maxpos"].cpu().numpy().astype(np.float64)
    cnt = res["cnt"].cpu().numpy().astype(np.float64)
    prs = np.zeros((nq, nk))
    for j, kap in enumerate(kappas):
        kq = np.minimum(maxpos, float(kap))                       # test.py:139
        with np.errstate(divide="ignore", invalid="ignore"):
            prs[:, j] = np.where(float(kap) <= maxpos, cnt[:, j], nrel) / kq
    ok = ~np.isnan(aps)                                           # test.py:122-126: empty queries are skipped
    if not ok.any():
        return float("nan"), aps, np.full(nk, np.nan), prs
    return float(np.sum(aps[ok]) / ok.sum()), aps, np.sum(prs[ok], axis=0) / ok.sum(), prs


def _label_bitmasks(labels):
    """Multi-hot [N, C <= 63] 0/1 matrix -> int64 bit masks, or None when it is not binary."""
    lab = _np(labels)
    if lab.ndim != 2 or lab.shape[1] > 63 or not np.isin(lab, (0, 1)).all():
        return None
    return (lab.astype(np.int64) << np.arange(lab.shape[1], dtype=np.int64)[None, :]).sum(axis=1)


# ---- test.py:38-54 ---------------------------------------------------------------------------
def retrieval_accuracy(output, target, topk=(1,), topk_ids=None):
    """R@k in percent (list of 0-d float32 values, like the reference's tensors).

    `output` is the [N,N] score matrix of the reference call; pass ``topk_ids=[N,>=max k]``
    (ranked ids with self already excluded, e.g. from FlatIndex.search) to skip the matrix."""
    target_np = _np(target)
    maxk = max(topk)
    if topk_ids is None:
        scores = _np(output)
        order = np.argsort(-scores, axis=1, kind="stable")[:, :maxk]
    else:
        order = _np(topk_ids)[:, :maxk]
    correct = target_np[order] == target_np[:, None]                 # [N, maxk]
    n = target_np.shape[0]
    res = []
    for k in topk:
        hit = np.float32(np.count_nonzero(correct[:, :k].any(axis=1)))
        val = np.float32(hit * np.float32(100.0 / n))
        res.append(torch.tensor(val) if torch is not None else val)
    return res


# ---- test.py:58-92 ---------------------------------------------------------------------------
def compute_ap(ranks, nres):
    """Trapezoidal AP from zero-based ranks of the positives."""
    ranks = np.asarray(ranks, dtype=np.float64)
    if ranks.size == 0:
        return 0
    j = np.arange(ranks.size, dtype=np.float64)
    p0 = np.where(ranks == 0, 1.0, j / np.where(ranks == 0, 1.0, ranks))
    p1 = (j + 1.0) / (ranks + 1.0)
    return float(np.sum((p0 + p1) * (1.0 / nres) / 2.0))


# ---- test.py:95-146 --------------------------------------------------------------------------
def compute_map(ranks, gnd, kappas=[]):
    """mAP, per-query AP, mean precision@kappas, per-query precision@kappas.

    `ranks` is [db_size, n_queries] (column i = ranking of query i).  Quirks of the
    reference kept: the query itself counts as a positive (it sits at the last rank because
    its score was -inf, test.py:119,1081); precision@kappa divides by
    min(largest 1-based positive rank, kappa) (test.py:139)."""
    if _is_cuda(ranks) and len(kappas) <= 8 and len(gnd):
        return _compute_map_device(ranks, gnd, list(kappas))
    ranks = _np(ranks)
    gnd = _np(gnd)
    nq = len(gnd)
    nk = len(kappas)
    if nq == 0:
        return float("nan"), np.zeros(0), np.zeros(nk), np.zeros((0, nk))
    rel = gnd[ranks] == gnd[None, :]                                  # [db, nq]
    nres = np.array([np.count_nonzero(gnd == g) for g in gnd], dtype=np.float64)
    cum = np.cumsum(rel, axis=0, dtype=np.float64)                    # 1-based count at each rank
    r = np.arange(ranks.shape[0], dtype=np.float64)[:, None]
    j = cum - 1.0
    with np.errstate(divide="ignore", invalid="ignore"):
        p0 = np.where(r == 0, 1.0, j / np.where(r == 0, 1.0, r))
    p1 = (j + 1.0) / (r + 1.0)
    aps = np.sum(np.where(rel, (p0 + p1) * 0.5, 0.0), axis=0) / nres
    prs = np.zeros((nq, nk))
    if nk:
        pos1 = np.where(rel, r + 1.0, 0.0)                            # 1-based ranks of positives
        maxpos = pos1.max(axis=0)
        for jk, kap in enumerate(kappas):
            kq = np.minimum(maxpos, float(kap))
            prs[:, jk] = np.sum(rel & (r + 1.0 <= kq[None, :]), axis=0) / kq
    # every query has at least itself as a positive, so the reference's "nempty" is 0
    return float(np.sum(aps) / nq), aps, np.sum(prs, axis=0) / nq, prs


# ---- test.py:149-162 -------------------------------------------------------------------------
def majority_vote(retrieved_labels):
    """Most frequent label; count ties go to the label seen first (the best-ranked one)."""
    if len(retrieved_labels) == 0:
        return None
    best, best_cnt, seen = None, 0, {}
    for lab in list(retrieved_labels):
        seen[lab] = seen.get(lab, 0) + 1
    for lab in list(retrieved_labels):                                # first-seen order
        if seen[lab] > best_cnt:
            best, best_cnt = lab, seen[lab]
    return best


def _prf(true, pred):
    labs = np.unique(np.concatenate([true, pred]))
    tp = np.array([np.count_nonzero((pred == c) & (true == c)) for c in labs], dtype=np.float64)
    fp = np.array([np.count_nonzero((pred == c) & (true != c)) for c in labs], dtype=np.float64)
    fn = np.array([np.count_nonzero((pred != c) & (true == c)) for c in labs], dtype=np.float64)
    with np.errstate(divide="ignore", invalid="ignore"):
        p = np.where(tp + fp > 0, tp / (tp + fp), 0.0)
        r = np.where(tp + fn > 0, tp / (tp + fn), 0.0)
        f = np.where(p + r > 0, 2 * p * r / (p + r), 0.0)
    w = tp + fn
    return p, r, f, w


# ---- test.py:165-223 -------------------------------------------------------------------------
def compute_classification_metrics(labels, dists, k_values=[1, 5, 10, 15, 20], ranks=None):
    """Majority-vote classification metrics per k (percent), keys as in the reference.

    `dists` is the reference's [N,N] score matrix (higher = more similar); pass
    ``ranks=[db, N]`` (column per query, as torch.argsort(dists, dim=0, descending=True)) to
    reuse a ranking.  sklearn's macro/weighted precision/recall/F1 (zero_division=0) and
    accuracy are computed directly."""
    labels_np = _np(labels)
    if ranks is None:
        ranks = np.argsort(-_np(dists), axis=0, kind="stable")
    ranks = _np(ranks)
    n = labels_np.shape[0]
    results = {}
    for k in k_values:
        top = labels_np[ranks[:k, :]]                                 # [k, N]
        pred = np.array([majority_vote(top[:, i]) for i in range(n)])
        p, r, f, w = _prf(labels_np, pred)
        results[k] = {
            "precision_macro": float(p.mean() * 100.0),
            "recall_macro": float(r.mean() * 100.0),
            "f1_macro": float(f.mean() * 100.0),
            "precision_weighted": float((p * w).sum() / w.sum() * 100.0),
            "recall_weighted": float((r * w).sum() / w.sum() * 100.0),
            "f1_weighted": float((f * w).sum() / w.sum() * 100.0),
            "accuracy": float(np.count_nonzero(pred == labels_np) / n * 100.0),
        }
    return results


# ---- test.py:941-985 -------------------------------------------------------------------------
def compute_map_multilabel(dists, labels, threshold=0.5, ranks=None):
    """Jaccard-thresholded mAP over the full ranking (query itself never relevant).
    `ranks` (optional) is [db, N] column-per-query like np.argsort(-dists, axis=0)."""
    if _is_cuda(ranks):
        masks = _label_bitmasks(labels)
        if masks is not None:
            n = masks.shape[0]
            res = rank_metrics_device(ranks.t(), masks, masks, (), query_ids=np.arange(n),
                                      jaccard_threshold=float(threshold), standard_ap=True)
            aps = res["ap"].cpu().numpy()
            aps = aps[~np.isnan(aps)]
            return np.mean(aps) if aps.size else 0
    lab = _np(labels).astype(np.float64)
    n = lab.shape[0]
    inter = lab @ lab.T
    rows = lab.sum(axis=1).reshape(-1, 1)
    jac = inter / (rows + rows.T - inter + 1e-8)
    if ranks is None:
        ranks = np.argsort(-_np(dists), axis=0, kind="stable")
    ranks = _np(ranks)
    rel = jac > threshold
    np.fill_diagonal(rel, False)
    aps = []
    pos = np.arange(1, n + 1, dtype=np.float64)
    for i in range(n):
        tot = np.count_nonzero(rel[i])
        if tot > 0:
            sr = rel[i][ranks[:, i]]
            aps.append(float(np.sum(np.cumsum(sr)[sr] / pos[sr]) / tot))
    return np.mean(aps) if aps else 0


# ---- evaluate_nih_zilliz.py:12-64 ------------------------------------------------------------
def jaccard_score(query_label, gallery_label):
    q = np.asarray(query_label, dtype=np.float32)
    g = np.asarray(gallery_label, dtype=np.float32)
    return float((q * g).sum()) / (float(np.clip(q + g, 0.0, 1.0).sum()) + 1e-8)


def precision_at_k(binary_relevance, k):
    if not len(binary_relevance):
        return 0.0
    k = min(k, len(binary_relevance))
    return float(np.mean(binary_relevance[:k]))


def recall_at_k(binary_relevance, total_positives, k):
    if total_positives <= 0:
        return 0.0
    k = min(k, len(binary_relevance))
    return float(np.sum(binary_relevance[:k]) / total_positives)


def _average_precision(rel, scores):
    """sklearn.metrics.average_precision_score for binary relevance (ties share a threshold)."""
    rel = np.asarray(rel, dtype=np.float64)
    scores = np.asarray(scores, dtype=np.float64)
    order = np.argsort(-scores, kind="stable")
    rel, scores = rel[order], scores[order]
    last = np.r_[np.flatnonzero(np.diff(scores)), len(scores) - 1]   # last index of each tie group
    tp = np.cumsum(rel)[last]
    precision = tp / (last + 1.0)
    recall = tp / rel.sum()
    return float(np.sum(np.diff(np.r_[0.0, recall]) * precision))


def evaluate_results(items, jaccard_threshold, ks):
    aps = []
    pks = {k: [] for k in ks}
    rks = {k: [] for k in ks}
    for item in items:
        hits = item["results"]
        rel = [1.0 if jaccard_score(item["query_label_vector"], h["label_vector"]) > jaccard_threshold
               else 0.0 for h in hits]
        tp = int(sum(rel))
        if tp > 0:
            aps.append(_average_precision(rel, [h["score"] for h in hits]))
        for k in ks:
            pks[k].append(precision_at_k(rel, k))
            rks[k].append(recall_at_k(rel, tp, k))
    out = {"mAP": float(np.mean(aps) * 100.0) if aps else 0.0,
           "num_queries": float(len(items)), "num_valid_ap_queries": float(len(aps))}
    for k in ks:
        out[f"P@{k}"] = float(np.mean(pks[k]) * 100.0) if pks[k] else 0.0
        out[f"R@{k}"] = float(np.mean(rks[k]) * 100.0) if rks[k] else 0.0
    return out


# ---- fusion_eval/fuse.py:11-52 ---------------------------------------------------------------
def l2_normalize(embeddings, eps=1e-12):
    norms = np.maximum(np.linalg.norm(embeddings, axis=1, keepdims=True), eps)
    return embeddings / norms


def concat_fusion(conv_embeddings, dino_embeddings):
    return l2_normalize(np.concatenate([l2_normalize(conv_embeddings), l2_normalize(dino_embeddings)], axis=1))


@dataclass(frozen=True)
class WeightedSumResult:
    embeddings: Optional[np.ndarray]
    skipped_reason: Optional[str] = None


def weighted_sum_fusion(conv_embeddings, dino_embeddings, alpha):
    if conv_embeddings.shape[1] != dino_embeddings.shape[1]:
        return WeightedSumResult(None, "weighted_sum_skipped_dimension_mismatch:"
                                 f" conv_dim={conv_embeddings.shape[1]}, dino_dim={dino_embeddings.shape[1]}")
    fused = alpha * l2_normalize(conv_embeddings) + (1.0 - alpha) * l2_normalize(dino_embeddings)
    return WeightedSumResult(l2_normalize(fused))


# ---- fusion_eval/metrics.py:12-94 ------------------------------------------------------------
def compute_similarity_matrix(embeddings):
    normalized = l2_normalize(np.asarray(embeddings).astype(np.float32))
    return normalized @ normalized.T


def rank_indices(similarity):
    similarity = np.array(similarity, copy=True)
    np.fill_diagonal(similarity, -np.inf)
    return np.argsort(-similarity, axis=1, kind="stable")


def evaluate_retrieval_metrics_from_similarity(similarity, labels, image_paths, k_values=(1, 5, 10), ranks=None):
    """Standard AP / mP@k / R@k with the self-match removed by image path.
    `ranks` (optional): [N,N] row-per-query ranking replacing argsort of `similarity`."""
    if ranks is None:
        similarity = np.asarray(similarity)
        if similarity.ndim != 2 or similarity.shape[0] != similarity.shape[1]:
            raise ValueError("Similarity matrix must be square")
        n = similarity.shape[0]
    else:
        n = ranks.shape[0]
    if len(labels) != len(image_paths) or len(labels) != n:
        raise ValueError("Labels, image_paths, and similarity matrix must have matching sizes")
    ks = sorted(set(int(k) for k in k_values))
    if _is_cuda(ranks) and len(ks) <= 8 and len(set(image_paths)) == n:
        # unique paths: "remove the query's own path" == drop id qi from its list
        _, lab_ids = np.unique(np.asarray(labels), return_inverse=True)
        res = rank_metrics_device(ranks, lab_ids, lab_ids, ks, query_ids=np.arange(n), drop_self=True,
                                  standard_ap=True)
        aps = np.nan_to_num(res["ap"].cpu().numpy(), nan=0.0)
        cnt = res["cnt"].cpu().numpy().astype(np.float64)
        out = {"num_samples": float(n), "mAP": float(np.mean(aps) * 100.0)}
        for j, k in enumerate(ks):
            out[f"mP@{k}"] = float(np.mean(cnt[:, j] / k) * 100.0)
            out[f"R@{k}"] = float(np.mean(cnt[:, j] > 0) * 100.0)
        return out
    if ranks is not None:
        ranks = _np(ranks)
    if ranks is None:
        ranks = rank_indices(similarity)
    labels_np = np.asarray(labels)
    paths_np = np.asarray(image_paths)
    aps = []
    pk = {k: [] for k in ks}
    rk = {k: [] for k in ks}
    for qi in range(n):
        order = ranks[qi]
        order = order[paths_np[order] != paths_np[qi]]
        rel = labels_np[order] == labels_np[qi]
        nrel = int(np.sum(labels_np == labels_np[qi]) - 1)
        if nrel <= 0:
            aps.append(0.0)
            for k in ks:
                pk[k].append(0.0)
                rk[k].append(0.0)
            continue
        hitpos = np.flatnonzero(rel)
        aps.append(float(np.sum(np.cumsum(rel.astype(np.int32))[hitpos] / (hitpos + 1)) / nrel) if len(hitpos) else 0.0)
        for k in ks:
            h = int(np.sum(rel[:k]))
            pk[k].append(h / k)
            rk[k].append(1.0 if h > 0 else 0.0)
    out = {"num_samples": float(n), "mAP": float(np.mean(aps) * 100.0)}
    for k in ks:
        out[f"mP@{k}"] = float(np.mean(pk[k]) * 100.0)
        out[f"R@{k}"] = float(np.mean(rk[k]) * 100.0)
    return out


def evaluate_retrieval_metrics(embeddings, labels, image_paths, k_values=(1, 5, 10)):
    return evaluate_retrieval_metrics_from_similarity(compute_similarity_matrix(embeddings), labels,
                                                      image_paths, k_values)
