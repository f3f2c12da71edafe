"""Late fusion of two embedding galleries and the file formats around it.

Mirrors (paths into /root/reference):
  EmbeddingRecord, AlignedEmbeddings, EmbeddingSource, FileEmbeddingSource,
  build_embedding_source, align_embedding_sources          fusion_eval/align.py:16-230
  ExperimentResult, run_late_fusion_experiments, normalize_similarity_matrix,
  confidence_based_fusion, top12_margin                     fusion_eval/evaluate.py:19-214
  load_query_set                                            retrieval_analysis/comparison.py:41-84

MI355X design.  Every score-level fusion the reference evaluates is, per query q,
    fused[q, g] = wa_q * (a_q . a_g) + wb_q * (b_q . b_g) + const_q
(alpha / 1-alpha; zscore: divided by the row's std; minmax: by its range; confidence fusion: a
per-query alpha from the top1-top2 margins), i.e. ONE inner product between the query
[wa_q a_q ; wb_q b_q] and the gallery row [a_g ; b_g].  So both galleries live in HBM once, concatenated
in a single FlatIndex, every fusion variant is a search of that index with re-weighted queries
(bf16 MFMA candidate GEMM + fp64 re-rank, or the exact full ranking), and the row statistics come
from the same index: mean = q . mean(gallery) and E[s^2] = q^T (G^T G / N) q are linear algebra on the
resident rows, min / max / top-2 are k = 1..2 searches.  const_q does not change a query's ranking.
Scores are exact fp64 where the reference rounds its similarity matrices to fp32, so rankings can
differ inside fp32 near-ties (tests state the tolerance).
"""
import csv
import json
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, Iterable, List, Mapping, Optional, Sequence

import numpy as np
import torch

from . import metrics as _m
from .index import FlatIndex
from .metrics import concat_fusion, l2_normalize, weighted_sum_fusion


# ---- fusion_eval/align.py:16-41 ---------------------------------------------------------------
@dataclass(frozen=True)
class EmbeddingRecord:
    image_path: str
    label: Optional[str]
    embedding: np.ndarray
    source_name: str
    raw: Mapping[str, Any]


@dataclass(frozen=True)
class AlignedEmbeddings:
    image_paths: List[str]
    labels: List[str]
    conv_embeddings: np.ndarray
    dino_embeddings: np.ndarray
    coverage: Dict[str, List[str]]


@dataclass(frozen=True)
class ExperimentResult:
    experiment_name: str
    num_samples: int
    metrics: Dict[str, float]
    skipped: bool = False
    skipped_reason: Optional[str] = None


class EmbeddingSource:
    def fetch_all(self) -> List[EmbeddingRecord]:
        raise NotImplementedError


# ---- fusion_eval/align.py:96-140 --------------------------------------------------------------
class FileEmbeddingSource(EmbeddingSource):
    """.json ({"records": [{image_path, label, embedding}]} or a bare list) / .npz
    ({image_paths, labels, embeddings})."""

    def __init__(self, path, source_name):
        self.path = Path(path)
        self.source_name = source_name

    def fetch_all(self):
        suffix = self.path.suffix.lower()
        if suffix == ".json":
            with self.path.open("r", encoding="utf-8") as handle:
                data = json.load(handle)
            rows = data.get("records", data) if isinstance(data, dict) else data
            return [EmbeddingRecord(image_path=row["image_path"], label=row.get("label"),
                                    embedding=np.asarray(row["embedding"], dtype=np.float32),
                                    source_name=self.source_name, raw=row) for row in rows]
        if suffix == ".npz":
            payload = np.load(self.path, allow_pickle=True)
            image_paths = payload["image_paths"].tolist()
            labels = payload["labels"].tolist() if "labels" in payload else [None] * len(image_paths)
            return [EmbeddingRecord(image_path=p, label=l, embedding=np.asarray(e, dtype=np.float32),
                                    source_name=self.source_name, raw={})
                    for p, l, e in zip(image_paths, labels, payload["embeddings"])]
        raise ValueError(f"Unsupported embedding file format: {self.path}")


class CollectionEmbeddingSource(EmbeddingSource):
    """The reference's MilvusEmbeddingSource (align.py:44-93) over a mirx.retriever.Collection:
    all rows of the resident index with their host-side metadata."""

    def __init__(self, collection, name=None):
        self.collection = collection
        self.name = name or collection.name

    def fetch_all(self):
        n = self.collection.num_entities
        if n == 0:
            return []
        rows, _ = self.collection.index.rows(0, n)
        rows = rows.cpu().numpy()
        meta = self.collection._meta
        return [EmbeddingRecord(image_path=meta["image_path"][i], label=meta["label"][i], embedding=rows[i],
                                source_name=self.name, raw={"id": i}) for i in range(n)
                if meta["image_path"][i] is not None]


MilvusEmbeddingSource = CollectionEmbeddingSource


def build_embedding_source(config):
    """align.py:143-152.  {"type": "file", "path", "name"} or {"type": "collection"|"milvus",
    "collection": <mirx.retriever.Collection>, "name"} (there is no server to connect to)."""
    source_type = config.get("type", "milvus")
    if source_type == "file":
        return FileEmbeddingSource(path=config["path"], source_name=config["name"])
    if source_type in ("milvus", "collection"):
        if "collection" not in config:
            raise ValueError("in-process build: pass the resident Collection as config['collection']")
        return CollectionEmbeddingSource(config["collection"], config.get("name"))
    raise ValueError(f"Unsupported source type: {source_type}")


def save_embedding_file(path, image_paths, labels, embeddings):
    """Write a gallery in the reference's FileEmbeddingSource formats (.npz or .json)."""
    path = Path(path)
    emb = embeddings.detach().cpu().numpy() if torch.is_tensor(embeddings) else np.asarray(embeddings)
    emb = emb.astype(np.float32)
    if not (len(image_paths) == len(labels) == emb.shape[0]):
        raise ValueError("column lengths differ")
    if path.suffix.lower() == ".npz":
        np.savez(path, image_paths=np.array([str(p) for p in image_paths]), labels=np.array(list(labels)),
                 embeddings=emb)
    elif path.suffix.lower() == ".json":
        with path.open("w", encoding="utf-8") as handle:
            json.dump({"records": [{"image_path": str(p), "label": l, "embedding": [float(v) for v in e]}
                                   for p, l, e in zip(image_paths, labels, emb)]}, handle)
    else:
        raise ValueError(f"Unsupported embedding file format: {path}")


def ingest_embedding_file(collection, path, batch_size=65536):
    """Load an embedding dump straight into a resident Collection (the insert loop of
    ingest_embeddings.py:399-411 without the model).  Returns the number of rows inserted."""
    records = FileEmbeddingSource(path, collection.name).fetch_all()
    for s in range(0, len(records), batch_size):
        chunk = records[s:s + batch_size]
        collection.insert([[r.image_path for r in chunk], [r.label for r in chunk],
                           np.stack([r.embedding for r in chunk])])
    collection.flush()
    return len(records)


# ---- retrieval_analysis/comparison.py:41-84 (paths only) ---------------------------------------
def load_query_set(path):
    path = Path(path)
    suffix = path.suffix.lower()
    if suffix == ".json":
        with path.open("r", encoding="utf-8") as handle:
            data = json.load(handle)
        if isinstance(data, dict):
            data = data.get("queries", data.get("results", []))
        return [item.get("image_path", item.get("query_image_path")) for item in data
                if item.get("image_path", item.get("query_image_path"))]
    if suffix == ".csv":
        with path.open("r", encoding="utf-8", newline="") as handle:
            return [row.get("image_path") or row.get("query_image_path") for row in csv.DictReader(handle)
                    if row.get("image_path") or row.get("query_image_path")]
    out = []
    with path.open("r", encoding="utf-8") as handle:
        for line in handle:
            s = line.strip()
            if s and not s.startswith("#"):
                out.append(s.split()[0])
    return out


# ---- fusion_eval/align.py:155-230 --------------------------------------------------------------
def _index_records(records: Iterable[EmbeddingRecord], source_name):
    indexed = {}
    for record in records:
        if record.image_path in indexed:
            raise ValueError(f"Duplicate image_path found in {source_name}: {record.image_path}")
        indexed[record.image_path] = record
    return indexed


def align_embedding_sources(conv_source, dino_source, query_set_path=None, strict_label_check=True):
    conv_records = _index_records(conv_source.fetch_all(), "ConvNeXt")
    dino_records = _index_records(dino_source.fetch_all(), "DINO")
    conv_paths, dino_paths = set(conv_records), set(dino_records)
    coverage = {"present_in_conv_only": sorted(conv_paths - dino_paths),
                "present_in_dino_only": sorted(dino_paths - conv_paths),
                "present_in_both": sorted(conv_paths & dino_paths)}
    if query_set_path:
        target_paths = [p for p in load_query_set(query_set_path) if p in conv_paths and p in dino_paths]
    else:
        target_paths = coverage["present_in_both"]
    labels, conv_emb, dino_emb, final_paths = [], [], [], []
    for image_path in target_paths:
        c, d = conv_records[image_path], dino_records[image_path]
        if strict_label_check and c.label != d.label:
            raise ValueError(f"Label mismatch for image_path={image_path}: conv={c.label!r}, dino={d.label!r}")
        final_paths.append(image_path)
        labels.append(c.label or d.label or "unknown")
        conv_emb.append(c.embedding)
        dino_emb.append(d.embedding)
    if not final_paths:
        raise ValueError("No aligned samples found across the requested sources")
    return AlignedEmbeddings(image_paths=final_paths, labels=labels,
                             conv_embeddings=np.stack(conv_emb).astype(np.float32),
                             dino_embeddings=np.stack(dino_emb).astype(np.float32), coverage=coverage)


# ---- fusion_eval/evaluate.py:150-214: matrix-level helpers (host, for callers that hold matrices) ----
def normalize_similarity_matrix(similarity, mode="none"):
    if mode == "none":
        return similarity.astype(np.float32, copy=True)
    similarity = similarity.astype(np.float32, copy=True)
    diag = np.diag(similarity).copy()
    if mode == "zscore":
        stds = np.maximum(np.std(similarity, axis=1, keepdims=True), 1e-12)
        normalized = (similarity - np.mean(similarity, axis=1, keepdims=True)) / stds
    elif mode == "minmax":
        mins = np.min(similarity, axis=1, keepdims=True)
        scales = np.maximum(np.max(similarity, axis=1, keepdims=True) - mins, 1e-12)
        normalized = (similarity - mins) / scales
    else:
        raise ValueError(f"Unsupported score normalization mode: {mode}. Use one of: none, zscore, minmax")
    np.fill_diagonal(normalized, diag)
    return normalized


def top12_margin(similarity):
    if similarity.shape[1] < 2:
        raise ValueError("Need at least two gallery scores per query for confidence margin")
    top2 = np.partition(similarity, kth=-2, axis=1)[:, -2:]
    return np.max(top2, axis=1) - np.min(top2, axis=1)


def confidence_based_fusion(conv_similarity, dino_similarity):
    if conv_similarity.shape != dino_similarity.shape:
        raise ValueError("Conv and DINO similarity matrices must have the same shape")
    conv_scores = conv_similarity.astype(np.float32, copy=True)
    dino_scores = dino_similarity.astype(np.float32, copy=True)
    np.fill_diagonal(conv_scores, -np.inf)
    np.fill_diagonal(dino_scores, -np.inf)
    conv_conf, dino_conf = top12_margin(conv_scores), top12_margin(dino_scores)
    alpha = conv_conf / (conv_conf + dino_conf + 1e-8)
    fused = alpha[:, None] * conv_scores + (1.0 - alpha[:, None]) * dino_scores
    return {"similarity": fused, "conv_selected_queries": int(np.sum(alpha >= 0.5)),
            "dino_selected_queries": int(np.sum(alpha < 0.5)), "alpha_mean": float(np.mean(alpha)),
            "alpha_std": float(np.std(alpha))}


# ---- the device path -----------------------------------------------------------------------------
class LateFusionIndex:
    """Two galleries (rows L2-normalised here, like evaluate.py:40-41) resident as one concatenated
    inner-product index; see the module docstring."""

    def __init__(self, conv_embeddings, dino_embeddings, device=None):
        dev = torch.device("cuda", 0 if device is None else device) if not isinstance(device, torch.device) else device
        a = torch.as_tensor(np.asarray(conv_embeddings, dtype=np.float32) if not torch.is_tensor(conv_embeddings)
                            else conv_embeddings).to(dev, torch.float32)
        b = torch.as_tensor(np.asarray(dino_embeddings, dtype=np.float32) if not torch.is_tensor(dino_embeddings)
                            else dino_embeddings).to(dev, torch.float32)
        if a.shape[0] != b.shape[0]:
            raise ValueError("Conv and DINO galleries must have the same number of rows")
        from .index import l2_normalize_
        self.a = l2_normalize_(a.contiguous().clone())
        self.b = l2_normalize_(b.contiguous().clone())
        self.da, self.db, self.n = a.shape[1], b.shape[1], a.shape[0]
        self.device = dev
        self.index = FlatIndex(self.da + self.db, "IP", dev.index or 0)
        self.index.add(torch.cat([self.a, self.b], dim=1))
        self._moments = None

    def __len__(self):
        return self.n

    # -- queries against one or both halves ----------------------------------------------------
    def _queries(self, qa, qb, wa=None, wb=None):
        nq = (qa if qa is not None else qb).shape[0]
        q = torch.zeros((nq, self.da + self.db), dtype=torch.float32, device=self.device)
        if qa is not None:
            q[:, :self.da] = qa if wa is None else qa * wa.to(torch.float32)[:, None]
        if qb is not None:
            q[:, self.da:] = qb if wb is None else qb * wb.to(torch.float32)[:, None]
        return q

    def row_statistics(self, qa, qb, exclude_ids=None):
        """Per query and half: mean / std (population) / min / max of its score row over the WHOLE
        gallery, and the top1 - top2 margin with `exclude_ids` left out (evaluate.py:158-172,192-197).
        -> dict of fp64 CUDA tensors keyed 'a_mean', 'a_std', 'a_min', 'a_max', 'a_margin', 'b_...'."""
        if self._moments is None:
            ga, gb = self.a.double(), self.b.double()
            self._moments = (ga.mean(0), ga.t() @ ga / self.n, gb.mean(0), gb.t() @ gb / self.n)
        ma, Ma, mb, Mb = self._moments
        out = {}
        for key, q, mean_vec, second, half in (("a", qa, ma, Ma, 0), ("b", qb, mb, Mb, 1)):
            qd = q.double()
            mu = qd @ mean_vec
            ex2 = ((qd @ second) * qd).sum(1)
            out[f"{key}_mean"] = mu
            out[f"{key}_std"] = (ex2 - mu * mu).clamp_min(0).sqrt()
            full = self._queries(q, None) if half == 0 else self._queries(None, q)
            hi, _ = self.index.search(full, 1, return_f64=True)
            lo, _ = self.index.search(-full, 1, return_f64=True)
            out[f"{key}_max"], out[f"{key}_min"] = hi[:, 0], -lo[:, 0]
            if self.n - (1 if exclude_ids is not None else 0) >= 2:
                top2, _ = self.index.search(full, 2, exclude_ids=exclude_ids, return_f64=True)
                out[f"{key}_margin"] = top2[:, 0] - top2[:, 1]
        return out

    def fusion_weights(self, qa, qb, alpha=None, score_normalization="none", exclude_ids=None):
        """(wa, wb, info): per-query weights of the two halves for score fusion with a fixed `alpha`
        (evaluate.py:60-78) or, with alpha=None, the confidence fusion of evaluate.py:180-203."""
        if score_normalization not in ("none", "zscore", "minmax"):
            raise ValueError(f"Unsupported score normalization mode: {score_normalization}. "
                             "Use one of: none, zscore, minmax")
        nq = qa.shape[0]
        one = torch.ones(nq, dtype=torch.float64, device=self.device)
        st = self.row_statistics(qa, qb, exclude_ids) if (alpha is None or score_normalization != "none") else {}
        if score_normalization == "zscore":
            sa, sb = 1.0 / st["a_std"].clamp_min(1e-12), 1.0 / st["b_std"].clamp_min(1e-12)
        elif score_normalization == "minmax":
            sa = 1.0 / (st["a_max"] - st["a_min"]).clamp_min(1e-12)
            sb = 1.0 / (st["b_max"] - st["b_min"]).clamp_min(1e-12)
        else:
            sa, sb = one, one
        info = {}
        if alpha is None:
            if "a_margin" not in st:
                raise ValueError("Need at least two gallery scores per query for confidence margin")
            ca, cb = st["a_margin"] * sa, st["b_margin"] * sb           # margins of the NORMALISED rows
            al = ca / (ca + cb + 1e-8)
            info = {"alpha": al, "conv_selected_queries": int((al >= 0.5).sum()),
                    "dino_selected_queries": int((al < 0.5).sum())}
        else:
            al = one * float(alpha)
        return al * sa, (1.0 - al) * sb, info

    def search(self, qa, qb, k, alpha=0.5, score_normalization="none", exclude_ids=None):
        """Fused top-k for external queries (both halves unit-norm): -> (scores fp32 [nq,k], ids, info).
        Scores are the weighted inner products (without the per-query constant of zscore/minmax)."""
        wa, wb, info = self.fusion_weights(qa, qb, alpha, score_normalization, exclude_ids)
        sc, ids = self.index.search(self._queries(qa, qb, wa, wb), k, exclude_ids=exclude_ids)
        return sc, ids, info

    def rank_self(self, alpha=0.5, score_normalization="none"):
        """Full fused ranking of the gallery against itself, self last: -> (ranks [N,N] CUDA, info)."""
        me = torch.arange(self.n, device=self.device)
        wa, wb, info = self.fusion_weights(self.a, self.b, alpha, score_normalization, me)
        return self.index.rank_all(self._queries(self.a, self.b, wa, wb), exclude_ids=me), info


def _rank_embeddings(embeddings, device):
    e = torch.as_tensor(np.asarray(embeddings, dtype=np.float32)).to(device)
    from .index import l2_normalize_
    e = l2_normalize_(e.contiguous().clone())
    ix = FlatIndex(e.shape[1], "IP", device.index or 0)
    ix.add(e)
    return ix.rank_all(e, exclude_ids=torch.arange(e.shape[0], device=device))


def run_late_fusion_experiments(aligned, alpha_values=(0.2, 0.4, 0.5, 0.6, 0.8), k_values=(1, 5, 10),
                                include_score_fusion=True, score_normalization="none",
                                include_confidence_fusion=True, device=None):
    """fusion_eval/evaluate.py:30-147 with every similarity matrix, argsort and metric loop on the GPU:
    same experiment names, order and metric keys."""
    dev = torch.device("cuda", 0 if device is None else device) if not isinstance(device, torch.device) else device
    n = len(aligned.image_paths)
    results = []

    def metrics_of(ranks):
        return _m.evaluate_retrieval_metrics_from_similarity(None, aligned.labels, aligned.image_paths, k_values,
                                                             ranks=ranks)

    baselines = {"convnext_baseline": l2_normalize(aligned.conv_embeddings),
                 "dino_baseline": l2_normalize(aligned.dino_embeddings),
                 "concat_fusion": concat_fusion(aligned.conv_embeddings, aligned.dino_embeddings)}
    for name, emb in baselines.items():
        results.append(ExperimentResult(name, n, metrics_of(_rank_embeddings(emb, dev))))
    fused = None
    if include_score_fusion or include_confidence_fusion:
        fused = LateFusionIndex(aligned.conv_embeddings, aligned.dino_embeddings, dev)
    if include_score_fusion:
        for alpha in alpha_values:
            ranks, _ = fused.rank_self(alpha, score_normalization)
            results.append(ExperimentResult(f"score_fusion_alpha_{alpha:.1f}", n, metrics_of(ranks)))
    if include_confidence_fusion:
        ranks, info = fused.rank_self(None, score_normalization)
        metrics = metrics_of(ranks)
        metrics["conv_selected_queries"] = float(info["conv_selected_queries"])
        metrics["dino_selected_queries"] = float(info["dino_selected_queries"])
        results.append(ExperimentResult("confidence_fusion_top12_margin", n, metrics))
    for alpha in alpha_values:
        fusion = weighted_sum_fusion(aligned.conv_embeddings, aligned.dino_embeddings, alpha)
        if fusion.embeddings is None:
            results.append(ExperimentResult(f"weighted_sum_alpha_{alpha:.1f}", n, {}, True, fusion.skipped_reason))
            continue
        results.append(ExperimentResult(f"weighted_sum_alpha_{alpha:.1f}", n,
                                        metrics_of(_rank_embeddings(fusion.embeddings, dev))))
    return results


# ---- test.py:599-647 (text-similarity re-ranking of the top-k image results) --------------------
def text_rerank_scores(embeds, concept_image_embeds, text_embeds, labels, rerank_k, text_weight):
    """The re-scored matrix ``dists`` of test.py:599-623 on the device, in one vectorised pass.

    embeds [N, D] (unit rows, retrieval backbone), concept_image_embeds [N, E] and text_embeds [C, E]
    (unit rows, the text tower's image / class-prompt features), labels [N] int.  For every query i the
    ``min(rerank_k, N)`` best images by image similarity (ties: lowest id) get
    ``alpha * sim[i, j] + (1 - alpha) * (concept_image[j] . text[labels[i]])`` with alpha = text_weight;
    j == i and every other entry keep the image similarity; the diagonal becomes -inf.  Scores are fp64
    (the reference keeps fp32: rankings can differ inside fp32 near-ties).  Returns the [N, N] fp64 tensor on
    the embeddings' device; rows are queries for retrieval_accuracy, COLUMNS are what the reference's
    ``argsort(dists, dim=0)`` ranks for compute_map -- kept as is by text_rerank_evaluate."""
    e = torch.as_tensor(embeds).double()
    dev = e.device
    n = e.shape[0]
    lab = torch.as_tensor(labels).to(dev).long()
    sim = e @ e.t()
    k = min(int(rerank_k), n)
    dists = sim.clone()
    if k > 0:
        top = torch.sort(sim, dim=1, descending=True, stable=True).indices[:, :k]          # [N, k]
        rows = torch.arange(n, device=dev)[:, None].expand(n, k)
        it = torch.as_tensor(concept_image_embeds).to(dev).double() @ torch.as_tensor(text_embeds).to(dev).double().t()
        text_score = it[top, lab[:, None].expand(n, k)]                                     # [N, k]: it[j, labels[i]]
        base = sim.gather(1, top)
        fused = float(text_weight) * base + (1.0 - float(text_weight)) * text_score
        dists.scatter_(1, top, torch.where(top != rows, fused, base))
    dists.fill_diagonal_(float("-inf"))
    return dists


def text_rerank_evaluate(embeds, concept_image_embeds, text_embeds, labels, rerank_k=20, text_weight=0.7,
                         kappas=(1, 5, 10)):
    """test.py:599-647: R@K from the rows of the re-scored matrix, mAP / mP@K from the ranking of its columns
    (``argsort(dists, dim=0, descending=True)``, ties -> lowest id), all on the device."""
    dists = text_rerank_scores(embeds, concept_image_embeds, text_embeds, labels, rerank_k, text_weight)
    lab = torch.as_tensor(labels).to(dists.device).long()
    kmax = min(max(kappas), dists.shape[0])
    top_ids = torch.sort(dists, dim=1, descending=True, stable=True).indices[:, :kmax]
    accuracy = _m.retrieval_accuracy(None, lab, topk=tuple(kappas), topk_ids=top_ids)
    ranks = torch.sort(dists, dim=0, descending=True, stable=True).indices            # [N, N]: column q = query q
    m_ap, aps, pr, prs = _m.compute_map(ranks, lab, list(kappas))
    return {"accuracy": np.array([float(a) for a in accuracy], dtype=np.float32), "mAP": m_ap, "aps": aps, "pr": pr,
            "prs": prs}


# ---- retrieval_analysis/rerank.py:10-26 ----------------------------------------------------------
class Reranker:
    """Hook of retrieval_analysis.compare_models: ``rerank(query, results) -> iterable of results``."""

    def rerank(self, query, results):
        raise NotImplementedError


class IdentityReranker(Reranker):
    def rerank(self, query, results):
        return list(results)
