"""Late fusion of two embedding galleries and the file formats around it.

Mirrors (paths into /root/reference):
  EmbeddingRecord, AlignedEmbeddings, EmbeddingSource, FileEmbeddingSource,
  build_embedding_source, align_embedding_sources          fusion_eval/align.py:16-230
  ExperimentResult, run_late_fusion_experiments, normalize_similarity_matrix,
  confidence_based_fusion, top12_margin                     fusion_eval/evaluate.py:19-214
  load_query_set                                            retrieval_analysis/comparison.py:41-84 (mirx.adapter)
The host half is written against the FORMAT (dump files, record fields, error conditions), column-wise: a source is a
table (paths, labels, one matrix), alignment is two dictionaries and two gathers.

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
import json
from dataclasses import dataclass
from pathlib import Path
from typing import Any, Dict, List, Mapping, Optional

import numpy as np
import torch

from . import metrics as _m
from .index import FlatIndex
from .metrics import concat_fusion, l2_normalize, weighted_sum_fusion


# ---- public record types (field names are the contract: fusion_eval/align.py:16-41, evaluate.py:19-27) ---------------
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


# ---- embedding sources: columnar (paths, labels, one [N, D] matrix), never one Python object per row -----------------
@dataclass
class _Table:
    """One embedding set as columns.  `raw` holds the per-row source dicts when the format has them (.json)."""
    name: str
    paths: List[str]
    labels: List[Optional[str]]
    matrix: np.ndarray                      # float32 [N, D]
    raw: Optional[List[Mapping[str, Any]]] = None

    def records(self):
        raw = self.raw or [{}] * len(self.paths)
        return [EmbeddingRecord(image_path=p, label=l, embedding=self.matrix[i], source_name=self.name, raw=raw[i])
                for i, (p, l) in enumerate(zip(self.paths, self.labels))]

    def row_of(self):
        """image_path -> row; a path that occurs twice is an error (the alignment key must be unique)."""
        index = {}
        for i, p in enumerate(self.paths):
            if index.setdefault(p, i) != i:
                raise ValueError(f"{self.name}: image_path {p!r} occurs more than once")
        return index


class EmbeddingSource:
    """fetch_all() -> list[EmbeddingRecord] (the reference's interface); table() -> the same data as columns."""

    def table(self) -> _Table:
        raise NotImplementedError

    def fetch_all(self) -> List[EmbeddingRecord]:
        return self.table().records()


def _read_embedding_file(path: Path, name: str) -> _Table:
    """The two dump formats of the reference (fusion_eval/align.py:96-140):
    .npz  arrays `image_paths` [N], `embeddings` [N, D], optional `labels` [N];
    .json {"records": [{"image_path", "label"?, "embedding": [...]}, ...]} or the bare list."""
    kind = path.suffix.lower()
    if kind == ".npz":
        with np.load(path, allow_pickle=True) as z:
            paths = [str(p) for p in z["image_paths"].tolist()]
            labels = z["labels"].tolist() if "labels" in z.files else [None] * len(paths)
            matrix = np.asarray(z["embeddings"], dtype=np.float32)
        return _Table(name, paths, labels, matrix.reshape(len(paths), -1))
    if kind == ".json":
        with path.open("r", encoding="utf-8") as fh:
            doc = json.load(fh)
        rows = doc["records"] if isinstance(doc, dict) and "records" in doc else doc
        matrix = np.asarray([r["embedding"] for r in rows], dtype=np.float32)
        return _Table(name, [r["image_path"] for r in rows], [r.get("label") for r in rows],
                      matrix.reshape(len(rows), -1), raw=list(rows))
    raise ValueError(f"{path}: embedding dumps are .npz or .json")


class FileEmbeddingSource(EmbeddingSource):
    def __init__(self, path, source_name):
        self.path = Path(path)
        self.source_name = source_name

    def table(self):
        return _read_embedding_file(self.path, self.source_name)


class CollectionEmbeddingSource(EmbeddingSource):
    """The reference's MilvusEmbeddingSource (align.py:44-93) over a mirx.retriever.Collection: every row of the
    resident index (ONE device read) with its host-side metadata."""

    def __init__(self, collection, name=None):
        self.collection = collection
        self.name = name or collection.name

    def table(self):
        n = self.collection.num_entities
        meta = self.collection._meta
        if n == 0:
            return _Table(self.name, [], [], np.zeros((0, self.collection.dim), np.float32))
        rows = self.collection.index.rows(0, n)[0].cpu().numpy()
        keep = [i for i in range(n) if meta["image_path"][i] is not None]
        labels = meta.get("label", [None] * n)
        return _Table(self.name, [meta["image_path"][i] for i in keep], [labels[i] for i in keep], rows[keep],
                      raw=[{"id": i} for i in keep])


MilvusEmbeddingSource = CollectionEmbeddingSource


def build_embedding_source(config):
    """{"type": "file", "path", "name"} or {"type": "collection" | "milvus", "collection": <Collection>, "name"}.
    There is no server to dial: a milvus-typed source must hand over the resident Collection."""
    kind = config.get("type", "milvus")
    if kind == "file":
        return FileEmbeddingSource(config["path"], config["name"])
    if kind in ("milvus", "collection"):
        if config.get("collection") is None:
            raise ValueError("in-process build: pass the resident Collection as config['collection']")
        return CollectionEmbeddingSource(config["collection"], config.get("name"))
    raise ValueError(f"unknown embedding source type {kind!r} (file | collection | milvus)")


def save_embedding_file(path, image_paths, labels, embeddings):
    """Write a gallery in either dump format, readable by the reference's FileEmbeddingSource."""
    path = Path(path)
    matrix = (embeddings.detach().cpu().numpy() if torch.is_tensor(embeddings) else np.asarray(embeddings)).astype(np.float32)
    paths = [str(p) for p in image_paths]
    labels = list(labels)
    if not (len(paths) == len(labels) == matrix.shape[0]):
        raise ValueError("image_paths, labels and embeddings must have one entry per row")
    kind = path.suffix.lower()
    if kind == ".npz":
        np.savez(path, image_paths=np.array(paths), labels=np.array(labels), embeddings=matrix)
    elif kind == ".json":
        doc = {"records": [{"image_path": p, "label": l, "embedding": row.tolist()} for p, l, row in zip(paths, labels, matrix)]}
        with path.open("w", encoding="utf-8") as fh:
            json.dump(doc, fh)
    else:
        raise ValueError(f"{path}: embedding dumps are .npz or .json")


def ingest_embedding_file(collection, path, batch_size=65536):
    """Load an embedding dump straight into a resident Collection (the insert loop of ingest_embeddings.py:399-411
    without the model): whole column slices per insert.  Returns the number of rows inserted."""
    t = _read_embedding_file(Path(path), collection.name)
    for s in range(0, len(t.paths), batch_size):
        collection.insert([t.paths[s:s + batch_size], t.labels[s:s + batch_size], t.matrix[s:s + batch_size]])
    collection.flush()
    return len(t.paths)


def load_query_set(path):
    """Paths of an ordered query set (.json / .csv / text; retrieval_analysis/comparison.py:41-84).  The records with
    their labels: mirx.adapter.load_query_set."""
    from .adapter import load_query_set as _records
    return [q.image_path for q in _records(path)]


def _table_of(source):
    if isinstance(source, EmbeddingSource) and type(source).table is not EmbeddingSource.table:
        return source.table()
    recs = source.fetch_all()                              # foreign sources that only implement fetch_all()
    name = recs[0].source_name if recs else "source"
    dim = recs[0].embedding.shape[-1] if recs else 0
    return _Table(name, [r.image_path for r in recs], [r.label for r in recs],
                  np.stack([r.embedding for r in recs]).astype(np.float32) if recs else np.zeros((0, dim), np.float32))


def align_embedding_sources(conv_source, dino_source, query_set_path=None, strict_label_check=True):
    """Rows of the two sources matched by image_path (fusion_eval/align.py:155-230): `coverage` lists what each side
    has; the aligned set is the sorted intersection, or the query set's order restricted to it; a label that differs
    between the sides raises unless strict_label_check is off (then the ConvNeXt side's label wins); an empty result
    raises.  Done with two dictionaries and two fancy-indexed gathers."""
    conv, dino = _table_of(conv_source), _table_of(dino_source)
    conv.name, dino.name = "ConvNeXt", "DINO"
    ci, di = conv.row_of(), dino.row_of()
    both = sorted(ci.keys() & di.keys())
    coverage = {"present_in_conv_only": sorted(ci.keys() - di.keys()),
                "present_in_dino_only": sorted(di.keys() - ci.keys()), "present_in_both": both}
    order = both if not query_set_path else [p for p in load_query_set(query_set_path) if p in ci and p in di]
    if not order:
        raise ValueError("the two sources share no image_path (after the query-set filter): nothing to fuse")
    crow = np.fromiter((ci[p] for p in order), dtype=np.int64, count=len(order))
    drow = np.fromiter((di[p] for p in order), dtype=np.int64, count=len(order))
    labels = []
    for p, i, j in zip(order, crow, drow):
        lc, ld = conv.labels[i], dino.labels[j]
        if strict_label_check and lc != ld:
            raise ValueError(f"{p}: the sources disagree on the label ({lc!r} vs {ld!r})")
        labels.append(lc or ld or "unknown")
    return AlignedEmbeddings(image_paths=order, labels=labels, conv_embeddings=conv.matrix[crow].astype(np.float32),
                             dino_embeddings=dino.matrix[drow].astype(np.float32), coverage=coverage)


# ---- matrix-level score fusion (for callers that already hold similarity matrices; the resident-index path below does
# not build them).  Arithmetic is fp32 in the reference's operation order (fusion_eval/evaluate.py:152-214): tests compare
# bit for bit with its outputs. -----------------------------------------------------------------------------------------
def _row_affine(sim, mode):
    """(offset, scale) per row such that normalised = (sim - offset) / scale."""
    if mode == "zscore":
        return np.mean(sim, axis=1, keepdims=True), np.maximum(np.std(sim, axis=1, keepdims=True), 1e-12)
    if mode == "minmax":
        lo = np.min(sim, axis=1, keepdims=True)
        return lo, np.maximum(np.max(sim, axis=1, keepdims=True) - lo, 1e-12)
    raise ValueError(f"score normalization {mode!r}: use one of none, zscore, minmax")


def normalize_similarity_matrix(similarity, mode="none"):
    """Row-wise z-score / min-max of a similarity matrix; the diagonal keeps its original values."""
    sim = similarity.astype(np.float32, copy=True)
    if mode == "none":
        return sim
    offset, scale = _row_affine(sim, mode)
    out = (sim - offset) / scale
    idx = np.arange(min(sim.shape))
    out[idx, idx] = sim[idx, idx]
    return out


def top12_margin(similarity):
    """Best minus second-best score of every row."""
    if similarity.shape[1] < 2:
        raise ValueError("a confidence margin needs at least two gallery scores per query")
    two = np.partition(similarity, kth=-2, axis=1)[:, -2:]
    return np.max(two, axis=1) - np.min(two, axis=1)


def confidence_based_fusion(conv_similarity, dino_similarity):
    """alpha_q = margin_conv / (margin_conv + margin_dino + 1e-8) per query (self excluded), fused = alpha conv +
    (1 - alpha) dino."""
    if conv_similarity.shape != dino_similarity.shape:
        raise ValueError("the two similarity matrices differ in shape")
    mats = []
    for m in (conv_similarity, dino_similarity):
        m = m.astype(np.float32, copy=True)
        np.fill_diagonal(m, -np.inf)
        mats.append(m)
    mc, md = top12_margin(mats[0]), top12_margin(mats[1])
    alpha = mc / (mc + md + 1e-8)
    return {"similarity": alpha[:, None] * mats[0] + (1.0 - alpha[:, None]) * mats[1],
            "conv_selected_queries": int(np.sum(alpha >= 0.5)), "dino_selected_queries": int(np.sum(alpha < 0.5)),
            "alpha_mean": float(np.mean(alpha)), "alpha_std": float(np.std(alpha))}


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
from .adapter import IdentityReranker, Reranker  # noqa: E402,F401  (retrieval_analysis/rerank.py:10-25)
