"""Batch search boundary of the reference's retrieval-analysis tools, over a resident Collection.

Mirrors (paths into /root/reference):
  MilvusCollectionConfig, QueryRecord, RetrievedItem, SearchResult,
  MilvusCollectionAdapter.{list_image_paths, fetch_record_by_image_path, fetch_records_by_image_paths,
                           search_by_embedding, search_by_embeddings}     retrieval_analysis/milvus_adapter.py:11-306
  compare_collection_coverage, filter_present_queries                     retrieval_analysis/milvus_adapter.py:309-336
  Reranker, IdentityReranker                                              retrieval_analysis/rerank.py:10-25
  load_query_set                                                          retrieval_analysis/comparison.py:41-84

The reference talks to a remote Milvus through `MilvusClient`; here the "cluster" is a mirx.retriever.Collection whose
rows live in HBM, and one `search_by_embeddings` call is ONE batched exact search (bf16 MFMA candidates + fp64
re-rank) instead of an IVF_FLAT probe per request.  Behaviour kept from the reference:
  * `exclude_self` asks for top_k + 1 hits and drops every hit whose image_path equals the query's (a path match, not
    an id match, so duplicated paths behave as they do there); the list is cut to top_k AFTER the reranker ran;
  * `reranker.rerank(query=..., results=...)` is called once per query with the self-filtered hits;
  * `batch_size` splits the request; `search_params` are accepted and ignored (the search is exhaustive);
  * hits carry `score == distance` (COSINE / IP similarity; L2: Euclidean distance) and the raw hit dict.
"""
import csv
import json
from dataclasses import dataclass, field
from pathlib import Path
from typing import Any, Dict, Iterable, List, Optional, Sequence

import numpy as np
import torch


@dataclass(frozen=True)
class MilvusCollectionConfig:
    """Field names and (unused here) connection settings of one collection."""
    name: str
    collection_name: str
    uri: Optional[str] = None
    token: Optional[str] = None
    user: Optional[str] = None
    password: Optional[str] = None
    db_name: Optional[str] = None
    host: Optional[str] = None
    port: Optional[int] = None
    vector_field: str = "embedding"
    id_field: str = "id"
    image_path_field: str = "image_path"
    label_field: str = "label"
    output_fields: Sequence[str] = field(default_factory=lambda: ("id", "image_path", "label"))


@dataclass
class QueryRecord:
    image_path: str
    label: Optional[str] = None


@dataclass
class RetrievedItem:
    id: Optional[Any]
    image_path: Optional[str]
    label: Optional[str]
    score: Optional[float]
    distance: Optional[float]
    raw: Dict[str, Any] = field(default_factory=dict)


@dataclass
class SearchResult:
    query: QueryRecord
    query_source: str
    retrieved: List[RetrievedItem]
    query_embedding: Sequence[float]


class Reranker:
    """One-method hook: rerank(query, results) -> iterable of RetrievedItem (a Protocol in the reference)."""

    def rerank(self, query, results):
        raise NotImplementedError


class IdentityReranker(Reranker):
    def rerank(self, query, results):
        return list(results)


def load_query_set(path):
    """Ordered QueryRecord list from .json ({"queries"|"results": [...]} or a list; keys image_path /
    query_image_path, label), .csv (columns image_path|query_image_path, label|query_label) or whitespace text
    (`path [label]`, `#` comments)."""
    path = Path(path)
    kind = path.suffix.lower()
    out = []
    if kind == ".json":
        with path.open("r", encoding="utf-8") as fh:
            items = json.load(fh)
        if isinstance(items, dict):
            items = items.get("queries", items.get("results", []))
        for it in items:
            p = it.get("image_path", it.get("query_image_path"))
            if p:
                out.append(QueryRecord(image_path=p, label=it.get("label")))
    elif kind == ".csv":
        with path.open("r", encoding="utf-8", newline="") as fh:
            for row in csv.DictReader(fh):
                p = row.get("image_path") or row.get("query_image_path")
                if p:
                    out.append(QueryRecord(image_path=p, label=row.get("label", row.get("query_label"))))
    else:
        with path.open("r", encoding="utf-8") as fh:
            for line in fh:
                tok = line.split()
                if tok and not tok[0].startswith("#"):
                    out.append(QueryRecord(image_path=tok[0], label=tok[1] if len(tok) > 1 else None))
    return out


class MilvusCollectionAdapter:
    """`MilvusCollectionAdapter(config, collection=<mirx.retriever.Collection>)`.  Without `collection` the reference
    would open a network client; there is none here, so that raises."""

    def __init__(self, config, collection=None):
        if collection is None:
            raise ValueError(f"{config.name}: in-process build -- pass the resident Collection as collection=")
        self.config = config
        self.collection = collection

    # -- metadata ------------------------------------------------------------------------------------------------
    def _paths(self):
        return self.collection._meta[self.config.image_path_field]

    def list_image_paths(self, batch_size=1000):
        return [p for p in self._paths() if p]

    def _row(self, i, include_embedding, vec=None):
        cfg = self.config
        row = {cfg.id_field: i}
        for f in cfg.output_fields:
            if f in self.collection._meta:
                row[f] = self.collection._meta[f][i]
        row.setdefault(cfg.image_path_field, self._paths()[i])
        if cfg.label_field in self.collection._meta:
            row.setdefault(cfg.label_field, self.collection._meta[cfg.label_field][i])
        if include_embedding:
            if vec is None:
                vec = self.collection.index.rows(i, 1)[0][0].cpu().numpy()
            row[cfg.vector_field] = vec
        return row

    def _rows_of(self, wanted):
        """image_path -> list of row numbers, for the requested paths only."""
        hits = {}
        for i, p in enumerate(self._paths()):
            if p in wanted:
                hits.setdefault(p, []).append(i)
        return hits

    def fetch_record_by_image_path(self, image_path, include_embedding=True):
        rows = self._rows_of({image_path}).get(image_path, [])
        if not rows:
            return None
        if len(rows) > 1:
            raise ValueError(f"{self.config.name}: multiple rows found for image_path={image_path}")
        return self._row(rows[0], include_embedding)

    def fetch_records_by_image_paths(self, image_paths, include_embedding=True, batch_size=100):
        wanted = {p for p in image_paths if p}
        found = self._rows_of(wanted)
        for p, rows in found.items():
            if len(rows) > 1:
                raise ValueError(f"{self.config.name}: multiple rows found for image_path={p}")
        out = {}
        if not found:
            return out
        vecs = None
        if include_embedding:
            order = sorted(r[0] for r in found.values())
            lo, hi = order[0], order[-1] + 1
            block = self.collection.index.rows(lo, hi - lo)[0].cpu().numpy()      # one device read
            vecs = {i: block[i - lo] for i in order}
        for p, rows in found.items():
            out[p] = self._row(rows[0], include_embedding, None if vecs is None else vecs[rows[0]])
        return out

    # -- search --------------------------------------------------------------------------------------------------
    def _fields(self, metadata_fields):
        fields = list(metadata_fields or self.config.output_fields)
        for f in (self.config.image_path_field, self.config.label_field):
            if f not in fields:
                fields.append(f)
        return fields

    def _item(self, hit, fields):
        entity = {f: hit.entity.get(f) for f in fields if f != self.config.id_field}
        raw = {"id": hit.id, "distance": hit.distance, "entity": entity}
        return RetrievedItem(id=hit.id, image_path=entity.get(self.config.image_path_field),
                             label=entity.get(self.config.label_field), score=hit.distance, distance=hit.distance, raw=raw)

    def search_by_embeddings(self, queries, query_embeddings, top_k, search_params=None, reranker=None,
                             exclude_self=True, metadata_fields=None, batch_size=None):
        if not queries:
            return []
        if len(queries) != len(query_embeddings):
            raise ValueError("queries and query_embeddings must have the same length")
        fields = self._fields(metadata_fields)
        limit = top_k + 1 if exclude_self else top_k
        step = max(1, int(batch_size or len(queries)))
        emb = query_embeddings if torch.is_tensor(query_embeddings) else np.asarray(query_embeddings, dtype=np.float32)
        results = []
        for s in range(0, len(queries), step):
            chunk_q = queries[s:s + step]
            chunk_e = emb[s:s + step]
            all_hits = self.collection.search(data=chunk_e, anns_field=self.config.vector_field, param=search_params,
                                              limit=limit, output_fields=fields)
            for q, e, hits in zip(chunk_q, query_embeddings[s:s + step], all_hits):
                items = [self._item(h, fields) for h in hits]
                if exclude_self:
                    items = [it for it in items if it.image_path != q.image_path]
                if reranker is not None:
                    items = list(reranker.rerank(query=q, results=items))
                results.append(SearchResult(query=q, query_source=self.config.name, retrieved=items[:top_k],
                                            query_embedding=e))
        return results

    def search_by_embedding(self, query, query_embedding, top_k, search_params=None, reranker=None, exclude_self=True,
                            metadata_fields=None):
        emb = query_embedding[None] if torch.is_tensor(query_embedding) else [query_embedding]
        res = self.search_by_embeddings([query], emb, top_k, search_params, reranker, exclude_self, metadata_fields)[0]
        res.query_embedding = query_embedding
        return res


def compare_collection_coverage(conv_adapter, dino_adapter):
    conv, dino = set(conv_adapter.list_image_paths()), set(dino_adapter.list_image_paths())
    return {"conv_only": sorted(conv - dino), "dino_only": sorted(dino - conv), "present_in_both": sorted(conv & dino)}


def filter_present_queries(queries: Iterable[QueryRecord], coverage):
    both = set(coverage["present_in_both"])
    split = {"valid": [], "missing": []}
    for q in queries:
        split["valid" if q.image_path in both else "missing"].append(q)
    return split
