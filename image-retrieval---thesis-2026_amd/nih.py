"""NIH ChestX-ray14 multi-label gallery: ingest, query and evaluation helpers over a resident Collection.

Mirrors (paths into /root/reference):
  NIH_RETRIEVAL_PATHOLOGIES                                          read_data.py:20-35
  normalize_nih_label, parse_nih_labels_from_path, load_npy_as_pil,
  resolve_npy_paths, build_collection_name, create_nih_collection,
  build_model_and_transform, encode_npy_paths, insert_rows,
  search_collection                                                  nih_zilliz_utils.py:25-280
  build_nih_val_transform, get_backbone_image_config                 nih_multilabel_retrieval.py:48-70
  BACKBONE_SPECS                                                     nih_multilabel_training.py:35-54
  the per-query search loop                                          query_nih_zilliz.py:49-71
  evaluate_map                                                       nih_multilabel_training.py:66-99

The reference keeps the gallery in a Zilliz cluster (IVF_FLAT) and sends ONE search request per query image; here the
gallery is a mirx.retriever.Collection in HBM and `query_gallery` answers every query of a batch with one exact
search (or one full ranking when top_k is 0, the reference's "retrieve the whole gallery" mode).  The JSON items
`query_gallery` returns are what `evaluate_nih_zilliz.evaluate_results` / mirx.metrics.evaluate_results consume.
"""
import json
from dataclasses import dataclass
from pathlib import Path
from typing import Callable
from urllib.parse import unquote

import numpy as np
import torch
import torch.nn.functional as F

from .retriever import Collection, default_transform

EMBEDDING_DIM = 256
NIH_RETRIEVAL_PATHOLOGIES = ["Atelectasis", "Cardiomegaly", "Effusion", "Infiltration", "Mass", "Nodule", "Pneumonia",
                             "Pneumothorax", "Consolidation", "Edema", "Emphysema", "Fibrosis", "Pleural Thickening",
                             "Hernia"]
_FILE_TOKEN = "Chest_X-ray_"
_NIH_FIELDS = ("image_name", "label_text", "label_vector_json")


def normalize_nih_label(label_name):
    s = label_name.strip().replace("%20", " ")
    return s.replace("_", " ").replace("-", " ").lower()


def parse_nih_labels_from_path(image_path, pathology_names=None):
    """`..Chest_X-ray_<Label|Label..>_<n>.npy` -> (canonical label names, multi-hot list).  The label part is
    URL-quoted and `|`-separated; unknown pathologies raise ValueError, as does a name without the token."""
    names = list(pathology_names or NIH_RETRIEVAL_PATHOLOGIES)
    canon = {normalize_nih_label(n): n for n in names}
    for alias in ("pleural_thickening", "pleural thickening", "pleuralthickening"):
        canon.setdefault(alias, "Pleural Thickening")
    canon["pleuralthickening"] = "Pleural Thickening"
    slot = {n: i for i, n in enumerate(names)}
    stem = Path(image_path).stem
    at = stem.find(_FILE_TOKEN)
    if at < 0:
        raise ValueError(f"Unsupported NIH file name '{Path(image_path).name}'. Expected token '{_FILE_TOKEN}'.")
    encoded = stem[at + len(_FILE_TOKEN):].rsplit("_", 1)[0]
    found, hot, unknown = [], [0.0] * len(names), []
    for raw in unquote(encoded).split("|"):
        raw = raw.strip()
        name = canon.get(normalize_nih_label(raw))
        if name is None or name not in slot:
            unknown.append(raw)
            continue
        hot[slot[name]] = 1.0
        found.append(name)
    if unknown:
        raise ValueError(f"Unknown pathologies in '{Path(image_path).name}': {unknown}.")
    return found, hot


def load_npy_as_pil(image_path):
    """A stored array ([H,W], [H,W,1|3] or [1|3,H,W]; uint8 or anything min-max scalable) -> 8-bit grey PIL image."""
    from PIL import Image
    a = np.asarray(np.load(image_path))
    if a.ndim == 3 and a.shape[0] in (1, 3):
        a = np.moveaxis(a, 0, -1)
    if a.ndim == 3 and a.shape[-1] == 1:
        a = a[..., 0]
    if a.dtype != np.uint8:
        a = a.astype(np.float32)
        lo, hi = float(a.min()), float(a.max())
        if hi <= lo:
            a = np.zeros(a.shape, dtype=np.uint8)
        else:
            a = np.clip((a - lo) / (hi - lo) * 255.0, 0.0, 255.0).astype(np.uint8)
    return Image.fromarray(a).convert("L")


def resolve_npy_paths(data_dir, image_list_file=None):
    """Manifest lines `path[,anything]` (relative paths are under data_dir) or every *.npy under data_dir, sorted."""
    if image_list_file:
        paths = []
        with open(image_list_file, "r", encoding="utf-8") as fh:
            for line in fh:
                first = line.strip().split(",")[0].strip()
                if first:
                    p = Path(first)
                    paths.append(str(p if p.is_absolute() else Path(data_dir) / p))
    else:
        paths = sorted(str(p) for p in Path(data_dir).rglob("*.npy"))
    if not paths:
        raise ValueError("No .npy files found for NIH ingestion/query.")
    return paths


def build_collection_name(model_name, suffix):
    return f"nih_{model_name}_{suffix}"


def get_backbone_image_config(backbone_type):
    if backbone_type == "dinov2":
        return {"image_size": 518, "resize_size": 518}
    if backbone_type == "convnextv2":
        return {"image_size": 384, "resize_size": 432}
    raise ValueError(f"Unsupported backbone_type: {backbone_type}")


def build_nih_val_transform(image_size, resize_size):
    """convert('RGB') -> Resize(resize_size) -> CenterCrop(image_size) -> ToTensor -> Normalize(ImageNet)."""
    return default_transform(image_size, resize=resize_size)


@dataclass
class BackboneSpec:
    name: str
    model_builder: Callable
    default_backbone_name: str


def _specs():
    from .model import ConvNeXtV2MultiLabelRetrievalModel, DINOv2MultiLabelRetrievalModel
    return {
        "dinov2": BackboneSpec("dinov2", lambda num_labels, backbone_name, pretrained: DINOv2MultiLabelRetrievalModel(
            num_labels=num_labels, backbone_name=backbone_name, pretrained=pretrained), "vit_base_patch14_dinov2.lvd142m"),
        "convnextv2": BackboneSpec("convnextv2", lambda num_labels, backbone_name, pretrained:
                                   ConvNeXtV2MultiLabelRetrievalModel(num_labels=num_labels, backbone_name=backbone_name,
                                                                      pretrained=pretrained),
                                   "convnextv2_base.fcmae_ft_in22k_in1k_384"),
    }


BACKBONE_SPECS = _specs()

_COLLECTIONS = {}        # name -> Collection: the in-process stand-in for the cluster's catalogue


def create_nih_collection(collection_name, drop_old=False, metric_type="COSINE", index_type="IVF_FLAT", nlist=1024,
                          device=None):
    """Schema id / image_path / image_name / label_text / label_vector_json / embedding[256]; index_type and nlist are
    accepted for compatibility (the search is exhaustive)."""
    if drop_old:
        _COLLECTIONS.pop(collection_name, None)
    col = _COLLECTIONS.get(collection_name)
    if col is None:
        col = Collection(collection_name, EMBEDDING_DIM, "NIH multi-label gallery embeddings", metric_type=metric_type,
                         device=device, extra_fields=_NIH_FIELDS, has_label=False)
        _COLLECTIONS[collection_name] = col
    col.load()
    return col


def get_nih_collection(collection_name):
    """`Collection(name)` of the reference (query_nih_zilliz.py:30): an existing gallery by name."""
    if collection_name not in _COLLECTIONS:
        raise ValueError(f"Collection {collection_name} does not exist")
    return _COLLECTIONS[collection_name]


def build_model_and_transform(model_name, checkpoint_path, backbone_name, device, num_labels=14):
    spec = BACKBONE_SPECS[model_name]
    cfg = get_backbone_image_config(model_name)
    transform = build_nih_val_transform(cfg["image_size"], cfg["resize_size"])
    model = spec.model_builder(num_labels, backbone_name or spec.default_backbone_name, False).to(device)
    if checkpoint_path:
        ckpt = torch.load(checkpoint_path, map_location=device)
        for key in ("state_dict", "state-dict"):
            if isinstance(ckpt, dict) and key in ckpt:
                ckpt = ckpt[key]
                break
        model.load_state_dict(ckpt, strict=False)
    model.eval()
    return model, transform


def encode_npy_paths(model, transform, image_paths, device, batch_size, progress_desc="Encoding"):
    """-> one dict per image: image_path, image_name, label_names, multi_hot, embedding (float32 [256])."""
    rows = []
    for s in range(0, len(image_paths), batch_size):
        chunk = image_paths[s:s + batch_size]
        meta, tensors = [], []
        for p in chunk:
            names, hot = parse_nih_labels_from_path(p)
            tensors.append(transform(load_npy_as_pil(p)))
            meta.append({"image_path": p, "image_name": Path(p).name, "label_names": names, "multi_hot": hot})
        with torch.no_grad():
            emb = model(torch.stack(tensors).to(device, non_blocking=True))["embedding"].detach().cpu().numpy()
        for m, e in zip(meta, emb):
            m["embedding"] = e.astype(np.float32)
            rows.append(m)
    return rows


def insert_rows(collection, rows):
    collection.insert([[r["image_path"] for r in rows], [r["image_name"] for r in rows],
                       ["|".join(r["label_names"]) for r in rows], [json.dumps(r["multi_hot"]) for r in rows],
                       np.stack([np.asarray(r["embedding"], dtype=np.float32) for r in rows])])
    collection.flush()


def _hit_dict(h):
    vec = h.entity.get("label_vector_json")
    return {"id": h.id, "score": float(h.distance), "image_path": h.entity.get("image_path"),
            "image_name": h.entity.get("image_name"), "label_text": h.entity.get("label_text"),
            "label_vector": json.loads(vec) if vec is not None else None}


def search_collection(collection, query_vector, top_k, nprobe=10):
    hits = collection.search(data=[query_vector], anns_field="embedding",
                             param={"metric_type": "COSINE", "params": {"nprobe": nprobe}}, limit=top_k,
                             output_fields=["image_path", "image_name", "label_text", "label_vector_json"])
    return [_hit_dict(h) for h in hits[0]]


def query_gallery(collection, query_rows, top_k=0, nprobe=10, batch_size=4096):
    """The loop of query_nih_zilliz.py:49-71 as batched searches: -> the list of result items that script writes as
    JSON (query_image_path, query_image_name, query_label_names, query_label_vector, results).  top_k <= 0 retrieves
    the full gallery ranking."""
    k = top_k if top_k and top_k > 0 else collection.num_entities
    fields = ["image_path", "image_name", "label_text", "label_vector_json"]
    out = []
    for s in range(0, len(query_rows), batch_size):
        chunk = query_rows[s:s + batch_size]
        emb = np.stack([np.asarray(r["embedding"], dtype=np.float32) for r in chunk])
        all_hits = collection.search(data=emb, limit=k, output_fields=fields)
        for r, hits in zip(chunk, all_hits):
            out.append({"query_image_path": r["image_path"], "query_image_name": r["image_name"],
                        "query_label_names": r["label_names"], "query_label_vector": r["multi_hot"],
                        "results": [_hit_dict(h) for h in hits]})
    return out


def evaluate_map(model, data_loader, device, jaccard_threshold=0.4):
    """nih_multilabel_training.evaluate_map: mAP (percent) over all pairs of the loader's images, relevance =
    Jaccard(labels) > threshold.  Quirks kept: the query itself stays in its list with similarity -1 and counts as a
    relevant item whenever its own Jaccard (1 for any labelled image) passes the threshold; queries without a relevant
    item are skipped.  On a GPU the ranking and the AP run on the device (FlatIndex.rank_all + mirx_rank_metrics)."""
    from .metrics import _average_precision, _label_bitmasks, rank_metrics_device
    model.eval()
    embs, labs = [], []
    with torch.no_grad():
        for images, batch_labels in data_loader:
            embs.append(model(images.to(device, non_blocking=True))["embedding"])
            labs.append(batch_labels.cpu())
    emb = F.normalize(torch.cat(embs, dim=0).float(), dim=1)
    labels = torch.cat(labs, dim=0)
    n = labels.shape[0]
    masks = _label_bitmasks(labels)
    if emb.is_cuda and masks is not None:
        from .index import FlatIndex
        ix = FlatIndex(emb.shape[1], "IP", emb.device.index)
        ix.add(emb)
        ranks = ix.rank_all(emb, exclude_ids=torch.arange(n))          # self last, like its score of -1
        ap = rank_metrics_device(ranks, masks, masks, (), jaccard_threshold=float(jaccard_threshold),
                                 standard_ap=True)["ap"].cpu().numpy()
        ap = ap[~np.isnan(ap)]
        return float(np.mean(ap) * 100.0) if ap.size else 0.0
    emb = emb.cpu()
    sim = emb @ emb.t()
    sim.fill_diagonal_(-1)
    lab = labels.float()
    aps = []
    for i in range(n):
        inter = (lab[i] * lab).sum(dim=1)
        union = (lab[i] + lab).clamp(max=1).sum(dim=1)
        rel = ((inter / (union + 1e-8)) > jaccard_threshold).float().numpy()
        if rel.sum() > 0:
            aps.append(_average_precision(rel, sim[i].numpy()))
    return float(np.mean(aps) * 100.0) if aps else 0.0
