"""In-process replacement for the reference's Milvus layer, same object protocol.

Mirrors (paths into /root/reference):
  MODEL_CONFIGS, MilvusManager                      milvus/milvus_setup.py:19-252
  MilvusRetriever.search / batch_search             milvus/milvus_retrieval.py:15-140
  get_model_and_transform                           milvus/milvus_retrieval.py:143-200
  collection.insert([paths, labels, embeddings])    ingest_embeddings.py:399-411
  search_by_embeddings                              retrieval_analysis/milvus_adapter.py:218-275 (mirx.adapter)
  NIH helpers                                       mirx.nih (nih_zilliz_utils.py)
The "server" is a device-resident FlatIndex (libmirx); searches are EXACT (the reference's
IVF_FLAT nlist=1024 / nprobe=10 index is approximate, milvus_setup.py:191-213) and
`search_params` / `nprobe` are accepted and ignored.  Metadata (image_path, label) stays on
the host, keyed by the int64 ids the index returns.

Errors follow the reference: unknown model type -> ValueError; connect() swallows failures
and returns False; everything else raises ordinary exceptions (callers wrap each query in
try/except, evaluate_test_dataset_milvus.py:586-590).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .index import FlatIndex

MODEL_CONFIGS = {
    "densenet121": {"embedding_dim": 1024, "description": "DenseNet121 image embeddings"},
    "resnet50": {"embedding_dim": 2048, "description": "ResNet50 image embeddings"},
    "convnextv2": {"embedding_dim": 1024, "description": "ConvNeXtV2 image embeddings"},
    "convnextv2_sra": {"embedding_dim": 1024, "description": "ConvNeXtV2_SRA image embeddings"},
    "dinov2": {"embedding_dim": 512, "description": "DINOv2 image embeddings"},
    "medsiglip": {"embedding_dim": 512, "description": "MedSigLIP image embeddings"},
}
for _name, _cfg in MODEL_CONFIGS.items():
    _suffix = "isic_image_retrieval_medsiglip" if _name == "medsiglip" else f"image_retrieval_{_name}"
    _cfg["collection_names"] = {"default": _suffix, "isic": f"isic_image_retrieval_{_name}",
                                "covid": f"covid_image_retrieval_{_name}"}


class _Entity:
    def __init__(self, fields):
        self._f = fields

    def get(self, name, default=None):
        return self._f.get(name, default)


class Hit:
    """pymilvus-like hit: .id, .distance, .entity.get(field)."""

    def __init__(self, id_, distance, fields):
        self.id = id_
        self.distance = distance
        self.entity = _Entity(fields)


class Collection:
    """The slice of pymilvus.Collection the reference uses: insert / flush / load / search /
    num_entities / name / description, backed by a FlatIndex."""

    def __init__(self, name, dim, description="", metric_type="COSINE", device=None, extra_fields=(), has_label=True):
        self.name = name
        self.description = description
        self.dim = dim
        self.metric_type = metric_type
        # column order of insert(): the default schema is id, image_path, label, embedding (milvus_setup.py:169-176)
        # followed by any extra scalar fields; the NIH schema (nih_zilliz_utils.py:149-156, has_label=False) is id,
        # image_path, <extra fields>, embedding
        if has_label:
            self.schema = ["id", "image_path", "label", "embedding"] + list(extra_fields)
        else:
            self.schema = ["id", "image_path"] + list(extra_fields) + ["embedding"]
        self._device = device
        self._index = None
        self._meta = {f: [] for f in self.schema if f not in ("id", "embedding")}

    def _ensure(self):
        if self._index is None:
            self._index = FlatIndex(self.dim, self.metric_type, self._device)
        return self._index

    @property
    def num_entities(self):
        return len(self._meta["image_path"])

    @property
    def index(self):
        return self._ensure()

    def create_index(self, field_name="embedding", index_params=None):
        metric = (index_params or {}).get("metric_type", self.metric_type)
        if self._index is not None and len(self._index) and metric != self.metric_type:
            raise ValueError("metric_type cannot change after rows were inserted")
        self.metric_type = metric
        return True

    def load(self):
        self._ensure()

    def flush(self):
        return None

    def insert(self, columns):
        """columns = one list per schema field after `id`, in schema order: [image_paths, labels, embeddings]
        (ingest_embeddings.py:402-408) for the default schema.  Returns the assigned ids (auto_id: previous size + i)."""
        fields = self.schema[1:]
        if len(columns) != len(fields):
            raise ValueError(f"expected {len(fields)} columns ({', '.join(fields)}), got {len(columns)}")
        cols = dict(zip(fields, columns))
        emb = cols.pop("embedding")
        emb = torch.as_tensor(np.asarray(emb, dtype=np.float32) if not torch.is_tensor(emb) else emb)
        if emb.dim() != 2 or emb.shape[1] != self.dim:
            raise ValueError(f"embedding dimension mismatch: expected {self.dim}, got {tuple(emb.shape)}")
        if any(len(c) != emb.shape[0] for c in cols.values()):
            raise ValueError("column lengths differ")
        first = self.num_entities
        self._ensure().add(emb)
        for f, col in cols.items():
            self._meta[f].extend((str(p) for p in col) if f == "image_path" else col)
        return list(range(first, first + emb.shape[0]))

    def search(self, data, anns_field="embedding", param=None, limit=10, output_fields=None, exclude_ids=None):
        """-> list (one per query) of lists of Hit, best first.  `distance` follows Milvus:
        COSINE/IP -> similarity, L2 -> Euclidean distance (>= 0)."""
        q = torch.as_tensor(np.asarray(data, dtype=np.float32) if not torch.is_tensor(data) else data)
        if q.dim() == 1:
            q = q[None]
        fields = output_fields or []
        if self.num_entities == 0:
            return [[] for _ in range(q.shape[0])]
        limit = min(int(limit), self.num_entities)
        if limit > 1024:
            ids, sc = self._ensure().rank_all(q, exclude_ids=exclude_ids, with_scores=True)
            ids, sc = ids[:, :limit], sc[:, :limit]
        else:
            sc, ids = self._ensure().search(q, limit, exclude_ids=exclude_ids)
        ids, sc = ids.cpu().numpy(), sc.cpu().numpy()
        out = []
        for qi in range(q.shape[0]):
            hits = []
            for j in range(ids.shape[1]):
                i = int(ids[qi, j])
                if i < 0 or not np.isfinite(sc[qi, j]):
                    continue
                dist = float(sc[qi, j]) if self.metric_type in ("COSINE", "IP") else float(-sc[qi, j])
                hits.append(Hit(i, dist, {f: self._meta[f][i] for f in fields if f in self._meta}))
            out.append(hits)
        return out


class MilvusManager:
    """Same surface as milvus_setup.MilvusManager; "connecting" binds a GPU instead of a server."""

    def __init__(self, uri=None, token=None, user="", password="", dataset="default", device=None):
        self.uri, self.token, self.user, self.password = uri, token, user, password
        self.dataset = dataset
        self.collections = {}
        self.device = device
        self._store = {}
        self.connected = False

    def get_model_config(self, model_type):
        if model_type not in MODEL_CONFIGS:
            raise ValueError(f"Unknown model type: {model_type}")
        config = dict(MODEL_CONFIGS[model_type])
        names = config.pop("collection_names", {})
        config["collection_name"] = names.get(self.dataset, names.get("default"))
        if not config["collection_name"]:
            raise ValueError(f"No collection name configured for model '{model_type}' and dataset '{self.dataset}'")
        return config

    def connect(self):
        try:
            from . import _lib
            _lib.load()
            if not torch.cuda.is_available():
                raise RuntimeError("no GPU visible")
            self.connected = True
            return True
        except Exception as e:  # reference behaviour: report and return False (milvus_setup.py:135-137)
            print(f"Connection failed: {e}")
            return False

    def disconnect(self):
        self.connected = False

    def create_collection(self, model_type, drop_old=False):
        config = self.get_model_config(model_type)
        name = config["collection_name"]
        if drop_old and name in self._store:
            del self._store[name]
        if name not in self._store:
            self._store[name] = Collection(name, config["embedding_dim"], config["description"], device=self.device)
        self.collections[model_type] = self._store[name]
        return self._store[name]

    def create_index(self, model_type, index_type="IVF_FLAT", metric_type="COSINE", nlist=1024):
        if model_type not in self.collections:
            raise ValueError(f"Collection for {model_type} not loaded")
        # index_type / nlist are accepted for compatibility: the search is always exhaustive
        self.collections[model_type].create_index("embedding", {"metric_type": metric_type})
        return True

    def load_collection(self, model_type):
        if model_type not in self.collections:
            name = self.get_model_config(model_type)["collection_name"]
            if name not in self._store:
                raise ValueError(f"Collection {name} does not exist")
            self.collections[model_type] = self._store[name]
        self.collections[model_type].load()
        return self.collections[model_type]

    def get_collection_info(self, model_type):
        if model_type not in self.collections:
            self.load_collection(model_type)
        c = self.collections[model_type]
        return {"name": c.name, "num_entities": c.num_entities, "description": c.description, "schema": c.schema}

    def setup_all_collections(self, drop_old=False, metric_type="COSINE"):
        for model_type in MODEL_CONFIGS:
            self.create_collection(model_type, drop_old=drop_old)
            self.create_index(model_type, metric_type=metric_type)
            self.load_collection(model_type)


IMAGENET_MEAN, IMAGENET_STD = (0.485, 0.456, 0.406), (0.229, 0.224, 0.225)
SIGLIP_MEAN, SIGLIP_STD = (0.5, 0.5, 0.5), (0.5, 0.5, 0.5)


def default_transform(img_size=224, mean=IMAGENET_MEAN, std=IMAGENET_STD, resize=None):
    """convert('RGB') -> Resize(shorter side) -> CenterCrop -> ToTensor -> Normalize(mean, std): the pipeline of
    milvus_retrieval.py:176-198 without torchvision (PIL + torch only).  Resize defaults follow the reference
    (448 -> 512, 384 -> 432, otherwise 256); MedSigLIP normalises with mean = std = 0.5 (milvus_retrieval.py:177-178),
    every other model with the ImageNet statistics."""
    if resize is None:
        resize = {448: 512, 384: 432}.get(img_size, 256)
    # numpy float32 arithmetic (the same IEEE operations as ToTensor + Normalize, bit for bit): torch's CPU operators cost
    # 10+ ms per call on a 150 k-element image when the intra-op thread pool is larger than the cores the process may use
    mean = np.asarray(mean, dtype=np.float32).reshape(3, 1, 1)
    std = np.asarray(std, dtype=np.float32).reshape(3, 1, 1)

    def pixels(img):
        """convert -> Resize -> CenterCrop: the 8-bit pixels [3, S, S] that ToTensor + Normalize would then turn into floats"""
        from PIL import Image
        img = img.convert("RGB")
        w, h = img.size
        if w <= h:
            nw, nh = resize, int(resize * h / w)
        else:
            nw, nh = int(resize * w / h), resize
        img = img.resize((nw, nh), Image.BILINEAR)
        left, top = int(round((nw - img_size) / 2.0)), int(round((nh - img_size) / 2.0))
        img = img.crop((left, top, left + img_size, top + img_size))
        return np.ascontiguousarray(np.asarray(img, dtype=np.uint8).transpose(2, 0, 1))

    def tf(img):
        x = pixels(img).astype(np.float32) / np.float32(255.0)
        return torch.from_numpy(np.ascontiguousarray((x - mean) / std))

    # MilvusRetriever.search hands the 8-bit pixels to a model that normalises them itself with the same constants (DenseNet121:
    # inside the stem kernel, bit-identical to the float path -- tests/test_model_gpu.py): a quarter of the bytes to the device
    # and no float pass over the image on the host
    tf.pixels = pixels
    tf.mean, tf.std = tuple(float(v) for v in mean.ravel()), tuple(float(v) for v in std.ravel())
    return tf


class MilvusRetriever:
    """milvus_retrieval.MilvusRetriever with an exact in-process search."""

    def __init__(self, manager, model_type, model, transform):
        self.manager = manager
        self.model_type = model_type
        self.model = model
        self.transform = transform
        self.collection = None

    def load_collection(self):
        if self.collection is None:
            self.manager.load_collection(self.model_type)
            self.collection = self.manager.collections[self.model_type]
        return self.collection

    def _device(self):
        if hasattr(self.model, "device"):
            return self.model.device
        try:
            return next(self.model.parameters()).device
        except StopIteration:
            return torch.device("cuda")

    def _query_tensor(self, img):
        """transform(img)[None] -- or, when the transform is default_transform and the model normalises 8-bit input itself with
        the transform's constants, the 8-bit pixels (same embedding bit for bit, a quarter of the bytes, no host float pass)."""
        tf, m = self.transform, self.model
        px = getattr(tf, "pixels", None)
        if px is not None and getattr(m, "accepts_uint8", False) and self._device().type == "cuda" and not m.training:
            mean, std = getattr(m, "input_mean", None), getattr(m, "input_std", None)
            if mean is not None and tuple(float(v) for v in mean) == tf.mean and tuple(float(v) for v in std) == tf.std:
                return torch.from_numpy(px(img)).unsqueeze(0)
        return tf(img).unsqueeze(0)

    def embed(self, images):
        """Batched F.normalize(model(x)) for a [B,3,H,W] tensor (milvus_retrieval.py:60-63)."""
        with torch.no_grad():
            out = self.model(images.to(self._device()))
            if isinstance(out, dict):
                out = out["embedding"]
            return F.normalize(out, p=2, dim=1)

    def search(self, query_image_path, top_k=10, search_params=None, metric_type="COSINE"):
        """-> (results, query_embedding).  results: best-first dicts {id, image_path, label,
        distance, similarity}; similarity = distance for COSINE/IP, 1 - d^2/2 for L2."""
        if isinstance(query_image_path, str):
            from PIL import Image
            img = Image.open(query_image_path).convert("RGB")
        else:
            img = query_image_path
        query_embedding = self.embed(self._query_tensor(img))
        if self.collection is None:
            self.load_collection()
        if metric_type not in ("COSINE", "IP", "L2"):
            raise ValueError(f"Unknown metric type: {metric_type}")
        if metric_type != self.collection.metric_type and {metric_type, self.collection.metric_type} != {"COSINE", "IP"}:
            # Milvus rejects a search whose metric differs from the index's (milvus_retrieval.py:75-86)
            raise ValueError(f"metric type not match: index has {self.collection.metric_type}, search asked {metric_type}")
        hits = self.collection.search(data=query_embedding, anns_field="embedding", param=search_params,
                                      limit=top_k, output_fields=["image_path", "label"])[0]
        return [self._format(h, metric_type) for h in hits], query_embedding

    @staticmethod
    def _format(hit, metric_type):
        d = hit.distance
        sim = d if metric_type in ("COSINE", "IP") else 1.0 - (d * d) / 2.0
        return {"id": hit.id, "image_path": hit.entity.get("image_path"), "label": hit.entity.get("label"),
                "distance": d, "similarity": sim}

    def batch_search(self, query_image_paths, top_k=10, search_params=None):
        """One batched embed + one batched search (the reference loops search(); results equal)."""
        if len(query_image_paths) == 0:
            return []
        from PIL import Image
        imgs = [Image.open(p).convert("RGB") if isinstance(p, str) else p for p in query_image_paths]
        emb = self.embed(torch.stack([self.transform(i) for i in imgs]))
        if self.collection is None:
            self.load_collection()
        all_hits = self.collection.search(data=emb, limit=top_k, output_fields=["image_path", "label"])
        return [[self._format(h, self.collection.metric_type) for h in hits] for hits in all_hits]

    def adapter(self, name=None):
        """This retriever's collection seen through retrieval_analysis' MilvusCollectionAdapter (mirx.adapter)."""
        from .adapter import MilvusCollectionAdapter, MilvusCollectionConfig
        if self.collection is None:
            self.load_collection()
        return MilvusCollectionAdapter(MilvusCollectionConfig(name=name or self.model_type,
                                                              collection_name=self.collection.name),
                                       collection=self.collection)

    def search_by_embeddings(self, queries, query_embeddings, top_k, search_params=None, reranker=None,
                             exclude_self=True, metadata_fields=None, batch_size=None):
        """retrieval_analysis/milvus_adapter.py:218-275, same arguments and return type (list[SearchResult]): one
        batched exact search; self dropped by image_path after fetching top_k + 1; `reranker.rerank(query, results)`
        applied before the cut to top_k."""
        return self.adapter().search_by_embeddings(queries, query_embeddings, top_k, search_params, reranker,
                                                   exclude_self, metadata_fields, batch_size)

    def search_ids(self, query_embeddings, top_k=10, exclude_ids=None):
        """Device-side batch form (no reference counterpart): -> best-first dicts per query, self excluded by id."""
        if self.collection is None:
            self.load_collection()
        hits = self.collection.search(data=query_embeddings, limit=top_k, output_fields=["image_path", "label"],
                                      exclude_ids=exclude_ids)
        return [[self._format(h, self.collection.metric_type) for h in hs] for hs in hits]


def search_collection(collection, query_vector, top_k, nprobe=10):
    """nih_zilliz_utils.search_collection (lives in mirx.nih with the other NIH helpers)."""
    from .nih import search_collection as _sc
    return _sc(collection, query_vector, top_k, nprobe)


def get_model_and_transform(model_type, model_weights, embedding_dim, device):
    """milvus_retrieval.get_model_and_transform for the model families built so far."""
    from .model import build_model
    model, img_size = build_model(model_type, embedding_dim=embedding_dim)
    if model_weights:
        checkpoint = torch.load(model_weights, map_location=device)
        for key in ("state-dict", "state_dict"):
            if isinstance(checkpoint, dict) and key in checkpoint:
                checkpoint = checkpoint[key]
        model.load_state_dict(checkpoint, strict=False)
    model.eval()
    model.to(device)
    if model_type == "medsiglip":                  # SigLIP normalisation (milvus_retrieval.py:176-181)
        return model, default_transform(img_size, SIGLIP_MEAN, SIGLIP_STD)
    return model, default_transform(img_size)
