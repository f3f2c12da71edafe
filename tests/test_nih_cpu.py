"""NIH helpers, query-set loader and transforms against golden outputs of the reference's own functions
(tests/golden/make_golden_r2.py: nih_zilliz_utils.py:25-280, retrieval_analysis/comparison.py:41-84).  CPU only."""
import json
import os

import numpy as np
import pytest
import torch


@pytest.fixture(scope="module")
def nih(golden_dir):
    with open(os.path.join(golden_dir, "nih_helpers.json")) as fh:
        return json.load(fh)


def test_parse_labels_and_names(nih):
    from mirx.nih import NIH_RETRIEVAL_PATHOLOGIES, build_collection_name, normalize_nih_label, parse_nih_labels_from_path
    assert len(NIH_RETRIEVAL_PATHOLOGIES) == 14
    for case in nih["parsed"]:
        names, hot = parse_nih_labels_from_path(case["path"])
        assert names == case["label_names"] and hot == case["multi_hot"], case["path"]
    for case in nih["errors"]:
        if case["raises"]:
            with pytest.raises(ValueError):
                parse_nih_labels_from_path(case["path"])
        else:
            parse_nih_labels_from_path(case["path"])
    assert build_collection_name("dinov2", "gallery") == nih["collection_name"]
    for k, v in nih["normalize"].items():
        assert normalize_nih_label(k) == v


def test_load_npy_as_pil(golden_dir, tmp_path):
    from mirx.nih import load_npy_as_pil
    z = np.load(os.path.join(golden_dir, "nih_load_npy.npz"))
    for key in [k[3:] for k in z.files if k.startswith("in_")]:
        p = tmp_path / f"{key}.npy"
        np.save(p, z["in_" + key])
        img = load_npy_as_pil(str(p))
        assert img.mode == "L"
        np.testing.assert_array_equal(np.asarray(img), z["out_" + key], err_msg=key)


def test_resolve_npy_paths(nih, tmp_path):
    from mirx.nih import resolve_npy_paths
    os.makedirs(tmp_path / "g" / "sub")
    for rel in ("g/b.npy", "g/a.npy", "g/sub/c.npy"):
        np.save(tmp_path / rel, np.zeros((2, 2), np.uint8))
    with open(tmp_path / "list.txt", "w") as fh:
        fh.write("sub/c.npy, 3\n\n" + str(tmp_path / "g" / "a.npy") + "\n b.npy ,x,y\n")
    got = [os.path.relpath(p, tmp_path) for p in resolve_npy_paths(str(tmp_path / "g"), str(tmp_path / "list.txt"))]
    assert got == nih["resolve_manifest"]
    assert [os.path.relpath(p, tmp_path) for p in resolve_npy_paths(str(tmp_path / "g"))] == nih["resolve_walk"]
    assert nih["resolve_empty_raises"]
    with pytest.raises(ValueError):
        resolve_npy_paths(str(tmp_path / "g" / "sub" / "nothing"))


def test_encode_insert_and_search_shapes(nih, tmp_path):
    from mirx.nih import encode_npy_paths, insert_rows, search_collection
    enc = nih["encode"]
    paths = []
    for name, arr in zip(enc["names"], enc["arrays"]):
        p = tmp_path / name
        np.save(p, np.asarray(arr, dtype=np.uint8))
        paths.append(str(p))
    proj = torch.tensor(enc["proj"], dtype=torch.float32)

    def transform(image):
        return torch.as_tensor(np.asarray(image, dtype=np.float32) / 255.0).reshape(1, 6, 6)

    class _Enc(torch.nn.Module):
        def forward(self, x):
            return {"embedding": torch.nn.functional.normalize(x.flatten(1) @ proj, dim=1)}

    rows = encode_npy_paths(_Enc(), transform, paths, torch.device("cpu"), batch_size=2)
    assert len(rows) == len(enc["rows"])
    for got, want, p in zip(rows, enc["rows"], paths):
        assert got["image_path"] == p and got["image_name"] == want["image_name"]
        assert got["label_names"] == want["label_names"] and got["multi_hot"] == want["multi_hot"]
        assert got["embedding"].dtype == np.float32
        np.testing.assert_allclose(got["embedding"], np.asarray(want["embedding"], dtype=np.float32), atol=1e-7)

    class _Col:
        inserted, flushed = None, 0

        def insert(self, cols):
            self.inserted = cols

        def flush(self):
            self.flushed += 1

    col = _Col()
    insert_rows(col, rows)
    assert col.flushed == 1 and col.inserted[0] == paths
    assert [col.inserted[1], col.inserted[2], col.inserted[3]] == nih["insert_columns"][:3]
    np.testing.assert_allclose(np.asarray(col.inserted[4]), np.asarray(nih["insert_columns"][3]), atol=1e-7)

    class _Hit:
        def __init__(self, i, dist, fields):
            self.id, self.distance = i, dist
            self.entity = fields

    class _SearchCol:
        def search(self, data, anns_field, param, limit, output_fields):
            self.seen = {"anns_field": anns_field, "param": param, "limit": limit, "output_fields": output_fields}
            return [[_Hit(11, 0.75, {"image_path": "/g/x.npy", "image_name": "x.npy", "label_text": "Mass",
                                     "label_vector_json": json.dumps([0.0, 1.0])}),
                     _Hit(4, 0.5, {"image_path": "/g/y.npy", "image_name": "y.npy", "label_text": "",
                                   "label_vector_json": json.dumps([0.0, 0.0])})]]

    sc = _SearchCol()
    hits = search_collection(sc, [0.1, 0.2], top_k=2, nprobe=7)
    assert sc.seen == nih["search_collection"]["seen"] and hits == nih["search_collection"]["hits"]


def test_load_query_set(golden_dir, tmp_path):
    from mirx.adapter import QueryRecord, load_query_set
    with open(os.path.join(golden_dir, "query_sets.json")) as fh:
        qs = json.load(fh)
    for name, case in qs.items():
        p = tmp_path / name
        p.write_text(case["content"])
        got = load_query_set(p)
        assert all(isinstance(q, QueryRecord) for q in got)
        assert [[q.image_path, q.label] for q in got] == case["records"], name


def test_transforms_per_model_type():
    """milvus_retrieval.py:176-198: MedSigLIP normalises with mean = std = 0.5, every other model with the ImageNet
    statistics; resize 256 / 432 / 512 by crop size; NIH transforms resize 518 -> 518 and 432 -> 384."""
    from PIL import Image
    from mirx.nih import build_nih_val_transform, get_backbone_image_config
    from mirx.retriever import IMAGENET_MEAN, IMAGENET_STD, SIGLIP_MEAN, SIGLIP_STD, default_transform
    rng = np.random.default_rng(0)
    img = Image.fromarray((rng.random((300, 260, 3)) * 255).astype(np.uint8))
    raw = None
    for size, resize in ((224, 256), (384, 432), (448, 512)):
        nw, nh = (resize, int(resize * 300 / 260))
        want = img.convert("RGB").resize((nw, nh), Image.BILINEAR)
        left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
        raw = torch.from_numpy(np.asarray(want.crop((left, top, left + size, top + size)), dtype=np.uint8).copy())
        raw = raw.permute(2, 0, 1).float() / 255.0
        x = default_transform(size)(img)
        assert x.shape == (3, size, size)
        torch.testing.assert_close(x, (raw - torch.tensor(IMAGENET_MEAN).view(3, 1, 1)) / torch.tensor(IMAGENET_STD).view(3, 1, 1))
    xs = default_transform(448, SIGLIP_MEAN, SIGLIP_STD)(img)
    torch.testing.assert_close(xs, (raw - 0.5) / 0.5)
    assert float(xs.min()) >= -1.0 and float(xs.max()) <= 1.0
    assert get_backbone_image_config("dinov2") == {"image_size": 518, "resize_size": 518}
    assert get_backbone_image_config("convnextv2") == {"image_size": 384, "resize_size": 432}
    with pytest.raises(ValueError):
        get_backbone_image_config("resnet")
    assert build_nih_val_transform(384, 432)(img.convert("L")).shape == (3, 384, 384)


def test_get_model_and_transform_medsiglip_uses_siglip_stats(monkeypatch):
    import mirx.retriever as R
    import mirx.model as M
    seen = {}

    class _Tiny(torch.nn.Module):
        def forward(self, x):
            return x.flatten(1)[:, :4]

    monkeypatch.setattr(M, "build_model", lambda model_type, embedding_dim=None, **kw: (_Tiny(), {"medsiglip": 448}.get(model_type, 224)))
    real = R.default_transform

    def spy(img_size=224, mean=R.IMAGENET_MEAN, std=R.IMAGENET_STD, resize=None):
        seen["args"] = (img_size, tuple(mean), tuple(std))
        return real(img_size, mean, std, resize)

    monkeypatch.setattr(R, "default_transform", spy)
    R.get_model_and_transform("medsiglip", None, 512, "cpu")
    assert seen["args"] == (448, (0.5, 0.5, 0.5), (0.5, 0.5, 0.5))
    R.get_model_and_transform("densenet121", None, None, "cpu")
    assert seen["args"] == (224, R.IMAGENET_MEAN, R.IMAGENET_STD)


def test_densenet_cache_follows_weight_changes_through_a_parent_module():
    """folded weights must be rebuilt when the parameters change by ANY route (here: load_state_dict on a
    parent module and an in-place edit), not only through this module's own load_state_dict / train / _apply."""
    from mirx.model import DenseNet121
    torch.manual_seed(0)
    m = DenseNet121().eval()
    c1 = m._cache()
    assert m._cache() is c1                                           # unchanged weights: same cache
    parent = torch.nn.Sequential(m)
    sd = {k: v.clone() for k, v in parent.state_dict().items()}
    sd["0.densenet121.0.norm5.bias"] += 1.0
    parent.load_state_dict(sd)
    c2 = m._cache()
    assert c2 is not c1
    torch.testing.assert_close(c2["norm5"][1], c1["norm5"][1] + 1.0)
    with torch.no_grad():
        m.densenet121[0].denseblock1.denselayer1.conv1.weight.mul_(2.0)
    c3 = m._cache()
    assert c3 is not c2
    torch.testing.assert_close(c3["denseblock1"]["denselayer1"][2], 2.0 * c2["denseblock1"]["denselayer1"][2])
