"""File sources / alignment of mirx.fusion (host logic, no GPU) against what the reference's own
FileEmbeddingSource / align_embedding_sources produced (tests/golden/fusion_align.json), and the
matrix-level helpers against the oracle."""
import json
import os

import numpy as np
import pytest

from oracle import fusion as of

GOLD = os.path.join(os.path.dirname(__file__), "golden")
SRC = os.path.join(GOLD, "fusion_sources")


def test_file_sources_read_the_reference_formats():
    from mirx import fusion as mf
    ref = json.load(open(os.path.join(GOLD, "fusion_align.json")))
    conv = mf.build_embedding_source({"type": "file", "path": os.path.join(SRC, "conv.npz"), "name": "convnext"})
    dino = mf.build_embedding_source({"type": "file", "path": os.path.join(SRC, "dino.json"), "name": "dino"})
    for src, want in ((conv, ref["conv_records"]), (dino, ref["dino_records"])):
        got = src.fetch_all()
        assert [r.image_path for r in got] == [w["image_path"] for w in want]
        assert [r.label for r in got] == [w["label"] for w in want]
        assert [r.source_name for r in got] == [w["source_name"] for w in want]
        assert all(r.embedding.dtype == np.float32 for r in got)
        np.testing.assert_array_equal(np.stack([r.embedding for r in got]).astype(np.float64),
                                      np.asarray([w["embedding"] for w in want]))
    al = mf.align_embedding_sources(conv, dino)
    w = ref["aligned"]
    assert al.image_paths == w["image_paths"] and al.labels == w["labels"] and al.coverage == w["coverage"]
    np.testing.assert_array_equal(al.conv_embeddings.astype(np.float64), np.asarray(w["conv_embeddings"]))
    np.testing.assert_array_equal(al.dino_embeddings.astype(np.float64), np.asarray(w["dino_embeddings"]))


def test_alignment_error_conditions_of_the_reference(tmp_path):
    """Same error CONDITIONS and exception type (ValueError) as fusion_eval/align.py; the wording is this build's own."""
    from mirx import fusion as mf
    e = np.eye(3, dtype=np.float32)
    mf.save_embedding_file(tmp_path / "a.npz", ["x", "y", "z"], ["l0", "l1", "l0"], e)
    mf.save_embedding_file(tmp_path / "b.json", ["x", "y", "q"], ["l0", "OTHER", "l0"], e)
    a, b = mf.FileEmbeddingSource(tmp_path / "a.npz", "a"), mf.FileEmbeddingSource(tmp_path / "b.json", "b")
    with pytest.raises(ValueError, match="y: the sources disagree on the label"):
        mf.align_embedding_sources(a, b)
    al = mf.align_embedding_sources(a, b, strict_label_check=False)
    assert al.image_paths == ["x", "y"] and al.labels == ["l0", "l1"]
    assert al.coverage["present_in_conv_only"] == ["z"] and al.coverage["present_in_dino_only"] == ["q"]
    mf.save_embedding_file(tmp_path / "dup.npz", ["x", "x"], ["l0", "l0"], e[:2])
    with pytest.raises(ValueError, match="ConvNeXt: image_path 'x' occurs more than once"):
        mf.align_embedding_sources(mf.FileEmbeddingSource(tmp_path / "dup.npz", "d"), b)
    mf.save_embedding_file(tmp_path / "none.npz", ["u"], ["l0"], e[:1])
    with pytest.raises(ValueError, match="share no image_path"):
        mf.align_embedding_sources(mf.FileEmbeddingSource(tmp_path / "none.npz", "n"), b)
    with pytest.raises(ValueError, match="embedding dumps are .npz or .json"):
        mf.FileEmbeddingSource(tmp_path / "a.csv", "a").fetch_all()
    with pytest.raises(ValueError, match="unknown embedding source type"):
        mf.build_embedding_source({"type": "parquet"})
    # query-set restriction keeps the query file's order (align.py:176-178)
    (tmp_path / "q.txt").write_text("# comment\ny label\nmissing\nx\n")
    al = mf.align_embedding_sources(a, b, query_set_path=tmp_path / "q.txt", strict_label_check=False)
    assert al.image_paths == ["y", "x"]


@pytest.mark.parametrize("ext", [".npz", ".json"])
def test_exported_files_round_trip_and_match_the_oracle_reader(tmp_path, ext):
    from mirx import fusion as mf
    rng = np.random.default_rng(0)
    emb = rng.standard_normal((17, 9)).astype(np.float32)
    paths = [f"dir/im{i}.png" for i in range(17)]
    labels = [["a", "b"][i % 2] for i in range(17)]
    mf.save_embedding_file(tmp_path / f"g{ext}", paths, labels, emb)
    got = mf.FileEmbeddingSource(tmp_path / f"g{ext}", "g").fetch_all()
    ora = of.read_embedding_file(tmp_path / f"g{ext}")
    assert [r.image_path for r in got] == paths == [o[0] for o in ora]
    assert [r.label for r in got] == labels == [o[1] for o in ora]
    np.testing.assert_array_equal(np.stack([r.embedding for r in got]), emb)
    np.testing.assert_array_equal(np.stack([o[2] for o in ora]), emb)


def test_matrix_helpers_match_oracle():
    from mirx import fusion as mf
    z = np.load(os.path.join(GOLD, "fusion_experiments.npz"), allow_pickle=True)
    c, d = z["d24_d16_conv"], z["d24_d16_dino"]
    cs, ds = of.similarity_matrix(of.om.l2_normalize_np(c)), of.similarity_matrix(of.om.l2_normalize_np(d))
    for mode in ("none", "zscore", "minmax"):
        np.testing.assert_array_equal(mf.normalize_similarity_matrix(cs, mode), of.normalize_similarity(cs, mode))
    np.testing.assert_array_equal(mf.normalize_similarity_matrix(cs, "zscore"), z["d24_d16_conv_sim_zscore"])
    got = mf.confidence_based_fusion(cs, ds)
    np.testing.assert_array_equal(got["similarity"], z["d24_d16_conf_similarity"])
    assert [got["conv_selected_queries"], got["dino_selected_queries"]] == z["d24_d16_conf_counts"].tolist()
    with pytest.raises(ValueError, match="use one of none, zscore, minmax"):
        mf.normalize_similarity_matrix(cs, "softmax")
    with pytest.raises(ValueError, match="differ in shape"):
        mf.confidence_based_fusion(cs, ds[:5])
    with pytest.raises(ValueError, match="at least two"):
        mf.top12_margin(cs[:, :1])
