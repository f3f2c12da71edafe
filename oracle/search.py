"""Exhaustive-search oracle (numpy + the C restatement in search_ref.c).

Test infrastructure only -- see oracle/__init__.py.  Follows the reference call sites
test.py:1080-1090 (``-torch.cdist`` / ``fill_diagonal_`` / ``topk`` / ``argsort``) and
test_nonclip.py:151 (cosine form); semantics pinned as fp64 score, ties -> lowest id.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

METRIC_IP = 0      # inner product == cosine on unit rows
METRIC_NEG_L2 = 1  # ranks by negative squared L2; reported value is -sqrt


def build():
    """Compile search_ref.c -> oracle/_build/liboracle.so (no-op when up to date)."""
    subprocess.run(["make", "-s", "-C", _HERE], check=True)
    return os.path.join(_HERE, "_build", "liboracle.so")


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "_build", "liboracle.so")
        if not os.path.exists(path) or (
                os.path.exists(os.path.join(_HERE, "search_ref.c"))
                and os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "search_ref.c"))):
            build()
        L = ctypes.CDLL(path)
        i64, p = ctypes.c_int64, ctypes.c_void_p
        L.mirx_oracle_scores.argtypes = [p, i64, p, i64, ctypes.c_int, ctypes.c_int, p]
        L.mirx_oracle_topk.argtypes = [p, i64, p, i64, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       p, p, p, p]
        L.mirx_oracle_rank_all.argtypes = [p, i64, p, i64, ctypes.c_int, ctypes.c_int, p, p, p]
        L.mirx_oracle_l2_normalize.argtypes = [p, i64, ctypes.c_int, p]
        L.mirx_oracle_bf16.argtypes = [ctypes.c_float]
        L.mirx_oracle_bf16.restype = ctypes.c_uint16
        _LIB = L
    return _LIB


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 2 or a.shape[1] % 4:
        raise ValueError("expected [n, dim] with dim % 4 == 0")
    return a


def _ptr(a):
    return None if a is None else a.ctypes.data_as(ctypes.c_void_p)


def scores(q, g, metric=METRIC_IP):
    """fp64 ranking scores [nq, n] in the lane-tree order (bit-identical to the HIP re-rank)."""
    q, g = _f32(q), _f32(g)
    out = np.empty((q.shape[0], g.shape[0]), dtype=np.float64)
    lib().mirx_oracle_scores(_ptr(q), q.shape[0], _ptr(g), g.shape[0], q.shape[1], metric, _ptr(out))
    return out


def topk(q, g, k, metric=METRIC_IP, exclude=None, ids=None):
    """(scores fp64 [nq,k], ids int64 [nq,k]); higher score first, ties -> lower id."""
    q, g = _f32(q), _f32(g)
    ex = None if exclude is None else np.ascontiguousarray(exclude, dtype=np.int64)
    idv = None if ids is None else np.ascontiguousarray(ids, dtype=np.int64)
    os_ = np.empty((q.shape[0], k), dtype=np.float64)
    oi = np.empty((q.shape[0], k), dtype=np.int64)
    lib().mirx_oracle_topk(_ptr(q), q.shape[0], _ptr(g), g.shape[0], q.shape[1], metric, k,
                           _ptr(ex), _ptr(idv), _ptr(os_), _ptr(oi))
    return os_, oi


def rank_all(q, g, metric=METRIC_IP, exclude=None, with_scores=False):
    """Full ranking [nq, n] (row = query), excluded row last."""
    q, g = _f32(q), _f32(g)
    ex = None if exclude is None else np.ascontiguousarray(exclude, dtype=np.int64)
    oi = np.empty((q.shape[0], g.shape[0]), dtype=np.int64)
    sc = np.empty((q.shape[0], g.shape[0]), dtype=np.float64) if with_scores else None
    lib().mirx_oracle_rank_all(_ptr(q), q.shape[0], _ptr(g), g.shape[0], q.shape[1], metric,
                               _ptr(ex), _ptr(oi), _ptr(sc))
    return (oi, sc) if with_scores else oi


def l2_normalize(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().mirx_oracle_l2_normalize(_ptr(x), x.shape[0], x.shape[1], _ptr(y))
    return y


def to_bf16_bits(x):
    """fp32 -> bf16 (round to nearest even) as uint16, vectorised restatement of mirx_oracle_bf16."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7FFF + ((u >> 16) & 1)
    return (u >> 16).astype(np.uint16)


def reported_value(rank_scores, metric):
    """What the boundary reports to callers: metric 0 the dot product, metric 1 -sqrt(dist^2)."""
    s = np.asarray(rank_scores, dtype=np.float64)
    if metric == METRIC_IP:
        return s
    return -np.sqrt(np.maximum(-s, 0.0))


# ---- pure-numpy cross-checks (small inputs only) -----------------------------------------

def scores_numpy_lane_tree(q, g, metric=METRIC_IP):
    """Same order as search_ref.c written with numpy; validates the C build."""
    q = np.asarray(q, dtype=np.float64)
    g = np.asarray(g, dtype=np.float64)
    nq, dim = q.shape
    n = g.shape[0]
    nchunk = dim // 4
    lanes = np.zeros((nq, n, 64), dtype=np.float64)
    for c in range(nchunk):
        l = c % 64
        for e in range(4):
            x = q[:, None, 4 * c + e]
            y = g[None, :, 4 * c + e]
            if metric == METRIC_IP:
                lanes[:, :, l] = lanes[:, :, l] + x * y      # product exact in fp64
            else:
                d = x - y
                # fma(d, d, acc): emulate with exact splitting is overkill for a cross-check;
                # callers compare with a 1-ulp tolerance for metric 1.
                lanes[:, :, l] = lanes[:, :, l] + d * d
    idx = np.arange(64)
    for off in (32, 16, 8, 4, 2, 1):
        lanes = lanes + lanes[:, :, idx ^ off]
    s = lanes[:, :, 0]
    return s if metric == METRIC_IP else -s


def topk_blas_f64(q, g, k, exclude=None):
    """fp64 BLAS matmul + stable sort: an independent statement of 'fp64, ties -> lowest id'."""
    s = np.asarray(q, dtype=np.float64) @ np.asarray(g, dtype=np.float64).T
    if exclude is not None:
        for i, e in enumerate(exclude):
            if 0 <= e < s.shape[1]:
                s[i, e] = -np.inf
    order = np.argsort(-s, axis=1, kind="stable")[:, :k]
    return np.take_along_axis(s, order, axis=1), order
