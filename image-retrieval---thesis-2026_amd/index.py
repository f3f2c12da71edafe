"""FlatIndex -- device-resident exhaustive index over the C ABI (include/mirx.h).

Stands where the reference uses a Milvus collection (milvus/milvus_setup.py:139-222,
milvus/milvus_retrieval.py:80-86) or the inline torch brute force (test.py:1080-1090).
Torch is plumbing only: tensors own the query/result memory, `data_ptr()` crosses the ABI.
"""
import ctypes

import torch

from . import _lib
from ._lib import METRIC_IP, METRIC_NEG_L2, MirxError

_METRICS = {"COSINE": METRIC_IP, "IP": METRIC_IP, "L2": METRIC_NEG_L2,
            "cosine": METRIC_IP, "ip": METRIC_IP, "l2": METRIC_NEG_L2, "cdist": METRIC_NEG_L2}


def metric_code(metric):
    if isinstance(metric, int):
        return metric
    try:
        return _METRICS[metric]
    except KeyError:
        raise ValueError(f"Unknown metric type: {metric}") from None


def _stream_ptr(device):
    return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class FlatIndex:
    """Exact top-k over fp32 rows resident on one GPU.

    Ranking semantics (pinned by oracle/search_ref.c): fp64 score, ties -> lower id.
    `metric`: 'COSINE'/'IP' (dot product; cosine for unit rows) or 'L2' (ranks by distance,
    reports -||q-g||_2 like the reference's `-torch.cdist`).
    """

    def __init__(self, dim, metric="COSINE", device=None):
        if not torch.cuda.is_available():
            raise MirxError("FlatIndex needs a GPU: libmirx has no CPU path")
        self._lib = _lib.load()
        if device is None:
            device = torch.cuda.current_device()
        self.device = torch.device("cuda", device if isinstance(device, int) else torch.device(device).index or 0)
        self.dim = int(dim)
        self.metric = metric_code(metric)
        h = ctypes.c_void_p()
        _lib.check(self._lib.mirx_index_create(self.dim, self.metric, self.device.index, ctypes.byref(h)),
                   "mirx_index_create")
        self._h = h

    def __del__(self):
        h = getattr(self, "_h", None)
        if h:
            self._lib.mirx_index_destroy(h)
            self._h = None

    def __len__(self):
        return int(self._lib.mirx_index_size(self._h))

    @property
    def ntotal(self):
        return len(self)

    def set_option(self, option, value):
        _lib.check(self._lib.mirx_index_set_option(self._h, option, int(value)), "mirx_index_set_option")

    def reserve(self, rows):
        _lib.check(self._lib.mirx_index_reserve(self._h, int(rows)), "mirx_index_reserve")

    def add(self, rows, ids=None):
        """Append rows: torch tensor (cpu or this GPU) or anything `torch.as_tensor` accepts."""
        rows = torch.as_tensor(rows)
        if rows.dim() != 2 or rows.shape[1] != self.dim:
            raise ValueError(f"expected [n, {self.dim}] rows, got {tuple(rows.shape)}")
        rows = rows.detach().to(torch.float32).contiguous()
        if rows.is_cuda and rows.device != self.device:
            rows = rows.to(self.device)
        idp = None
        if ids is not None:
            ids = torch.as_tensor(ids).detach().to(torch.int64).contiguous()
            if ids.numel() != rows.shape[0]:
                raise ValueError("ids and rows disagree in length")
            if ids.is_cuda and ids.device != self.device:
                ids = ids.to(self.device)
            idp = ctypes.c_void_p(ids.data_ptr())
        if rows.is_cuda:
            torch.cuda.current_stream(self.device).synchronize()
        _lib.check(self._lib.mirx_index_add(self._h, ctypes.c_void_p(rows.data_ptr()), rows.shape[0], idp),
                   "mirx_index_add")

    def _prep_queries(self, q, exclude_ids):
        q = torch.as_tensor(q)
        if q.dim() == 1:
            q = q[None]
        if q.dim() != 2 or q.shape[1] != self.dim:
            raise ValueError(f"expected [nq, {self.dim}] queries, got {tuple(q.shape)}")
        q = q.detach().to(device=self.device, dtype=torch.float32).contiguous()
        ex = None
        if exclude_ids is not None:
            ex = torch.as_tensor(exclude_ids).detach().to(device=self.device, dtype=torch.int64).contiguous()
            if ex.numel() != q.shape[0]:
                raise ValueError("exclude_ids needs one id per query")
        return q, ex

    def search(self, q, k, exclude_ids=None, return_f64=False):
        """-> (scores [nq,k] float32 reported values, ids [nq,k] int64) on the index device.

        With return_f64=True the first tensor holds the fp64 ranking scores instead
        (dot product, or NEGATIVE SQUARED distance for the L2 metric)."""
        q, ex = self._prep_queries(q, exclude_ids)
        nq = q.shape[0]
        ids = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        exp = ctypes.c_void_p(ex.data_ptr()) if ex is not None else None
        with torch.cuda.device(self.device):
            st = _stream_ptr(self.device)
            if return_f64:
                sc = torch.empty((nq, k), dtype=torch.float64, device=self.device)
                rc = self._lib.mirx_index_search_f64(self._h, ctypes.c_void_p(q.data_ptr()), nq, int(k), exp,
                                                     ctypes.c_void_p(sc.data_ptr()),
                                                     ctypes.c_void_p(ids.data_ptr()), st)
            else:
                sc = torch.empty((nq, k), dtype=torch.float32, device=self.device)
                rc = self._lib.mirx_index_search(self._h, ctypes.c_void_p(q.data_ptr()), nq, int(k), exp,
                                                 ctypes.c_void_p(sc.data_ptr()),
                                                 ctypes.c_void_p(ids.data_ptr()), st)
        _lib.check(rc, "mirx_index_search")
        return sc, ids

    def search_begin(self, q, k, exclude_ids=None):
        """First half of search(): enqueue the whole first pass on the current stream and return without waiting.
        -> a handle for search_end(); the index must not be used for anything else in between."""
        q, ex = self._prep_queries(q, exclude_ids)
        nq = q.shape[0]
        ids = torch.empty((nq, k), dtype=torch.int64, device=self.device)
        sc = torch.empty((nq, k), dtype=torch.float32, device=self.device)
        s64 = torch.empty((nq, k), dtype=torch.float64, device=self.device)
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mirx_index_search_begin(self._h, ctypes.c_void_p(q.data_ptr()), nq, int(k),
                                                         ctypes.c_void_p(ex.data_ptr()) if ex is not None else None,
                                                         ctypes.c_void_p(sc.data_ptr()), ctypes.c_void_p(s64.data_ptr()),
                                                         ctypes.c_void_p(ids.data_ptr()), _stream_ptr(self.device)),
                       "mirx_index_search_begin")
        return (q, ex, sc, s64, ids)                         # keeps the buffers alive until search_end

    def search_end(self, handle, return_f64=False):
        """Second half: wait for the first pass's counters (host wait on an event), run the rare follow-up passes.
        -> (scores, ids) like search()."""
        _lib.check(self._lib.mirx_index_search_end(self._h), "mirx_index_search_end")
        _, _, sc, s64, ids = handle
        return (s64 if return_f64 else sc), ids

    def rank_all(self, q, exclude_ids=None, with_scores=False):
        """Full ranking [nq, ntotal] (row = query, excluded id last)."""
        q, ex = self._prep_queries(q, exclude_ids)
        nq, n = q.shape[0], len(self)
        ids = torch.empty((nq, n), dtype=torch.int64, device=self.device)
        sc = torch.empty((nq, n), dtype=torch.float32, device=self.device) if with_scores else None
        with torch.cuda.device(self.device):
            rc = self._lib.mirx_index_rank_all(
                self._h, ctypes.c_void_p(q.data_ptr()), nq,
                ctypes.c_void_p(ex.data_ptr()) if ex is not None else None,
                ctypes.c_void_p(ids.data_ptr()),
                ctypes.c_void_p(sc.data_ptr()) if sc is not None else None, _stream_ptr(self.device))
        _lib.check(rc, "mirx_index_rank_all")
        return (ids, sc) if with_scores else ids

    def last_stats(self):
        s = _lib.SearchStats()
        with torch.cuda.device(self.device):
            _lib.check(self._lib.mirx_index_last_stats(self._h, _stream_ptr(self.device), ctypes.byref(s)),
                       "mirx_index_last_stats")
        return s.as_dict()

    def last_timings(self):
        """Stage -> milliseconds of the last search (needs set_option(OPT_PROFILE, 1))."""
        arr = (ctypes.c_float * len(_lib.STAGES))()
        _lib.check(self._lib.mirx_index_last_timings(self._h, arr), "mirx_index_last_timings")
        return dict(zip(_lib.STAGES, [float(v) for v in arr]))

    def rows(self, first=0, n=None):
        n = len(self) - first if n is None else n
        out = torch.empty((n, self.dim), dtype=torch.float32, device=self.device)
        ids = torch.empty((n,), dtype=torch.int64, device=self.device)
        _lib.check(self._lib.mirx_index_get_rows(self._h, first, n, ctypes.c_void_p(out.data_ptr()),
                                                 ctypes.c_void_p(ids.data_ptr())), "mirx_index_get_rows")
        return out, ids


def l2_normalize_(x):
    """In-place F.normalize(x, dim=1) on a CUDA fp32 tensor (model.py:83)."""
    if not (x.is_cuda and x.dtype == torch.float32 and x.is_contiguous() and x.dim() == 2):
        raise ValueError("l2_normalize_ expects a contiguous CUDA float32 [n, d] tensor")
    lib = _lib.load()
    with torch.cuda.device(x.device):
        _lib.check(lib.mirx_l2_normalize(ctypes.c_void_p(x.data_ptr()), x.shape[0], x.shape[1],
                                         _stream_ptr(x.device)), "mirx_l2_normalize")
    return x


def topk_merge(scores_f64, ids, metric=METRIC_IP):
    """Merge [nshard, nq, k] per-shard hits (fp64 ranking scores) -> (f64 [nq,k], f32 reported, ids)."""
    if scores_f64.dim() != 3 or scores_f64.shape != ids.shape:
        raise ValueError("expected [nshard, nq, k] scores and ids")
    ns, nq, k = scores_f64.shape
    scores_f64 = scores_f64.contiguous()
    ids = ids.contiguous()
    dev = scores_f64.device
    o64 = torch.empty((nq, k), dtype=torch.float64, device=dev)
    o32 = torch.empty((nq, k), dtype=torch.float32, device=dev)
    oid = torch.empty((nq, k), dtype=torch.int64, device=dev)
    lib = _lib.load()
    with torch.cuda.device(dev):
        _lib.check(lib.mirx_topk_merge(ctypes.c_void_p(scores_f64.data_ptr()), ctypes.c_void_p(ids.data_ptr()),
                                       ns, nq, k, metric_code(metric), ctypes.c_void_p(o64.data_ptr()),
                                       ctypes.c_void_p(o32.data_ptr()), ctypes.c_void_p(oid.data_ptr()),
                                       _stream_ptr(dev)), "mirx_topk_merge")
    return o64, o32, oid
