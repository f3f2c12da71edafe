"""ctypes binding of libmirx.so -- the only way the package reaches the HIP kernels.

There is no CPU fallback: if the library is missing or a symbol of include/mirx.h is absent
the import raises, and every wrapper turns a negative return code into MirxError.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MIRX_LIB_PATH", os.path.join(_HERE, "libmirx.so"))   # override: kernel experiments only

METRIC_IP = 0
METRIC_NEG_L2 = 1
TIER_AUTO = 0
TIER_EXACT_ONLY = 3
OPT_TIERS = 1
OPT_SAMPLE_RANK = 2
OPT_FORCE_TAU = 3
OPT_PROFILE = 4
TUNE_CONV1X1_SMALL_MAX_WG = 1      # mirx_set_tuning keys
TUNE_CONV3X3_SMALL_MAX_WG = 2
STAGES = ("prep", "sample", "gemm", "finalize", "exact")
FORCE_TAU_OFF = 0x7FC00000


class MirxError(RuntimeError):
    """A libmirx call failed (message from mirx_last_error())."""


class SearchStats(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int64) for n in
                ("nq", "tier1_answered", "exact_answered", "candidates", "reranked",
                 "overflowed", "incomplete")]

    def as_dict(self):
        return {n: int(getattr(self, n)) for n, _ in self._fields_}


# name -> (restype, argtypes): every symbol include/mirx.h declares
_vp, _i64, _int = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
SYMBOLS = {
    "mirx_last_error": (ctypes.c_char_p, []),
    "mirx_version": (_int, []),
    "mirx_set_tuning": (_int, [_int, _i64]),
    "mirx_index_create": (_int, [_int, _int, _int, ctypes.POINTER(_vp)]),
    "mirx_index_destroy": (None, [_vp]),
    "mirx_index_add": (_int, [_vp, _vp, _i64, _vp]),
    "mirx_index_reserve": (_int, [_vp, _i64]),
    "mirx_index_size": (_i64, [_vp]),
    "mirx_index_dim": (_int, [_vp]),
    "mirx_index_set_option": (_int, [_vp, _int, _i64]),
    "mirx_index_get_rows": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "mirx_index_search": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _vp]),
    "mirx_index_search_f64": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _vp]),
    "mirx_index_search_begin": (_int, [_vp, _vp, _i64, _int, _vp, _vp, _vp, _vp, _vp]),
    "mirx_index_search_end": (_int, [_vp]),
    "mirx_index_last_stats": (_int, [_vp, _vp, ctypes.POINTER(SearchStats)]),
    "mirx_index_last_timings": (_int, [_vp, ctypes.POINTER(ctypes.c_float)]),
    "mirx_index_rank_all": (_int, [_vp, _vp, _i64, _vp, _vp, _vp, _vp]),
    "mirx_topk_merge": (_int, [_vp, _vp, _int, _i64, _int, _int, _vp, _vp, _vp, _vp]),
    "mirx_rank_metrics": (_int, [_vp, _i64, _i64, _i64, _vp, _i64, _vp, _vp, _int, _int, ctypes.c_double, _int,
                                 ctypes.POINTER(ctypes.c_int32), _int, _vp, _vp, _vp, _vp, _vp]),
    "mirx_linear_split2h": (_int, [_vp, _i64, _int, _vp, _vp, _int, _int, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp]),
    "mirx_rows_to_terms": (_int, [_vp, _i64, _int, _i64, ctypes.c_float, _vp, _vp]),
    "mirx_layernorm_patch2_nhwc": (_int, [_vp, _i64, _int, _int, _int, _vp, _vp, ctypes.c_float, _vp, _vp]),
    "mirx_layernorm_terms": (_int, [_vp, _i64, _int, _vp, _vp, ctypes.c_float, ctypes.c_float, _vp, _vp]),
    "mirx_linear_terms": (_int, [_vp, _i64, _int, _vp, _vp, _int, _int, _vp, _vp, ctypes.c_float, _vp, _vp, ctypes.c_float, _vp, _i64,
                                 _vp]),
    "mirx_linear_terms_workspace_bytes": (_i64, [_i64, _int, _int]),
    "mirx_linear_split3": (_int, [_vp, _i64, _int, _vp, _vp, _int, _int, _vp, _vp, _vp, _vp]),
    "mirx_linear_split3_nchw": (_int, [_vp, _i64, _int, _int, _vp, _vp, _int, _vp, _vp, _vp, _vp]),
    "mirx_linear_split2h_nchw": (_int, [_vp, _i64, _int, _int, _vp, _vp, _int, _vp, _vp, ctypes.c_float, _vp, ctypes.c_float, _vp,
                                        _vp]),
    "mirx_linear_split2h_gelu_grn": (_int, [_vp, _i64, _int, _int, _vp, _vp, _int, ctypes.c_float, ctypes.c_float, _vp, _vp, _vp, _vp]),
    "mirx_linear_split2h_grn_rows": (_int, [_vp, _i64, _int, _int, _vp, _vp, _int, _vp, _vp, ctypes.c_float, _vp, ctypes.c_float, _vp,
                                            _vp]),
    "mirx_grn_norm_nhwc": (_int, [_vp, _i64, _int, _int, _vp, _vp]),
    "mirx_grn_scale": (_int, [_vp, _vp, _i64, _int, ctypes.c_float, _vp, _vp, _vp]),
    "mirx_conv1x1_bn_relu_split3": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _vp, _i64, _vp]),
    "mirx_conv1x1_bn_relu_split2h": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _vp, _i64,
                                            _vp, ctypes.c_float, ctypes.c_float, _vp, _i64, _i64, _vp]),
    "mirx_layernorm": (_int, [_vp, _i64, _int, _vp, _vp, ctypes.c_float, _vp, _int, _vp]),
    "mirx_patchify_nchw": (_int, [_vp, _i64, _int, _int, _int, _int, _vp, _vp, ctypes.c_float, _vp, _int, _vp]),
    "mirx_attention_small": (_int, [_vp, _i64, _vp, _vp, _i64, _vp, _i64, _int, _int, _int, _int, ctypes.c_float, _vp, _vp]),
    "mirx_range_absmax": (_int, [_vp, _i64, _i64, _vp, _vp]),
    "mirx_range_absmax_u8": (_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp]),
    "mirx_stem_conv7_bn_relu_pool_split2h_u8_into": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _vp, _i64, _vp, _vp,
                                                            _vp]),
    "mirx_stem_conv7_bn_relu_pool_split2h_into": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _int, _vp, _i64, _vp, _vp, _vp]),
    "mirx_conv1x1_bn_relu_split2h_terms": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp, _vp, ctypes.c_float,
                                                  ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _i64, _vp]),
    "mirx_dense_layer_fused": (_int, [_vp, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _vp, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp]),
    "mirx_conv3x3_direct_terms_nchw": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _i64, _vp, _vp, _i64, _vp]),
    "mirx_conv3x3_direct_terms_nchw_pool": (_int, [_vp, _vp, _vp, _i64, _int, _vp, _i64, _vp, _vp, _i64, _vp, _vp, _vp, _i64, _vp]),
    "mirx_conv3x3_small_launch": (_int, [_i64, _int]),
    "mirx_conv3x3_winograd_split3_nchw": (_int, [_vp, _vp, _i64, _int, _vp, _i64, _vp]),
    "mirx_conv3x3_direct_split3_nchw": (_int, [_vp, _vp, _i64, _int, _vp, _i64, _vp]),
    "mirx_conv3x3_winograd_nchw": (_int, [_vp, _vp, _i64, _int, _vp, _i64, _vp]),
    "mirx_attention_qkv_f32": (_int, [_vp, _i64, _int, _int, _int, ctypes.c_float, _vp, _vp]),
    "mirx_attention_qkv_f32_split2h": (_int, [_vp, _i64, _int, _int, _int, ctypes.c_float, ctypes.c_float, ctypes.c_float, _vp, _vp]),
    "mirx_attention_qkv_f32_split2h_terms": (_int, [_vp, _i64, _int, _int, _int, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                                    ctypes.c_float, _vp, _vp]),
    "mirx_attention_qkv_f32_split3": (_int, [_vp, _i64, _int, _int, _int, ctypes.c_float, _vp, _vp]),
    "mirx_l2_normalize": (_int, [_vp, _i64, _int, _vp]),
    "mirx_bn_relu_gap_l2norm": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp]),
    "mirx_bn_relu_nchw": (_int, [_vp, _i64, _vp, _vp, _i64, _int, _int, _vp, _vp]),
    "mirx_bn_relu_avgpool2": (_int, [_vp, _i64, _vp, _vp, _i64, _int, _int, _int, _vp, _i64, _vp]),
    "mirx_bn_relu_avgpool2_into": (_int, [_vp, _i64, _vp, _vp, _i64, _int, _int, _int, _vp, _i64, _i64, _vp]),
    "mirx_conv1x1_bn_relu": (_int, [_vp, _i64, _int, _vp, _vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp]),
    "mirx_dwconv7x7_nchw_to_nhwc": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp]),
    "mirx_dwconv7x7_nhwc": (_int, [_vp, _vp, _vp, _i64, _int, _int, _int, _vp, _vp]),
    "mirx_stem_conv7_bn_relu_pool_split3": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _vp, _vp]),
    "mirx_stem_conv7_bn_relu_pool": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _int, _vp, _vp]),
}

_lib = None
ABI_VERSION = 305          # include/mirx.h MIRX_VERSION this binding was written against


def load():
    """Load libmirx.so once; raise MirxError with build instructions when it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MirxError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            f"or `make -C {os.path.join(_HERE, 'csrc')}` (needs hipcc; there is no CPU fallback)")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SYMBOLS.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:  # pragma: no cover
            raise MirxError(f"libmirx.so does not export {name}") from e
        fn.restype = res
        fn.argtypes = args
    if lib.mirx_version() != ABI_VERSION:
        raise MirxError(f"{LIB_PATH} is ABI version {lib.mirx_version()}, this package binds version {ABI_VERSION}: "
                        f"rebuild with `make -C {os.path.join(_HERE, 'csrc')}`")
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != 0:
        msg = load().mirx_last_error()
        raise MirxError(f"{what} failed ({rc}): {msg.decode() if msg else ''}")
