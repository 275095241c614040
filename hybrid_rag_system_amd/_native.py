"""ctypes binding of libhrs.so (the C ABI declared in include/hrs.h).

The library holds gfx950 code only.  There is no CPU fallback: if the shared object is missing, or
no MI355X is present when a context is created, the call raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HRS_LIBRARY") or os.path.join(_HERE, "libhrs.so")  # HRS_LIBRARY: another build of the same ABI (A/B timing)

SPACES = {"l2": 0, "cosine": 1, "ip": 2}
SHADOWS = {"none": 0, "bf16": 1, "i8": 2, "auto": 3}
MODE_AUTO, MODE_EXACT, MODE_MFMA = 0, 1, 2
MAX_K = 128
PAD_KEY = 0xFFFFFFFFFFFFFFFF

EXPORTS = [
    "hrs_version",
    "hrs_last_error",
    "hrs_ctx_create",
    "hrs_ctx_destroy",
    "hrs_ctx_set_peer",
    "hrs_tokenizer_create",
    "hrs_tokenizer_release",
    "hrs_tokenize",
    "hrs_corpus_register",
    "hrs_corpus_release",
    "hrs_corpus_append",
    "hrs_corpus_shadow",
    "hrs_corpus_stats",
    "hrs_score_topk",
    "hrs_score_topk_begin",
    "hrs_score_topk_finish",
    "hrs_score_topk_static",
    "hrs_score_topk_after",
    "hrs_score_stats",
    "hrs_profile_enable",
    "hrs_profile_read",
    "hrs_encoder_load",
    "hrs_encoder_release",
    "hrs_encode",
    "hrs_row_mask_update",
    "hrs_postings_register",
    "hrs_postings_release",
    "hrs_keyword_counts",
    "hrs_postings_register_inverted",
    "hrs_postings_release_inverted",
    "hrs_match_counts",
    "hrs_score_topk_levels",
    "hrs_scan_work",
    "hrs_classes_finish",
    "hrs_fuse_sort",
    "hrs_bm25_scores",
    "hrs_fuse_sort_scores",
    "hrs_merge_topk",
    "hrs_merge_topk_i64",
    "hrs_unpack_keys",
    "hrs_synth_rows",
    "hrs_synth_postings",
]


class EncoderConfigC(C.Structure):
    _fields_ = [("vocab", C.c_int), ("hidden", C.c_int), ("layers", C.c_int), ("heads", C.c_int), ("inter", C.c_int), ("max_pos", C.c_int), ("eps", C.c_float)]


class HrsError(RuntimeError):
    pass


_lib = None


def lib():
    """Load libhrs.so once; raise (never fall back) when it is not there."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise HrsError(
                f"{LIB_PATH} is missing: build it with `make -C {os.path.join(_HERE, 'csrc')}` "
                "(hipcc --offload-arch=gfx950); this package has no CPU path"
            )
        L = C.CDLL(LIB_PATH)
        vp, i32, i64, u64, dbl = C.c_void_p, C.c_int, C.c_int64, C.c_uint64, C.c_double
        sig = {
            "hrs_version": ([], i32),
            "hrs_last_error": ([C.c_char_p, C.c_size_t], i32),
            "hrs_ctx_create": ([i32, C.POINTER(vp)], i32),
            "hrs_ctx_destroy": ([vp], i32),
            "hrs_ctx_set_peer": ([vp, vp], i32),
            "hrs_tokenizer_create": ([C.c_char_p, i64, i32, i32, i32, C.POINTER(vp)], i32),
            "hrs_tokenizer_release": ([vp], i32),
            "hrs_tokenize": ([vp, C.c_char_p, vp, i32, i32, vp, vp, vp, i32], i32),
            "hrs_corpus_register": ([vp, vp, i64, i32, i64, vp, C.POINTER(vp)], i32),
            "hrs_corpus_release": ([vp, vp], i32),
            "hrs_corpus_append": ([vp, vp, vp, i64, vp], i32),
            "hrs_score_topk_after": ([vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp, vp], i32),
            "hrs_corpus_shadow": ([vp, vp, i32, vp], i32),
            "hrs_corpus_stats": ([vp, vp, vp, C.POINTER(dbl), C.POINTER(dbl)], i32),
            "hrs_score_topk": ([vp, vp, vp, i32, i32, i32, vp, i32, vp, vp, vp, vp], i32),
            "hrs_score_topk_begin": ([vp, vp, vp, i32, i32, i32, vp, i32, vp], i32),
            "hrs_score_topk_finish": ([vp, vp, vp, vp, vp], i32),
            "hrs_score_topk_static": ([vp, vp, vp, i32, i32, i32, vp, vp, vp, vp, vp], i32),
            "hrs_score_stats": ([vp, vp, C.POINTER(i64)], i32),
            "hrs_profile_enable": ([vp, i32], i32),
            "hrs_profile_read": ([vp, C.POINTER(dbl), C.POINTER(i64)], i32),
            "hrs_encoder_load": ([vp, vp, vp, vp, C.POINTER(vp)], i32),
            "hrs_encoder_release": ([vp, vp], i32),
            "hrs_encode": ([vp, vp, vp, vp, i32, i32, vp, vp], i32),
            "hrs_row_mask_update": ([vp, i64, vp, vp, i32, i32, vp, i32, vp, vp], i32),
            "hrs_postings_register": ([vp, vp, vp, i64, i64, C.POINTER(vp)], i32),
            "hrs_postings_release": ([vp, vp], i32),
            "hrs_postings_register_inverted": ([vp, vp, vp, i64, i64, C.POINTER(vp)], i32),
            "hrs_postings_release_inverted": ([vp, vp], i32),
            "hrs_match_counts": ([vp, vp, vp, vp, i32, i64, vp, vp], i32),
            "hrs_score_topk_levels": ([vp, vp, vp, i32, i32, i32, vp, i64, i32, vp, vp], i32),
            "hrs_scan_work": ([vp, vp, vp], i32),
            "hrs_classes_finish": ([vp, vp, vp, i32, i32, i32, dbl, i32, vp, vp, vp, vp, vp], i32),
            "hrs_keyword_counts": ([vp, vp, vp, vp, vp, i32, i32, vp, vp], i32),
            "hrs_fuse_sort": ([vp, vp, vp, vp, i32, i32, dbl, i32, vp, vp, vp], i32),
            "hrs_bm25_scores": ([vp, vp, vp, i64, dbl, dbl, dbl, vp, vp, vp, i32, i32, vp, vp], i32),
            "hrs_fuse_sort_scores": ([vp, vp, vp, i32, i32, dbl, i32, vp, vp, vp], i32),
            "hrs_merge_topk": ([vp, vp, vp, i32, i32, i32, vp, vp, vp], i32),
            "hrs_merge_topk_i64": ([vp, vp, vp, i32, i32, i32, vp, vp, vp], i32),
            "hrs_unpack_keys": ([vp, vp, i64, vp, vp, vp], i32),
            "hrs_synth_rows": ([vp, u64, i64, i64, i32, i32, vp, vp], i32),
            "hrs_synth_postings": ([vp, u64, i64, i64, i32, i32, vp, vp, vp], i32),
        }
        for name, (args, res) in sig.items():
            fn = getattr(L, name)
            fn.argtypes = args
            fn.restype = res
        _lib = L
    return _lib


def last_error():
    buf = C.create_string_buffer(512)
    lib().hrs_last_error(buf, 512)
    return buf.value.decode(errors="replace")


def check(rc, what):
    if rc != 0:
        raise HrsError(f"{what} failed ({rc}): {last_error()}")
