"""ctypes binding of libsuggest_hip.so (include/suggest_hip.h).

There is no CPU fallback: if the HIP library is missing or fails to load this module raises.
Import torch *before* this module when torch is used in the same process, so that both resolve
the same libamdhip64 (SONAME libamdhip64.so.7).
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("SG_LIB_NAME", "libsuggest_hip.so"))   # (SG_LIB_NAME: A/B timing against another build)

SG_COUNT_REF_PANIC = 0xFFFFFFFF
SG_COUNT_REF_DEADLOCK = 0xFFFFFFFE
SG_COUNT_TOO_LONG = 0xFFFFFFFD
SG_MAX_QUERY_TERMS = 65536
SG_MAX_TOPK = 65536

EXPORTS = [
    "sg_index_build", "sg_index_build_device", "sg_index_build_ex", "sg_index_digest", "sg_index_load_reference", "sg_index_upload", "sg_suggest_batch", "sg_suggest_batch_device", "sg_autocomplete_batch",
    "sg_autocomplete_batch_device", "sg_index_retain", "sg_index_release", "sg_last_error", "sg_index_stats",
    "sg_tokenize", "sg_term_string", "sg_index_list", "sg_index_lists", "sg_suggest_algorithmic_bytes",
    "sg_lm_load_google", "sg_lm_build_google", "sg_lm_retain", "sg_lm_release", "sg_lm_num_words", "sg_lm_word", "sg_lm_word_id", "sg_lm_score",
    "sg_lm_score_word_ids", "sg_lm_next_score", "sg_lm_tokenize", "sg_spell_index_build", "sg_spell_predict_batch", "sg_spell_predict_batch_device",
    "sg_index_replicate", "sg_index_replicas", "sg_suggest_batch_multi", "sg_autocomplete_batch_multi", "sg_suggest_one", "sg_autocomplete_one", "sg_autocomplete_one_from", "sg_autocomplete_batch_from",
    "sg_lm_load_google_ex", "sg_lm_load_binary", "sg_lm_level", "sg_lm_order", "sg_index_tune", "sg_index_forward", "sg_autocomplete_algorithmic_bytes", "sg_debug_pairsort", "sg_debug_tune_choice", "sg_debug_pipe_shape", "sg_debug_tune_index", "sg_debug_replica_devices",
    "sg_host_alloc", "sg_host_free", "sg_suggest_submit", "sg_suggest_submit_on", "sg_autocomplete_submit", "sg_ticket_wait",
    "sg_metric_tables_create", "sg_metric_tables_retain", "sg_metric_tables_release", "sg_suggest_batch_tables", "sg_suggest_batch_from", "sg_index_launch_stats", "sg_index_pipe_stats", "sg_index_pipe_volumes",
]
SG_COUNT_LM_ERROR = 0xFFFFFFFC


class SgDesc(C.Structure):
    _fields_ = [("ngram_size", C.c_uint32), ("wrap_start", C.c_char_p), ("wrap_end", C.c_char_p), ("pad", C.c_char_p),
                ("alphabet", C.POINTER(C.c_char_p)), ("n_alphabet", C.c_uint32)]


class SgStats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("n_docs", "n_segments", "n_terms", "n_lists", "n_postings", "n_postings_raw",
                                          "posting_bytes", "table_bytes", "device_bytes")]


class SuggestHipError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("libsuggest_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C suggest_amd/csrc` (hipcc --offload-arch=gfx950). There is no CPU fallback." % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32, dbl = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int, C.c_double
    if hasattr(L, "sg_index_build"): L.sg_index_build.argtypes = [vp, vp, u32, C.POINTER(SgDesc), C.POINTER(vp)]
    if hasattr(L, "sg_index_build_device"): L.sg_index_build_device.argtypes = [vp, vp, u32, C.POINTER(SgDesc), i32, C.POINTER(vp)]
    if hasattr(L, "sg_index_build_ex"): L.sg_index_build_ex.argtypes = [vp, vp, u32, C.POINTER(SgDesc), u32, i32, C.POINTER(vp)]
    if hasattr(L, "sg_index_digest"): L.sg_index_digest.argtypes = [vp, vp]
    if hasattr(L, "sg_index_load_reference"): L.sg_index_load_reference.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(SgDesc), C.POINTER(vp)]
    if hasattr(L, "sg_index_upload"): L.sg_index_upload.argtypes = [vp, i32]
    if hasattr(L, "sg_index_replicate"): L.sg_index_replicate.argtypes = [vp, vp, u32]
    if hasattr(L, "sg_index_tune"): L.sg_index_tune.argtypes = [vp, C.c_char_p, i32]
    if hasattr(L, "sg_debug_pairsort"): L.sg_debug_pairsort.argtypes = [i32, vp, u32, vp]
    if hasattr(L, "sg_debug_tune_choice"): L.sg_debug_tune_choice.argtypes = [dbl, dbl, vp]
    if hasattr(L, "sg_debug_pipe_shape"): L.sg_debug_pipe_shape.argtypes = [dbl, dbl, i32, i32, dbl, vp]
    if hasattr(L, "sg_debug_tune_index"): L.sg_debug_tune_index.argtypes = [vp, vp, vp]
    if hasattr(L, "sg_debug_replica_devices"): L.sg_debug_replica_devices.argtypes = [vp, u32, vp]
    if hasattr(L, "sg_autocomplete_algorithmic_bytes"): L.sg_autocomplete_algorithmic_bytes.argtypes = [vp, vp, vp, u32, u32, C.POINTER(u64)]
    if hasattr(L, "sg_index_forward"): L.sg_index_forward.argtypes = [vp, u32, u32, u32, vp, vp, vp]
    if hasattr(L, "sg_index_replicas"): L.sg_index_replicas.argtypes = [vp, vp, u32]
    if hasattr(L, "sg_index_replicas"): L.sg_index_replicas.restype = u32
    if hasattr(L, "sg_suggest_batch"): L.sg_suggest_batch.argtypes = [vp, vp, vp, u32, i32, dbl, u32, vp, vp, vp]
    if hasattr(L, "sg_suggest_batch_multi"): L.sg_suggest_batch_multi.argtypes = [vp, vp, vp, u32, i32, dbl, u32, vp, vp, vp]
    if hasattr(L, "sg_autocomplete_batch_multi"): L.sg_autocomplete_batch_multi.argtypes = [vp, vp, vp, u32, u32, vp, vp]
    if hasattr(L, "sg_suggest_one"): L.sg_suggest_one.argtypes = [vp, C.c_char_p, u32, i32, dbl, u32, vp, vp, vp]
    if hasattr(L, "sg_autocomplete_one"): L.sg_autocomplete_one.argtypes = [vp, C.c_char_p, u32, u32, vp, vp]
    if hasattr(L, "sg_suggest_batch_device"): L.sg_suggest_batch_device.argtypes = [vp, vp, vp, u32, i32, dbl, u32, vp, vp, vp, vp]
    if hasattr(L, "sg_autocomplete_batch"): L.sg_autocomplete_batch.argtypes = [vp, vp, vp, u32, u32, vp, vp]
    if hasattr(L, "sg_autocomplete_batch_from"): L.sg_autocomplete_batch_from.argtypes = [vp, vp, vp, u32, u32, u32, vp, vp]
    if hasattr(L, "sg_autocomplete_one_from"): L.sg_autocomplete_one_from.argtypes = [vp, C.c_char_p, u32, u32, u32, vp, vp]
    if hasattr(L, "sg_autocomplete_batch_device"): L.sg_autocomplete_batch_device.argtypes = [vp, vp, vp, u32, u32, vp, vp, vp]
    if hasattr(L, "sg_lm_load_google"): L.sg_lm_load_google.argtypes = [C.c_char_p, u32, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), u32, C.POINTER(vp)]
    if hasattr(L, "sg_lm_load_google_ex"): L.sg_lm_load_google_ex.argtypes = [C.c_char_p, u32, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), u32, i32, C.POINTER(vp)]
    if hasattr(L, "sg_lm_load_binary"): L.sg_lm_load_binary.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), u32, C.POINTER(vp)]
    if hasattr(L, "sg_lm_level"): L.sg_lm_level.argtypes = [vp, u32, vp, u32, C.POINTER(u32), vp, u32, C.POINTER(u32), C.POINTER(u32)]
    if hasattr(L, "sg_lm_order"): L.sg_lm_order.argtypes = [vp]
    if hasattr(L, "sg_lm_order"): L.sg_lm_order.restype = u32
    if hasattr(L, "sg_lm_build_google"): L.sg_lm_build_google.argtypes = [C.c_char_p, u64, u32, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), u32, C.POINTER(C.c_char_p), u32, C.c_char_p]
    if hasattr(L, "sg_lm_retain"): L.sg_lm_retain.argtypes = [vp]
    if hasattr(L, "sg_lm_retain"): L.sg_lm_retain.restype = None
    if hasattr(L, "sg_lm_release"): L.sg_lm_release.argtypes = [vp]
    if hasattr(L, "sg_lm_release"): L.sg_lm_release.restype = None
    if hasattr(L, "sg_lm_num_words"): L.sg_lm_num_words.argtypes = [vp]
    if hasattr(L, "sg_lm_num_words"): L.sg_lm_num_words.restype = u32
    if hasattr(L, "sg_lm_word"): L.sg_lm_word.argtypes = [vp, u32, C.c_char_p, u32]
    if hasattr(L, "sg_lm_word_id"): L.sg_lm_word_id.argtypes = [vp, C.c_char_p, u32]
    if hasattr(L, "sg_lm_word_id"): L.sg_lm_word_id.restype = u32
    for f in (L.sg_lm_score, L.sg_lm_score_word_ids):  # noqa
        f.argtypes = [vp, vp, u32]
        f.restype = dbl
    if hasattr(L, "sg_lm_next_score"): L.sg_lm_next_score.argtypes = [vp, vp, u32, u32, i32, C.POINTER(dbl)]
    if hasattr(L, "sg_lm_tokenize"): L.sg_lm_tokenize.argtypes = [vp, C.c_char_p, u32, C.c_char_p, u32]
    if hasattr(L, "sg_spell_index_build"): L.sg_spell_index_build.argtypes = [vp, C.POINTER(SgDesc), i32, C.POINTER(vp)]
    if hasattr(L, "sg_spell_predict_batch"): L.sg_spell_predict_batch.argtypes = [vp, vp, vp, vp, u32, u32, dbl, vp, vp]
    if hasattr(L, "sg_spell_predict_batch_device"): L.sg_spell_predict_batch_device.argtypes = [vp, vp, vp, vp, u32, C.c_uint64, u32, dbl, vp, vp, vp]
    if hasattr(L, "sg_index_retain"): L.sg_index_retain.argtypes = [vp]
    if hasattr(L, "sg_index_retain"): L.sg_index_retain.restype = None
    if hasattr(L, "sg_index_release"): L.sg_index_release.argtypes = [vp]
    if hasattr(L, "sg_index_release"): L.sg_index_release.restype = None
    if hasattr(L, "sg_last_error"): L.sg_last_error.restype = C.c_char_p
    if hasattr(L, "sg_index_stats"): L.sg_index_stats.argtypes = [vp, C.POINTER(SgStats)]
    if hasattr(L, "sg_tokenize"): L.sg_tokenize.argtypes = [vp, C.c_char_p, u32, i32, vp, u32]
    if hasattr(L, "sg_term_string"): L.sg_term_string.argtypes = [vp, u64, C.c_char_p, u32]
    if hasattr(L, "sg_index_list"): L.sg_index_list.argtypes = [vp, u32, u64, vp, u64, C.POINTER(u64)]
    if hasattr(L, "sg_index_list"): L.sg_index_list.restype = C.c_int64
    if hasattr(L, "sg_index_lists"): L.sg_index_lists.argtypes = [vp, vp, vp, u64]
    if hasattr(L, "sg_index_lists"): L.sg_index_lists.restype = u64
    if hasattr(L, "sg_suggest_algorithmic_bytes"): L.sg_suggest_algorithmic_bytes.argtypes = [vp, vp, vp, u32, i32, dbl, u32, C.POINTER(u64)]
    if hasattr(L, "sg_host_alloc"): L.sg_host_alloc.argtypes = [u64, C.POINTER(vp)]
    if hasattr(L, "sg_host_free"): L.sg_host_free.argtypes = [vp]
    if hasattr(L, "sg_host_free"): L.sg_host_free.restype = None
    if hasattr(L, "sg_suggest_submit"): L.sg_suggest_submit.argtypes = [vp, vp, vp, u32, i32, dbl, u32, vp, vp, vp, C.POINTER(vp)]
    if hasattr(L, "sg_suggest_submit_on"): L.sg_suggest_submit_on.argtypes = [vp, u32, vp, vp, u32, i32, dbl, u32, vp, vp, vp, C.POINTER(vp)]
    if hasattr(L, "sg_autocomplete_submit"): L.sg_autocomplete_submit.argtypes = [vp, vp, vp, u32, u32, u32, vp, vp, C.POINTER(vp)]
    if hasattr(L, "sg_ticket_wait"): L.sg_ticket_wait.argtypes = [vp]
    if hasattr(L, "sg_index_launch_stats"): L.sg_index_launch_stats.argtypes = [vp, vp]
    if hasattr(L, "sg_index_pipe_stats"): L.sg_index_pipe_stats.argtypes = [vp, vp]
    if hasattr(L, "sg_index_pipe_volumes"): L.sg_index_pipe_volumes.argtypes = [vp, vp]
    if hasattr(L, "sg_metric_tables_create"): L.sg_metric_tables_create.argtypes = [vp, u32, vp, vp, vp, vp, C.POINTER(vp)]
    if hasattr(L, "sg_metric_tables_retain"): L.sg_metric_tables_retain.argtypes = [vp]
    if hasattr(L, "sg_metric_tables_retain"): L.sg_metric_tables_retain.restype = None
    if hasattr(L, "sg_metric_tables_release"): L.sg_metric_tables_release.argtypes = [vp]
    if hasattr(L, "sg_metric_tables_release"): L.sg_metric_tables_release.restype = None
    if hasattr(L, "sg_suggest_batch_tables"): L.sg_suggest_batch_tables.argtypes = [vp, vp, vp, u32, vp, u32, vp, vp, vp]
    if hasattr(L, "sg_suggest_batch_from"): L.sg_suggest_batch_from.argtypes = [vp, vp, vp, u32, i32, dbl, vp, u32, u32, vp, vp, vp, vp]
    _lib = L
    return L


def check(rc):
    if rc < 0:
        raise SuggestHipError(rc, lib().sg_last_error().decode("utf-8", "replace"))
    return rc
