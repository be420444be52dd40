"""ctypes binding of libsbv2_hip.so (include/sbv2_hip.h).  Fails loudly: there is no CPU fallback."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libsbv2_hip.so")

i64p = C.POINTER(C.c_int64)
f32p = C.POINTER(C.c_float)


class Sbv2Batch(C.Structure):
    """struct sbv2_batch (include/sbv2_hip.h)."""
    _fields_ = [
        ("n", C.c_int64), ("t_lens", i64p), ("x_tst", i64p), ("tones", i64p), ("lang_ids", i64p), ("sids", i64p),
        ("style_vectors", f32p), ("bert", f32p),
        ("sdp_ratio", C.c_float), ("length_scale", C.c_float), ("noise_scale", C.c_float), ("noise_scale_w", C.c_float),
        ("noise_seed", C.c_uint64), ("forced_durations", i64p),
    ]


#: every symbol include/sbv2_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "sbv2_last_error": (C.c_char_p, []),
    "sbv2_device_count": (C.c_int, []),
    "sbv2_bert_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "sbv2_bert_destroy": (None, [C.c_void_p]),
    "sbv2_bert_hidden": (C.c_int64, [C.c_void_p]),
    "sbv2_bert_gemm_parts": (C.c_int, [C.c_void_p]),
    "sbv2_bert_predict": (C.c_int, [C.c_void_p, i64p, i64p, C.c_int64, f32p]),
    "sbv2_bert_predict_batch": (C.c_int, [C.c_void_p, C.c_int64, i64p, i64p, i64p, f32p]),
    "sbv2_vits_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p)]),
    "sbv2_vits_destroy": (None, [C.c_void_p]),
    "sbv2_vits_hop": (C.c_int64, [C.c_void_p]),
    "sbv2_vits_bert_dim": (C.c_int64, [C.c_void_p]),
    "sbv2_vits_style_dim": (C.c_int64, [C.c_void_p]),
    "sbv2_vits_decoder_mode": (C.c_int, [C.c_void_p]),
    "sbv2_vits_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "sbv2_vits_synthesize": (C.c_int, [C.c_void_p, f32p, i64p, i64p, i64p, C.c_int64, C.c_int64, f32p, C.c_float, C.c_float,
                                       C.c_float, C.c_float, C.c_uint64, C.POINTER(f32p), i64p]),
    "sbv2_pcm_free": (None, [f32p]),
    "sbv2_vits_synthesize_batch": (C.c_int, [C.c_void_p, C.POINTER(Sbv2Batch), i64p]),
    "sbv2_vits_fetch_pcm": (C.c_int, [C.c_void_p, f32p, C.c_int64]),
    "sbv2_vits_pcm_device": (C.c_void_p, [C.c_void_p, i64p]),
    "sbv2_vits_copy_pcm_device": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "sbv2_sync": (C.c_int, [C.c_void_p]),
    "sbv2_prof_begin": (C.c_int, []),
    "sbv2_prof_end": (C.c_int, [C.c_char_p, C.c_int64]),
    "sbv2_vits_fetch_durations": (C.c_int, [C.c_void_p, i64p, f32p, C.c_int64]),
    "sbv2_vits_set_trace": (C.c_int, [C.c_void_p, C.c_int]),
    "sbv2_vits_get_trace": (C.c_int, [C.c_void_p, C.c_char_p, C.c_int64, f32p, C.c_int64, i64p, i64p]),
    "sbv2_pipeline_create": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]),
    "sbv2_pipeline_destroy": (None, [C.c_void_p]),
    "sbv2_pipeline_run": (C.c_int, [C.c_void_p, C.POINTER(Sbv2Batch), i64p, i64p, i64p, i64p]),
    "sbv2_pipeline_sync": (C.c_int, [C.c_void_p]),
    "sbv2_pipeline_fetch_pcm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int]),
    "sbv2_pipeline_fetch_pcm_ticket": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int]),
    "sbv2_pipeline_last_ticket": (C.c_int64, [C.c_void_p]),
    "sbv2_pipeline_wait": (C.c_int, [C.c_void_p, C.c_int64]),
    "sbv2_host_alloc": (C.c_void_p, [C.c_size_t]),
    "sbv2_host_free": (None, [C.c_void_p]),
    "sbv2_deal": (C.c_int, [C.c_int64, i64p, C.c_int, C.POINTER(C.c_int32)]),
    "sbv2_gather_plan": (C.c_int, [C.c_int64, i64p, C.POINTER(C.c_int32), C.c_int, i64p, i64p]),
    "sbv2_comm_unique_id": (C.c_int, [C.c_char_p]),
    "sbv2_comm_create": (C.c_int, [C.c_char_p, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_void_p)]),
    "sbv2_comm_destroy": (None, [C.c_void_p]),
    "sbv2_comm_rank": (C.c_int, [C.c_void_p]),
    "sbv2_comm_world": (C.c_int, [C.c_void_p]),
    "sbv2_comm_barrier": (C.c_int, [C.c_void_p]),
    "sbv2_comm_max_f64": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "sbv2_comm_gather_pcm": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int, C.c_void_p, C.c_int64, i64p]),
    "sbv2_node_create": (C.c_int, [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.POINTER(C.c_int), C.c_int, C.POINTER(C.c_void_p)]),
    "sbv2_node_destroy": (None, [C.c_void_p]),
    "sbv2_node_devices": (C.c_int, [C.c_void_p]),
    "sbv2_node_uses_rccl": (C.c_int, [C.c_void_p]),
    "sbv2_node_synthesize": (C.c_int, [C.c_void_p, C.POINTER(Sbv2Batch), i64p, i64p, i64p, i64p, C.c_void_p, C.c_int64]),
    "sbv2_node_last_deal": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int64]),
    "sbv2_stream_begin": (C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(Sbv2Batch), i64p, i64p, i64p, C.c_int64, C.POINTER(C.c_void_p), i64p]),
    "sbv2_stream_next": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, i64p]),
    "sbv2_stream_uses_graph": (C.c_int, [C.c_void_p]),
    "sbv2_stream_workspace_bytes": (C.c_int64, [C.c_void_p]),
    "sbv2_stream_end": (None, [C.c_void_p]),
    "sbv2_parse_sbv2file": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t), C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "sbv2_bytes_free": (None, [C.c_void_p]),
    "sbv2_style_load": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), i64p, i64p]),
    "sbv2_aivmx_style_vectors": (C.c_int, [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p), i64p, i64p]),
    "sbv2_style_vector": (C.c_int, [f32p, C.c_int64, C.c_int64, C.c_int64, C.c_float, f32p]),
    "sbv2_debug_import_to_container": (C.c_int, [C.c_void_p, C.c_size_t, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]),
    "sbv2_debug_bucket_table": (C.c_int, [C.c_int64, C.c_int64, C.c_int64, C.POINTER(C.c_int32)]),
    "sbv2_debug_conv1d": (C.c_int, [C.c_int, f32p, f32p, f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, f32p]),
    "sbv2_debug_conv_transpose1d": (C.c_int, [C.c_int, f32p, f32p, f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                              C.c_int64, C.c_float, f32p]),
    "sbv2_debug_conv1d_cl": (C.c_int, [C.c_int, f32p, f32p, f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_int,
                                       C.c_int64, f32p, f32p]),
    "sbv2_debug_time_conv1d": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, f32p]),
    "sbv2_debug_set_skinny_max": (C.c_int, [C.c_int]),
    "sbv2_debug_set_clx": (C.c_int, [C.c_int]),
    "sbv2_debug_set_ksplit": (C.c_int, [C.c_int]),
    "sbv2_debug_set_flash_parts": (C.c_int, [C.c_int]),
    "sbv2_debug_conv1d_clx": (C.c_int, [C.c_int, f32p, f32p, f32p, f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_float,
                                        C.c_int64, f32p, f32p, f32p]),
    "sbv2_debug_conv_cl_clock": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_double, C.POINTER(C.c_double)]),
    "sbv2_debug_clx_timeline": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_uint64), C.c_int64,
                                          C.POINTER(C.c_int64), C.POINTER(C.c_double)]),
    "sbv2_debug_set_respair_clx": (C.c_int, [C.c_int]),
    "sbv2_debug_f16x3_saturation": (C.c_int, [C.c_int, C.c_int, C.POINTER(C.c_uint64)]),
    "sbv2_debug_respair": (C.c_int, [C.c_int, f32p, f32p, f32p, f32p, f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_float, C.c_int,
                                     C.c_int, f32p]),
    "sbv2_debug_set_resbranch": (C.c_int, [C.c_int]),
    "sbv2_debug_set_upx": (C.c_int, [C.c_int]),
    "sbv2_debug_conv_transpose1d_clx": (C.c_int, [C.c_int, f32p, f32p, f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_void_p,
                                                  C.c_int64, C.c_int64, f32p, f32p, f32p]),
    "sbv2_debug_resbranch": (C.c_int, [C.c_int, f32p, f32p, f32p, C.c_int64, C.c_int64, C.c_int64, i64p, C.c_void_p, C.c_int64, C.c_float, C.c_int, C.c_int,
                                       C.c_int64, f32p, f32p, C.c_void_p, C.c_int64]),
    "sbv2_debug_respair_clock": (C.c_int, [C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_double), C.c_int]),
    "sbv2_debug_gemm_bfs_alt": (C.c_int, [C.c_int, f32p, f32p, f32p, f32p, f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int64, f32p, f32p]),
    "sbv2_debug_gemm_bfs": (C.c_int, [C.c_int, f32p, f32p, f32p, f32p, C.c_int64, C.c_int64, C.c_int64, C.c_int, C.c_int, C.c_int, C.c_int64,
                                      f32p, f32p]),
}

_lib = None


class Sbv2Error(RuntimeError):
    """Mirror of sbv2_core::error::Error::OtherError (crates/sbv2_core/src/error.rs:29-30)."""


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise Sbv2Error(f"{LIB_PATH} is missing: build it with __graft_entry__.build() (there is no CPU fallback)")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)   # AttributeError here = the library does not export what the header declares
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc != 0:
        raise Sbv2Error(lib().sbv2_last_error().decode(errors="replace"))
