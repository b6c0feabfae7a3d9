"""ctypes binding of ``libt3hip.so`` (declarations: include/t3_hip.h).

The library is the product: there is no Python or CPU fallback.  ``load()`` raises if the
shared object is missing, and every call raises ``T3Error`` with the library's message
when it reports a non-zero status.
"""
from __future__ import annotations

import ctypes as C
import os
from functools import lru_cache

LIB_NAME = "libt3hip.so"
ABI_VERSION = 4


class T3Error(RuntimeError):
    pass


class T3Config(C.Structure):
    _fields_ = [
        ("n_layers", C.c_int32), ("d_model", C.c_int32), ("n_heads", C.c_int32), ("head_dim", C.c_int32),
        ("d_ff", C.c_int32), ("text_vocab", C.c_int32), ("speech_vocab", C.c_int32),
        ("text_pos_rows", C.c_int32), ("speech_pos_rows", C.c_int32), ("max_model_len", C.c_int32),
        ("max_num_seqs", C.c_int32), ("cfg_paired", C.c_int32), ("max_prefill_rows", C.c_int32),
        ("device", C.c_int32), ("use_graphs", C.c_int32), ("weight_int8", C.c_int32),
        ("cfg_scale", C.c_float), ("rms_eps", C.c_float), ("kv_pages", C.c_int64),
        ("prefix_sharing", C.c_int32), ("max_voices", C.c_int32),
    ]


class T3Sampling(C.Structure):
    _fields_ = [
        ("temperature", C.c_float), ("top_p", C.c_float), ("min_p", C.c_float), ("repetition_penalty", C.c_float),
        ("max_tokens", C.c_int32), ("stop_token", C.c_int32), ("has_seed", C.c_int32), ("n_extra_stop", C.c_int32),
        ("seed", C.c_uint64), ("extra_stop_tokens", C.c_int32 * 4),
    ]


class T3StepStats(C.Structure):
    _fields_ = [
        ("prefill_requests", C.c_int32), ("prefill_rows", C.c_int32), ("decode_requests", C.c_int32),
        ("decode_rows", C.c_int32), ("decode_ctx_tokens", C.c_int64), ("free_pages", C.c_int32),
        ("waiting", C.c_int32), ("running", C.c_int32), ("prefill_outstanding", C.c_int32),
        ("decode_gpu_ms", C.c_float), ("prefill_gpu_ms", C.c_float),
    ]


_P = C.c_void_p
_I32P = C.POINTER(C.c_int32)


class T3GemmDesc(C.Structure):
    _fields_ = [
        ("a_dev", C.c_void_p), ("w_dev", C.c_void_p), ("w_scale_dev", C.c_void_p), ("out_dev", C.c_void_p),
        ("slab_dev", C.c_void_p),
        ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("ldc", C.c_int32),
        ("epilogue", C.c_int32), ("prompt_form", C.c_int32),
        ("H", C.c_int32), ("hd", C.c_int32), ("layer", C.c_int32), ("max_pages", C.c_int32),
        ("rope_dev", C.c_void_p), ("pos_dev", C.c_void_p), ("stream_dev", C.c_void_p), ("kv_pool_dev", C.c_void_p),
        ("layer_stride_elems", C.c_int64), ("block_table_dev", C.c_void_p),
        ("a_tiled", C.c_int32), ("out_tiled", C.c_int32),
        ("fn_slab_dev", C.c_void_p), ("fn_res_in_dev", C.c_void_p), ("fn_res_out_dev", C.c_void_p),
        ("fn_norm_w_dev", C.c_void_p), ("fn_nsplit", C.c_int32), ("fn_eps", C.c_float),
        ("fn_pair_scale_dev", C.c_void_p),
    ]


# name -> (argtypes, restype); every symbol include/t3_hip.h declares
SIGNATURES = {
    "t3_last_error": ([], C.c_char_p),
    "t3_abi_version": ([], C.c_int),
    "t3_engine_create": ([C.POINTER(T3Config), C.POINTER(_P)], C.c_int),
    "t3_engine_create_on_stream": ([C.POINTER(T3Config), _P, C.POINTER(_P)], C.c_int),
    "t3_engine_load_tensor": ([_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32, C.c_int32], C.c_int),
    "t3_engine_load_tensor_dev": ([_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32, C.c_int32], C.c_int),
    "t3_engine_finalize": ([_P], C.c_int),
    "t3_engine_destroy": ([_P], C.c_int),
    "t3_engine_add_request": ([_P, C.c_int64, _I32P, C.c_int32, C.POINTER(C.c_float), C.POINTER(T3Sampling)], C.c_int),
    "t3_engine_step": ([_P, C.POINTER(C.c_int64), _I32P, _I32P, C.c_int32, _I32P], C.c_int),
    "t3_engine_set_schedule_policy": ([_P, C.c_int32], C.c_int),
    "t3_engine_set_prefill_chunk": ([_P, C.c_int32], C.c_int),
    "t3_engine_abort": ([_P, C.c_int64], C.c_int),
    "t3_engine_has_unfinished": ([_P], C.c_int),
    "t3_engine_get_stats": ([_P, C.POINTER(T3StepStats)], C.c_int),
    "t3_engine_set_timing": ([_P, C.c_int32], C.c_int),
    "t3_engine_set_kernel_timing": ([_P, C.c_int32], C.c_int),
    "t3_kernel_class_count": ([], C.c_int),
    "t3_kernel_class_name": ([C.c_int32], C.c_char_p),
    "t3_engine_get_kernel_times": ([_P, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                    C.c_int32], C.c_int),
    "t3_condenc_create": ([C.c_int32, C.POINTER(_P)], C.c_int),
    "t3_condenc_load_tensor": ([_P, C.c_char_p, C.POINTER(C.c_float), C.POINTER(C.c_int64), C.c_int32], C.c_int),
    "t3_condenc_encode": ([_P, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int32, C.c_float, C.POINTER(C.c_float)], C.c_int),
    "t3_condenc_emotion_row": ([_P, C.c_float, C.POINTER(C.c_float)], C.c_int),
    "t3_condenc_destroy": ([_P], C.c_int),
    "t3_engine_set_keep_logits": ([_P, C.c_int32], C.c_int),
    "t3_engine_last_logits": ([_P, C.c_int64, C.POINTER(C.c_float)], C.c_int),
    "t3_op_repack_weight": ([_P, _P, C.c_int32, C.c_int32, _P], C.c_int),
    "t3_op_add_rmsnorm": ([_P, _P, _P, _P, C.c_int32, C.c_int32, C.c_float, _P], C.c_int),
    "t3_op_gemm": ([_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P], C.c_int),
    "t3_op_gemm_slab": ([_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P], C.c_int),
    "t3_op_slab_split": ([], C.c_int),
    "t3_op_set_prompt_gemm_form": ([C.c_int32], C.c_int),
    "t3_op_slab_add_rmsnorm": ([_P, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, C.c_float, _P], C.c_int),
    "t3_op_slab_add_rmsnorm_tiled": ([_P, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, C.c_float, _P], C.c_int),
    "t3_op_gemm_silu": ([_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P], C.c_int),
    "t3_op_gemm_qkv_rope": ([_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int64,
                             C.c_int32, _P, C.c_int32, _P], C.c_int),
    "t3_op_gemm_prompt_slab": ([_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P], C.c_int),
    "t3_op_gemm_prompt_silu": ([_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P], C.c_int),
    "t3_op_gemm_prompt_qkv_rope": ([_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, _P, C.c_int64,
                                    C.c_int32, _P, C.c_int32, _P], C.c_int),
    "t3_op_quantize_weight": ([_P, _P, _P, C.c_int32, C.c_int32, _P], C.c_int),
    "t3_op_gemm_ex": ([C.POINTER(T3GemmDesc), _P], C.c_int),
    "t3_op_attn_decode": ([_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, C.c_int32, _P,
                           C.c_int32, _P], C.c_int),
    "t3_op_attn_decode_oproj": ([_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, C.c_int32, _P,
                                 C.c_int32, _P, C.c_int32, C.c_int32, _P, C.c_int32, _P], C.c_int),
    "t3_op_attn_states_floats": ([C.c_int32], C.c_int),
    "t3_op_attn_decode_split_oproj": ([_P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, C.c_int32, _P,
                                       C.c_int32, _P, _P, _P, C.c_int32, _P], C.c_int),
    "t3_op_oproj_split": ([], C.c_int),
    "t3_op_head_slab_add_rmsnorm": ([_P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_float, _P], C.c_int),
    "t3_op_attn_prefill": ([_P, _P, C.c_int32, _P, _I32P, C.c_int32, C.c_int32, _P], C.c_int),
    "t3_op_attn_prefill_static": ([_P, _P, C.c_int32, _P, _I32P, C.c_int32, C.c_int32, _P], C.c_int),
    "t3_op_attn_decode_prefix": ([_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int64, C.c_int32, _P,
                                  C.c_int32, _P, C.c_int32, _P], C.c_int),
    "t3_op_attn_prefill_prefix": ([_P, _P, C.c_int32, _P, _I32P, C.c_int32, C.c_int32, _P, C.c_int32, _P, C.c_int64,
                                   C.c_int32, _P], C.c_int),
    "t3_op_sample_greedy": ([_P, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_float, _P, _P, C.c_int32, _P, _P, _P],
                            C.c_int),
}


def lib_path() -> str:
    return os.environ.get("T3_HIP_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib", LIB_NAME)


@lru_cache(maxsize=1)
def load() -> C.CDLL:
    path = lib_path()
    if not os.path.isfile(path):
        raise T3Error(f"{path} not found: build it with `make -C chatterbox_vllm_amd/csrc` "
                      "(or __graft_entry__.build()); there is no fallback path")
    # PyTorch-ROCm ships its own HIP / HSA runtime.  Loading it first makes this library bind to that same copy;
    # the other order puts two runtimes into the process and the second one finds no device.
    import torch  # noqa: F401
    lib = C.CDLL(path)
    for name, (argtypes, restype) in SIGNATURES.items():
        fn = getattr(lib, name)            # AttributeError here = library does not match the header
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.t3_abi_version() != ABI_VERSION:
        raise T3Error(f"{path}: ABI version {lib.t3_abi_version()} != expected {ABI_VERSION}")
    return lib


def check(lib: C.CDLL, status: int) -> None:
    if status != 0:
        msg = lib.t3_last_error().decode("utf-8", "replace")
        if status == 2:
            raise ValueError(msg)        # request rejected (e.g. prompt longer than max_model_len)
        raise T3Error(msg)


def ptr(t) -> C.c_void_p:
    """Device/host address of a torch tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())
