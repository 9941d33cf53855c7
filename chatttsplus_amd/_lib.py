"""ctypes binding of libctts_hip.so (include/ctts_hip.h).  The product path fails loudly when the
HIP library is missing or no GPU is visible -- there is no CPU fallback."""
from __future__ import annotations

import ctypes as C
import os

from . import build as _build

LIB_PATH = os.environ.get("CTTS_HIP_LIB", _build.LIB_PATH)      # developer knob: A/B builds of the same ABI

DTYPE_F32 = 0
DTYPE_F16 = 1
MAX_BATCH = 128
MAX_ADAPTERS = 8
NUM_VQ = 4


class GptCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("hidden", "inter", "heads", "layers", "vocab_code", "num_vq", "max_batch", "max_seq", "dtype")]


class SamplerCfg(C.Structure):
    _fields_ = [
        ("temperature", C.c_float * NUM_VQ),
        ("top_p_threshold", C.c_float),
        ("top_k", C.c_int32),
        ("min_tokens_to_keep", C.c_int32),
        ("use_penalty", C.c_int32),
        ("penalty_table", C.c_float * 17),
        ("past_window", C.c_int32),
        ("max_input_ids", C.c_int32),
        ("eos_token", C.c_int32),
        ("min_new_token", C.c_int32),
        ("max_new_token", C.c_int32),
        ("infer_text", C.c_int32),
    ]


class GenIO(C.Structure):
    _fields_ = [
        ("ids", C.c_void_p),
        ("hiddens", C.c_void_p),
        ("finish", C.c_void_p),
        ("end_idx", C.c_void_p),
        ("noise", C.c_void_p),
        ("n_draws", C.c_int32),
        ("seed", C.c_uint64),
        ("utt_ids", C.c_void_p),
        ("row_limits", C.c_void_p),
    ]


class VocCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("dvae_idim", "dvae_hidden", "dvae_bn", "dvae_layers", "n_mels", "vocos_dim",
                                          "vocos_inter", "vocos_layers", "n_fft", "hop", "max_frames", "max_batch",
                                          "vq_groups", "vq_residuals")] + [("vq_levels", C.c_int32 * 4)]


class EncCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_mels", "dim", "enc_hidden", "enc_bn", "enc_layers", "enc_odim", "vq_groups", "vq_residuals",
                                          "n_fft", "hop", "max_samples", "pre_bound")]


# every symbol include/ctts_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
SYMBOLS = [
    ("ctts_last_error", C.c_char_p, []),
    ("ctts_version", C.c_int, []),
    ("ctts_gpt_create", C.c_int, [C.POINTER(GptCfg), C.POINTER(_P)]),
    ("ctts_gpt_destroy", None, [_P]),
    ("ctts_gpt_set_option", C.c_int, [_P, C.c_char_p, C.c_int]),
    ("ctts_gpt_get_option", C.c_int, [_P, C.c_char_p, C.POINTER(C.c_int)]),
    ("ctts_gpt_debug_read", C.c_int, [_P, C.c_char_p, _P, C.c_size_t, C.POINTER(C.c_size_t), _P]),
    ("ctts_gpt_set_weight", C.c_int, [_P, C.c_char_p, _P, C.c_size_t]),
    ("ctts_gpt_merge_lora", C.c_int, [_P, C.c_int, C.c_char_p, _P, _P, C.c_int, C.c_float]),
    ("ctts_gpt_set_adapter", C.c_int, [_P, C.c_int, C.c_int, C.c_char_p, _P, _P, C.c_int, C.c_float]),
    ("ctts_gpt_clear_adapter", C.c_int, [_P, C.c_int]),
    ("ctts_gpt_set_row_adapters", C.c_int, [_P, _P, C.c_int]),
    ("ctts_gpt_finalize", C.c_int, [_P]),
    ("ctts_gpt_kv_bytes", C.c_size_t, [_P]),
    ("ctts_gpt_bind_kv", C.c_int, [_P, _P, C.c_size_t]),
    ("ctts_gpt_set_rope", C.c_int, [_P, _P, C.c_int]),
    ("ctts_gpt_embed", C.c_int, [_P, _P, _P, C.c_int, C.c_int, _P, C.c_int, _P, _P]),
    ("ctts_gpt_begin", C.c_int, [_P, C.c_int, C.c_int, _P, C.POINTER(SamplerCfg), C.POINTER(GenIO), _P]),
    ("ctts_gpt_prefill", C.c_int, [_P, _P, _P]),
    ("ctts_gpt_sample", C.c_int, [_P, _P]),
    ("ctts_gpt_restart", C.c_int, [_P, _P]),
    ("ctts_gpt_decode", C.c_int, [_P, C.c_int, C.c_int, _P]),
    ("ctts_gpt_progress", C.c_int, [_P, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _P]),
    ("ctts_gpt_progress_enqueue", C.c_int, [_P, _P, _P]),
    ("ctts_gpt_saturations", C.c_int, [_P, C.POINTER(C.c_int32), _P]),
    ("ctts_gpt_rows_enqueue", C.c_int, [_P, _P, _P]),
    ("ctts_gpt_compact", C.c_int, [_P, _P, C.c_int, _P]),
    ("ctts_sampler_noise", C.c_int, [C.c_uint64, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, _P, _P]),
    ("ctts_gpt_admit", C.c_int, [_P, C.c_int, _P, C.c_int, _P, _P, _P, _P, _P, _P, _P]),
    ("ctts_gpt_admit_adapters", C.c_int, [_P, C.c_int, _P, _P, _P]),
    ("ctts_gpt_logits", C.c_int, [_P, _P, _P]),
    ("ctts_gpt_force_ids", C.c_int, [_P, _P, _P]),
    ("ctts_sampler_run", C.c_int, [C.POINTER(SamplerCfg), _P, _P, C.c_int, _P, C.c_int, C.c_int, C.c_int, _P, _P]),
    ("ctts_gpt_time_decode", C.c_int, [_P, C.c_int, C.POINTER(C.c_float), _P]),
    ("ctts_gpt_step_bytes", C.c_double, [_P, C.c_int, C.c_double]),
    ("ctts_voc_create", C.c_int, [C.POINTER(VocCfg), C.POINTER(_P)]),
    ("ctts_voc_destroy", None, [_P]),
    ("ctts_voc_set_weight", C.c_int, [_P, C.c_char_p, _P, C.c_size_t]),
    ("ctts_voc_finalize", C.c_int, [_P]),
    ("ctts_dvae_decode", C.c_int, [_P, _P, C.c_int, _P, _P]),
    ("ctts_vocos_decode", C.c_int, [_P, _P, C.c_int, _P, _P]),
    ("ctts_synth_batch", C.c_int, [_P, _P, _P, C.c_int, _P, _P]),
    ("ctts_dvae_decode_codes", C.c_int, [_P, _P, C.c_int, _P, _P]),
    ("ctts_synth_batch_codes", C.c_int, [_P, _P, _P, C.c_int, _P, _P]),
    ("ctts_enc_create", C.c_int, [C.POINTER(EncCfg), C.POINTER(_P)]),
    ("ctts_enc_destroy", None, [_P]),
    ("ctts_enc_set_weight", C.c_int, [_P, C.c_char_p, _P, C.c_size_t]),
    ("ctts_enc_finalize", C.c_int, [_P]),
    ("ctts_dvae_encode", C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P]),
]

_lib = None


class HipBackendError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Loads the in-tree library; raises HipBackendError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise HipBackendError(
            f"{LIB_PATH} is missing: build it with `python -m chatttsplus_amd.build` "
            "(there is no CPU fallback for infer_type='hip')")
    lib = C.CDLL(LIB_PATH)
    for name, res, args in SYMBOLS:
        fn = getattr(lib, name)          # AttributeError if the header and the library disagree
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().ctts_last_error()
        raise HipBackendError(f"{what or 'libctts_hip'} failed: {msg.decode() if msg else 'unknown error'}")
