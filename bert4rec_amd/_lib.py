"""ctypes binding of libb4r_hip.so (include/b4r.h).  There is NO CPU fallback: if the HIP library is missing or a
call fails, an exception is raised.  This is the stub a maintainer of the reference would add to bind the library
(INTEGRATION.md)."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
# B4R_LIB_PATH: another build of the same HIP library (kernel experiments: A/B of two builds inside one GPU session)
LIB_PATH = os.environ.get("B4R_LIB_PATH") or os.path.join(_HERE, "libb4r_hip.so")


class B4RError(RuntimeError):
    pass


class ModelConfig(C.Structure):
    _fields_ = [("vocab_size", C.c_int32), ("hidden_size", C.c_int32), ("num_layers", C.c_int32),
                ("num_heads", C.c_int32), ("inner_dim", C.c_int32), ("max_seq_len", C.c_int32),
                ("output_dropout", C.c_float), ("attention_dropout", C.c_float), ("ln_eps", C.c_float)]


class Batch(C.Structure):
    _fields_ = [("input_word_ids", C.c_void_p), ("input_mask", C.c_void_p), ("masked_lm_positions", C.c_void_p),
                ("masked_lm_ids", C.c_void_p), ("B", C.c_int32), ("L", C.c_int32), ("P", C.c_int32)]


class AdamWConfig(C.Structure):
    _fields_ = [("init_lr", C.c_float), ("end_lr", C.c_float), ("num_train_steps", C.c_int32),
                ("num_warmup_steps", C.c_int32), ("weight_decay_rate", C.c_float), ("beta_1", C.c_float),
                ("beta_2", C.c_float), ("epsilon", C.c_float), ("clip_norm", C.c_float), ("decay_mask", C.c_void_p)]


class GemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int32), ("B", C.c_void_p), ("ldb", C.c_int32), ("C", C.c_void_p),
                ("ldc", C.c_int32), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("b_is_nk", C.c_int32),
                ("epilogue", C.c_int32), ("bias", C.c_void_p), ("C2", C.c_void_p), ("ldc2", C.c_int32),
                ("R", C.c_void_p), ("ldr", C.c_int32), ("qscale", C.c_float), ("qcols", C.c_int32),
                ("rng", C.c_void_p), ("drop_stream", C.c_uint32), ("drop_rate", C.c_float), ("a_dropout", C.c_int32),
                ("c_pad_scratch", C.c_int32), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_mean", C.c_void_p),
                ("ln_rstd", C.c_void_p), ("ln_eps", C.c_float), ("ln_z", C.c_void_p), ("ln_ldz", C.c_int32),
                ("ln_dgamma", C.c_void_p), ("ln_dbeta", C.c_void_p), ("ln_ids", C.c_void_p), ("ln_table", C.c_void_p),
                ("ln_pos", C.c_void_p), ("ln_L", C.c_int32), ("ln_V", C.c_int32), ("C3", C.c_void_p), ("ldc3", C.c_int32),
                ("a_gather_idx", C.c_void_p), ("a_gather_add_per", C.c_int64), ("a_gather_per", C.c_int32), ("a_copy", C.c_void_p),
                ("a_copy_ld", C.c_int32)]


class GemmTnDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("lda", C.c_int32), ("B", C.c_void_p), ("ldb", C.c_int32), ("out", C.c_void_p),
                ("ldo", C.c_int32), ("R", C.c_int32), ("Mo", C.c_int32), ("No", C.c_int32), ("colsum", C.c_void_p),
                ("colsum_a", C.c_void_p), ("rng", C.c_void_p), ("drop_stream", C.c_uint32), ("drop_rate", C.c_float),
                ("b_dropout", C.c_int32), ("accumulate", C.c_int32), ("dgrad_w", C.c_void_p), ("dgrad_ldw", C.c_int32),
                ("dgrad_out", C.c_void_p), ("dgrad_ldo", C.c_int32), ("dgrad_gelu_pre", C.c_void_p), ("dgrad_ldg", C.c_int32)]


class AttnBlockDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("heads", C.c_int32), ("x", C.c_void_p),
                ("input_mask", C.c_void_p), ("Wqkv", C.c_void_p), ("bqkv", C.c_void_p), ("Wo", C.c_void_p), ("bo", C.c_void_p),
                ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float), ("rng", C.c_void_p),
                ("probs_stream", C.c_uint32), ("probs_rate", C.c_float), ("out_stream", C.c_uint32), ("out_rate", C.c_float),
                ("qkv", C.c_void_p), ("ctx", C.c_void_p), ("lse", C.c_void_p), ("keep_bits", C.c_void_p), ("z1", C.c_void_p),
                ("x1", C.c_void_p), ("mean1", C.c_void_p), ("rstd1", C.c_void_p),
                ("emb_ids", C.c_void_p), ("emb_table", C.c_void_p), ("emb_pos", C.c_void_p), ("emb_gamma", C.c_void_p),
                ("emb_beta", C.c_void_p), ("emb_vocab", C.c_int32), ("emb_eps", C.c_float), ("emb_stream", C.c_uint32),
                ("emb_rate", C.c_float), ("emb_x", C.c_void_p), ("emb_mean", C.c_void_p), ("emb_rstd", C.c_void_p),
                ("out_slot_positions", C.c_void_p), ("out_slots", C.c_int32)]


class AttnBlockBwdDesc(C.Structure):
    _fields_ = [("B", C.c_int32), ("L", C.c_int32), ("H", C.c_int32), ("heads", C.c_int32), ("x", C.c_void_p), ("dz1", C.c_void_p),
                ("ctx", C.c_void_p), ("lse", C.c_void_p), ("keep_bits", C.c_void_p), ("input_mask", C.c_void_p),
                ("Wqkv", C.c_void_p), ("bqkv", C.c_void_p), ("Wo", C.c_void_p), ("rng", C.c_void_p), ("probs_stream", C.c_uint32),
                ("probs_rate", C.c_float), ("out_stream", C.c_uint32), ("out_rate", C.c_float), ("prev_z", C.c_void_p),
                ("prev_mean", C.c_void_p), ("prev_rstd", C.c_void_p), ("prev_gamma", C.c_void_p), ("emb_ids", C.c_void_p),
                ("emb_table", C.c_void_p), ("emb_pos", C.c_void_p), ("emb_vocab", C.c_int32), ("emb_stream", C.c_uint32),
                ("emb_rate", C.c_float), ("dqkv", C.c_void_p), ("dx_prev", C.c_void_p), ("dprev_gamma", C.c_void_p),
                ("scratch", C.c_void_p), ("dWqkv", C.c_void_p), ("dbqkv", C.c_void_p), ("dw_scratch", C.c_void_p),
                ("dWo", C.c_void_p), ("dbo", C.c_void_p), ("dz1_slot_positions", C.c_void_p), ("dz1_slot_ids", C.c_void_p),
                ("dz1_slots", C.c_int32)]


class FfnDesc(C.Structure):
    _fields_ = [("N", C.c_int32), ("H", C.c_int32), ("I", C.c_int32), ("x1", C.c_void_p), ("W1", C.c_void_p), ("b1", C.c_void_p),
                ("W2", C.c_void_p), ("b2", C.c_void_p), ("ln_gamma", C.c_void_p), ("ln_beta", C.c_void_p), ("ln_eps", C.c_float),
                ("rng", C.c_void_p), ("drop_stream", C.c_uint32), ("drop_rate", C.c_float), ("z2", C.c_void_p), ("x2", C.c_void_p),
                ("mean2", C.c_void_p), ("rstd2", C.c_void_p), ("dz2", C.c_void_p), ("z1", C.c_void_p), ("mean1", C.c_void_p),
                ("rstd1", C.c_void_p), ("ln1_gamma", C.c_void_p), ("dz1", C.c_void_p), ("dW1", C.c_void_p), ("db1", C.c_void_p),
                ("dW2", C.c_void_p), ("db2", C.c_void_p), ("dln1_gamma", C.c_void_p), ("scratch", C.c_void_p),
                ("rows", C.c_void_p), ("n_rows", C.c_void_p), ("max_rows", C.c_int32), ("row_slot", C.c_void_p),
                ("slot_grad", C.c_void_p), ("dln_gamma", C.c_void_p), ("dz2_rows", C.c_void_p),
                ("slot_positions", C.c_void_p), ("slot_ids", C.c_void_p), ("slots_per_seq", C.c_int32), ("seq_len", C.c_int32),
                ("ln1_beta", C.c_void_p)]


# b4r_train_state: 16 x 32-bit words; word indices of the float fields
STATE_WORDS = 16
ST_SEED, ST_STEP_LO, ST_STEP = 0, 1, 2  # step is int64 at words 2..3
ST_LOSS_SUM, ST_VALID, ST_CORRECT_MASKED, ST_CORRECT_ALL, ST_SLOTS_ALL, ST_SQNORM, ST_GRAD_NORM, ST_LR = range(4, 12)

EPI_NONE, EPI_BIAS, EPI_BIAS_QSCALE, EPI_BIAS_GELU, EPI_BIAS_DROP_RES, EPI_GELU_BWD, EPI_ADD_RES, EPI_BIAS_TANH = range(8)
EPI_BIAS_DROP_RES_LN, EPI_ADD_RES_LN_BWD, EPI_BIAS_GELU_LN = 8, 9, 10
FLAG_TRAINING, FLAG_POOLER, FLAG_FUSED_HEAD, FLAG_GRAD_TAIL, FLAG_HEAD_ROWS_ONLY, FLAG_LOSS_SUMS = 1, 2, 4, 8, 16, 32
FLAG_ENCODER_ONLY = 64
LOSS_FUSED_HEAD = 2
GEMM_F32, GEMM_BF16X3 = 0, 1

_P, _I32, _I64, _F, _U32 = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_uint32

# name -> (restype, argtypes).  Every symbol declared in include/b4r.h appears here (tests check the two lists agree).
PROTOTYPES = {
    "b4r_version": (C.c_int, []),
    "b4r_last_error": (C.c_size_t, [C.c_char_p, C.c_size_t]),
    "b4r_param_total_floats": (_I64, [C.POINTER(ModelConfig)]),
    "b4r_param_decay_floats": (_I64, [C.POINTER(ModelConfig)]),
    "b4r_param_count": (_I32, [C.POINTER(ModelConfig)]),
    "b4r_param_info": (C.c_int, [C.POINTER(ModelConfig), _I32, C.c_char_p, C.c_size_t, C.POINTER(_I64),
                                 C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "b4r_pooler_floats": (_I64, [C.POINTER(ModelConfig)]),
    "b4r_workspace_bytes": (_I64, [C.POINTER(ModelConfig), _I32, _I32, _I32]),
    "b4r_workspace_bytes_encoder": (_I64, [C.POINTER(ModelConfig), _I32, _I32, _I32]),
    "b4r_workspace_region": (C.c_int, [C.POINTER(ModelConfig), _I32, _I32, _I32, C.c_char_p, C.POINTER(_I64),
                                       C.POINTER(_I32), C.POINTER(_I32), C.POINTER(_I32)]),
    "b4r_fused_head_supported": (_I32, [C.POINTER(ModelConfig)]),
    "b4r_forward": (C.c_int, [C.POINTER(ModelConfig), C.POINTER(Batch), _P, _P, _P, _I64, _P, _I32, _P]),
    "b4r_loss": (C.c_int, [C.POINTER(ModelConfig), C.POINTER(Batch), _P, _I64, _P, _I32, _P]),
    "b4r_backward": (C.c_int, [C.POINTER(ModelConfig), C.POINTER(Batch), _P, _P, _P, _I64, _P, _I32, _P]),
    "b4r_optimizer_step": (C.c_int, [C.POINTER(ModelConfig), C.POINTER(AdamWConfig), _P, _P, _P, _P, _P, _I64, _P, _P]),
    "b4r_optimizer_step_reduced": (C.c_int, [C.POINTER(ModelConfig), C.POINTER(AdamWConfig), _P, _P, _P, _P, _P, _I64, _P, _P]),
    "b4r_state_begin_step": (C.c_int, [_P, _P]),
    "b4r_train_step": (C.c_int, [C.POINTER(ModelConfig), C.POINTER(AdamWConfig), C.POINTER(Batch), _P, _P, _P, _P, _P,
                                 _I64, _P, _P]),
    "b4r_rank_scratch_bytes": (_I64, [_I32, _I32]),
    "b4r_rank_candidates": (C.c_int, [_P, _I32, _P, _P, _P, _I32, _I32, _P, _I32, _I32, _P, _P, _P, _P, _P, _I64, _P]),
    "b4r_rank_metrics": (C.c_int, [_P, _I32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _I32, _P, _P, _P]),
    "b4r_mlm_transform_rows": (C.c_int, [C.POINTER(ModelConfig), _P, _P, _I64, _P, _I32, _P, _P, _P]),
    "b4r_embed_ln_fwd": (C.c_int, [_P, _I32, _I32, _P, _I32, _P, _P, _P, _I32, _F, _P, _P, _P, _P, _F, _P]),
    "b4r_ln_fwd": (C.c_int, [_P, _I32, _I32, _P, _P, _F, _P, _P, _P, _P]),
    "b4r_ln_bwd_scratch_floats": (_I64, [_I32, _I32]),
    "b4r_ln_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _P, _P, _P, _P, _P]),
    "b4r_set_gemm_mode": (C.c_int, [C.c_int]),
    "b4r_get_gemm_mode": (C.c_int, []),
    "b4r_gemm_f32": (C.c_int, [C.POINTER(GemmDesc), _P]),
    "b4r_gemm_ln_supported": (C.c_int, [C.POINTER(GemmDesc)]),
    "b4r_gemm_ln_bwd_partial_floats": (_I64, [_I32]),
    "b4r_gemm_tn_scratch_floats": (_I64, [_I32, _I32, _I32]),
    "b4r_gemm_tn_f32": (C.c_int, [C.POINTER(GemmTnDesc), _P, _P]),
    "b4r_gemm_tn_dgrad_supported": (C.c_int, [C.POINTER(GemmTnDesc)]),
    "b4r_attn_fwd": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P, _P, _U32, _F, _P, _P]),
    "b4r_attn_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _F, _P, _P, _U32, _F, _P, _P]),
    "b4r_attn_keep_words": (C.c_int64, [_I32, _I32, _I32]),
    "b4r_attn_block_supported": (_I32, [_I32, _I32, _I32]),
    "b4r_attn_block_fwd": (C.c_int, [C.POINTER(AttnBlockDesc), _P]),
    "b4r_attn_block_bwd_supported": (_I32, [_I32, _I32, _I32]),
    "b4r_attn_block_bwd_scratch_floats": (_I64, [_I32]),
    "b4r_attn_block_bwd_dw_scratch_floats": (_I64, [_I32]),
    "b4r_attn_block_bwd": (C.c_int, [C.POINTER(AttnBlockBwdDesc), _P]),
    "b4r_mlm_rows": (C.c_int, [_P, _P, _I32, _I32, _I32, _P, _P, _P, _P]),
    "b4r_encoder_layer_supported": (_I32, [_I32, _I32, _I32, _I32]),
    "b4r_encoder_layer_bwd_scratch_floats": (_I64, [_I32]),
    "b4r_encoder_layer_fwd": (C.c_int, [C.POINTER(AttnBlockDesc), C.POINTER(FfnDesc), _P]),
    "b4r_encoder_layer_bwd": (C.c_int, [C.POINTER(FfnDesc), C.POINTER(AttnBlockBwdDesc), _P, _P, _P, _P, _P, _P]),
    "b4r_ffn_block_supported": (_I32, [_I32, _I32]),
    "b4r_ffn_block_bwd_scratch_floats": (_I64, [_I32]),
    "b4r_ffn_block_fwd": (C.c_int, [C.POINTER(FfnDesc), _P]),
    "b4r_ffn_block_bwd": (C.c_int, [C.POINTER(FfnDesc), _P]),
    "b4r_ffn_wide_supported": (_I32, [_I32, _I32]),
    "b4r_ffn_wide_scratch_floats": (_I64, [_I32, _I32]),
    "b4r_ffn_wide_fwd": (C.c_int, [C.POINTER(FfnDesc), _P, _P, _P]),
    "b4r_ffn_wide_bwd": (C.c_int, [C.POINTER(FfnDesc), _P, _P, _P, _I32, _P]),
    "b4r_gather_rows": (C.c_int, [_P, _I32, _P, _I64, _I32, _I32, _I32, _P, _P]),
    "b4r_scatter_add_rows": (C.c_int, [_P, _P, _I64, _I32, _I32, _I32, _P, _I32, _P, _P]),
    "b4r_mlm_head_fused_scratch_floats": (C.c_int64, [_I32, _I32, _I32]),
    "b4r_mlm_head_fused_fwd": (C.c_int, [_P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _P, _P, _I32, _P]),
    "b4r_mlm_head_fused_bwd": (C.c_int, [_P, _P, _P, _P, _P, _I32, _I32, _I32, _P, _P, _P, _P]),
    "b4r_timing_begin": (C.c_int, [_P, _I32]),
    "b4r_timing_end": (C.c_int, [C.POINTER(_I32), C.POINTER(C.c_float), C.c_char_p, _I32, _I32]),
    "b4r_mask_batch": (C.c_int, [_P, _P, _P, _I32, _I32, _I32, _I32, C.c_double, _F, _F, _I32, C.c_uint64, _P, _P, _P, _P, _P, _P, _P]),
    "b4r_uniform_from_hash": (_F, [_U32]),
    "b4r_sample_candidates": (C.c_int, [_P, _I32, _P, _I32, _P, _I32, _I32, C.c_uint64, _P, _P]),
    "b4r_attn32_set_min_len": (_I32, [_I32]),
    "b4r_attn32_set_core_fwd": (_I32, [_I32]),
    "b4r_sample_candidates_flagged": (C.c_int, [_P, _I32, _P, _I32, _P, _I32, _I32, C.c_uint64, _P, _P, _P]),
    "b4r_softmax_ce": (C.c_int, [_P, _I32, _I32, _I32, _P, _P, _P, _I32, _P]),
    "b4r_global_sqnorm": (C.c_int, [_P, _I64, _P, _P, _P]),
    "b4r_adamw_step": (C.c_int, [C.POINTER(AdamWConfig), _P, _P, _P, _P, _I64, _I64, _P, _P]),
}

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load the HIP library.  Fails loudly: the product has no other compute path."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise B4RError(f"{LIB_PATH} is missing: build it first (python -c 'import __graft_entry__ as g; g.build()' "
                       f"or bert4rec_amd/build.py). bert4rec_amd has no CPU fallback.")
    # The library and torch must share ONE HIP runtime in the process (device pointers, streams and events cross the
    # boundary).  torch ships its own libamdhip64 and registers it under the same soname the library links against, so torch
    # has to be imported first: the loader then resolves the library's dependency to the copy torch already loaded.  Loading
    # the library first pulls in /opt/rocm's copy next to torch's, and its launches fail with "no ROCm-capable device".
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def last_error() -> str:
    buf = C.create_string_buffer(512)
    load().b4r_last_error(buf, 512)
    return buf.value.decode("utf-8", "replace")


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        raise B4RError(f"{what or 'b4r call'} failed with code {rc}: {last_error()}")
