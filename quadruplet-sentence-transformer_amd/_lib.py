"""ctypes binding of libqst.so (include/qst.h, include/qst_kernels.h).

There is no fallback: if the shared library is missing the import of anything
numeric fails with an explicit error (build it with `python -c "import
__graft_entry__ as g; g.build()"` or `make -C quadruplet-sentence-transformer_amd/csrc`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libqst.so")

c_i64p = C.POINTER(C.c_int64)
c_i32p = C.POINTER(C.c_int32)
c_f32p = C.POINTER(C.c_float)
vp = C.c_void_p


class QstError(RuntimeError):
    pass


class QstConfig(C.Structure):
    _fields_ = [("arch", C.c_int32), ("vocab_size", C.c_int32), ("hidden_size", C.c_int32),
                ("num_layers", C.c_int32), ("num_heads", C.c_int32), ("intermediate_size", C.c_int32),
                ("max_position", C.c_int32), ("type_vocab_size", C.c_int32), ("layer_norm_eps", C.c_float),
                ("normalize", C.c_int32), ("rel_buckets", C.c_int32), ("rel_max_distance", C.c_int32),
                ("pad_token_id", C.c_int32), ("precision", C.c_int32)]


class QstDrop(C.Structure):
    """include/qst_kernels.h: dropout mask = f(state {seed lo, seed hi, step, 0} on the device, site, element index)."""
    _fields_ = [("state", vp), ("site", C.c_uint32), ("thr16", C.c_uint32)]


class QstAttnDesc(C.Structure):
    _fields_ = [("qkv", vp), ("mask", vp), ("rel_pos", vp), ("nseq", C.c_int32), ("L", C.c_int32), ("A", C.c_int32),
                ("d", C.c_int32), ("ctx", vp), ("lse", vp), ("dctx", vp), ("dqkv", vp), ("drel", vp), ("delta_scratch", vp),
                ("force_split", C.c_int32), ("drop", QstDrop)]


class QstGemmArgs(C.Structure):
    _fields_ = [("A", vp), ("B", vp), ("C", vp), ("C2", vp), ("aux", vp), ("bias", vp), ("resid", vp),
                ("colsum", vp), ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("lda", C.c_int32),
                ("ldb", C.c_int32), ("ldc", C.c_int32), ("ldr", C.c_int32), ("splits", C.c_int32), ("bscale", vp),
                ("drop", QstDrop), ("drop_where", C.c_int32), ("C3", vp), ("C4", vp), ("B2", vp), ("b2_n0", C.c_int32), ("sat16", C.c_int32)]


class QstLnEpi(C.Structure):
    _fields_ = [("gamma", vp), ("beta", vp), ("eps", C.c_float), ("xhat", vp), ("rstd", vp), ("partials", vp)]


class QstFfnArgs(C.Structure):
    _fields_ = [("A", vp), ("B1", vp), ("B2", vp), ("bias1", vp), ("bias2", vp), ("resid", vp), ("aux", vp),
                ("save_gp", vp), ("save_h", vp), ("C", vp), ("C2", vp), ("M", C.c_int32), ("H", C.c_int32), ("I", C.c_int32),
                ("diag", C.c_int32)]


class QstLnReduceBatch(C.Structure):
    _fields_ = [("count", C.c_int32), ("H", C.c_int32), ("nblocks", C.c_int32), ("partials", vp * 32), ("dgamma", vp * 32),
                ("dbeta", vp * 32), ("nblocks_each", C.c_int32 * 32)]


class QstTnGroup(C.Structure):
    _fields_ = [("nprob", C.c_int32), ("splits", C.c_int32), ("total_tiles", C.c_int32), ("ranges_per_xcd", C.c_int32),
                ("tiles", C.c_int32 * 8),
                ("prob", QstGemmArgs * 8)]


# name -> (restype, argtypes). Every symbol the two public headers declare.
SIGNATURES = {
    "qst_strerror": (C.c_char_p, [C.c_int]),
    "qst_last_hip_error": (C.c_int, []),
    "qst_version": (C.c_int, []),
    "qst_arena_elems": (C.c_int64, [C.POINTER(QstConfig)]),
    "qst_arena_num_segments": (C.c_int, [C.POINTER(QstConfig)]),
    "qst_arena_segment": (C.c_int, [C.POINTER(QstConfig), C.c_int, C.POINTER(C.c_char_p), c_i64p, c_i64p, c_i32p, c_i32p]),
    "qst_shadow_elems": (C.c_int64, [C.POINTER(QstConfig)]),
    "qst_encoder_create": (C.c_int, [C.POINTER(QstConfig), C.POINTER(vp)]),
    "qst_encoder_destroy": (None, [vp]),
    "qst_encoder_saved_bytes": (C.c_size_t, [vp, C.c_int, C.c_int, C.c_int]),
    "qst_encoder_bwd_workspace_bytes": (C.c_size_t, [vp, C.c_int, C.c_int]),
    "qst_refresh_shadow": (C.c_int, [vp, vp, vp, vp]),
    "qst_refresh_shadow_mx": (C.c_int, [vp, vp, vp, vp]),
    "qst_encoder_forward": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_size_t, C.c_int, vp]),
    "qst_encoder_backward": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t, vp]),
    "qst_encoder_backward_partial": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t,
                                               C.c_int, C.c_int, C.c_int, C.c_int, vp]),
    "qst_encoder_backward_stage": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.c_size_t, vp, C.c_size_t,
                                             C.c_int, C.c_int, C.c_int, vp]),
    "qst_quadruplet_loss": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float,
                                      C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp, vp]),
    "qst_clip_adamw_step": (C.c_int, [vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                      C.c_float, C.c_float, C.c_int64, vp, vp, vp]),
    "qst_shadow8_bytes": (C.c_int64, [C.POINTER(QstConfig)]),
    "qst_clip_adamw_step_sched": (C.c_int, [vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                            C.c_float, C.c_float, C.c_int64, C.c_int64, vp, vp, vp, vp]),
    "qst_topk_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "qst_topk_scores": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, C.c_size_t, vp]),
    "qst_topk_scores_capped": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp, vp,
                                         C.c_size_t, vp]),
    "qst_score_workspace_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "qst_score_matrix": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, C.c_int64, vp, C.c_size_t, vp]),
    # kernel level (include/qst_kernels.h)
    "qst_gemm_nt": (C.c_int, [C.POINTER(QstGemmArgs), C.c_int, vp]),
    "qst_gemm_nt_f8": (C.c_int, [C.POINTER(QstGemmArgs), C.c_int, vp]),
    "qst_quant_mx": (C.c_int, [vp, C.c_int, C.c_int64, C.c_int, vp, vp, vp]),
    "qst_gemm_nt_ln_supported": (C.c_int, [C.c_int]),
    "qst_gemm_nt_ln": (C.c_int, [C.POINTER(QstGemmArgs), C.POINTER(QstLnEpi), C.c_int, vp]),
    "qst_gemm_nt_ln_block_rows": (C.c_int, [C.c_int]),
    "qst_gemm_nt_ln_block_rows_m": (C.c_int, [C.c_int, C.c_int]),
    "qst_gemm_nt8_ln_block_rows": (C.c_int, [C.c_int, C.c_int]),
    "qst_gemm8_stagger": (C.c_int, [C.c_int]),
    "qst_gemm8_ln_store": (C.c_int, [C.c_int]),
    "qst_gemm_nt8_ln_supported": (C.c_int, [C.c_int]),
    "qst_gemm_nt8_ln": (C.c_int, [C.POINTER(QstGemmArgs), C.POINTER(QstLnEpi), C.c_int, vp]),
    "qst_gemm_nt8_ln_timeouts": (C.c_int, []),
    "qst_gemm_nt8_f8_ln": (C.c_int, [C.POINTER(QstGemmArgs), C.POINTER(QstLnEpi), vp]),
    "qst_ffn_chain_supported": (C.c_int, [C.c_int, C.c_int]),
    "qst_ffn_chain": (C.c_int, [C.POINTER(QstFfnArgs), C.POINTER(QstLnEpi), C.c_int, vp]),
    "qst_gemm_tn": (C.c_int, [C.POINTER(QstGemmArgs), vp]),
    "qst_gemm_tn_group": (C.c_int, [C.POINTER(QstTnGroup), vp]),
    "qst_gemm_nt8_supported": (C.c_int, [C.POINTER(QstGemmArgs), C.c_int]),
    "qst_gemm_nt8": (C.c_int, [C.POINTER(QstGemmArgs), C.c_int, C.c_int, vp]),
    "qst_gemm_tn8_group": (C.c_int, [C.POINTER(QstTnGroup), vp]),
    "qst_gemm_nt8_f8": (C.c_int, [C.POINTER(QstGemmArgs), C.c_int, C.c_int, vp]),
    "qst_gemm8_mode": (C.c_int, [C.c_int]),
    "qst_embed_ln_fwd": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "qst_embed_ln_fwd_drop": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp,
                                        C.POINTER(QstDrop), vp]),
    "qst_ln_fwd_mx": (C.c_int, [vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "qst_embed_ln_fwd_mx": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "qst_ln_fwd_mx_train": (C.c_int, [vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp, vp]),
    "qst_embed_ln_fwd_mx_train": (C.c_int, [vp, vp, vp, vp, vp, vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp,
                                            vp, vp]),
    "qst_ln_fwd": (C.c_int, [vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "qst_ln_bwd_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int]),
    "qst_ln_bwd_reduce_batch": (C.c_int, [vp, vp]),
    "qst_ln_bwd": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, vp]),
    "qst_ln_bwd_drop": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, vp, vp, C.POINTER(QstDrop), C.POINTER(QstDrop), vp]),
    "qst_embed_bwd": (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "qst_position_ids": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "qst_forward_prologue": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "qst_pool_norm_fwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "qst_pool_norm_bwd": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "qst_attention_fwd": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "qst_attention_bwd": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "qst_attention_fwd_ex": (C.c_int, [C.POINTER(QstAttnDesc), vp]),
    "qst_attention_bwd_ex": (C.c_int, [C.POINTER(QstAttnDesc), vp]),
    "qst_dropout_multipliers": (C.c_int, [C.POINTER(QstDrop), C.c_int, C.c_int64, vp, vp]),
    "qst_abi_sizeof": (C.c_int64, [C.c_int]),
    "qst_normalize_rows": (C.c_int, [vp, C.c_int, C.c_int, vp, vp]),
    "qst_dropout_init": (C.c_int, [vp, C.c_uint64, vp]),
    "qst_dropout_advance": (C.c_int, [vp, vp]),
    "qst_encoder_set_dropout": (C.c_int, [vp, C.c_float, C.c_float, vp]),
    "qst_encoder_set_ffn_chain": (C.c_int, [vp, C.c_int]),
    "qst_encoder_set_ln_fusion": (C.c_int, [vp, C.c_int]),
    "qst_transpose_f32": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, C.c_int, vp]),
    "qst_gelu_f32": (C.c_int, [vp, C.c_int64, vp, vp]),
    "qst_gelu_bwd_f32": (C.c_int, [vp, vp, C.c_int64, vp, vp]),
    "qst_colsum_f32": (C.c_int, [vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "qst_embed_sum_f32": (C.c_int, [vp, vp, vp, vp, vp, vp, C.c_int, C.c_int, vp, vp]),
    "qst_ln_bwd_f32": (C.c_int, [vp, vp, vp, C.c_float, C.c_int, C.c_int, vp, vp, vp, vp]),
    "qst_attention_bwd_f32": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "qst_attention_bwd_f32_drop": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp]),
    "qst_attention_bwd_x3_scratch_bytes": (C.c_size_t, [C.c_int, C.c_int, C.c_int]),
    "qst_attention_bwd_x3": (C.c_int, [vp, vp, vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp, vp, vp]),
    "qst_dropout_apply_f32": (C.c_int, [vp, vp, vp, C.c_int64, vp, vp]),
    "qst_comm_unique_id": (C.c_int, [vp]),
    "qst_comm_init": (C.c_int, [C.c_int, C.c_int, vp, C.POINTER(vp)]),
    "qst_allreduce_bucket": (C.c_int, [vp, vp, C.c_int64, C.c_int, vp]),
    "qst_comm_rank": (C.c_int, [vp]),
    "qst_comm_world": (C.c_int, [vp]),
    "qst_comm_destroy": (None, [vp]),
    "qst_comm_last_error": (C.c_char_p, []),
    "qst_rel_bucket_host": (C.c_int, [C.c_int, C.c_int, C.c_int]),
    "qst_rel_bias_fwd": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "qst_rel_bias_bwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "qst_rel_pos_fwd": (C.c_int, [vp, vp, C.c_int, C.c_int, vp, vp]),
    "qst_rel_pos_bwd": (C.c_int, [vp, vp, C.c_int, C.c_int, C.c_int, vp, vp]),
    "qst_topk_rows": (C.c_int, [vp, C.c_int64, vp, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
    "qst_shadow_matrix": (C.c_int, [vp, C.c_int, C.c_int, vp, vp, vp]),
    "qst_shadow_all": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, vp]),
    "qst_gemm_nt_x3": (C.c_int, [C.POINTER(QstGemmArgs), C.c_int, vp]),
    "qst_gemm_tn_x3": (C.c_int, [C.POINTER(QstGemmArgs), vp]),
    "qst_attention_fwd_x3": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp]),
    "qst_attention_fwd_x3_drop": (C.c_int, [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, vp]),
}

# f16-operand twins (include/qst_kernels.h, "f16-operand twins"): same signatures, IEEE half where the original has bf16
F16_TWINS = ["qst_gemm_nt", "qst_gemm_nt_ln", "qst_gemm_nt8_ln_supported", "qst_gemm_nt8_ln_block_rows", "qst_gemm_nt8_ln", "qst_gemm_nt8_ln_timeouts", "qst_ffn_chain", "qst_gemm_tn", "qst_gemm_tn_group", "qst_gemm_nt8_supported",
             "qst_gemm_nt8", "qst_gemm_tn8_group", "qst_embed_ln_fwd", "qst_embed_ln_fwd_drop", "qst_ln_fwd", "qst_ln_bwd",
             "qst_ln_bwd_drop", "qst_attention_fwd", "qst_attention_bwd", "qst_attention_fwd_ex", "qst_attention_bwd_ex",
             "qst_shadow_all", "qst_shadow_matrix"]
for _n in F16_TWINS:
    SIGNATURES[_n + "_f16"] = SIGNATURES[_n]
SIGNATURES["qst_shadow_all_split_f16"] = (C.c_int, [vp, vp, vp, vp, C.c_int, C.c_int, vp])
SIGNATURES["qst_amp_scaler_init"] = (C.c_int, [vp, C.c_float, vp])
SIGNATURES["qst_clip_adamw_step_amp"] = (C.c_int, [vp, vp, vp, vp, vp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                                  C.c_float, C.c_float, C.c_int64, C.c_int64, vp, vp, C.c_float, C.c_float,
                                                  C.c_int32, vp, vp, vp])

_lib: Optional[C.CDLL] = None


def load() -> C.CDLL:
    """Load libqst.so and bind every declared symbol; raises QstError if the extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise QstError(f"HIP extension not built: {LIB_PATH} is missing (no CPU fallback exists). "
                       f"Run `make -C {os.path.join(_HERE, 'csrc')}` or __graft_entry__.build().")
    # torch first: its wheel bundles a HIP runtime, and libqst.so must bind to THAT copy (device pointers and streams come
    # from torch). Loaded the other way round -- libqst.so pulling /opt/rocm's runtime in before torch is imported -- the
    # process holds two runtimes and the second one to initialise sees no device (`python __graft_entry__.py smoke`, round 5).
    import torch  # noqa: F401
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status: int, what: str = "") -> None:
    if status != 0:
        lib = load()
        msg = lib.qst_strerror(status).decode()
        extra = f" (hip error {lib.qst_last_hip_error()})" if status == -4 else ""
        raise QstError(f"{what or 'libqst'}: {msg}{extra}")


def make_config(cfg, precision: int = 0) -> QstConfig:
    return QstConfig(cfg.arch, cfg.vocab_size, cfg.hidden_size, cfg.num_layers, cfg.num_heads,
                     cfg.intermediate_size, cfg.max_position, cfg.type_vocab_size, cfg.layer_norm_eps,
                     int(cfg.normalize), cfg.rel_buckets, cfg.rel_max_distance, cfg.pad_token_id, precision)


def kfn(lib, name: str, op: str = "bf16"):
    """The kernel-level entry point `name` for 16-bit operand type `op` ("bf16" or "f16": the _f16 twin)."""
    return getattr(lib, name + ("_f16" if op == "f16" else ""))


def ptr(t) -> Optional[int]:
    """data_ptr of a torch tensor (or None)."""
    return None if t is None else t.data_ptr()


def current_stream_ptr() -> int:
    import torch
    return torch.cuda.current_stream().cuda_stream
