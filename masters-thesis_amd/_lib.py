"""ctypes binding of the C-ABI kernel library (include/tnt_hip.h).

There is NO fallback: if ``csrc/libtnt_hip.so`` is missing or a symbol is absent the
import of the library raises.  ``import torch`` happens first on purpose: torch bundles
its own libamdhip64 (soname libamdhip64.so.7) and the kernel library must bind to the
same HIP runtime instance so that torch's streams and device pointers are valid in it.
"""
import ctypes as C
import os

import torch  # noqa: F401  (must precede CDLL, see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
# TNT_HIP_LIB: A/B builds of the same library for tools/ (e.g. a variant compiled with an experiment macro)
LIB_PATH = os.environ.get("TNT_HIP_LIB") or os.path.join(_HERE, "csrc", "libtnt_hip.so")

P = C.c_void_p          # any device pointer
I32, I64, U32, U64, F32 = C.c_int32, C.c_int64, C.c_uint32, C.c_uint64, C.c_float

# name -> argtypes (restype is always int32); order mirrors include/tnt_hip.h
SIGNATURES = {
    "tnt_version": [],
    "tnt_gemm_f32": [P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, F32, I32, I32, P, P],
    "tnt_gemm_fused_f32": [P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, P],
    "tnt_gemm_fused_cfg": [I32, I32, I32, I32, I32, I32],
    "tnt_gemm3_f32": [P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, I32, P, P, P],
    "tnt_gemm3_plan": [I32, I32, I32, I32, I32, I32, I32, P, P],
    "tnt_gemm3_work_floats": [I32, I32, I32, I32, I32],
    "tnt_gemm3_sync_words": [I32, I32, I32, I32],
    "tnt_gemm3_work_arm": [P, I64, P],
    "tnt_gemm3_pair_supported": [I32, I32, I32, I32, I32, I32],
    "tnt_gemm3_pair_f32": [P, P, P],
    "tnt_span_sqnorm_lr_f32": [P, P, P, P, P, P, P, I32, P, P, P, F32, F32, P, P],
    "tnt_dense_gram_norm_spans_lr_f32": [P, P, P, P, I32, P, I32, F32, P, I32, I32, I32, P, P, P, P, P, P, P, I32, P, P, P, F32, F32, P, P],
    "tnt_dense_dw_adam_fin_f32": [P, P, P, P, P, F32, P, I32, I32, P, P, F32, F32, F32, F32, P, I32, I32, I32, I32, P],
    "tnt_adam_fin_f32": [P, P, P, P, P, P, P, P, I32, F32, F32, P, P, I32, P, I32, P, P],
    "tnt_dropout_mask4_u8": [P, I64, I32, F32, U64, U32, U32, P, P],
    "tnt_dropout_f32": [P, P, I32, I32, I32, I32, I32, I32, I32, F32, U64, U32, U32, P, P],
    "tnt_dropout_metric_f32": [P, P, I32, I32, I32, I32, I32, I32, I32, F32, U64, U32, U32, P, P, P, I32, I32, I32, P],
    "tnt_dropout2_f32": [P, P, I32, I32, I32, I32, I32, I32, I32, F32, U32, I32, I32, I32, I32, F32, U32, U64, U32, P, P],
    "tnt_act_bwd_f32": [P, P, P, I64, I32, F32, P],
    "tnt_bn_nchunk": [I32],
    "tnt_batchnorm_fwd_f32": [P, P, P, P, P, P, P, P, I32, I32, I32, I32, F32, F32, P, P],
    "tnt_batchnorm_bwd_f32": [P, P, P, P, P, P, P, I32, I32, I32, I32, P, P],
    "tnt_batchnorm_fwd_drop_f32": [P, P, P, P, P, P, P, P, I32, I32, I32, I32, F32, F32, P, F32, U64, U32, P, P],
    "tnt_batchnorm_bwd_act_f32": [P, P, P, P, P, P, P, I32, I32, I32, I32, P, P, F32, P],
    "tnt_batchnorm_stats_f32": [P, I32, I32, P, P],
    "tnt_batchnorm_apply_stats_f32": [P, I32, P, P, P, P, P, P, P, P, I32, I32, I32, F32, F32, P, P],
    "tnt_batchnorm_dx_f32": [P, I32, P, P, P, P, P, P, I32, I32, I32, P],
    "tnt_layernorm_fwd_f32": [P, P, P, P, P, P, I32, I32, I32, F32, P],
    "tnt_layernorm_bwd_f32": [P, P, P, P, P, P, P, I32, I32, I32, P, P],
    "tnt_colsum_f32": [P, P, I32, I32, I32, P, P],
    "tnt_bias_act_drop_bwd_f32": [P, P, P, P, I32, I32, I32, I32, F32, I32, I32, I32, F32, U64, U32, P, P, P, I32, I32, I32, P],
    "tnt_colsum2_f32": [P, P, I32, I32, I32, P, P, I32, I32, I32, P],
    "tnt_colsum4_f32": [P, P, I32, I32, I32, P, P, I32, I32, I32, P, P, I32, I32, I32, P, P, I32, I32, I32, P],
    "tnt_embedding_fwd_f32": [P, P, P, I32, I32, I32, I32, I32, P],
    "tnt_embedding_bwd_parts": [I32, I32, I32],
    "tnt_attention_front_bwd_parts": [I32, I32, I32],
    "tnt_attention_metric_parts": [I32, I32],
    "tnt_embedding_bwd_sparse_f32": [P, P, P, P, P, I32, I32, I32, I32, I32, F32, U64, U32, P, I32, P],
    "tnt_embedding_bwd_f32": [P, P, P, P, P, I32, I32, I32, I32, I32, P],
    "tnt_lstm_seq_supported": [I32, I32],
    "tnt_lstm_seq_fwd_f32": [P, P, P, P, P, P, I32, I32, P, P, I32, I32, I32, P, P, P],
    "tnt_lstm_seq_bwd_work_floats": [I32, I32],
    "tnt_lstm_seq_bwd_f32": [P, P, P, I32, I32, P, P, P, P, I64, I32, I32, I32, P, P, P],
    "tnt_lstm_step_fwd_f32": [P, P, P, P, P, P, I32, P, I32, I32, P, P, P, P, P, I32, I32, P, P],
    "tnt_ln_lstm_cell_fwd_f32": [P, P, P, P, P, P, P, P, P, P, P, I32, I32, F32, P],
    "tnt_ln_lstm_cell_bwd_f32": [P, P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, P],
    "tnt_lstm_step_bwd_f32": [P, P, P, P, P, P, P, P, I32, I32, P, P, P, P, P, P, P, I32, I32, P, I32, P, P],
    "tnt_softmax_cce_f32": [P, P, P, P, P, P, I32, I32, I32, F32, I32, I32, P],
    "tnt_onehot_argmax_f32": [P, P, I32, I32, I32, P],
    "tnt_beam_topk_f32": [P, P, P, I32, I32, I32, I32, I32, P, P, P, P, P],
    "tnt_argmax_rows_f32": [P, P, I32, I32, I32, P],
    "tnt_enc_tail_fwd_f32": [P, P, P, P, P, P, P, P, I32, I32, I32, I32, F32, F32, F32, F32, U64, U32, U32, P, P],
    "tnt_enc_tail_bwd_f32": [P, P, P, P, P, P, P, P, P, I32, I32, I32, F32, F32, F32, U64, U32, U32, P, P],
    "tnt_enc_tail_bwd_drop_f32": [P, P, P, P, P, P, P, P, P, I32, I32, I32, F32, F32, F32, U64, U32, U32, P, P, I32, I32, I32, I32,
                                  I32, I32, F32, U32, P],
    "tnt_embedding_fwd_drop_f32": [P, P, P, P, I32, I32, I32, I32, I32, F32, U64, U32, U32, P, P],
    "tnt_embedding_fwd_drop2_f32": [P, P, P, P, I32, I32, I32, I32, I32, F32, U64, U32, U32, P, F32, U32, I32, I32, P],
    "tnt_dense_dw_skinny_f32": [P, P, P, I32, I32, I32, I32, P],
    "tnt_dense_dw_sqnorm_f32": [P, P, P, F32, P, I32, I32, I32, I32, I32, P],
    "tnt_dense_dw_adam_f32": [P, P, P, P, P, F32, P, P, P, F32, F32, F32, F32, P, I32, I32, I32, I32, P],
    "tnt_dense_fwd_stream_gram_f32": [P, P, P, P, P, I32, I32, I32, I32, I32, I32, P],
    "tnt_dense_gram_norm_f32": [P, P, P, P, I32, P, I32, F32, P, I32, I32, I32, P],
    "tnt_dense_gram_norm_spans_f32": [P, P, P, P, I32, P, I32, F32, P, I32, I32, I32, P, P, P, P, P, P, P, I32, P],
    "tnt_dense_fwd_stream_f32": [P, P, P, I32, I32, I32, I32, I32, I32, P],
    "tnt_enc_tail_fwd_sk_f32": [P, I32, P, P, F32, P, P, P, P, P, P, P, I32, I32, I32, I32, F32, F32, F32, F32, U64, U32,
                                U32, P, P],
    "tnt_enc_tail_fwd_sk_emb_f32": [P, I32, P, P, F32, P, P, P, P, P, P, P, I32, I32, I32, I32, F32, F32, F32, F32, U64, U32,
                                    U32, P, P, P, P, I32, I32, I32, F32, U32, P],
    "tnt_gru_step_fwd_f32": [P, P, P, P, P, P, I32, I32, P],
    "tnt_gru_step_bwd_f32": [P, P, P, P, P, P, P, P, P, I32, I32, P],
    "tnt_lc_seq_fwd_f32": [P, P, P, P, P, P, P, P, P, P, P, I64, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, F32, F32, F32,
                           I32, U64, U32, U32, P, P, P, P, P],
    "tnt_lc_seq_fwd_work_floats": [I32],
    "tnt_lc_seq_bwd_f32": [P, P, P, P, P, P, P, I64, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, F32, F32, F32,
                           I32, U64, U32, U32, P, F32, P, P, P],
    "tnt_lc_seq_fwd_drop_f32": [P, P, P, P, P, P, P, P, P, P, P, I64, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, F32, F32,
                                F32, I32, U64, U32, U32, P, P, F32, U32, P, P, P, P],
    "tnt_lc_seq_bwd_drop_f32": [P, P, P, P, P, P, P, I64, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, I32, F32, F32,
                                F32, I32, U64, U32, U32, P, F32, F32, U32, P, P, P],
    "tnt_lc_seq_bwd_work_floats": [I32, I32],
    "tnt_attention_front_bwd_f32": [P, P, P, P, P, P, P, P, I32, I32, I32, F32, P],
    "tnt_attention_front_bwd_drop_f32": [P, P, P, P, P, P, P, P, I32, I32, I32, F32, F32, U64, U32, P, P],
    "tnt_gemm_blas_f32": [P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, P],
    "tnt_gemm_lt_f32": [P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, P],
    "tnt_locally_dense_fwd_split_f32": [P, I32, P, P, P, P, I32, P, P, P, P, P, I32, I32, I32, F32, I32, P],
    "tnt_locally_dense_bwd_split_f32": [P, I32, P, P, P, P, I32, P, P, P, I32, I32, I32, I32, P],
    "tnt_sum2_f32": [P, P, P, P, I32, F32, P],
    "tnt_stage_batch_f32": [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, P, I32, P],
    "tnt_stage_batch_h16": [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, P, I32, P],
    "tnt_stage_batch_masks_f32": [P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, P, I32, P, I64, I32, F32, U64, U32, P, P],
    "tnt_sample_rows_f32": [P, P, I32, I32, I32, F32, I32, U64, U32, U32, P, P],
    "tnt_sqdiff_mean_f32": [P, P, I64, F32, P],
    "tnt_sum_f32": [P, P, I32, F32, P],
    "tnt_l2_total_f32": [P, P, I32, P, P],
    "tnt_seg_sqnorm_f32": [P, P, P, P, P, P, P, P, P, P, P, I32, I32, P],
    "tnt_span_sqnorm_f32": [P, P, P, P, P, P, P, I32, P],
    "tnt_step_finalize_f32": [P, P, P, P, P, P, I32, P, P, P, P, I32, F32, P, P, I32, P, P, I32, P, P, I32, F32, P, P, P, P,
                              F32, F32, P, P],
    "tnt_adam_f32": [P, P, P, P, P, P, P, P, P, P, I32, F32, P, F32, F32, F32, F32, P, P],
    "tnt_adam_ring_f32": [P, P, P, P, P, P, P, P, P, P, I32, F32, P, F32, F32, F32, F32, P, P, I32, P, I32, P, P],
    "tnt_sgd_f32": [P, P, P, P, P, P, P, P, P, I32, F32, P, F32, F32, P, P],
    "tnt_agc_f32": [P, P, P, P, P, P, P, I32, P, P, I32, I32, I32, P, P, F32, F32, P],
    "tnt_colsq_f32": [P, P, I32, I32, I32, P],
    "tnt_step_tick": [P, P, P, P, F32, F32, P, P],
    "tnt_sam_f32": [P, P, P, P, P, P, P, P, P, I32, I32, F32, I32, P],
    "tnt_gemm_f32_tile": [P, P, P, P, P, I32, I32, I32, I32, I32, I32, I32, I32, I32, F32, I32, I32, P, I32, I32, P],
    "tnt_block_dense_dx_f32": [P, P, P, I32, I32, I32, I32, P],
    "tnt_locally_dense_fwd_f32": [P, I32, P, P, P, P, P, P, I32, I32, I32, F32, P],
    "tnt_locally_dense_bwd_f32": [P, I32, P, P, P, P, P, I32, I32, I32, P],
    "tnt_attention_step_fwd_f32": [P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, F32, F32, F32,
                                   I32, U64, U32, U32, U32, P, P, P],
    "tnt_attention_step_bwd_f32": [P, P, P, P, P, P, P, P, P, P, P, P, I32, I32, I32, I32, I32, F32, F32, F32,
                                   I32, U64, U32, U32, U32, P, P, P, P, I32, P, F32, I32, P],
    "tnt_attention_metric_f32": [P, P, P, I32, I32, I32, I64, P],
}

_lib = None


class KernelLibraryError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes library.  Raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise KernelLibraryError(
            f"HIP kernel library not built: {LIB_PATH} is missing. Run "
            "`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C masters-thesis_amd/csrc`). "
            "There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise KernelLibraryError(f"{LIB_PATH} does not export {name}") from e
        fn.argtypes = argtypes
        fn.restype = I32
    _lib = lib
    return lib


def check(rc, name):
    if rc != 0:
        raise KernelLibraryError(f"{name} failed with code {rc}")
