"""ctypes binding of ``csrc/libmtam_hip.so`` (the C ABI in include/mtam_hip.h).

There is no CPU fallback: if the shared library is missing or an entry point
fails, the caller gets an exception.  The product path never imports
``oracle``.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libmtam_hip.so")

c_int, c_float, c_void_p, c_size_t = ctypes.c_int, ctypes.c_float, ctypes.c_void_p, ctypes.c_size_t
P = c_void_p       # device pointer

# name -> (restype, argtypes); every symbol include/mtam_hip.h declares
SIGNATURES = {
    "mtam_last_error": (ctypes.c_char_p, []),
    "mtam_version": (c_int, []),
    "mtam_arch": (ctypes.c_char_p, []),
    "mtam_gemm_sq_partials": (c_int, [c_int, c_int]),
    "mtam_gemm_f32_dual": (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, c_int, c_int, P, c_int, P, c_int,
                                   P, c_int, c_int, P, P, P, c_int, P]),
    "mtam_gemm_f32": (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, P, c_int, P, c_int, c_int,
                              P, P, P, c_int, c_int, P]),
    "mtam_gemm_f32_batched": (c_int, [c_int, c_int, c_int, c_int, c_int, P, c_int, ctypes.c_long, ctypes.c_long,
                                      P, c_int, ctypes.c_long, ctypes.c_long, P, c_int, ctypes.c_long,
                                      ctypes.c_long, c_int, c_int, c_int, P]),
    "mtam_colsum_atomic": (c_int, [P, c_int, c_int, c_int, P, P]),
    "mtam_emb_gather_partials": (c_int, [c_int, c_int]),
    "mtam_emb_gather_fwd": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, P, P, P, P, c_int, c_int, c_int,
                                    P, P, P, P, P]),
    "mtam_emb_gather_fwd_clear": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, P, P, P, P, c_int, c_int, c_int,
                                          P, P, P, P, P, c_size_t, P, c_size_t, P]),
    "mtam_emb_gather_fwd_item16": (c_int, [P, P, c_int, P, c_int, P, c_int, P, c_int, P, P, P, P, c_int, c_int, c_int,
                                           P, P, P, P, P, c_size_t, P, c_size_t, P]),
    "mtam_emb_scatter_partials": (c_int, [c_int, c_int]),
    "mtam_emb_scatter_add_bwd": (c_int, [P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_float, c_int,
                                         P, c_int, P, c_int, P, c_int, P, c_int, P, P]),
    "mtam_emb_scatter_add_bwd_postab": (c_int, [P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_float, c_int,
                                                P, c_int, P, c_int, P, c_int, P, c_int, P, P]),
    "mtam_emb_scatter_add_bwd_fused": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_float, c_int,
                                               P, c_int, P, c_int, P, c_int, P, c_int, P, P]),
    "mtam_emb_scatter_add_bwd_range": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_float, c_int,
                                               P, c_int, P, c_int, P, c_int, P, c_int, P, c_int, c_int, c_int, P]),
    "mtam_emb_scatter_add_bwd_norm": (c_int, [P, P, P, P, P, P, P, P, P, P, P, P, P, c_int, c_int, c_float, c_int,
                                              P, c_int, P, c_int, P, c_int, P, c_int, P, P, P]),
    "mtam_rows_gather_range": (c_int, [P, c_int, c_int, P, ctypes.c_long, P, P]),
    "mtam_seq_chain_gather_partials": (c_int, [c_int, c_int]),
    "mtam_seq_chain_gather_fwd": (c_int, [P, c_int, P, c_int, P, c_int, P, c_int, P, P, P, P, c_int, c_int, c_int,
                                          P, P, P, c_int, P, P, c_int, P, P, P, c_int, P, P, P, P, P, c_size_t, P,
                                          c_size_t, P, P]),
    "mtam_seq_chain_images_elems": (c_size_t, [c_int, c_int]),
    "mtam_seq_chain_image_offset": (c_size_t, [c_int, c_int]),
    "mtam_split_weight_images": (c_int, [P, c_int, c_int, P, P]),
    "mtam_split_weight_rows": (c_int, [P, c_int, c_int, P, P]),
    "mtam_seq_chain_bwd_max_k": (c_int, []),
    "mtam_seq_chain_bwd": (c_int, [P, c_int, P, c_int, P, P, c_int, P, P, P, P, P]),
    "mtam_tagru_fwd": (c_int, [P, P, P, P, P, P, P, c_int, c_int, P, P, P, P]),
    "mtam_tagru_bwd": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, P, P, P, P, P]),
    "mtam_tagru_fwd_kv": (c_int, [P, P, P, P, P, P, P, c_int, c_int, P, P, P, P, P, c_int, P, P, P]),
    "mtam_gru_weight_image_floats": (c_int, []),
    "mtam_gru_weight_image_pos": (c_int, [c_int, c_int, c_int]),
    "mtam_gru_weight_image": (c_int, [P, P, P, P]),
    "mtam_tagru_bwd_dkv": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, P, P, P, P, P, c_int, P, P, P]),
    "mtam_tagru_seqrec_fwd": (c_int, [P, P, P, P, c_int, c_int, P, P, P, P]),
    "mtam_tagru_seqrec_bwd": (c_int, [P, P, P, P, P, P, c_int, c_int, P, P, P, P, P]),
    "mtam_tsr_time_inputs_fwd": (c_int, [P, P, P, c_int, P, P]),
    "mtam_tsr_time_inputs_bwd": (c_int, [P, P, P, P, c_int, P, P]),
    "mtam_ta_attn_decode_save_floats": (c_int, [c_int, c_int]),
    "mtam_ta_attn_decode_fwd": (c_int, [P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P, P,
                                        c_int, c_int, c_int, P, P, P, P, P, P, P]),
    "mtam_ta_attn_decode_bwd": (c_int, [P, P, P, P, c_int, c_int, c_int, P, P, P, P, P, P, P,
                                        c_int, c_int, c_int, c_int, P, P, P, P, P, P, P, P, P, P, P]),
    "mtam_layer_norm_fwd": (c_int, [P, P, P, P, c_float, c_int, c_int, P, P, P]),
    "mtam_ta_selfattn_gate_softmax_fwd": (c_int, [P, P, P, P, P, c_int, c_int, c_int, P, P, P, P]),
    "mtam_ta_selfattn_gate_softmax_bwd": (c_int, [P, P, P, P, P, P, P, P, P, c_int, c_int, c_int, P, P, P]),
    "mtam_seq_row_gather": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "mtam_seq_row_scatter": (c_int, [P, P, c_int, c_int, c_int, P, P]),
    "mtam_relu_bwd_inplace": (c_int, [P, P, c_size_t, P]),
    "mtam_layer_norm_bwd": (c_int, [P, P, P, c_int, P, P, P]),
    "mtam_softmax_ce_partials": (c_int, [c_int, c_int]),
    "mtam_softmax_ce": (c_int, [P, c_int, P, c_int, c_int, c_float, P, P, P, P, P]),
    "mtam_loss_reduce": (c_int, [P, c_int, P, c_int, c_float, c_float, P, P]),
    "mtam_softmax_ce_loss": (c_int, [P, c_int, P, c_int, c_int, c_float, P, P, P, P, P, c_int, c_float, c_float,
                                     P, P]),
    "mtam_topk": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P]),
    "mtam_topk_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mtam_topk_ws": (c_int, [P, c_int, c_int, c_int, c_int, P, P, P, P]),
    "mtam_partials_sum": (c_int, [P, c_int, c_float, P, c_int, P]),
    "mtam_clip_scale_sq": (c_int, [P, c_int, c_float, P, P, P, P]),
    "mtam_topk_stream_segments": (c_int, [c_int]),
    "mtam_topk_stream_workspace_bytes": (c_size_t, [c_int, c_int, c_int]),
    "mtam_topk_stream_slab": (c_int, [P, c_int, c_int, c_int, c_int, c_int, c_int, P, P]),
    "mtam_topk_stream_finish": (c_int, [P, c_int, c_int, c_int, P, P, P]),
    "mtam_f32_to_bf16": (c_int, [P, c_size_t, P, c_size_t, P]),
    "mtam_score16_batch_pad": (c_int, [c_int]),
    "mtam_score16_partials": (c_int, [c_int, c_int]),
    "mtam_score16_sq_partials": (c_int, [c_int]),
    "mtam_score16_lse": (c_int, [P, P, P, c_int, c_int, P, P, P, P]),
    "mtam_score16_bwd": (c_int, [P, P, P, P, c_int, c_int, c_float, P, P, P, P]),
    "mtam_seq_chain_fwd": (c_int, [P, P, P, c_int, P, P, c_int, P, P, c_int, P, P, P, P, P, P]),
    "mtam_score32_set_split_min_rows": (None, [ctypes.c_long]),
    "mtam_score32_partials": (c_int, [c_int, c_int]),
    "mtam_score32_sq_partials": (c_int, [c_int]),
    "mtam_score32_lse": (c_int, [P, P, P, c_int, c_int, P, c_int, P, P, P]),
    "mtam_score32_lse_range": (c_int, [P, P, P, c_int, c_int, c_int, P, c_int, P, P, P]),
    "mtam_score32_bwd_range": (c_int, [P, P, P, P, c_int, c_int, c_int, c_float, P, P, P, c_int, P]),
    "mtam_score32_bwd": (c_int, [P, P, P, P, c_int, c_int, c_float, P, P, P, c_int, P]),
    "mtam_score32_set_fused": (None, [c_int]),
    "mtam_score32_train_is_fused": (c_int, [c_int, c_int]),
    "mtam_score32_train_work_floats": (ctypes.c_long, [c_int, c_int]),
    "mtam_score32_train_work_init": (c_int, [P, ctypes.c_long, c_int, c_int, P]),
    "mtam_score32_train": (c_int, [P, P, P, c_int, c_int, c_float, P, ctypes.c_long, P, P, P, P, P, c_int, P]),
    "mtam_score16_logits": (c_int, [P, P, c_int, c_int, P, ctypes.c_long, P]),
    "mtam_sqnorm_blocks": (c_int, [c_size_t]),
    "mtam_sqnorm_partial": (c_int, [P, c_size_t, P, P]),
    "mtam_clip_scale": (c_int, [P, c_int, c_float, P, P, P, P]),
    "mtam_sqnorm_clip_scale": (c_int, [P, c_size_t, P, c_int, c_int, c_float, P, P, P, P, P, c_int, P, c_int,
                                       c_float, c_float, P, P]),
    "mtam_adam_block": (c_int, []),
    "mtam_adam": (c_int, [P, P, P, P, c_size_t, P, P, c_size_t, P]),
    "mtam_adam_bf16copy": (c_int, [P, P, P, P, c_size_t, P, P, c_size_t, P, c_size_t, P]),
    "mtam_adam_images": (c_int, [P, P, P, P, c_size_t, P, P, c_size_t, P, c_size_t, P, c_int, P]),
    "mtam_adam_images_clip": (c_int, [P, P, P, P, c_size_t, P, c_int, c_float, P, P, c_size_t, P, c_size_t, P, c_int,
                                      P]),
    "mtam_adam_images_clip_feed": (c_int, [P, P, P, P, c_size_t, P, c_int, c_float, P, P, c_size_t, P, c_size_t, P,
                                           c_int, P, c_int, c_int, P, P, P]),
    "mtam_adam_clip_max_partials": (c_int, []),
    "mtam_sqnorm_state_loss": (c_int, [P, c_size_t, P, c_int, P, P, P, c_int, P, c_int, c_float, c_float, P, P]),
    "mtam_opt_update": (c_int, [c_int, P, P, P, P, c_size_t, P, P, c_size_t, c_size_t, P]),
    "mtam_gemm_tn_atomic_grouped": (c_int, [c_int, P, P]),
    "mtam_colsum_atomic_multi": (c_int, [c_int, P, P]),
    "mtam_weight_grads": (c_int, [c_int, P, c_int, P, P]),
}



class GemmDesc(ctypes.Structure):
    _fields_ = [("A", c_void_p), ("lda", c_int), ("B", c_void_p), ("ldb", c_int), ("C", c_void_p),
                ("ldc", c_int), ("M", c_int), ("N", c_int), ("K", c_int), ("split_k", c_int)]


class WeightImages(ctypes.Structure):
    """MtamWeightImages (include/mtam_hip.h): one weight matrix of the flat space and where its bf16 images go."""
    _fields_ = [("begin", c_size_t), ("K", c_int), ("N", c_int), ("images", c_void_p), ("images_r", c_void_p),
                ("gru_which", c_int)]


class NormRider(ctypes.Structure):
    """MtamNormRider (include/mtam_hip.h): the clip's partial pass riding in the scatter-add launch."""
    _fields_ = [("g", c_void_p), ("n", c_size_t), ("partials", c_void_p), ("offset", c_int), ("lr", c_void_p),
                ("adam_state", c_void_p), ("l2_partial", c_void_p), ("n_l2", c_int), ("ce", c_void_p), ("B", c_int),
                ("reg", c_float), ("ce_scale", c_float), ("loss", c_void_p)]


class ColsumJob(ctypes.Structure):
    _fields_ = [("in_", c_void_p), ("rows", c_int), ("cols", c_int), ("ld", c_int), ("out", c_void_p)]


_lib = None


class MtamHipError(RuntimeError):
    pass


def load():
    """Load the library once; raise if it was not built (no silent fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MtamHipError(
            "libmtam_hip.so not found at %s -- build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C mtamrecommender_amd/csrc`" % LIB_PATH)
    # One HIP runtime per process: PyTorch ships its own libamdhip64 and the device memory comes from it, so it
    # has to be the copy this library binds to.  Loaded the other way round (this library first, against
    # /opt/rocm's runtime, torch afterwards) the process holds two runtimes and the first launch from here
    # fails with "no ROCm-capable device is detected".
    import torch  # noqa: F401
    lib = ctypes.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        msg = load().mtam_last_error()
        raise MtamHipError("%s failed (%d): %s" % (what, rc, msg.decode() if msg else ""))
