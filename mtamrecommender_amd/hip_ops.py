"""Torch-tensor front end of the C ABI: pointer/shape plumbing only.

Every function takes device tensors, checks dtype/contiguity, and calls the
matching ``mtam_*`` entry point on torch's current HIP stream.  PyTorch is used
for device memory and streams; all arithmetic happens in libmtam_hip.so.
"""
import ctypes

import torch

from . import _lib

EPI_STORE, EPI_BIAS, EPI_BIAS_RELU, EPI_RELU_ADD, EPI_ACCUM, EPI_ACCUM_MASK, EPI_ATOMIC, EPI_ACCUM2_MASK, \
    EPI_STORE_SQ = range(9)
D = 128


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _p(t, dtype=torch.float32):
    if t is None:
        return None
    if not t.is_cuda:
        raise _lib.MtamHipError("expected a device tensor")
    if t.dtype != dtype:
        raise _lib.MtamHipError("expected dtype %s, got %s" % (dtype, t.dtype))
    if not t.is_contiguous():
        raise _lib.MtamHipError("expected a contiguous tensor")
    return ctypes.c_void_p(t.data_ptr())


def _pi(t):
    return _p(t, torch.int32)


GEMM_SPLIT_BF16 = 0x100      # MTAM_GEMM_SPLIT_BF16 (include/mtam_hip.h)


def gemm(a, b, c, trans_a=False, trans_b=False, epilogue=EPI_STORE, bias=None, aux_in=None,
         aux_out=None, split_k=1, M=None, N=None, K=None, lda=None, ldb=None, ldc=None, ld_aux=None, split=True):
    """C = op(A) op(B) (+ epilogue).  2-D tensors; leading dimensions default to row lengths.
    Sub-matrix views are expressed with explicit M/N/K/ld* over a base tensor slice
    obtained by ``tensor.view(-1)[offset:]``.  ``split``: products on the bf16 matrix cores from operands split
    three ways (fp32-equivalent); the evaluation logits pass ``split=False`` (k-ordered fp32 fmaf chain)."""
    lib = _lib.load()
    if split:
        epilogue |= GEMM_SPLIT_BF16
    if lda is None:
        lda = a.shape[-1]
    if ldb is None:
        ldb = b.shape[-1]
    if ldc is None:
        ldc = c.shape[-1]
    if M is None:
        M = a.shape[1] if trans_a else a.shape[0]
    if K is None:
        K = a.shape[0] if trans_a else a.shape[1]
    if N is None:
        N = b.shape[0] if trans_b else b.shape[1]
    if ld_aux is None:
        ld_aux = aux_in.shape[-1] if aux_in is not None else 0
    rc = lib.mtam_gemm_f32(int(trans_a), int(trans_b), M, N, K, _p(a), lda, _p(b), ldb, _p(c), ldc,
                           epilogue, _p(bias), _p(aux_in), _p(aux_out), ld_aux, split_k, _stream())
    _lib.check(rc, "mtam_gemm_f32")


def gemm_dual(a, b, a2, b2, c, trans_b=True, epilogue=EPI_STORE, bias=None, aux_in=None, aux_out=None, split=True):
    """C = A op(B) + A2 op(B2) (+ epilogue); A [M,K], A2 [M,K2] row-major, B/B2 as in ``gemm``."""
    lib = _lib.load()
    if split:
        epilogue |= GEMM_SPLIT_BF16
    M, K, K2 = a.shape[0], a.shape[1], a2.shape[1]
    N = b.shape[0] if trans_b else b.shape[1]
    ld_aux = aux_in.shape[-1] if aux_in is not None else 0
    rc = lib.mtam_gemm_f32_dual(0, int(trans_b), M, N, K, _p(a), a.shape[-1], _p(b), b.shape[-1], K2, _p(a2),
                                a2.shape[-1], _p(b2), b2.shape[-1], _p(c), c.shape[-1], epilogue, _p(bias),
                                _p(aux_in), _p(aux_out), ld_aux, _stream())
    _lib.check(rc, "mtam_gemm_f32_dual")


def gemm_sq_partials(M, N):
    return _lib.load().mtam_gemm_sq_partials(M, N)


def gemm_batched(a, b, c, M, N, K, lda, sa, ldb, sb, ldc, sc, batch, trans_a=False, trans_b=False,
                 epilogue=EPI_STORE, split=True):
    """batch = (batch0, batch1); sa/sb/sc = (stride over batch0, stride over batch1) in elements.
    a, b, c: base tensors (any shape, contiguous storage).  ``split``: as in ``gemm``."""
    lib = _lib.load()
    if split:
        epilogue |= GEMM_SPLIT_BF16
    rc = lib.mtam_gemm_f32_batched(int(trans_a), int(trans_b), M, N, K, _p(a), lda, sa[0], sa[1], _p(b), ldb,
                                   sb[0], sb[1], _p(c), ldc, sc[0], sc[1], batch[0], batch[1], epilogue,
                                   _stream())
    _lib.check(rc, "mtam_gemm_f32_batched")


def colsum_atomic(x, out, rows=None, cols=None, ld=None):
    lib = _lib.load()
    rows = x.shape[0] if rows is None else rows
    cols = x.shape[1] if cols is None else cols
    ld = x.shape[-1] if ld is None else ld
    _lib.check(lib.mtam_colsum_atomic(_p(x), rows, cols, ld, _p(out), _stream()), "mtam_colsum_atomic")


def emb_gather_partials(B, L):
    return _lib.load().mtam_emb_gather_partials(B, L)


def emb_gather_fwd(item_table, cat_table, pos_table, user_table, item_ids, cat_ids, pos_ids, user_ids,
                   B, L, with_user, ic_out, pos_out, user_out, l2_partial, clear=(), item16=None):
    """clear: up to two flat float tensors zeroed by the same launch (the step's gradient accumulators).
    item16: bf16 image of the item table to read the item rows from (mixed precision)."""
    lib = _lib.load()
    ca = clear[0] if len(clear) > 0 else None
    cb = clear[1] if len(clear) > 1 else None
    rc = lib.mtam_emb_gather_fwd_item16(_p(item_table), _p(item16, torch.bfloat16) if item16 is not None else None,
                                        item_table.shape[0], _p(cat_table), cat_table.shape[0],
                                        _p(pos_table), pos_table.shape[0], _p(user_table), user_table.shape[0],
                                        _pi(item_ids), _pi(cat_ids), _pi(pos_ids), _pi(user_ids), B, L,
                                        int(with_user), _p(ic_out), _p(pos_out), _p(user_out), _p(l2_partial),
                                        _p(ca), ca.numel() if ca is not None else 0,
                                        _p(cb), cb.numel() if cb is not None else 0, _stream())
    _lib.check(rc, "mtam_emb_gather_fwd")


def emb_scatter_partials(B, L):
    return _lib.load().mtam_emb_scatter_partials(B, L)


def emb_scatter_add_bwd(d_ic, d_pos, ic, pos, user, item_ids, cat_ids, pos_ids, user_ids, seq_len, B, L,
                        reg, with_user, g_item, g_cat, g_pos, g_user, slot_sq_partial, pos_table=None, d_z=None,
                        W4=None, item_range=None, norm=None):
    """pos_table given (and pos None): the looked-up position rows were never written out; their L2 term reads
    the table through the ids (steps whose forward is seq_chain_gather_fwd).  d_z and W4 given (and d_ic None): the
    [item | category] gradient rows are computed inside the kernel, d_z . W4^T per 128-slot chunk.
    item_range = (lo, hi): only item slots with lo <= id < hi are added (a data-parallel rank's own item rows).
    norm = dict(g, n, partials, offset, lr, adam_state, l2_partial, ce, B, reg, ce_scale, loss): what
    sqnorm_state_loss does rides along as extra workgroups (mtam_emb_scatter_add_bwd_norm)."""
    lib = _lib.load()
    if norm is not None:
        assert item_range is None
        nr = _lib.NormRider()
        nr.g, nr.n, nr.partials, nr.offset = norm["g"].data_ptr(), int(norm["n"]), norm["partials"].data_ptr(), int(norm["offset"])
        nr.lr = norm["lr"].data_ptr() if norm.get("lr") is not None else None
        nr.adam_state = norm["adam_state"].data_ptr() if norm.get("adam_state") is not None else None
        l2, ce, loss = norm.get("l2_partial"), norm.get("ce"), norm.get("loss")
        nr.l2_partial, nr.n_l2 = (l2.data_ptr(), l2.numel()) if l2 is not None else (None, 0)
        nr.ce, nr.B = (ce.data_ptr(), int(norm["B"])) if ce is not None else (None, 0)
        nr.reg, nr.ce_scale = float(norm.get("reg", 0.0)), float(norm.get("ce_scale", 0.0))
        nr.loss = loss.data_ptr() if loss is not None else None
        rc = lib.mtam_emb_scatter_add_bwd_norm(
            _p(d_ic), _p(d_z), _p(W4), _p(d_pos), _p(ic), _p(pos), _p(pos_table), _p(user), _pi(item_ids),
            _pi(cat_ids), _pi(pos_ids), _pi(user_ids), _pi(seq_len), B, L, float(reg), int(with_user), _p(g_item),
            g_item.shape[0], _p(g_cat), g_cat.shape[0], _p(g_pos), g_pos.shape[0], _p(g_user), g_user.shape[0],
            _p(slot_sq_partial), ctypes.byref(nr), _stream())
        _lib.check(rc, "mtam_emb_scatter_add_bwd_norm")
        return
    if item_range is not None:
        rc = lib.mtam_emb_scatter_add_bwd_range(
            _p(d_ic), _p(d_z), _p(W4), _p(d_pos), _p(ic), _p(pos), _p(pos_table), _p(user), _pi(item_ids),
            _pi(cat_ids), _pi(pos_ids), _pi(user_ids), _pi(seq_len), B, L, float(reg), int(with_user), _p(g_item),
            g_item.shape[0], _p(g_cat), g_cat.shape[0], _p(g_pos), g_pos.shape[0], _p(g_user), g_user.shape[0],
            _p(slot_sq_partial), int(item_range[0]), int(item_range[1]), 0, _stream())
        _lib.check(rc, "mtam_emb_scatter_add_bwd_range")
        return
    if d_z is not None:
        rc = lib.mtam_emb_scatter_add_bwd_fused(
            _p(d_ic), _p(d_z), _p(W4), _p(d_pos), _p(ic), _p(pos), _p(pos_table), _p(user), _pi(item_ids),
            _pi(cat_ids), _pi(pos_ids), _pi(user_ids), _pi(seq_len), B, L, float(reg), int(with_user), _p(g_item),
            g_item.shape[0], _p(g_cat), g_cat.shape[0], _p(g_pos), g_pos.shape[0], _p(g_user), g_user.shape[0],
            _p(slot_sq_partial), _stream())
        _lib.check(rc, "mtam_emb_scatter_add_bwd_fused")
        return
    if pos_table is not None:
        rc = lib.mtam_emb_scatter_add_bwd_postab(
            _p(d_ic), _p(d_pos), _p(ic), _p(pos), _p(pos_table), _p(user), _pi(item_ids), _pi(cat_ids), _pi(pos_ids),
            _pi(user_ids), _pi(seq_len), B, L, float(reg), int(with_user), _p(g_item), g_item.shape[0], _p(g_cat),
            g_cat.shape[0], _p(g_pos), g_pos.shape[0], _p(g_user), g_user.shape[0], _p(slot_sq_partial), _stream())
        _lib.check(rc, "mtam_emb_scatter_add_bwd_postab")
        return
    rc = lib.mtam_emb_scatter_add_bwd(_p(d_ic), _p(d_pos), _p(ic), _p(pos), _p(user), _pi(item_ids),
                                      _pi(cat_ids), _pi(pos_ids), _pi(user_ids), _pi(seq_len), B, L,
                                      float(reg), int(with_user), _p(g_item), g_item.shape[0], _p(g_cat),
                                      g_cat.shape[0], _p(g_pos), g_pos.shape[0], _p(g_user),
                                      g_user.shape[0], _p(slot_sq_partial), _stream())
    _lib.check(rc, "mtam_emb_scatter_add_bwd")


def emb_scatter_add_items_range(d_ic, ic, item_ids, seq_len, B, L, reg, g_item, slot_sq_partial, item_range):
    """Another rank's slots (its all-gathered d[item | category] rows, looked-up rows, item ids, lengths) applied to
    THIS rank's item rows [lo, hi) of the item gradient -- only the item halves are read."""
    rc = _lib.load().mtam_emb_scatter_add_bwd_range(
        _p(d_ic), None, None, None, _p(ic), None, None, None, _pi(item_ids), None, None, None, _pi(seq_len), B, L,
        float(reg), 0, _p(g_item), g_item.shape[0], None, 1, None, 1, None, 1, _p(slot_sq_partial),
        int(item_range[0]), int(item_range[1]), 1, _stream())
    _lib.check(rc, "mtam_emb_scatter_add_bwd_range")


def rows_gather_range(table_rows, row0, ids, out):
    """out[i] = table_rows[ids[i] - row0] where the range [row0, row0 + len(table_rows)) holds catalog row ids[i],
    zeros elsewhere."""
    _lib.check(_lib.load().mtam_rows_gather_range(_p(table_rows), int(row0), table_rows.shape[0], _pi(ids), ids.numel(),
                                                  _p(out), _stream()), "mtam_rows_gather_range")


def tagru_fwd(xproj, x, timelast, seq_len, wh_g, wh_c, tvec, B, L, hs, short_out, save, kv=None, w_image=None):
    """kv = (Wkv operand images, bkv, kv_out [B L, n_kv]): the K/V projection relu(x Wkv + bkv) rides in the same
    launch as extra workgroups on the CUs the recurrence leaves idle.  w_image: the recurrent weights in the lanes'
    register order (``gru_weight_image``): loaded straight into registers instead of going through LDS."""
    lib = _lib.load()
    img, bkv, kv_out = kv if kv is not None else (None, None, None)
    rc = lib.mtam_tagru_fwd_kv(_p(xproj), _p(x), _p(timelast), _pi(seq_len), _p(wh_g), _p(wh_c), _p(tvec),
                               B, L, _p(hs), _p(short_out), _p(save), _pb(img) if img is not None else None, _p(bkv),
                               kv_out.shape[1] if kv_out is not None else 0, _p(kv_out), _p(w_image), _stream())
    _lib.check(rc, "mtam_tagru_fwd")


def gru_weight_image_floats():
    return _lib.load().mtam_gru_weight_image_floats()


def gru_weight_image(wh_g, wh_c, image):
    """wh_g [128, 256], wh_c [128, 128] -> the forward's register-order image (one launch)."""
    _lib.check(_lib.load().mtam_gru_weight_image(_p(wh_g), _p(wh_c), _p(image), _stream()), "mtam_gru_weight_image")


def tagru_bwd(d_short, x, timelast, seq_len, wh_g, wh_c, tvec, save, B, L, d_xproj, rh, d_x,
              d_tvec_partial, d_hs=None, dkv=None):
    """dkv = (d_kv [B L, 256], images of Wkv's transpose, d_x_keys [B L, 128]): d_x_keys += d_kv Wkv^T rides in the
    same launch as extra workgroups.  (``d_x`` here is the GRU's own time-gate path output, d_xt.)"""
    lib = _lib.load()
    d_kv, img_t, d_xk = dkv if dkv is not None else (None, None, None)
    rc = lib.mtam_tagru_bwd_dkv(_p(d_short), _p(d_hs), _p(x), _p(timelast), _pi(seq_len), _p(wh_g), _p(wh_c), _p(tvec),
                                _p(save), B, L, _p(d_xproj), _p(rh), _p(d_x), _p(d_tvec_partial), _p(d_kv),
                                d_kv.shape[1] if d_kv is not None else 0, _pb(img_t) if img_t is not None else None,
                                _p(d_xk), _stream())
    _lib.check(rc, "mtam_tagru_bwd")


def tagru_seqrec_fwd(xproj5, seq_len, wh_g, wh_c, B, L, hs, short_out, save6=None):
    lib = _lib.load()
    _lib.check(lib.mtam_tagru_seqrec_fwd(_p(xproj5), _pi(seq_len), _p(wh_g), _p(wh_c), B, L, _p(hs), _p(short_out),
                                         _p(save6), _stream()), "mtam_tagru_seqrec_fwd")


def tagru_seqrec_bwd(d_short, seq_len, wh_g, wh_c, save6, B, L, d_xproj5, rh, d_xt, d_tvec_partial, d_hs=None):
    lib = _lib.load()
    _lib.check(lib.mtam_tagru_seqrec_bwd(_p(d_short), _p(d_hs), _pi(seq_len), _p(wh_g), _p(wh_c), _p(save6), B, L,
                                         _p(d_xproj5), _p(rh), _p(d_xt), _p(d_tvec_partial), _stream()),
               "mtam_tagru_seqrec_bwd")


def tsr_time_inputs_fwd(timenow, timelast, tvec4, R, tin):
    lib = _lib.load()
    _lib.check(lib.mtam_tsr_time_inputs_fwd(_p(timenow), _p(timelast), _p(tvec4), R, _p(tin), _stream()),
               "mtam_tsr_time_inputs_fwd")


def tsr_time_inputs_bwd(d_tin, tin, timenow, timelast, R, out):
    lib = _lib.load()
    _lib.check(lib.mtam_tsr_time_inputs_bwd(_p(d_tin), _p(tin), _p(timenow), _p(timelast), R, _p(out), _stream()),
               "mtam_tsr_time_inputs_bwd")


def ta_attn_decode_save_floats(L, H):
    return _lib.load().mtam_ta_attn_decode_save_floats(L, H)


def ta_attn_decode_fwd(dec_in, x, kv, ld_kv, k_off, v_off, t_query, t_keys, seq_len, wqt, bq, tparams,
                       ln_beta, ln_gamma, B, L, H, dec_out, save, head=None):
    """head = (beta, gamma, pred_out, head_save): fuse the model's head layer_norm (last block)."""
    lib = _lib.load()
    hb, hg, po, hs = head if head is not None else (None, None, None, None)
    rc = lib.mtam_ta_attn_decode_fwd(_p(dec_in), _p(x), _p(kv), ld_kv, k_off, v_off, _p(t_query),
                                     _p(t_keys), _pi(seq_len), _p(wqt), _p(bq), _p(tparams), _p(ln_beta),
                                     _p(ln_gamma), B, L, H, _p(dec_out), _p(save), _p(hb), _p(hg), _p(po),
                                     _p(hs), _stream())
    _lib.check(rc, "mtam_ta_attn_decode_fwd")


def ta_attn_decode_bwd(d_out, dec_in, x, kv, ld_kv, k_off, v_off, t_query, t_keys, seq_len, wqt, tparams,
                       ln_gamma, save, B, L, H, accumulate_dx, d_dec_in, d_kv, d_x, d_qt_pre,
                       d_tparams_partial, d_ln_partial, head=None):
    """head = (d_pred, head_gamma, head_save, d_head_partial): backward of the fused head layer_norm
    (d_out may then be None)."""
    lib = _lib.load()
    dp, hg, hs, dhp = head if head is not None else (None, None, None, None)
    rc = lib.mtam_ta_attn_decode_bwd(_p(d_out), _p(dec_in), _p(x), _p(kv), ld_kv, k_off, v_off,
                                     _p(t_query), _p(t_keys), _pi(seq_len), _p(wqt), _p(tparams),
                                     _p(ln_gamma), _p(save), B, L, H, int(accumulate_dx), _p(d_dec_in),
                                     _p(d_kv), _p(d_x), _p(d_qt_pre), _p(d_tparams_partial),
                                     _p(d_ln_partial), _p(dp), _p(hg), _p(hs), _p(dhp), _stream())
    _lib.check(rc, "mtam_ta_attn_decode_bwd")


def layer_norm_fwd(x, beta, gamma, eps, rows, y, save, resid=None, form=0):
    lib = _lib.load()
    _lib.check(lib.mtam_layer_norm_fwd(_p(x), _p(resid), _p(beta), _p(gamma), float(eps), form, rows, _p(y),
                                       _p(save), _stream()), "mtam_layer_norm_fwd")


def ta_selfattn_gate_softmax_fwd(s_raw, a, t, seq_len, tparams, B, L, H, w, dk, sg):
    lib = _lib.load()
    _lib.check(lib.mtam_ta_selfattn_gate_softmax_fwd(_p(s_raw), _p(a), _p(t), _pi(seq_len), _p(tparams), B, L, H,
                                                     _p(w), _p(dk), _p(sg), _stream()),
               "mtam_ta_selfattn_gate_softmax_fwd")


def ta_selfattn_gate_softmax_bwd(dw, w, s_raw, a, dk, sg, t, seq_len, tparams, B, L, H, d_a, g_tparams):
    lib = _lib.load()
    _lib.check(lib.mtam_ta_selfattn_gate_softmax_bwd(_p(dw), _p(w), _p(s_raw), _p(a), _p(dk), _p(sg), _p(t),
                                                     _pi(seq_len), _p(tparams), B, L, H, _p(d_a), _p(g_tparams),
                                                     _stream()), "mtam_ta_selfattn_gate_softmax_bwd")


def seq_row_gather(src, seq_len, offset, B, L, out):
    lib = _lib.load()
    _lib.check(lib.mtam_seq_row_gather(_p(src), _pi(seq_len), offset, B, L, _p(out), _stream()),
               "mtam_seq_row_gather")


def seq_row_scatter(d_out, seq_len, offset, B, L, d_src):
    lib = _lib.load()
    _lib.check(lib.mtam_seq_row_scatter(_p(d_out), _pi(seq_len), offset, B, L, _p(d_src), _stream()),
               "mtam_seq_row_scatter")


def relu_bwd_inplace(d, y, n):
    lib = _lib.load()
    _lib.check(lib.mtam_relu_bwd_inplace(_p(d), _p(y), n, _stream()), "mtam_relu_bwd_inplace")


def layer_norm_bwd(d_y, gamma, save, rows, d_x, d_bg):
    lib = _lib.load()
    _lib.check(lib.mtam_layer_norm_bwd(_p(d_y), _p(gamma), _p(save), rows, _p(d_x), _p(d_bg), _stream()),
               "mtam_layer_norm_bwd")


def softmax_ce_partials(B, V):
    return _lib.load().mtam_softmax_ce_partials(B, V)


def softmax_ce(logits, ld, target, B, V, grad_scale, lse, ce, d_logits, partial):
    lib = _lib.load()
    rc = lib.mtam_softmax_ce(_p(logits), ld, _pi(target), B, V, float(grad_scale), _p(lse), _p(ce),
                             _p(d_logits), _p(partial), _stream())
    _lib.check(rc, "mtam_softmax_ce")


def loss_reduce(l2_partial, n_l2, ce, B, reg, ce_scale, loss):
    lib = _lib.load()
    _lib.check(lib.mtam_loss_reduce(_p(l2_partial), n_l2, _p(ce), B, float(reg), float(ce_scale),
                                    _p(loss), _stream()), "mtam_loss_reduce")


def softmax_ce_loss(logits, ld, target, B, V, grad_scale, lse, ce, d_logits, partial, l2_partial, n_l2, reg,
                    ce_scale, loss):
    lib = _lib.load()
    rc = lib.mtam_softmax_ce_loss(_p(logits), ld, _pi(target), B, V, float(grad_scale), _p(lse), _p(ce),
                                  _p(d_logits), _p(partial), _p(l2_partial), n_l2, float(reg), float(ce_scale),
                                  _p(loss), _stream())     # loss may be None
    _lib.check(rc, "mtam_softmax_ce_loss")


def topk_workspace_bytes(rows, V, k):
    return _lib.load().mtam_topk_workspace_bytes(rows, V, k)


def topk(scores, ld, rows, V, k, idx_out, val_out=None, workspace=None):
    """workspace: a float32 tensor of topk_workspace_bytes(rows, V, k) bytes enables the two-level form
    for long rows (same result)."""
    lib = _lib.load()
    _lib.check(lib.mtam_topk_ws(_p(scores), ld, rows, V, k, _pi(idx_out), _p(val_out), _p(workspace), _stream()),
               "mtam_topk")


TOPK_STREAM_SEG = 65536        # MTAM_TOPK_STREAM_SEG


def topk_stream_workspace_bytes(rows, V, k):
    return _lib.load().mtam_topk_stream_workspace_bytes(rows, V, k)


def topk_stream_slab(slab_scores, ld, rows, col0, width, V, k, workspace):
    """Candidates of columns [col0, col0 + width) of a [rows, V] score matrix that is never stored whole."""
    lib = _lib.load()
    _lib.check(lib.mtam_topk_stream_slab(_p(slab_scores), ld, rows, col0, width, V, k, _p(workspace), _stream()),
               "mtam_topk_stream_slab")


def topk_stream_finish(workspace, rows, V, k, idx_out, val_out=None):
    lib = _lib.load()
    _lib.check(lib.mtam_topk_stream_finish(_p(workspace), rows, V, k, _pi(idx_out), _p(val_out), _stream()),
               "mtam_topk_stream_finish")


# ---- bf16 scoring (csrc/score16.hip): bf16 bit patterns travel as torch.bfloat16 tensors
def _pb(t):
    return _p(t, torch.bfloat16)


def f32_to_bf16(src, dst):
    """dst (bf16, numel >= src.numel()) = round-to-nearest-even(src); the tail of dst is zero-filled."""
    lib = _lib.load()
    _lib.check(lib.mtam_f32_to_bf16(_p(src), src.numel(), _pb(dst), dst.numel(), _stream()), "mtam_f32_to_bf16")


def score16_batch_pad(B):
    return _lib.load().mtam_score16_batch_pad(B)


def score16_partials(B, V):
    return _lib.load().mtam_score16_partials(B, V)


def score16_sq_partials(V):
    return _lib.load().mtam_score16_sq_partials(V)


def score16_lse(E16, P16, target, B, V, partial, lse, ce):
    lib = _lib.load()
    _lib.check(lib.mtam_score16_lse(_pb(E16), _pb(P16), _pi(target), B, V, _p(partial), _p(lse), _p(ce), _stream()),
               "mtam_score16_lse")


def score16_bwd(E16, P16, lse, target, B, V, scale, d_pred, dE, sq_partial=None):
    lib = _lib.load()
    _lib.check(lib.mtam_score16_bwd(_pb(E16), _pb(P16), _p(lse), _pi(target), B, V, float(scale), _p(d_pred),
                                    _p(dE), _p(sq_partial), _stream()), "mtam_score16_bwd")


def seq_chain_fwd(ic, W4, pos, R, Wkv, bkv, Wx, bx, zr, x, kv, xproj, w_images=None):
    """zr, x, kv (optional: Wkv None), xproj from [item | category] rows in one launch.  ``w_images``: the bf16
    operand images of the three weight matrices (``WeightImageSet.buf``); given, the products run as split-bf16."""
    lib = _lib.load()
    n_kv = Wkv.shape[1] if Wkv is not None else 0
    _lib.check(lib.mtam_seq_chain_fwd(_p(ic), _p(W4), _p(pos), R, _p(Wkv), _p(bkv), n_kv, _p(Wx), _p(bx),
                                      Wx.shape[1], _p(zr), _p(x), _p(kv) if n_kv else None, _p(xproj),
                                      _pb(w_images) if w_images is not None else None, _stream()),
               "mtam_seq_chain_fwd")


def seq_chain_images_elems(n_kv, n_x):
    return int(_lib.load().mtam_seq_chain_images_elems(int(n_kv), int(n_x)))


def seq_chain_image_offset(which, n_x):
    """First element of matrix ``which`` (0: W4, 1: Wkv, 2: Wx) in the image buffer [W4 | Wx | Wkv]."""
    return int(_lib.load().mtam_seq_chain_image_offset(int(which), int(n_x)))


def split_weight_images(W, images):
    """W [K, N] fp32 (contiguous) -> its three bf16 operand images (one launch)."""
    K, N = W.shape
    _lib.check(_lib.load().mtam_split_weight_images(_p(W), K, N, _pb(images), _stream()), "mtam_split_weight_images")


def split_weight_rows(W, images_r):
    """W [K, N] fp32 (contiguous) -> the three bf16 images of its transpose, the operands of products with W^T."""
    K, N = W.shape
    _lib.check(_lib.load().mtam_split_weight_rows(_p(W), K, N, _pb(images_r), _stream()), "mtam_split_weight_rows")


def seq_chain_bwd_max_k():
    return _lib.load().mtam_seq_chain_bwd_max_k()


def seq_chain_bwd(d_xproj, d_kv, d_xt, zr, R, d_x, d_z, d_ic, w_images_r):
    """d_x += d_xproj Wx^T + d_kv Wkv^T + d_xt; d_z = d_x where zr > 0; d_ic = d_z W4^T -- one launch."""
    n_kv = d_kv.shape[1] if d_kv is not None else 0
    _lib.check(_lib.load().mtam_seq_chain_bwd(_p(d_xproj), d_xproj.shape[1], _p(d_kv), n_kv, _p(d_xt), _p(zr), R,
                                              _p(d_x), _p(d_z), _p(d_ic), _pb(w_images_r), _stream()),
               "mtam_seq_chain_bwd")


def weight_image_descs(entries):
    """[(first element in the flat space, K, N, bf16 image tensor[, image tensor of the transpose])] -> the ctypes array
    mtam_adam_images takes."""
    arr = (_lib.WeightImages * max(1, len(entries)))()
    for i, e in enumerate(entries):
        begin, K, N, img = e[:4]
        arr[i].begin, arr[i].K, arr[i].N, arr[i].images = int(begin), int(K), int(N), img.data_ptr()
        arr[i].images_r = e[4].data_ptr() if len(e) > 4 and e[4] is not None else None
        # a sixth element "gru_g" / "gru_c": W is the GRU's wh_g / wh_c, `img` the fp32 register-order image
        arr[i].gru_which = {"gru_g": 1, "gru_c": 2}[e[5]] if len(e) > 5 and e[5] else 0
    return arr, len(entries)


def adam_images(p, m, v, g, n, scale, hyper, sparse_begin, descs, copy16=None, copy_begin=0):
    """mtam_adam / mtam_adam_bf16copy that also re-writes the weight matrices' bf16 operand images."""
    arr, n_w = descs
    _lib.check(_lib.load().mtam_adam_images(_p(p), _p(m), _p(v), _p(g), n, _p(scale), _p(hyper), int(sparse_begin),
                                            _pb(copy16) if copy16 is not None else None, int(copy_begin),
                                            ctypes.cast(arr, ctypes.c_void_p), n_w, _stream()), "mtam_adam_images")


def adam_clip_max_partials():
    return _lib.load().mtam_adam_clip_max_partials()


def adam_images_clip(p, m, v, g, n, norm_partials, n_partials, clip_norm, scale_out, hyper, sparse_begin, descs,
                     copy16=None, copy_begin=0, feed=None):
    """adam_images where every workgroup derives the clip scale from the norm's partials itself (no ticket launch
    before it: sqnorm_state_loss writes the partials); scale_out <- (scale, norm).  ``descs``: None = no images.
    ``feed`` = (ring [slots, words] int32, arena [words] int32, cursor [1] int32): one more workgroup copies ring slot
    cursor % slots into the arena for the NEXT step and advances the cursor (mtam_adam_images_clip_feed)."""
    arr, n_w = descs if descs is not None else (None, 0)
    if feed is not None:
        ring, arena, cursor = feed
        assert ring.dtype == torch.int32 and arena.dtype == torch.int32 and cursor.dtype == torch.int32
        assert ring.dim() == 2 and ring.is_contiguous() and ring.shape[1] == arena.numel() and cursor.numel() == 1
        _lib.check(_lib.load().mtam_adam_images_clip_feed(
            _p(p), _p(m), _p(v), _p(g), n, _p(norm_partials), int(n_partials), float(clip_norm), _p(scale_out),
            _p(hyper), int(sparse_begin), _pb(copy16) if copy16 is not None else None, int(copy_begin),
            ctypes.cast(arr, ctypes.c_void_p) if arr is not None else None, n_w, _pi(ring), int(ring.shape[0]),
            int(ring.shape[1]), _pi(arena), _pi(cursor), _stream()), "mtam_adam_images_clip_feed")
        return
    _lib.check(_lib.load().mtam_adam_images_clip(_p(p), _p(m), _p(v), _p(g), n, _p(norm_partials), int(n_partials),
                                                 float(clip_norm), _p(scale_out), _p(hyper), int(sparse_begin),
                                                 _pb(copy16) if copy16 is not None else None, int(copy_begin),
                                                 ctypes.cast(arr, ctypes.c_void_p) if arr is not None else None, n_w,
                                                 _stream()), "mtam_adam_images_clip")


def sqnorm_state_loss(g, n, partials, offset, lr, adam_state, l2_partial=None, n_l2=0, ce=None, B=0, reg=0.0,
                      ce_scale=0.0, loss=None):
    """Partial sums of squares of g's blocks -> partials[offset ..]; one more workgroup advances the Adam state and
    reduces the reported loss.  No ticket, no scale: adam_images_clip forms the norm."""
    _lib.check(_lib.load().mtam_sqnorm_state_loss(_p(g), n, _p(partials), int(offset), _p(lr), _p(adam_state),
                                                  _p(l2_partial), n_l2, _p(ce), B, float(reg), float(ce_scale),
                                                  _p(loss), _stream()), "mtam_sqnorm_state_loss")


def seq_chain_gather_partials(B, L):
    return _lib.load().mtam_seq_chain_gather_partials(B, L)


def seq_chain_gather_fwd(item_table, cat_table, pos_table, user_table, item_ids, cat_ids, pos_ids, user_ids, B, L,
                         with_user, W4, Wkv, bkv, Wx, bx, ic_out, user_out, l2_partial, zr, x, kv, xproj, clear=(),
                         w_images=None):
    """The four embedding lookups + dense4emb + K/V projection + GRU input projection in one launch; ic_out None
    in evaluation.  ``clear``: up to two float tensors zeroed on the side."""
    lib = _lib.load()
    n_kv = Wkv.shape[1] if Wkv is not None else 0
    ca = clear[0] if len(clear) > 0 else None
    cb = clear[1] if len(clear) > 1 else None
    rc = lib.mtam_seq_chain_gather_fwd(
        _p(item_table), item_table.shape[0], _p(cat_table), cat_table.shape[0], _p(pos_table), pos_table.shape[0],
        _p(user_table), user_table.shape[0], _pi(item_ids), _pi(cat_ids), _pi(pos_ids), _pi(user_ids), B, L,
        int(with_user), _p(W4), _p(Wkv), _p(bkv), n_kv, _p(Wx), _p(bx), Wx.shape[1], _p(ic_out), _p(user_out),
        _p(l2_partial), l2_partial.numel(), _p(zr), _p(x), _p(kv) if n_kv else None, _p(xproj),
        _p(ca), ca.numel() if ca is not None else 0, _p(cb), cb.numel() if cb is not None else 0,
        _pb(w_images) if w_images is not None else None, _stream())
    _lib.check(rc, "mtam_seq_chain_gather_fwd")


def score32_set_split_min_rows(min_rows):
    """Catalogs of at least ``min_rows`` rows are scored by the split-bf16 kernels (0 = never); sizes follow."""
    _lib.load().mtam_score32_set_split_min_rows(int(min_rows))


def score32_partials(B, V):
    return _lib.load().mtam_score32_partials(B, V)


def score32_sq_partials(V):
    return _lib.load().mtam_score32_sq_partials(V)


def score32_lse(E, pred, target, B, V, partial, lse, ce, row0=None):
    """row0 given: E is a row RANGE of the catalog starting at catalog row row0 (V rows); ``target`` holds catalog
    row numbers; lse = log-sum-exp over the range, ce = the target's logit (0 when the target is not in the range)."""
    lib = _lib.load()
    _lib.check(lib.mtam_score32_lse_range(_p(E), _p(pred), _pi(target), B, V, -1 if row0 is None else int(row0),
                                          _p(partial), partial.numel(), _p(lse), _p(ce), _stream()), "mtam_score32_lse")


def score32_bwd(E, pred, lse, target, B, V, scale, d_pred, dE, sq_partial=None, n_sq=None, row0=None):
    lib = _lib.load()
    _lib.check(lib.mtam_score32_bwd_range(_p(E), _p(pred), _p(lse), _pi(target), B, V,
                                          -1 if row0 is None else int(row0), float(scale), _p(d_pred), _p(dE),
                                          _p(sq_partial), 0 if sq_partial is None else
                                          (sq_partial.numel() if n_sq is None else int(n_sq)), _stream()),
               "mtam_score32_bwd")


def score32_set_fused(on):
    """Run-time form of MTAM_SCORE32_FUSED (process-global): whether score32_train may run as one launch."""
    _lib.load().mtam_score32_set_fused(1 if on else 0)


def score32_train_is_fused(B, V):
    """Whether score32_train runs as ONE launch at this size (csrc/score32.hip, x3::train_small_kernel)."""
    return bool(_lib.load().mtam_score32_train_is_fused(int(B), int(V)))


def score32_train_work_floats(B, V):
    return int(_lib.load().mtam_score32_train_work_floats(int(B), int(V)))


def score32_train_work(B, V, device="cuda"):
    """The work buffer of score32_train for this (B, V), prepared (mtam_score32_train_work_init)."""
    work = torch.empty(score32_train_work_floats(B, V), dtype=torch.float32, device=device)
    _lib.check(_lib.load().mtam_score32_train_work_init(_p(work), work.numel(), int(B), int(V), _stream()),
               "mtam_score32_train_work_init")
    return work


def score32_train(E, pred, target, B, V, scale, work, lse, ce, d_pred, dE, sq_partial=None, n_sq=None):
    """Training's scoring in one call: ce = lse - target logit, d_pred += G E, dE = G^T pred.  ``work``:
    score32_train_work(B, V), then left to this call."""
    lib = _lib.load()
    _lib.check(lib.mtam_score32_train(_p(E), _p(pred), _pi(target), B, V, float(scale), _p(work), work.numel(),
                                      _p(lse), _p(ce), _p(d_pred), _p(dE), _p(sq_partial),
                                      0 if sq_partial is None else (sq_partial.numel() if n_sq is None else int(n_sq)),
                                      _stream()), "mtam_score32_train")


def score16_logits(E16, P16, B, V, logits, ld):
    lib = _lib.load()
    _lib.check(lib.mtam_score16_logits(_pb(E16), _pb(P16), B, V, _p(logits), ld, _stream()), "mtam_score16_logits")


def sqnorm_blocks(n):
    return _lib.load().mtam_sqnorm_blocks(n)


def sqnorm_partial(g, n, partial):
    lib = _lib.load()
    _lib.check(lib.mtam_sqnorm_partial(_p(g), n, _p(partial), _stream()), "mtam_sqnorm_partial")


def clip_scale(partials, n_partials, clip_norm, scale, lr=None, adam_state=None):
    lib = _lib.load()
    _lib.check(lib.mtam_clip_scale(_p(partials), n_partials, float(clip_norm), _p(scale), _p(lr),
                                   _p(adam_state), _stream()), "mtam_clip_scale")


def partials_sum(partials, n, weight, out, accumulate=False):
    """out[0] (+)= weight * sum(partials[:n]) in float64 (one rank's share of the squared gradient norm)."""
    lib = _lib.load()
    _lib.check(lib.mtam_partials_sum(_p(partials), n, float(weight), _p(out, torch.float64), int(accumulate),
                                     _stream()), "mtam_partials_sum")


def clip_scale_sq(sq_total, n, clip_norm, scale, lr=None, adam_state=None):
    lib = _lib.load()
    _lib.check(lib.mtam_clip_scale_sq(_p(sq_total, torch.float64), n, float(clip_norm), _p(scale), _p(lr),
                                      _p(adam_state), _stream()), "mtam_clip_scale_sq")


def sqnorm_clip_scale(g, n, partials, offset, n_total, clip_norm, scale, lr, adam_state, ticket,
                      l2_partial=None, n_l2=0, ce=None, B=0, reg=0.0, ce_scale=0.0, loss=None):
    lib = _lib.load()
    rc = lib.mtam_sqnorm_clip_scale(_p(g), n, _p(partials), offset, n_total, float(clip_norm), _p(scale), _p(lr),
                                    _p(adam_state), _pi(ticket), _p(l2_partial), n_l2, _p(ce), B, float(reg),
                                    float(ce_scale), _p(loss), _stream())
    _lib.check(rc, "mtam_sqnorm_clip_scale")


def adam_block():
    return _lib.load().mtam_adam_block()


def adam(p, m, v, g, n, scale, hyper, sparse_begin):
    lib = _lib.load()
    _lib.check(lib.mtam_adam(_p(p), _p(m), _p(v), _p(g), n, _p(scale), _p(hyper), int(sparse_begin),
                             _stream()), "mtam_adam")


def adam_bf16copy(p, m, v, g, n, scale, hyper, sparse_begin, copy16, copy_begin):
    lib = _lib.load()
    _lib.check(lib.mtam_adam_bf16copy(_p(p), _p(m), _p(v), _p(g), n, _p(scale), _p(hyper), int(sparse_begin),
                                      _pb(copy16), int(copy_begin), _stream()), "mtam_adam_bf16copy")


OPT_KINDS = {"sgd": 0, "adadelta": 1, "rmsprop": 2}


def opt_update(kind, p, slot1, slot2, g, n, scale, lr, sparse_begin, rowskip_end):
    lib = _lib.load()
    _lib.check(lib.mtam_opt_update(OPT_KINDS[kind], _p(p), _p(slot1), _p(slot2), _p(g), n, _p(scale), _p(lr),
                                   int(sparse_begin), int(rowskip_end), _stream()), "mtam_opt_update")


def _gemm_descs(problems):
    arr = (_lib.GemmDesc * len(problems))()
    for d, q in zip(arr, problems):
        d.A, d.B, d.C = _p(q["A"]).value, _p(q["B"]).value, _p(q["C"]).value
        d.lda, d.ldb, d.ldc = q["lda"], q["ldb"], q["ldc"]
        d.M, d.N, d.K, d.split_k = q["M"], q["N"], q["K"], q.get("split_k", 1)
    return arr


def _colsum_jobs(jobs):
    arr = (_lib.ColsumJob * len(jobs))()
    for d, (x, rows, cols, ld, out) in zip(arr, jobs):
        d.in_, d.rows, d.cols, d.ld, d.out = _p(x).value, rows, cols, ld, _p(out).value
    return arr


def gemm_tn_atomic_grouped(problems):
    """problems: list of dicts(A, lda, B, ldb, C, ldc, M, N, K, split_k) with device tensors."""
    lib = _lib.load()
    arr = _gemm_descs(problems)
    _lib.check(lib.mtam_gemm_tn_atomic_grouped(len(problems), ctypes.byref(arr), _stream()),
               "mtam_gemm_tn_atomic_grouped")


def colsum_atomic_multi(jobs):
    """jobs: list of (in_tensor, rows, cols, ld, out_tensor)."""
    lib = _lib.load()
    arr = _colsum_jobs(jobs)
    _lib.check(lib.mtam_colsum_atomic_multi(len(jobs), ctypes.byref(arr), _stream()),
               "mtam_colsum_atomic_multi")


def weight_grads(problems, jobs):
    """Both of the above in one launch."""
    lib = _lib.load()
    a, b = _gemm_descs(problems), _colsum_jobs(jobs)
    _lib.check(lib.mtam_weight_grads(len(problems), ctypes.byref(a), len(jobs), ctypes.byref(b), _stream()),
               "mtam_weight_grads")
