"""Device-side training/eval step of the time-aware SELF-attention model (PISTRec):
``Time_Aware_self_Attention_model.build_model`` (Model/PISTRec_model.py:38-74), i.e.
embedding -> NB ``self_attention`` encoder blocks (Model/Modules/time_aware_attention.py:459-490,
215-456 with T_q = T_k = L) -> gather at the mask-token position -> contrib layer_norm ->
full-catalog softmax, without the user-embedding L2 term.

Here the attention products are real L x L contractions, so they run on the matrix
cores as batched fp32-MFMA GEMMs (one problem per (sample, head)); the gate / mask / softmax rows run
in ``mtam_ta_selfattn_gate_softmax_*``.  Parameters, loss, clipping and Adam are shared with the
MTAM path (same flat parameter space).
"""
import os

import torch

from .. import hip_ops as ops
from .time_aware_path import D, TimeAwarePath, _Batch


class _SABatch(_Batch):

    def __init__(self, path, B):
        super(_SABatch, self).__init__(path, B)
        dev, L, NB, H = path.device, path.L, path.NB, path.H
        R = B * L
        # zero-filled ONCE: some buffers are written only where a sample is alive (e.g. the GRU's saved
        # state rows) and read whole by a GEMM whose other operand is zero there -- 0 x stale garbage must
        # not be 0 x NaN (seen as a NaN weight gradient when the allocator handed out recycled memory)
        f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        self.enc = [self.x] + [f(R, D) for _ in range(NB)]
        self.qkv = [f(R, 3 * D) for _ in range(NB)]
        self.qt = [f(R, D) for _ in range(NB)]
        self.s_raw = [f(B, H, L, L) for _ in range(NB)]
        self.a = [f(B, L, L) for _ in range(NB)]
        self.w = [f(B, H, L, L) for _ in range(NB)]
        self.dk = [f(B, L, L) for _ in range(NB)]
        self.sg = [f(B, L, L) for _ in range(NB)]
        self.o = f(R, D)
        self.enc_ln_save = [f(R, D + 1) for _ in range(NB)]
        self.long = f(B, D)
        # backward
        self.d_long = f(B, D)
        self.d_a, self.d_b = f(R, D), f(R, D)            # ping-pong: gradient w.r.t. a block's output / input
        self.d_w = f(B, H, L, L)
        self.d_amat = f(B, L, L)
        self.d_qkv = f(R, 3 * D)
        self.d_qt_sa = f(R, D)


class SelfAttentionPath(TimeAwarePath):
    MODEL = "PISTRec"
    BATCH_CLASS = _SABatch
    WITH_USER = 0        # 1: the loss is base_model.output() (user embedding in the L2 sum), see UserL2SelfAttentionPath

    # ----------------------------------------------------------------- forward
    def forward(self, bt, training=True, score=True):
        B, R, L, NB, H = bt.B, bt.R, self.L, self.NB, self.H
        d = D // H
        fd, T = bt.feed, self.tables
        ops.emb_gather_fwd(T["item"], T["category"], T["position"], T["user"], fd["item_list"],
                           fd["category_list"], fd["position_list"], fd["user_id"], B, L, self.WITH_USER,
                           bt.ic, bt.pos, bt.user, bt.l2_partial,
                           # a training step's first kernel also clears its gradient accumulators
                           clear=(self.zero_prefix, bt.d_pred.view(-1)) if training else (), item16=self.item16)
        ops.gemm(bt.ic, self.seg("dense4emb/w"), bt.x, epilogue=ops.EPI_RELU_ADD, aux_in=bt.pos, aux_out=bt.zr)
        for i in range(NB):
            enc, qkv, qt = bt.enc[i], bt.qkv[i], bt.qt[i]
            ops.gemm(enc, self.seg("blk%d/wqkv" % i), qkv, epilogue=ops.EPI_BIAS_RELU, bias=self.seg("blk%d/bqkv" % i))
            ops.gemm(enc, self.seg("blk%d/wt" % i), qt)
            # s_raw[b,h] = Q_bh K_bh^T  (L x L, K = d)
            ops.gemm_batched(qkv, qkv.view(-1)[D:], bt.s_raw[i], L, L, d, 3 * D, (L * 3 * D, d), 3 * D,
                             (L * 3 * D, d), L, (H * L * L, L * L), (B, H), trans_b=True)
            # a[b] = (enc Wt)_b enc_b^T  (L x L, K = D)
            ops.gemm_batched(qt, enc, bt.a[i], L, L, D, D, (L * D, 0), D, (L * D, 0), L, (L * L, 0), (B, 1),
                             trans_b=True)
            ops.ta_selfattn_gate_softmax_fwd(bt.s_raw[i], bt.a[i], fd["time_list"], fd["seq_length"],
                                             self.seg("blk%d/tparams" % i), B, L, H, bt.w[i], bt.dk[i], bt.sg[i])
            # o[b, :, h*d:(h+1)*d] = W_bh V_bh  (L x d, K = L)
            ops.gemm_batched(bt.w[i], qkv.view(-1)[2 * D:], bt.o, L, d, L, L, (H * L * L, L * L), 3 * D,
                             (L * 3 * D, d), D, (L * D, d), (B, H))
            ln = self.seg("blk%d/ln" % i)
            ops.layer_norm_fwd(bt.o, ln[0], ln[1], 1e-8, R, bt.enc[i + 1], bt.enc_ln_save[i] if training else None,
                               resid=enc, form=1)
        ops.seq_row_gather(bt.enc[NB], fd["seq_length"], -1, B, L, bt.long)
        hl = self.seg("head/ln")
        ops.layer_norm_fwd(bt.long, hl[0], hl[1], 1e-12, B, bt.pred, bt.ln_save if training else None)
        if score:
            self.score_forward(bt, training)

    # ---------------------------------------------------------------- backward
    def backward(self, bt):
        B, R, L, NB, H = bt.B, bt.R, self.L, self.NB, self.H
        d = D // H
        fd, T, G = bt.feed, self.tables, self.grads
        gseg = lambda name: self.layout.view(G, name)
        part = bt.norm_partial
        sr = max(1, min(int(os.environ.get("MTAM_WGRAD_SPLIT", "16")), R // 256))
        self.score_backward(bt)
        ops.layer_norm_bwd(bt.d_pred, self.seg("head/ln")[1], bt.ln_save, B, bt.d_long, gseg("head/ln"))
        d_out, d_in = bt.d_a, bt.d_b
        ops.seq_row_scatter(bt.d_long, fd["seq_length"], -1, B, L, d_out)
        for i in reversed(range(NB)):
            enc, qkv, qt = bt.enc[i], bt.qkv[i], bt.qt[i]
            ln = self.seg("blk%d/ln" % i)
            # normalize(o + enc): d_in = d(o + enc) -- both d_o and the residual part of d_enc
            ops.layer_norm_bwd(d_out, ln[1], bt.enc_ln_save[i], R, d_in, gseg("blk%d/ln" % i))
            # dW_bh = dO_bh V_bh^T ; dV_bh = W_bh^T dO_bh
            ops.gemm_batched(d_in, qkv.view(-1)[2 * D:], bt.d_w, L, L, d, D, (L * D, d), 3 * D, (L * 3 * D, d), L,
                             (H * L * L, L * L), (B, H), trans_b=True)
            ops.gemm_batched(bt.w[i], d_in, bt.d_qkv.view(-1)[2 * D:], L, d, L, L, (H * L * L, L * L), D, (L * D, d),
                             3 * D, (L * 3 * D, d), (B, H), trans_a=True)
            ops.ta_selfattn_gate_softmax_bwd(bt.d_w, bt.w[i], bt.s_raw[i], bt.a[i], bt.dk[i], bt.sg[i],
                                             fd["time_list"], fd["seq_length"], self.seg("blk%d/tparams" % i), B, L, H,
                                             bt.d_amat, gseg("blk%d/tparams" % i))
            # dQ_bh = dS_bh K_bh ; dK_bh = dS_bh^T Q_bh
            ops.gemm_batched(bt.d_w, qkv.view(-1)[D:], bt.d_qkv, L, d, L, L, (H * L * L, L * L), 3 * D, (L * 3 * D, d),
                             3 * D, (L * 3 * D, d), (B, H))
            ops.gemm_batched(bt.d_w, qkv, bt.d_qkv.view(-1)[D:], L, d, L, L, (H * L * L, L * L), 3 * D, (L * 3 * D, d),
                             3 * D, (L * 3 * D, d), (B, H), trans_a=True)
            # d(qt)_b = dA_b enc_b ; d_enc_b += dA_b^T qt_b
            ops.gemm_batched(bt.d_amat, enc, bt.d_qt_sa, L, D, L, L, (L * L, 0), D, (L * D, 0), D, (L * D, 0), (B, 1))
            ops.gemm_batched(bt.d_amat, qt, d_in, L, D, L, L, (L * L, 0), D, (L * D, 0), D, (L * D, 0), (B, 1),
                             trans_a=True, epilogue=ops.EPI_ACCUM)
            ops.relu_bwd_inplace(bt.d_qkv, qkv, bt.d_qkv.numel())
            ops.gemm(bt.d_qkv, self.seg("blk%d/wqkv" % i), d_in, trans_b=True, epilogue=ops.EPI_ACCUM)
            ops.gemm(bt.d_qt_sa, self.seg("blk%d/wt" % i), d_in, trans_b=True, epilogue=ops.EPI_ACCUM)
            ops.gemm_tn_atomic_grouped([
                dict(A=enc, lda=D, B=bt.d_qkv, ldb=3 * D, C=gseg("blk%d/wqkv" % i), ldc=3 * D, M=D, N=3 * D, K=R, split_k=sr),
                dict(A=enc, lda=D, B=bt.d_qt_sa, ldb=D, C=gseg("blk%d/wt" % i), ldc=D, M=D, N=D, K=R, split_k=sr)])
            ops.colsum_atomic(bt.d_qkv, gseg("blk%d/bqkv" % i))
            d_out, d_in = d_in, d_out
        # d_out now holds d loss / d x; dense4emb and the tables
        bt.d_z.copy_(d_out)
        ops.relu_bwd_inplace(bt.d_z, bt.zr, bt.d_z.numel())
        fused_scatter = os.environ.get("MTAM_FUSED_SCATTER", "0") == "1"      # see TimeAwarePath.backward
        if not fused_scatter:
            ops.gemm(bt.d_z, self.seg("dense4emb/w"), bt.d_ic, trans_b=True)
        ops.gemm_tn_atomic_grouped([dict(A=bt.ic, lda=2 * D, B=bt.d_z, ldb=D, C=gseg("dense4emb/w"), ldc=D,
                                         M=2 * D, N=D, K=R, split_k=sr)])
        slot_part = part[self.nb_dense + self.nb_item:]
        ops.emb_scatter_add_bwd(None if fused_scatter else bt.d_ic, d_out, bt.ic, bt.pos, bt.user, fd["item_list"],
                                fd["category_list"], fd["position_list"], fd["user_id"], fd["seq_length"], B, L,
                                self.reg, self.WITH_USER, self.g_tab["item"], self.g_tab["category"],
                                self.g_tab["position"], self.g_tab["user"], slot_part,
                                d_z=bt.d_z if fused_scatter else None,
                                W4=self.seg("dense4emb/w") if fused_scatter else None)


class UserL2SelfAttentionPath(SelfAttentionPath):
    """The same encoder under ``base_model.output()`` (Model/base_model.py:300-328): the L2 sum also takes the
    user embedding rows, as ``Time_Aware_Self_Attention_Model`` does (Model/attention_baseline_models.py:47-65)."""
    WITH_USER = 1
