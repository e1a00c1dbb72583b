"""The device-side training/eval step of the time-aware path.

What ``sess.run([loss, merged, train_op], feed)`` (Model/base_model.py:159-164)
and the eval ``sess.run`` (:201-202) execute in the reference becomes, here, a
fixed sequence of ``mtam_*`` launches over pre-allocated HBM buffers:

  forward   gather -> dense4emb GEMM -> x-projection GEMM -> time-aware GRU
            -> K/V GEMM -> NB decoder blocks -> head LN -> logits GEMM
            -> softmax CE                                   
  backward  the same chain reversed; every weight gradient in ONE grouped
            split-K GEMM launch, every bias-like gradient in ONE column-sum
            launch, embedding rows by atomic scatter-add             (K11)
  update    global-norm clip -> ONE Adam launch over the flat parameter space
            (dense variables and the four tables)                 (K12, K13)

All trainable state lives in one flat float32 space
``[dense | pad | category | position | user | item]`` (parameters, gradients,
Adam m and v share the layout).  Nothing is allocated inside a step and Adam's
per-step scalars are kept on the device, so a step is captured once into a
hipGraph (``torch.cuda.CUDAGraph``) and replayed; the only per-step input is
the feed arena.  Data-parallel training calls ``allreduce_fn`` between
backward and update (``data_parallel.py``).
"""
import os

import numpy as np
import torch

from .. import hip_ops as ops
from .param_layout import DenseLayout

D = 128
INT_FIELDS = ("user_id", "item_list", "category_list", "position_list", "target_item_id", "seq_length")
FLOAT_FIELDS = ("time_list", "timelast_list", "target_item_time", "timenow_list")
TABLES = ("category", "position", "user", "item")        # order inside the flat space (item last)
MAX_GROUP = 16


def _chunks(seq, n):
    for i in range(0, len(seq), n):
        yield seq[i:i + n]


def arena_layout(B, L):
    """Word offsets of the feed fields inside one int32 arena: name -> (offset, count, shape, dtype),
    plus the arena size.  Every field starts on a 16-byte boundary."""
    fields = [("user_id", (B,), torch.int32), ("item_list", (B, L), torch.int32),
              ("category_list", (B, L), torch.int32), ("position_list", (B, L), torch.int32),
              ("target_item_id", (B,), torch.int32), ("seq_length", (B,), torch.int32),
              ("time_list", (B, L), torch.float32), ("timelast_list", (B, L), torch.float32),
              ("target_item_time", (B,), torch.float32), ("lr", (4,), torch.float32),
              # read by the T-SeqRec cell only (Model/Modules/time_aware_rnn.py:75-76,113-116)
              ("timenow_list", (B, L), torch.float32)]
    offsets, o = {}, 0
    for name, shape, dt in fields:
        n = int(np.prod(shape))
        offsets[name] = (o, n, shape, dt)
        o += (n + 3) // 4 * 4
    return offsets, o


class _Batch(object):
    """Per-batch-size device buffers (activations, gradients, saved state)."""

    def __init__(self, path, B):
        dev, L, NB, H = path.device, path.L, path.NB, path.H
        R = B * L
        V = path.item_rows
        # zero-filled ONCE: some buffers are written only where a sample is alive (e.g. the GRU's saved
        # state rows) and read whole by a GEMM whose other operand is zero there -- 0 x stale garbage must
        # not be 0 x NaN (seen as a NaN weight gradient when the allocator handed out recycled memory)
        f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
        self.B, self.R = B, R
        # feed: ONE arena of 4-byte words at a fixed address (ids, times, learning rate), so that a
        # step needs one host->device (or device->device) copy and can be replayed from a hipGraph.
        self.offsets, o = arena_layout(B, L)
        self.arena = torch.zeros(o, dtype=torch.int32, device=dev)
        # two pinned host copies used in turn, each guarded by an event recorded behind its host -> device copy:
        # with an asynchronous loss read-back the host fills batch t + 1 while the copy of batch t may still be
        # in flight
        self._host_ring = [torch.zeros(o, dtype=torch.int32).pin_memory() for _ in range(2)]
        self._host_events = [None, None]
        self._host_slot = 0
        self.host_arena = self._host_ring[0]
        self.feed = {name: self._view(self.arena, name) for name in self.offsets}
        self.host = {name: self._view(self.host_arena, name) for name in self.offsets}
        # forward activations
        self.ic, self.pos, self.user = f(R, 2 * D), f(R, D), f(B, D)
        self.zr, self.x = f(R, D), f(R, D)
        # x-projection width and saved-state slots: 5 D / 6 for the T-SeqRec cell (two hoisted time gates)
        self.xw = xw = 5 if path.cfg["gru"] == "seqrec" else 3
        self.nsave = nsave = 6 if path.cfg["gru"] == "seqrec" else 5
        self.xproj, self.hs, self.short = f(R, xw * D), f(R, D), f(B, D)
        self.gru_save = f(R, nsave * D)
        if path.cfg["gru"] == "seqrec":
            self.tin, self.d_tin, self.d_tvec4_rows = f(R, 2 * D), f(R, 2 * D), f(R, 4 * D)
        self.kv = f(R, 2 * NB * D)
        # decoder input: the short-term intent, layer-normed first in the via_* members
        self.short_n, self.short_ln_save = f(B, D), f(B, D + 1)
        self.dec = [self.short_n if path.cfg["short_ln"] else self.short] + [f(B, D) for _ in range(NB)]
        self.attn_save = [f(B, ops.ta_attn_decode_save_floats(L, H)) for _ in range(NB)]
        self.pred, self.ln_save = f(B, D), f(B, D + 1)
        # pred = what the catalog is scored with; ln_out = the head layer_norm's output.  The same buffer,
        # except under the output_concat head (MTAM_hybird), where pred = [short | ln_out] . output_w
        self.concat_head = path.cfg.get("head") == "concat"
        self.ln_out = f(B, D) if self.concat_head else self.pred
        # logits rows start on 16-byte boundaries (row stride = V rounded up to 4 floats) so that the
        # transposed read of d_logits in the item-gradient GEMM takes the vector-load path; `logits` is the
        # [B, V] view of that storage
        # Allocated on first use (the `logits_store` property): only the stored-logits fp32 training path and
        # the stored form of evaluation touch it -- 5.1 GB at 128 x 10 M, 25.6 GB at 128 x 50 M, 410 GB at the
        # preset test batch of 2,048 x 50 M, which logits-free training and slab-wise evaluation never need
        self.ld_logits = (V + 3) // 4 * 4
        self._path_V, self._dev, self._logits_store = V, dev, None
        self.eval_slab = None        # [B, slab] score scratch of the slab-wise evaluation
        self.lse, self.ce = f(B), f(B)
        if path.logits_free32:
            self.s32_partial = f(ops.score32_partials(B, V))
        if path.score_dtype == "bf16":
            self.pred16 = torch.zeros((ops.score16_batch_pad(B), D), dtype=torch.bfloat16, device=dev)
            self.s16_partial = f(ops.score16_partials(B, V))
        self.ce_partial = torch.zeros(ops.softmax_ce_partials(B, V) + 4, dtype=torch.float32, device=dev)
        self.l2_partial = f(ops.emb_gather_partials(B, L))
        # the fused lookups write 4 sums per 32-row stripe (800 at 6,400 rows, against the gather kernel's 4,832): the
        # step's last reduction reads one batch of loads instead of three
        self.l2_fused = self.l2_partial[:ops.seq_chain_gather_partials(B, L)]
        self.l2_live = self.l2_partial          # (what the last forward wrote: set by it)
        self.loss = f(3)
        # backward
        self.d_dec = [f(B, D) for _ in range(NB + 1)]
        self.d_kv = f(R, 2 * NB * D)
        self.d_x, self.d_z, self.d_xt = f(R, D), f(R, D), f(R, D)
        self.d_qt = [f(B, 2 * D) for _ in range(NB)]
        self.d_tp_partial = [f(B, 5 * L) for _ in range(NB)]
        self.d_ln_partial = [f(B, 2 * D) for _ in range(NB)]
        self.d_xproj, self.rh = f(R, xw * D), f(R, D)
        self.d_tvec_partial = f(B, 8 * D)
        self.d_ic = f(R, 2 * D)
        # [d_pred | d_x]: cleared together by the step's first kernel when the decoder's key gradient goes
        # to the GRU outputs (d_hs) and d_x only receives the GRU's input-path gradient
        self.d_clear = f(B * D + R * D)
        self.d_pred = self.d_clear[:B * D].view(B, D)
        self.d_ln_out = f(B, D) if self.concat_head else self.d_pred
        if path.cfg["keys"] == "gru":
            self.d_x = self.d_clear[B * D:].view(R, D)
            self.d_hs = f(R, D)
        self.d_short = f(B, D)
        self.d_head_partial = f(B, 2 * D)
        self.n_slot = ops.emb_scatter_partials(B, L)
        self.norm_partial = torch.zeros(path.nb_all + self.n_slot, dtype=torch.float32, device=dev)
        self.topk_idx = torch.zeros((B, 50), dtype=torch.int32, device=dev)
        self.topk_ws = None         # candidate scratch of the two-level top-K (long rows only)

    @property
    def logits_store(self):
        if self._logits_store is None:
            self._logits_store = torch.zeros((self.B, self.ld_logits), dtype=torch.float32, device=self._dev)
        return self._logits_store

    @property
    def logits(self):
        return self.logits_store[:, :self._path_V]

    def next_host_arena(self):
        """Switch to the other pinned arena (waiting, if need be, for the copy that last read it)."""
        self._host_slot = 1 - self._host_slot
        ev = self._host_events[self._host_slot]
        if ev is not None:
            ev.synchronize()
        self.host_arena = self._host_ring[self._host_slot]
        self.host = {name: self._view(self.host_arena, name) for name in self.offsets}

    def upload(self):
        """host arena -> device arena (asynchronous; the event guards the host buffer's next use)."""
        self.arena.copy_(self.host_arena, non_blocking=True)
        if self._host_events[self._host_slot] is None:
            self._host_events[self._host_slot] = torch.cuda.Event()
        self._host_events[self._host_slot].record()

    def _view(self, arena, name):
        o, n, shape, dt = self.offsets[name]
        v = arena[o:o + n]
        return (v.view(torch.float32) if dt == torch.float32 else v).view(*shape)


class FeedRing(object):
    """Packed feeds resident in HBM, consumed by the captured training step WITHOUT a copy in front of it.

    ``slots`` [n, words] int32: n packed feed arenas (``arena_layout``: ids, times, learning rate) -- a whole epoch of
    them (ml-1m's 6,040 training sequences are 48 arenas = 6 MB; a million sequences 1 GB of the 288), or a short
    ring a loader keeps ahead of the step.  The optimizer launch of step k carries one more workgroup that copies
    slot (cursor % n) into ``bt.arena`` -- the fixed address every kernel of the step reads -- and advances the
    cursor (``mtam_adam_images_clip_feed``): the feed of step k + 1 arrives in the shadow of step k's update, and a
    step is ONE graph launch with nothing in front of it (the reference hands every ``sess.run`` its feed_dict,
    Model/base_model.py:150-164; a device-to-device copy of the arena ahead of the graph cost 4.9 us of a 226 us
    step).  ``prime()`` puts the first slot into the arena by an ordinary copy; after it, ``consumed`` counts the
    steps launched and slot ``(consumed + 1) % n`` must be complete before the NEXT step is launched."""

    _serials = 0

    def __init__(self, bt, n_slots):
        self.bt, self.n = bt, int(n_slots)
        FeedRing._serials += 1
        self.serial = FeedRing._serials        # key of the graphs captured on this ring's addresses (never re-used, unlike id())
        self.words = bt.arena.numel()
        self.slots = torch.zeros((self.n, self.words), dtype=torch.int32, device=bt.arena.device)
        self.cursor = torch.zeros(1, dtype=torch.int32, device=bt.arena.device)
        self.consumed = 0
        self.primed = False
        self.taken = False          # the arena was handed to someone else's feed since the last prime()
        self.gate = None            # (first slot still in flight, event behind its copy): base_model.load_resident_epoch

    def put(self, slot, arena):
        """A packed arena (device or pinned host tensor of ``words`` int32) -> slot ``slot`` (stream-ordered)."""
        self.slots[slot % self.n].copy_(arena, non_blocking=True)

    def prime(self, first_slot=0):
        """slot ``first_slot`` -> arena, cursor -> the slot after it: the next step consumes ``first_slot``."""
        self.bt.arena.copy_(self.slots[first_slot % self.n])
        self.cursor.fill_(first_slot + 1)
        self.consumed = first_slot
        self.primed, self.taken = True, False

    def args(self):
        return (self.slots, self.bt.arena, self.cursor)


class TimeAwarePath(object):
    """Owns parameters, optimizer state and the kernel sequence for MTAM."""

    MODEL = "MTAM"
    BATCH_CLASS = None       # set below

    def __init__(self, tables, dense_tf, L, num_heads, num_blocks, regulation_rate, max_gradient_norm,
                 tf_compat_global_norm=True, device="cuda:0", optimizer="adam", variant=None,
                 score_dtype="f32"):
        self.device = dev = torch.device(device)
        self.L, self.H, self.NB = L, num_heads, num_blocks
        self.reg, self.clip = float(regulation_rate), float(max_gradient_norm)
        self.tf_compat = bool(tf_compat_global_norm)
        if optimizer not in ("adam", "sgd", "adadelta", "rmsprop"):
            raise ValueError("unknown optimizer %r" % (optimizer,))
        self.optimizer = optimizer
        if score_dtype not in ("f32", "bf16"):
            raise ValueError("score_dtype must be 'f32' or 'bf16' (got %r)" % (score_dtype,))
        # "bf16" (BASELINE.json configs[4]): the forward reads the item table from a bf16 copy -- history
        # gathers widen its rows to fp32, full-catalog scoring runs on bf16 MFMA with fp32 accumulation and
        # no stored logits (csrc/score16.hip); fp32 master weights, gradients and optimizer slots, fp32
        # activations on the sequence side, fp32 atomics in the scatter-add
        self.score_dtype = score_dtype
        from .variables import MTAM_VARIANTS
        if variant is not None:
            self.MODEL = variant
        self.cfg = MTAM_VARIANTS.get(self.MODEL, dict(gru=None, keys="x", short_ln=False, attention=True))
        self.layout = DenseLayout(self.MODEL, D, L, num_blocks)
        for k, v in tables.items():
            if v.shape[1] != D:
                raise ValueError("this build supports num_units == %d only (table %s has %d)" % (D, k, v.shape[1]))
        # ---- flat parameter space: [dense | pad to the Adam block | category | position | user | item]
        P = self.layout.total
        blk = ops.adam_block()
        self.n_dense = (P + blk - 1) // blk * blk
        self.tab_off, o = {}, self.n_dense
        for k in TABLES:
            self.tab_off[k] = o
            o += int(tables[k].size)
        self.n_total = o
        self.item_rows = tables["item"].shape[0]
        # the item region is allocated with its row count rounded up to a multiple of 8, so that the
        # data-parallel exchange can cut it into 1, 2, 4 or 8 equal row ranges (data_parallel.ShardedItemExchange);
        # the pad rows are zero, never looked up, scored or updated (every kernel gets the true counts)
        self.item_rows_pad = (self.item_rows + 7) // 8 * 8
        # ... plus one more row at the very end: under the flat data-parallel exchange the first floats of it in the
        # GRADIENT buffer carry the rank's loss terms through the same all-reduce as the gradients (loss_tail)
        self.n_items_end = o + (self.item_rows_pad - self.item_rows) * D
        self.n_alloc = self.n_items_end + D
        z = lambda n: torch.zeros(n, dtype=torch.float32, device=dev)
        self.flat_p, self.flat_g, self.flat_m, self.flat_v = (z(self.n_alloc), z(self.n_alloc), z(self.n_alloc),
                                                              z(self.n_alloc))
        self.params = self.flat_p[:P]
        self.grads = self.flat_g[:P]
        self.m, self.v = self.flat_m[:P], self.flat_v[:P]
        tview = lambda flat, k: flat[self.tab_off[k]:self.tab_off[k] + tables[k].size].view(*tables[k].shape)
        self.tables = {k: tview(self.flat_p, k) for k in TABLES}
        self.g_tab = {k: tview(self.flat_g, k) for k in TABLES}
        self.tm = {k: tview(self.flat_m, k) for k in TABLES}
        self.tv = {k: tview(self.flat_v, k) for k in TABLES}
        self.params.copy_(torch.from_numpy(self.layout.pack(dense_tf)))
        for k in TABLES:
            self.tables[k].copy_(torch.from_numpy(np.ascontiguousarray(tables[k], dtype=np.float32)))
        # everything before the item gradient is zeroed per step (the item gradient is overwritten
        # by the dense scoring GEMM)
        if optimizer == "rmsprop":
            self.flat_m.fill_(1.0)      # the "rms" slot starts at one [TF1.14 RMSPropOptimizer._create_slots]
        self.zero_prefix = self.flat_g[:self.tab_off["item"]]
        self.nb_dense = ops.sqnorm_blocks(self.n_dense)
        # the dense item gradient's squared norm comes out of its GEMM's epilogue (one partial per wave)
        self.nb_item = ops.gemm_sq_partials(self.item_rows, D)
        # fp32 TRAINING scores the catalog without stored logits at every size (csrc/score32.hip: the log-sum-exp pass,
        # then one pass that recomputes a slab's scores and forms both scoring gradients; SURVEY.md K9 asks for it
        # from 1 M rows).  Per step at B=128 against logits GEMM + softmax-CE + two gradient GEMMs: 3,709 rows 0.2722
        # vs 0.2740 ms, 8,000 rows 0.2812 vs 0.2830, 30,000 rows 0.3235 vs 0.3275, 60,000 rows 0.3669 vs 0.3828.
        # MTAM_TRAIN_STORED_LOGITS=1 keeps the four-launch form (evaluation always scores through the GEMM: its
        # k-ordered fmaf chain is the ranking contract)
        self.logits_free32 = score_dtype == "f32" and os.environ.get("MTAM_TRAIN_STORED_LOGITS", "0") != "1"
        if self.logits_free32:
            self.nb_item = ops.score32_sq_partials(self.item_rows)
        self.item16 = None
        if score_dtype == "bf16":
            self.nb_item = ops.score16_sq_partials(self.item_rows)
            self.item16 = torch.empty((self.item_rows, D), dtype=torch.bfloat16, device=dev)
            self.refresh_item16()
        self._init_weight_images()
        self.refresh_weight_images()
        self.nb_all = max(ops.sqnorm_blocks(self.n_total), self.nb_dense + self.nb_item)
        self.scale = z(2)
        self.ticket = torch.zeros(4, dtype=torch.int32, device=dev)
        # Adam state on the device: [lr_t, beta1, beta2, eps, beta1_power, beta2_power, -, -]
        self.adam_state = torch.tensor([0.0, 0.9, 0.999, 1e-8, 0.9, 0.999, 0.0, 0.0], dtype=torch.float32,
                                       device=dev)
        self._batches = {}
        # (Independent branches of the step were tried on side streams -- K/V projection next to the GRU,
        # weight-gradient GEMMs next to the serial chain: 388 us/step vs 366 us in one stream at B=128; the
        # cross-queue graph edges cost more than the overlap won, so the step is single-stream.)
        self.loss_tail = self.flat_g[self.n_items_end:self.n_items_end + 4]
        self.loss_in_tail = False       # data_parallel.attach(): the reported loss is the all-reduced tail
        self.allreduce_fn = None        # set by data_parallel.attach()
        self.sharded = None             # data_parallel.ShardedItemExchange: replaces allreduce_fn + clip_and_apply
        self.sharded_scoring = None     # data_parallel.ShardedScoringExchange: scoring itself is row-sharded
        self.dp_exchange = None
        self.world_size = 1             # the loss is a mean over world_size * B samples ...
        self.global_batch = None        # ... or over exactly this many when the ranks' batches differ in size

    # ----------------------------------------------------------------- helpers
    def gb(self, bt):
        """The number of samples the cross-entropy is a mean over: the GLOBAL batch (Model/base_model.py:322 is a
        reduce_mean over the batch; data-parallel ranks hold slices of it)."""
        return int(self.global_batch) if self.global_batch else bt.B * self.world_size

    def seg(self, name, flat=None):
        return self.layout.view(self.params if flat is None else flat, name)

    def feed_ring(self, bt, n_slots):
        """Attach a ring of ``n_slots`` HBM-resident packed feeds to ``bt`` (see FeedRing); ``bt.feed_ring = None``
        detaches it.  Adam steps on one GPU or under the flat data-parallel exchange (the optimizer launch behind the
        all-reduce carries the copy just the same); the other optimizers have no launch that could carry it."""
        if not self.ring_supported(bt):
            raise RuntimeError("feed ring: Adam steps whose optimizer launch forms the clip scale itself (one GPU, or "
                               "the flat data-parallel exchange); not the row-sharded exchanges, not the other "
                               "optimizers")
        bt.feed_ring = FeedRing(bt, n_slots)
        return bt.feed_ring

    def ring_supported(self, bt):
        """Can this batch's optimizer launch carry the next step's feed (mtam_adam_images_clip_feed)?  It is the
        launch clip_and_apply uses when the norm has few enough partials -- on one GPU and, under the flat
        data-parallel exchange, behind the all-reduce.  The row-sharded exchanges update through their own launches."""
        if self.optimizer != "adam" or self.sharded is not None or self.sharded_scoring is not None:
            return False
        n = self.nb_dense + self.nb_item + bt.n_slot if self.tf_compat else ops.sqnorm_blocks(self.n_total)
        return self._clip_in_adam(n)

    def batch(self, B):
        if B not in self._batches:
            self._batches[B] = (self.BATCH_CLASS or _Batch)(self, B)
            if self.loss_in_tail:
                self._batches[B].loss = self.loss_tail[:3]
        return self._batches[B]

    def fill_host(self, bt, feed, lr=None):
        for k in INT_FIELDS + FLOAT_FIELDS:
            h = bt.host[k]
            h.copy_(torch.from_numpy(np.ascontiguousarray(feed[k])).to(h.dtype).view(h.shape))
        bt.host["lr"][0] = float(np.float32(lr)) if lr is not None else 0.0   # f64 placeholder cast to f32

    def load_feed(self, feed, lr=None):
        """Host feed arrays (Embedding.make_feed_dic_new) [+ learning rate] -> the fixed
        device arena, one pinned host->device copy."""
        bt = self.batch(len(feed["user_id"]))
        bt.next_host_arena()
        self.fill_host(bt, feed, lr)
        bt.upload()
        return bt

    def stage(self, feed, lr):
        """A device-resident copy of one step's arena (bench: inputs already in HBM)."""
        bt = self.batch(len(feed["user_id"]))
        self.fill_host(bt, feed, lr)
        return bt.host_arena.to(self.device, non_blocking=False)

    def refresh_item16(self):
        """bf16 scoring copy of the item table <- fp32 master (after every update of the table)."""
        if self.item16 is not None:
            ops.f32_to_bf16(self.tables["item"].view(-1), self.item16.view(-1))

    def _init_weight_images(self):
        """bf16 operand images of dense4emb/w, kv/w and gru/wx for the forward's fused projection kernel
        (csrc/seq_chain.hip, split-bf16 products): one buffer [W4 | Wkv | Wx], re-written by the Adam launch that
        updates the weights (``mtam_adam_images``) and by ``refresh_weight_images`` whenever they change any other
        way.  MTAM_SEQ_CHAIN_X3=0 keeps the products on the fp32 MFMA."""
        self.wimg, self.wimg_r, self.wimg_descs, self._wimg_parts, self.gru_img = None, None, None, [], None
        segs = self.layout.segments
        if "gru/wx" not in segs or self.cfg["gru"] == "seqrec" or os.environ.get("MTAM_SEQ_CHAIN_X3", "1") == "0":
            return
        kv_from_x = self.cfg["attention"] and self.cfg["keys"] == "x"
        n_kv = segs["kv/w"].shape[1] if kv_from_x else 0
        n_x = segs["gru/wx"].shape[1]
        n_img = ops.seq_chain_images_elems(n_kv, n_x)
        # ... and the images of the same three matrices' TRANSPOSES behind them, same order and offsets: the B operands
        # of the backward's stripe kernel (mtam_seq_chain_bwd multiplies by W^T)
        both = torch.zeros(2 * n_img, dtype=torch.bfloat16, device=self.device)
        self.wimg, self.wimg_r = both[:n_img], both[n_img:]
        self.wimg_n_kv = n_kv
        names = ["dense4emb/w"] + (["kv/w"] if kv_from_x else []) + ["gru/wx"]
        which = {"dense4emb/w": 0, "kv/w": 1, "gru/wx": 2}
        for name in names:
            K, N = segs[name].shape
            o = ops.seq_chain_image_offset(which[name], n_x)
            self._wimg_parts.append((name, segs[name].offset, K, N, self.wimg[o:], self.wimg_r[o:]))
        entries = [part[1:] for part in self._wimg_parts]
        # ... and the GRU's recurrent weights in the order its forward's lanes hold them (csrc/gru_image.h): loaded
        # straight into registers, the ~5 us LDS route of every forward launch gone (MTAM_GRU_WEIGHT_IMAGE=0 keeps it)
        if self.cfg["gru"] in ("time", "plain") and os.environ.get("MTAM_GRU_WEIGHT_IMAGE", "1") != "0":
            self.gru_img = torch.zeros(ops.gru_weight_image_floats(), dtype=torch.float32, device=self.device)
            for name, kind in (("gru/wh_g", "gru_g"), ("gru/wh_c", "gru_c")):
                K, N = segs[name].shape
                entries.append((segs[name].offset, K, N, self.gru_img, None, kind))
        self.wimg_descs = ops.weight_image_descs(entries)

    def _wimg_of(self, name):
        """(operand images, images of the transpose) of one weight matrix."""
        for part in self._wimg_parts:
            if part[0] == name:
                return part[4], part[5]
        raise KeyError(name)

    def _stripe_bwd_on(self, bt):
        """The backward's sequence-side chain as one stripe kernel (csrc/seq_chain.hip): needs the transposes' images and
        n_x + n_kv within the staged stripe; MTAM_SEQ_CHAIN_BWD=0 / MTAM_FUSED_SCATTER=1 keep the GEMM launches."""
        kv_src = self.cfg["attention"] and self.cfg["keys"] == "x"
        return (self.wimg_r is not None and os.environ.get("MTAM_FUSED_SCATTER", "0") != "1"
                and (self.wimg_n_kv > 0) == bool(kv_src)
                and bt.xw * D + self.wimg_n_kv <= ops.seq_chain_bwd_max_k()
                and os.environ.get("MTAM_SEQ_CHAIN_BWD", "1") != "0")

    def _kv_role_on(self):
        """kv = relu(x Wkv + bkv) as extra workgroups of the GRU's forward launch (csrc/tagru.hip: kv_role) instead of
        inside the fused projection kernel; MTAM_KV_ROLES=0 keeps it there."""
        return (self.cfg["attention"] and self.cfg["keys"] == "x" and self.wimg is not None
                and self.cfg["gru"] in ("time", "plain") and os.environ.get("MTAM_KV_ROLES", "1") != "0")

    def _dkv_role_on(self, bt):
        """d_x += d_kv . Wkv^T as extra workgroups of the GRU's backward launch (csrc/tagru.hip: dkv_role; one decoder
        block, keys = x, the stripe kernel behind it); MTAM_KV_ROLES=0 leaves the term in the stripe kernel."""
        return (self._stripe_bwd_on(bt) and self.wimg_n_kv == 2 * D and self.cfg["gru"] in ("time", "plain")
                and os.environ.get("MTAM_KV_ROLES", "1") != "0")

    def refresh_weight_images(self):
        for name, _, _, _, img, img_r in self._wimg_parts:
            ops.split_weight_images(self.seg(name), img)
            ops.split_weight_rows(self.seg(name), img_r)
        if self.gru_img is not None:
            ops.gru_weight_image(self.seg("gru/wh_g"), self.seg("gru/wh_c"), self.gru_img)

    def refresh_derived(self):
        """Everything kept beside the fp32 parameters and derived from them: call after ANY change of
        ``flat_p`` that did not come from this path's own optimizer launch."""
        self.refresh_item16()
        self.refresh_weight_images()

    def _score_fused_on(self, bt):
        """Training's two scoring passes as ONE launch (csrc/score32.hip, x3::train_small_kernel): when the step runs
        forward and backward back to back (``bt.train_step``: nothing reads the loss terms in between) and the catalog
        is small enough for every slab to have a resident workgroup.  The cross entropies then come out of the
        BACKWARD's first launch."""
        if not (self.logits_free32 and getattr(bt, "train_step", False) and self.sharded_scoring is None):
            return False
        if getattr(bt, "s32_work", None) is None:
            if not ops.score32_train_is_fused(bt.B, self.item_rows):
                bt.s32_work = False
            else:
                bt.s32_work = ops.score32_train_work(bt.B, self.item_rows, self.device)
        return bt.s32_work is not False

    # ----------------------------------------------------------------- scoring (base_model.output)
    def score_forward(self, bt, training):
        """logits = pred . E^T (Model/base_model.py:309-312).  bf16 training keeps no logits: it goes
        straight to lse / cross entropy."""
        V = self.item_rows
        if self.logits_free32 and training:
            bt.score_in_backward = self._score_fused_on(bt)
            if not bt.score_in_backward:
                ops.score32_lse(self.tables["item"], bt.pred, bt.feed["target_item_id"], bt.B, V, bt.s32_partial,
                                bt.lse, bt.ce)
            return
        if self.score_dtype == "f32":
            ops.gemm(bt.pred, self.tables["item"], bt.logits_store, trans_b=True, split=False)
            return
        ops.f32_to_bf16(bt.pred.view(-1), bt.pred16.view(-1))
        if training:
            ops.score16_lse(self.item16, bt.pred16, bt.feed["target_item_id"], bt.B, V, bt.s16_partial, bt.lse, bt.ce)
        else:
            ops.score16_logits(self.item16, bt.pred16, bt.B, V, bt.logits_store, bt.ld_logits)

    def score_backward(self, bt):
        """dE (every row; its share of the TF global norm on the way out) and d_pred (accumulated)."""
        part, V = bt.norm_partial, self.item_rows
        sq = part[self.nb_dense:] if self.tf_compat else None
        if self.logits_free32 and getattr(bt, "score_in_backward", False):
            # small catalogs: the log-sum-exp pass, the loss terms and both gradients as ONE launch
            ops.score32_train(self.tables["item"], bt.pred, bt.feed["target_item_id"], bt.B, V, 1.0 / self.gb(bt),
                              bt.s32_work, bt.lse, bt.ce, bt.d_pred, self.g_tab["item"], sq, n_sq=self.nb_item)
            return
        if self.logits_free32:
            ops.score32_bwd(self.tables["item"], bt.pred, bt.lse, bt.feed["target_item_id"], bt.B, V,
                            1.0 / self.gb(bt), bt.d_pred, self.g_tab["item"], sq, n_sq=self.nb_item)
            return
        if self.score_dtype == "bf16":
            gb = self.gb(bt)
            ops.score16_bwd(self.item16, bt.pred16, bt.lse, bt.feed["target_item_id"], bt.B, V, 1.0 / gb,
                            bt.d_pred, self.g_tab["item"], sq)
            return
        # dense item gradient dE = G^T pred (every row) and its share of the TF global norm
        if self.tf_compat:
            ops.gemm(bt.logits_store, bt.pred, self.g_tab["item"], trans_a=True, epilogue=ops.EPI_STORE_SQ,
                     aux_out=sq, M=V)
        else:
            ops.gemm(bt.logits_store, bt.pred, self.g_tab["item"], trans_a=True, M=V)
        # d_pred = G E: split-K over the catalog: ~64 slices at ml-1m sizes; for large catalogs enough
        # slices (<= 1024) that 2,000+ workgroups stream the table (64 slices left one workgroup per CU:
        # 43 TFLOP/s at V = 1 M)
        split_v = max(1, min(64, (V + 127) // 128), min(1024, V // 2048))
        ops.gemm(bt.logits_store, self.tables["item"], bt.d_pred, epilogue=ops.EPI_ATOMIC, split_k=split_v, K=V)

    # ----------------------------------------------------------------- forward
    def forward(self, bt, training=True, score=True):
        """Model/MTAMRec_model.py:40-238 (the member is chosen by ``self.cfg``).  score=False stops at
        predict_behavior_emb (slab-wise evaluation scores the catalog itself)."""
        B, R, L, NB, H = bt.B, bt.R, self.L, self.NB, self.H
        fd, T, cfg = bt.feed, self.tables, self.cfg
        item_table, item_ids = T["item"], fd["item_list"]
        if training and self.sharded_scoring is not None and not self.sharded_scoring.replicate_table:
            # "sharded-table" data parallelism: this rank's copy of the item rows it does not own is stale; the rows of
            # the batch's history ids were fetched from their owners (ShardedScoringExchange.fetch_history_rows) and
            # are looked up by position
            item_table, item_ids = bt.item_rows, self.row_iota(bt)
        # a training step's first kernel also clears its gradient accumulators
        clear = (self.zero_prefix, bt.d_clear if cfg["keys"] == "gru" else bt.d_pred.view(-1)) if training else ()
        keys = bt.hs if cfg["keys"] == "gru" else bt.x        # user_history: what the decoder attends over
        kv_from_x = cfg["attention"] and cfg["keys"] == "x"    # keys/values of every block (before the GRU)
        # ... computed by extra workgroups of the GRU launch, on the CUs the recurrence leaves idle, instead of by the
        # fused projection kernel (csrc/tagru.hip: kv_role; MTAM_KV_ROLES=0 keeps it in the projection kernel)
        kv_role = self._kv_role_on()
        kv_in_chain = kv_from_x and not kv_role
        # dense4emb, the K/V projection and the GRU's input projection in ONE launch (a 32-row stripe of x
        # stays on its CU; 16-byte stores): 25.5 us against 35.4 us as three GEMMs at 6,400 rows.
        # MTAM_SEQ_CHAIN=0 keeps the three GEMMs.
        chain = cfg["gru"] != "seqrec" and os.environ.get("MTAM_SEQ_CHAIN", "1") != "0"
        # ... and the four embedding lookups folded into the same launch (fp32 item rows): the looked-up rows go
        # straight into the first product's LDS operand; the position rows are never written, the [item | category]
        # rows only in training (dense4emb's weight gradient reads them).  MTAM_FUSED_GATHER=0 keeps the gather kernel.
        bt.fused_gather = chain and self.item16 is None and os.environ.get("MTAM_FUSED_GATHER", "1") != "0"
        if bt.fused_gather:
            ops.seq_chain_gather_fwd(item_table, T["category"], T["position"], T["user"], item_ids,
                                     fd["category_list"], fd["position_list"], fd["user_id"], B, L, 1,
                                     self.seg("dense4emb/w"),
                                     self.seg("kv/w") if kv_in_chain else None, self.seg("kv/b") if kv_in_chain else None,
                                     self.seg("gru/wx"), self.seg("gru/bx"), bt.ic if training else None, bt.user,
                                     bt.l2_fused, bt.zr, bt.x, bt.kv if kv_in_chain else None, bt.xproj, clear=clear,
                                     w_images=self.wimg)
            bt.l2_live = bt.l2_fused          # 4 sums per 32-row stripe: what the loss reduction has to read
        else:
            ops.emb_gather_fwd(item_table, T["category"], T["position"], T["user"], item_ids,
                               fd["category_list"], fd["position_list"], fd["user_id"], B, L, 1,
                               bt.ic, bt.pos, bt.user, bt.l2_partial, clear=clear, item16=self.item16)
            bt.l2_live = bt.l2_partial
        if bt.fused_gather:
            pass
        elif chain:
            ops.seq_chain_fwd(bt.ic, self.seg("dense4emb/w"), bt.pos, R,
                              self.seg("kv/w") if kv_in_chain else None, self.seg("kv/b") if kv_in_chain else None,
                              self.seg("gru/wx"), self.seg("gru/bx"), bt.zr, bt.x, bt.kv if kv_in_chain else None,
                              bt.xproj, w_images=self.wimg)
        else:
            ops.gemm(bt.ic, self.seg("dense4emb/w"), bt.x, epilogue=ops.EPI_RELU_ADD, aux_in=bt.pos, aux_out=bt.zr)
            if kv_in_chain:
                ops.gemm(bt.x, self.seg("kv/w"), bt.kv, epilogue=ops.EPI_BIAS_RELU, bias=self.seg("kv/b"))
        if cfg["gru"] == "seqrec":
            # TimeAwareGRUCell_sigmoid: gate and candidate input halves (columns 0 .. 3 D) and the two
            # state-independent time gates  x Wk + tanh(t w + b) Wt + bias  (columns 3 D .. 5 D) in one buffer
            wx, bx, wt = self.seg("gru/wx"), self.seg("gru/bx"), self.seg("gru/tsr_wt")
            ops.tsr_time_inputs_fwd(fd["timenow_list"], fd["timelast_list"], self.seg("gru/tsr_tvec"), R, bt.tin)
            ops.gemm(bt.x, wx, bt.xproj, epilogue=ops.EPI_BIAS, bias=bx, N=3 * D)
            for j in range(2):
                ops.gemm(bt.x, wx.view(-1)[(3 + j) * D:], bt.xproj.view(-1)[(3 + j) * D:], epilogue=ops.EPI_BIAS,
                         bias=bx[(3 + j) * D:], N=D, K=D, ldb=5 * D, ldc=5 * D)
                ops.gemm(bt.tin.view(-1)[j * D:], wt[j], bt.xproj.view(-1)[(3 + j) * D:], epilogue=ops.EPI_ACCUM,
                         M=R, N=D, K=D, lda=2 * D, ldc=5 * D)
            ops.tagru_seqrec_fwd(bt.xproj, fd["seq_length"], self.seg("gru/wh_g"), self.seg("gru/wh_c"), B, L,
                                 bt.hs, bt.short, bt.gru_save if training else None)
        else:
            if not chain:
                ops.gemm(bt.x, self.seg("gru/wx"), bt.xproj, epilogue=ops.EPI_BIAS, bias=self.seg("gru/bx"))
            tvec = self.seg("gru/tvec") if cfg["gru"] == "time" else None
            kv_job = (self._wimg_of("kv/w")[0], self.seg("kv/b"), bt.kv) if kv_role else None
            ops.tagru_fwd(bt.xproj, bt.x, fd["timelast_list"], fd["seq_length"], self.seg("gru/wh_g"),
                          self.seg("gru/wh_c"), tvec, B, L, bt.hs, bt.short, bt.gru_save if training else None,
                          kv=kv_job, w_image=self.gru_img)
        if cfg["short_ln"]:
            sl = self.seg("short/ln")
            ops.layer_norm_fwd(bt.short, sl[0], sl[1], 1e-12, B, bt.short_n, bt.short_ln_save if training else None)
        if cfg["attention"] and cfg["keys"] == "gru":
            ops.gemm(bt.hs, self.seg("kv/w"), bt.kv, epilogue=ops.EPI_BIAS_RELU, bias=self.seg("kv/b"))
        hl = self.seg("head/ln")
        if cfg["attention"]:
            for i in range(NB):
                ln = self.seg("blk%d/ln" % i)
                # the last block also applies the head layer_norm (pred) in the same launch
                head = (hl[0], hl[1], bt.ln_out, bt.ln_save if training else None) if i == NB - 1 else None
                ops.ta_attn_decode_fwd(bt.dec[i], keys, bt.kv, 2 * NB * D, 2 * i * D, (2 * i + 1) * D,
                                       fd["target_item_time"], fd["time_list"], fd["seq_length"],
                                       self.seg("blk%d/wqt" % i), self.seg("blk%d/bq" % i),
                                       self.seg("blk%d/tparams" % i), ln[0], ln[1], B, L, H, bt.dec[i + 1],
                                       bt.attn_save[i] if training else None, head=head)
        else:
            ops.layer_norm_fwd(bt.short, hl[0], hl[1], 1e-12, B, bt.ln_out, bt.ln_save if training else None)
        if bt.concat_head:
            W = self.seg("head/output_w")
            ops.gemm_dual(bt.short, W[:D], bt.ln_out, W[D:], bt.pred, trans_b=False)
        if score:
            self.score_forward(bt, training)

    def loss_and_logit_grad(self, bt):
        if self.score_dtype == "bf16" or self.logits_free32:
            return              # lse and cross entropy came out of score_forward; G is formed inside score_backward
        B, V = bt.B, self.item_rows
        gb = self.gb(bt)
        # logits -> lse, ce; then d_logits in place
        # the loss scalar itself is reduced in the step epilogue (clip_and_apply), off the chain
        ops.softmax_ce_loss(bt.logits_store, bt.ld_logits, bt.feed["target_item_id"], B, V, 1.0 / gb, bt.lse, bt.ce,
                            bt.logits_store,
                            bt.ce_partial, bt.l2_live, bt.l2_live.numel(), self.reg, 1.0 / gb, None)

    # ---------------------------------------------------------------- backward
    def backward(self, bt, score=True):
        """score=False: d_pred is already there (data-parallel row-sharded scoring formed it and the rank's own rows
        of the item gradient); the scatter-add then touches this rank's item rows only."""
        B, R, L, NB, H = bt.B, bt.R, self.L, self.NB, self.H
        fd, T, G, cfg = bt.feed, self.tables, self.grads, self.cfg
        gseg = lambda name: self.layout.view(G, name)
        part = bt.norm_partial
        # split-K slices of the grouped weight-gradient launch: 12 at 6,400 rows = 480 workgroups, one resident wave of
        # them; measured per step 8: 0.2748, 10: 0.2697, 12: 0.2685, 14: 0.2708, 16: 0.2715, 24: 0.2779 ms
        sr = max(1, min(int(os.environ.get("MTAM_WGRAD_SPLIT", "12")), R // 256))
        prob = lambda A, lda, Bm, ldb, name, M, N, K, s: dict(A=A, lda=lda, B=Bm, ldb=ldb, C=gseg(name),
                                                              ldc=N, M=M, N=N, K=K, split_k=s)
        if score:
            self.score_backward(bt)      # d_pred -> head LN -> decoder blocks (last to first)
        if bt.concat_head:               # back through output_w: d(ln_out) now, d(short) after the decoder
            W = self.seg("head/output_w")
            ops.gemm(bt.d_pred, W[D:], bt.d_ln_out, trans_b=True)
        keys = bt.hs if cfg["keys"] == "gru" else bt.x
        d_keys = bt.d_hs if cfg["keys"] == "gru" else bt.d_x      # gradient of user_history
        problems, jobs = [], []
        if cfg["attention"]:
            for i in reversed(range(NB)):
                ln = self.seg("blk%d/ln" % i)
                # the last block starts from d_pred: backward of the fused head layer_norm
                head = (bt.d_ln_out, self.seg("head/ln")[1], bt.ln_save, bt.d_head_partial) if i == NB - 1 else None
                ops.ta_attn_decode_bwd(None if head else bt.d_dec[i + 1], bt.dec[i], keys, bt.kv, 2 * NB * D,
                                       2 * i * D, (2 * i + 1) * D, fd["target_item_time"], fd["time_list"],
                                       fd["seq_length"], self.seg("blk%d/wqt" % i), self.seg("blk%d/tparams" % i),
                                       ln[1], bt.attn_save[i], B, L, H, 0 if i == NB - 1 else 1,
                                       bt.d_dec[i], bt.d_kv, d_keys, bt.d_qt[i], bt.d_tp_partial[i],
                                       bt.d_ln_partial[i], head=head)
            jobs.append((bt.d_head_partial, B, 2 * D, 2 * D, gseg("head/ln").view(-1)))
            problems += [prob(keys, D, bt.d_kv, 2 * NB * D, "kv/w", D, 2 * NB * D, R, sr)] + \
                [prob(bt.dec[i], D, bt.d_qt[i], 2 * D, "blk%d/wqt" % i, D, 2 * D, B, 1) for i in range(NB)]
            jobs += [(bt.d_kv, R, 2 * NB * D, 2 * NB * D, gseg("kv/b"))]
            for i in range(NB):
                jobs += [(bt.d_qt[i], B, D, 2 * D, gseg("blk%d/bq" % i)),
                         (bt.d_tp_partial[i], B, 5 * L, 5 * L, gseg("blk%d/tparams" % i).view(-1)),
                         (bt.d_ln_partial[i], B, 2 * D, 2 * D, gseg("blk%d/ln" % i).view(-1))]
            # d(user_history) += d_kv . Wkv^T: needed before the GRU's backward when the keys are its outputs;
            # with x as keys it rides along in the d_x GEMM below (second source)
            if cfg["keys"] == "gru":
                ops.gemm(bt.d_kv, self.seg("kv/w"), d_keys, trans_b=True, epilogue=ops.EPI_ACCUM)
            d_short = bt.d_dec[0]
        else:
            ops.layer_norm_bwd(bt.d_ln_out, self.seg("head/ln")[1], bt.ln_save, B, bt.d_short, gseg("head/ln"))
            d_short = bt.d_short
        if bt.concat_head:
            W, gW = self.seg("head/output_w"), gseg("head/output_w")
            ops.gemm(bt.d_pred, W[:D], d_short, trans_b=True, epilogue=ops.EPI_ACCUM)
            problems += [dict(A=bt.short, lda=D, B=bt.d_pred, ldb=D, C=gW[:D], ldc=D, M=D, N=D, K=B, split_k=1),
                         dict(A=bt.ln_out, lda=D, B=bt.d_pred, ldb=D, C=gW[D:], ldc=D, M=D, N=D, K=B, split_k=1)]
        if cfg["short_ln"]:
            ops.layer_norm_bwd(d_short, self.seg("short/ln")[1], bt.short_ln_save, B, bt.d_short, gseg("short/ln"))
            d_short = bt.d_short
        # GRU back through time (its time-gate path goes to d_xt); with the GRU outputs as keys their
        # gradient enters every step
        xw, ns = bt.xw * D, bt.nsave * D
        if cfg["gru"] == "seqrec":
            ops.tagru_seqrec_bwd(d_short, fd["seq_length"], self.seg("gru/wh_g"), self.seg("gru/wh_c"), bt.gru_save,
                                 B, L, bt.d_xproj, bt.rh, bt.d_xt, bt.d_tvec_partial,
                                 d_hs=bt.d_hs if cfg["keys"] == "gru" else None)
            # back through the two time kernels and the tanh time inputs
            wt, g_wt = self.seg("gru/tsr_wt"), gseg("gru/tsr_wt")
            for j in range(2):
                ops.gemm(bt.d_xproj.view(-1)[(3 + j) * D:], wt[j], bt.d_tin.view(-1)[j * D:], trans_b=True,
                         M=R, N=D, K=D, lda=5 * D, ldc=2 * D)
                problems.append(dict(A=bt.tin.view(-1)[j * D:], lda=2 * D, B=bt.d_xproj.view(-1)[(3 + j) * D:],
                                     ldb=5 * D, C=g_wt[j], ldc=D, M=D, N=D, K=R, split_k=sr))
            ops.tsr_time_inputs_bwd(bt.d_tin, bt.tin, fd["timenow_list"], fd["timelast_list"], R, bt.d_tvec4_rows)
            jobs.append((bt.d_tvec4_rows, R, 4 * D, 4 * D, gseg("gru/tsr_tvec").view(-1)))
        else:
            tvec = self.seg("gru/tvec") if cfg["gru"] == "time" else None
            # d_x += d_kv . Wkv^T (the K/V projection's gradient towards x) rides with the GRU's backward launch as
            # extra workgroups; the stripe kernel below then runs without its d_kv source
            dkv_role = self._dkv_role_on(bt)
            ops.tagru_bwd(d_short, bt.x, fd["timelast_list"], fd["seq_length"], self.seg("gru/wh_g"),
                          self.seg("gru/wh_c"), tvec, bt.gru_save, B, L, bt.d_xproj, bt.rh, bt.d_xt,
                          bt.d_tvec_partial, d_hs=bt.d_hs if cfg["keys"] == "gru" else None,
                          dkv=(bt.d_kv, self._wimg_of("kv/w")[1], bt.d_x) if dkv_role else None)
        problems = [prob(bt.x, D, bt.d_xproj, xw, "gru/wx", D, xw, R, sr),
                    prob(bt.gru_save.view(-1)[4 * D:], ns, bt.d_xproj, xw, "gru/wh_g", D, 2 * D, R, sr),
                    prob(bt.rh, D, bt.d_xproj.view(-1)[2 * D:], xw, "gru/wh_c", D, D, R, sr)] + problems
        jobs = [(bt.d_xproj, R, xw, xw, gseg("gru/bx"))] + jobs
        if cfg["gru"] == "time":
            jobs.append((bt.d_tvec_partial, B, 8 * D, 8 * D, gseg("gru/tvec").view(-1)))
        # d_x (+)= d_xproj . Wx^T + d_xt, d_z = d_x where relu(z) > 0; then d[item|cat].  d_x already holds the
        # decoder's key gradient when the keys are x; otherwise it was cleared by the step's first kernel
        # (members without a decoder leave it untouched: cleared here)
        if not cfg["attention"]:
            bt.d_x.zero_()
        # MTAM_FUSED_SCATTER=1: d[item | category] = d_z . W4^T is formed inside the scatter-add, chunk by chunk (no
        # [R, 2D] round trip, one launch less).  Measured at B=128, L=50: 0.2859 ms per step against 0.2781 with the
        # separate 9 us GEMM -- a chunk's 64 dependent fp32 MFMAs and two operand round trips lengthen the 100
        # item / category workgroups of a launch that is latency-bound already -- so the GEMM stays the default
        fused_scatter = os.environ.get("MTAM_FUSED_SCATTER", "0") == "1"
        kv_src = cfg["attention"] and cfg["keys"] == "x"
        # the whole chain in ONE stripe kernel (d_z never leaves the CU between the two products; split-bf16 products
        # on the images of the weights' transposes): MTAM_SEQ_CHAIN_BWD=0 keeps the two GEMM launches
        stripe = self._stripe_bwd_on(bt)
        if stripe:
            in_stripe = kv_src and not self._dkv_role_on(bt)
            ops.seq_chain_bwd(bt.d_xproj, bt.d_kv if in_stripe else None, bt.d_xt, bt.zr, R, bt.d_x, bt.d_z, bt.d_ic,
                              self.wimg_r)
        elif kv_src:
            ops.gemm_dual(bt.d_xproj, self.seg("gru/wx"), bt.d_kv, self.seg("kv/w"), bt.d_x, trans_b=True,
                          epilogue=ops.EPI_ACCUM2_MASK, bias=bt.d_xt, aux_in=bt.zr, aux_out=bt.d_z)
        else:
            ops.gemm(bt.d_xproj, self.seg("gru/wx"), bt.d_x, trans_b=True, epilogue=ops.EPI_ACCUM2_MASK,
                     bias=bt.d_xt, aux_in=bt.zr, aux_out=bt.d_z)
        if not fused_scatter and not stripe:
            ops.gemm(bt.d_z, self.seg("dense4emb/w"), bt.d_ic, trans_b=True)
        problems.append(prob(bt.ic, 2 * D, bt.d_z, D, "dense4emb/w", 2 * D, D, R, sr))
        # every weight gradient and every bias-like gradient in ONE launch (more when a group overflows)
        pc, jc = list(_chunks(problems, MAX_GROUP)), list(_chunks(jobs, MAX_GROUP))
        ops.weight_grads(pc[0], jc[0])
        for chunk in pc[1:]:
            ops.gemm_tn_atomic_grouped(chunk)
        for chunk in jc[1:]:
            ops.colsum_atomic_multi(chunk)
        # tables: sparse rows on top of the dense item gradient
        slot_part = part[self.nb_dense + self.nb_item:]
        fused = getattr(bt, "fused_gather", False)         # the position rows were not written out: read by id
        ops.emb_scatter_add_bwd(None if fused_scatter else bt.d_ic, bt.d_x, bt.ic, None if fused else bt.pos, bt.user,
                                fd["item_list"], fd["category_list"], fd["position_list"], fd["user_id"],
                                fd["seq_length"], B, L, self.reg, 1, self.g_tab["item"], self.g_tab["category"],
                                self.g_tab["position"], self.g_tab["user"], slot_part,
                                pos_table=T["position"] if fused else None,
                                d_z=bt.d_z if fused_scatter else None,
                                W4=self.seg("dense4emb/w") if fused_scatter else None,
                                item_range=(self.sharded_scoring.row_lo, self.sharded_scoring.row_hi)
                                if self.sharded_scoring is not None else None,
                                norm=self._norm_rider(bt))

    # ------------------------------------------------------------------ update
    def _clip_in_adam(self, n_partials):
        """The clip scale formed inside the optimizer launch (no ticket launch before it)."""
        return self.optimizer == "adam" and n_partials <= ops.adam_clip_max_partials() and \
            os.environ.get("MTAM_CLIP_IN_ADAM", "1") != "0"

    def _norm_rider(self, bt):
        """Single-GPU training steps: the clip's partial pass over the dense gradient (+ Adam state, + reported loss)
        rides in the scatter-add launch, the last launch of the backward -- the weight gradients are complete by
        then and nothing is exchanged between backward and update.  None: clip_and_apply launches it."""
        bt.norm_rode = False
        if not (getattr(bt, "train_step_bwd", False) and self.tf_compat and self.allreduce_fn is None and
                self.sharded is None and self.sharded_scoring is None and not self.loss_in_tail and
                self._clip_in_adam(self.nb_dense + self.nb_item + bt.n_slot) and
                os.environ.get("MTAM_NORM_RIDER", "1") != "0"):
            return None
        bt.norm_rode = True
        return dict(g=self.flat_g, n=self.n_dense, partials=bt.norm_partial, offset=0, lr=bt.feed["lr"],
                    adam_state=self.adam_state, l2_partial=bt.l2_live, ce=bt.ce, B=bt.B, reg=self.reg,
                    ce_scale=1.0 / self.gb(bt), loss=bt.loss)

    def clip_and_apply(self, bt):
        part = bt.norm_partial
        if self.tf_compat:
            # dense variables + [item dense (measured before the scatter) + un-deduplicated slots]
            n_g, n = self.n_dense, self.nb_dense + self.nb_item + bt.n_slot
        else:
            n_g = self.n_total
            n = ops.sqnorm_blocks(self.n_total)
        gb = self.gb(bt)
        if self._clip_in_adam(n):
            # no arrival ticket: the first launch only writes partials (and, in one more workgroup, advances the Adam
            # state and reduces the loss) -- or has already ridden in the backward's last launch; every workgroup of the
            # optimizer launch forms the norm itself
            if not getattr(bt, "norm_rode", False):
                ops.sqnorm_state_loss(self.flat_g, n_g, part, 0, bt.feed["lr"], self.adam_state, bt.l2_live,
                                      bt.l2_live.numel(), bt.ce, bt.B, self.reg, 1.0 / gb,
                                      None if self.loss_in_tail else bt.loss)
            bt.norm_rode = False
            # (only the ring's own steps carry the feed role: an ordinary step on the same batch object brings its feed itself)
            ring = getattr(bt, "feed_ring", None) if getattr(bt, "ring_step", False) else None
            ops.adam_images_clip(self.flat_p, self.flat_m, self.flat_v, self.flat_g, self.n_total, part, n, self.clip,
                                 self.scale, self.adam_state, self.n_dense, self.wimg_descs, copy16=self.item16,
                                 copy_begin=self.tab_off["item"], feed=ring.args() if ring is not None else None)
            return
        if getattr(bt, "ring_step", False):
            raise RuntimeError("a feed ring needs the optimizer launch that forms the clip scale itself "
                               "(Adam, <= %d norm partials)" % ops.adam_clip_max_partials())
        ops.sqnorm_clip_scale(self.flat_g, n_g, part, 0, n, self.clip, self.scale, bt.feed["lr"],
                              self.adam_state, self.ticket, bt.l2_live, bt.l2_live.numel(), bt.ce, bt.B,
                              self.reg, 1.0 / gb, None if self.loss_in_tail else bt.loss)
        if self.optimizer == "adam":
            # the bf16 scoring copy of the item table (if any) and the bf16 operand images of the forward's fused
            # projection weights (if any) are re-written by the same launch
            if self.wimg_descs is not None:
                ops.adam_images(self.flat_p, self.flat_m, self.flat_v, self.flat_g, self.n_total, self.scale,
                                self.adam_state, self.n_dense, self.wimg_descs, copy16=self.item16,
                                copy_begin=self.tab_off["item"])
            elif self.item16 is not None:
                ops.adam_bf16copy(self.flat_p, self.flat_m, self.flat_v, self.flat_g, self.n_total, self.scale,
                                  self.adam_state, self.n_dense, self.item16, self.tab_off["item"])
            else:
                ops.adam(self.flat_p, self.flat_m, self.flat_v, self.flat_g, self.n_total, self.scale,
                         self.adam_state, self.n_dense)
            return
        # sparse (row-skipping) region: category, position, user; the item gradient has every row
        ops.opt_update(self.optimizer, self.flat_p, self.flat_m, self.flat_v, self.flat_g, self.n_total,
                       self.scale, bt.feed["lr"], self.n_dense, self.tab_off["item"])
        self.refresh_derived()

    def forward_backward_kernels(self, bt, update_follows=False):
        """``update_follows``: clip_and_apply comes right after (train_kernels on one GPU) -- the clip's partial pass,
        the Adam state's advance and the loss reduction may then ride in the backward's last launch."""
        bt.train_step = True             # forward and backward back to back: the scoring passes may merge
        try:
            self.forward(bt, training=True)
        finally:
            bt.train_step = False
        self.loss_and_logit_grad(bt)
        bt.train_step_bwd = bool(update_follows)
        try:
            self.backward(bt)
        finally:
            bt.train_step_bwd = False

    # the two halves of a step under data-parallel row-sharded scoring (the scoring passes and their collectives run
    # between them: data_parallel.ShardedScoringExchange.score)
    def row_iota(self, bt):
        """0 .. B L - 1 (int32): the "ids" of pre-fetched item rows; also allocates ``bt.item_rows`` [B L, 128]."""
        if getattr(bt, "iota", None) is None:
            bt.iota = torch.arange(bt.R, dtype=torch.int32, device=self.device)
            bt.item_rows = torch.zeros((bt.R, D), dtype=torch.float32, device=self.device)
        return bt.iota

    def forward_to_pred_kernels(self, bt):
        self.forward(bt, training=True, score=False)

    def backward_from_pred_kernels(self, bt):
        self.backward(bt, score=False)

    def ring_train_kernels(self, bt):
        """train_kernels of a step whose feed came from ``bt.feed_ring``: the optimizer launch hands over the next."""
        bt.ring_step = True
        try:
            self.train_kernels(bt)
        finally:
            bt.ring_step = False

    def ring_clip_and_apply(self, bt):
        """clip_and_apply of a ring-fed step (the update half of a data-parallel step in two graphs)."""
        bt.ring_step = True
        try:
            self.clip_and_apply(bt)
        finally:
            bt.ring_step = False

    def train_kernels(self, bt):
        """Everything between feed upload and loss read-back; capturable."""
        if self.sharded_scoring is not None:
            if not self.sharded_scoring.replicate_table:
                self.row_iota(bt)
                self.sharded_scoring.fetch_history_rows(bt)
            self.forward_to_pred_kernels(bt)
            self.sharded_scoring.score(bt)
            self.backward_from_pred_kernels(bt)
            self.sharded_scoring.exchange_and_apply(bt)
            return
        self.forward_backward_kernels(bt, update_follows=self.sharded is None and self.allreduce_fn is None)
        if self.sharded is not None:
            self.sharded.exchange_and_apply(bt)
            return
        if self.allreduce_fn is not None:
            self.allreduce_fn(self, bt)
        self.clip_and_apply(bt)

    # evaluation scores the catalog in one piece while the [B, V] matrix is small; beyond this many bytes it goes
    # slab by slab through a bounded scratch and keeps only per-segment top-k candidates
    EVAL_STORED_MAX_BYTES = int(os.environ.get("MTAM_EVAL_STORED_MAX_BYTES", str(1 << 30)))
    EVAL_SLAB_BYTES = int(os.environ.get("MTAM_EVAL_SLAB_BYTES", str(1 << 30)))

    def eval_slab_width(self, B):
        seg = ops.TOPK_STREAM_SEG
        w = max(seg, self.EVAL_SLAB_BYTES // (4 * B) // seg * seg)
        return min(w, (self.item_rows + seg - 1) // seg * seg)

    def eval_kernels(self, bt, k=50, stored=None):
        """Forward + top-k of predict_behavior_emb . item_table^T (Model/base_model.py:194-202).
        stored=True keeps the whole [B, V] logits (``bt.logits``); False scores slab by slab and never holds
        more than EVAL_SLAB_BYTES of scores; None picks by size.  Both give identical lists."""
        V = self.item_rows
        if stored is None:
            stored = bt.B * bt.ld_logits * 4 <= self.EVAL_STORED_MAX_BYTES
        if stored:
            self.forward(bt, training=False)
            nbytes = ops.topk_workspace_bytes(bt.B, V, k)
            if nbytes and (bt.topk_ws is None or bt.topk_ws.numel() * 4 < nbytes):
                bt.topk_ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
            ops.topk(bt.logits_store, bt.ld_logits, bt.B, V, k, bt.topk_idx,
                     workspace=bt.topk_ws if nbytes else None)
            return
        self.forward(bt, training=False, score=False)
        W = self.eval_slab_width(bt.B)
        if bt.eval_slab is None or bt.eval_slab.shape[1] != W:
            bt.eval_slab = torch.empty((bt.B, W), dtype=torch.float32, device=self.device)
        nbytes = ops.topk_stream_workspace_bytes(bt.B, V, k)
        if bt.topk_ws is None or bt.topk_ws.numel() * 4 < nbytes:
            bt.topk_ws = torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=self.device)
        if self.score_dtype == "bf16":
            ops.f32_to_bf16(bt.pred.view(-1), bt.pred16.view(-1))
        for col0 in range(0, V, W):
            width = min(W, V - col0)
            if self.score_dtype == "f32":
                # rows [col0, col0 + width) of the item table: every score is the same k-ordered fmaf chain
                # as in the one-piece product
                ops.gemm(bt.pred, self.tables["item"][col0:col0 + width], bt.eval_slab, trans_b=True, ldc=W, split=False)
            else:
                ops.score16_logits(self.item16[col0:col0 + width], bt.pred16, bt.B, width, bt.eval_slab, W)
            ops.topk_stream_slab(bt.eval_slab, W, bt.B, col0, width, V, k, bt.topk_ws)
        ops.topk_stream_finish(bt.topk_ws, bt.B, V, k, bt.topk_idx)

    # ------------------------------------------------------- weights in / out
    def dense_tf(self):
        return self.layout.unpack(self.params.detach().cpu().numpy())

    def grads_tf(self):
        out = self.layout.unpack(self.grads.detach().cpu().numpy())
        for k in TABLES:
            out["embedding_layer/" + k] = self.g_tab[k].detach().cpu().numpy()
        return out

    def tables_numpy(self):
        return {k: v.detach().cpu().numpy() for k, v in self.tables.items()}

    def optimizer_state(self):
        """Both optimizer slots over the TRUE parameter space ``[:n_total]`` (the item pad rows and the
        data-parallel loss tail behind it are allocation details, not state) + the device-side Adam scalars."""
        n = self.n_total
        return {"flat_m": self.flat_m[:n].cpu(), "flat_v": self.flat_v[:n].cpu(), "adam_state": self.adam_state.cpu(),
                "n_total": n}

    def load_optimizer_state(self, st):
        """Accepts a state of length n_total (this build) or longer (a checkpoint written when the whole
        allocation was saved: its first n_total entries are the state)."""
        n = self.n_total
        for name, flat in (("flat_m", self.flat_m), ("flat_v", self.flat_v)):
            src = st[name]
            if src.numel() < n:
                raise ValueError("optimizer state %s has %d entries, this model needs %d" % (name, src.numel(), n))
            flat[:n].copy_(src.reshape(-1)[:n])
            flat[n:].zero_()
        self.adam_state.copy_(st["adam_state"])
