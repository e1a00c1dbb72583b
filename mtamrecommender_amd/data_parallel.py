"""User-batch data parallelism over the GPUs of one node (RCCL over xGMI).

The reference has no multi-GPU path (run_server.py only fans out
independent experiments).  BASELINE.json's north_star asks for one: every rank
holds a full replica (tables included), takes its own slice of the global
batch, and gradients are summed with RCCL all-reduce before the clip + Adam
update, so that replicas stay bit-identical.

What is exchanged: the dense-parameter gradients AND the
table gradients -- the item table's gradient is dense because of the
full-catalog softmax, so replicas diverge if only the attention/GRU parameters
are reduced.  All gradients live in one flat buffer
[dense | category | position | user | item], so the exchange is ONE all-reduce
per step (one large collective suits the per-link-bound xGMI ring better than
many small ones).
The loss is a mean over the GLOBAL batch (each rank scales by 1/B_global), the
L2 term is a plain sum, so a sum all-reduce is exact.  The clip norm is the
true norm of the summed gradient (the TF IndexedSlices norm of App D-5 is a
single-process artefact and is only reproduced on one GPU).
"""
import os

import torch
import torch.distributed as dist

D = 128


def shard(records, rank, world):
    """Contiguous, equal slices of a global batch (drop nothing: sizes differ by at most 1)."""
    n = len(records)
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return records[lo:hi]


def keep_global_batch(global_size, world):
    """Whether a global batch is trained on.  One smaller than the world size would give some ranks an empty
    slice -- they could not run the step while the others wait inside the collective -- so it is dropped on EVERY
    rank (the test looks at the global size only, hence every rank agrees).  The reference keeps its last partial
    batch (DataHandle/get_input_data.py); with one rank so does this build."""
    return global_size >= max(1, world)


def allreduce_gradients(buffers, group=None):
    """Sum the given gradient buffers across ranks, in place."""
    for b in buffers:
        dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)


EXCHANGES = ("flat", "sharded", "sharded-scoring", "sharded-table")


def attach(path, world_size, group=None, force=False, shard_items=None, exchange=None):
    """Make ``path.train_kernels`` exchange gradients; call once after building the model.
    ``force`` installs the exchange even for one rank (tests the code path on one GPU).
    ``exchange`` (or MTAM_DP_EXCHANGE): "flat" = one all-reduce of every gradient; "sharded" = the item gradient as
    reduce-scatter + shard-owned update + all-gather (ShardedItemExchange); "sharded-scoring" = the item table
    row-sharded for scoring too, the dense item gradient never moves (ShardedScoringExchange); None = by size --
    "sharded" once the item table is at least MTAM_DP_SHARD_MIN_BYTES (default 64 MiB), "flat" below that.
    ``shard_items`` (True / False): the older spelling of "sharded" / "flat"."""
    if world_size <= 1 and not force:
        return
    path.tf_compat = False
    path.world_size = world_size
    exchange = exchange or os.environ.get("MTAM_DP_EXCHANGE") or None
    if exchange is None and shard_items is not None:
        exchange = "sharded" if shard_items else "flat"
    if exchange is None:
        big = path.item_rows * D * 4 >= int(os.environ.get("MTAM_DP_SHARD_MIN_BYTES", str(64 << 20)))
        exchange = "sharded" if big else "flat"
    if exchange not in EXCHANGES:
        raise ValueError("unknown data-parallel exchange %r (one of %s)" % (exchange, ", ".join(EXCHANGES)))
    path.dp_exchange = exchange
    rank = dist.get_rank(group) if dist.is_initialized() else 0
    if exchange == "sharded":
        path.sharded = ShardedItemExchange(path, max(world_size, 1), rank, group)
        return
    if exchange in ("sharded-scoring", "sharded-table"):
        if not path.logits_free32:
            raise ValueError("%s runs on the fp32 logits-free scoring kernels (score_dtype 'f32')" % exchange)
        path.sharded_scoring = ShardedScoringExchange(path, max(world_size, 1), rank, group,
                                                      replicate_table=exchange == "sharded-scoring")
        return

    # The reported loss is a sum over ranks too (reg * l2 is a plain sum, the cross entropy a mean over the GLOBAL
    # batch): each rank's terms ride in the tail of the gradient buffer through the same all-reduce, so every rank
    # reports the loss of the whole batch without a collective of its own.
    from . import hip_ops as ops
    path.loss_in_tail = True
    for bt in path._batches.values():
        bt.loss = path.loss_tail[:3]

    def exchange(p, bt):
        ops.loss_reduce(bt.l2_live, bt.l2_live.numel(), bt.ce, bt.B, p.reg, 1.0 / p.gb(bt), p.loss_tail)
        allreduce_gradients([p.flat_g[:p.n_items_end + 4]], group)

    path.allreduce_fn = exchange


class HipStepKernels(object):
    """The device kernels ShardedItemExchange drives (csrc/optim.hip through the C ABI).  A CPU twin with the
    same methods stands in for them in the 2-rank gloo test (tests/test_data_parallel_cpu.py)."""

    def __init__(self, path):
        from . import hip_ops as ops
        self.ops, self.p = ops, path
        self.partials = torch.zeros(ops.sqnorm_blocks(path.n_alloc) + 8, dtype=torch.float32, device=path.device)

    def sq_sum(self, g, weight, out, accumulate):
        """out[0] (+)= weight * sum(g^2) in float64."""
        n = g.numel()
        self.ops.sqnorm_partial(g, n, self.partials)
        self.ops.partials_sum(self.partials, self.ops.sqnorm_blocks(n), weight, out, accumulate)

    def clip_scale(self, sq_total, clip, scale, lr, adam_state):
        self.ops.clip_scale_sq(sq_total, 1, clip, scale, lr, adam_state)

    def loss(self, bt, reg, ce_scale):
        self.ops.loss_reduce(bt.l2_live, bt.l2_live.numel(), bt.ce, bt.B, reg, ce_scale, bt.loss)

    def adam(self, p, m, v, g, scale, hyper, sparse_begin):
        self.ops.adam(p, m, v, g, g.numel(), scale, hyper, sparse_begin)


class ShardedItemExchange(object):
    """Data-parallel exchange for large catalogs (BASELINE.json configs[3]).

    The full-catalog softmax makes the item table's gradient dense: 5.1 GB at 10 M items.  Instead of
    all-reducing it and running the same dense Adam on every rank, each rank OWNS a contiguous range of item rows:

      all-reduce      [dense | category | position | user] gradients (a few MB; replicated update)
      reduce-scatter  the item gradient by row range (in place: a rank receives the sum of its own rows)
      clip            every rank sums the squares of what it owns -- its item rows, and on rank 0 also the
                      replicated small part -- the doubles are all-reduced, the scale comes from the total
      Adam            on the small part (every rank, identical) and on the rank's own item rows only:
                      1 / world of the 28 B/element the dense update streams
      all-gather      the updated item rows (in place), then the bf16 scoring copy is refreshed locally

    Wire volume per rank equals the flat all-reduce's, 2 (G-1)/G x 4 V D bytes; what goes away is (G-1)/G of
    the item-table Adam.  Replicas stay bit-identical: every rank applies the same reduced values to the same
    rows.  The item region of the flat buffers is allocated with its row count rounded up to a multiple of 8
    (TimeAwarePath.item_rows_pad), so 1, 2, 4 and 8 ranks cut it evenly; pad rows stay zero.
    No run with more than one GPU exists yet (this build's GPU box has one): covered by a 2-rank gloo test on
    CPU tensors (bit-for-bit against the replicated update) and a 1-rank RCCL test through the real kernels.
    """

    def __init__(self, path, world, rank, group=None, kernels=None):
        self.p, self.world, self.rank, self.group = path, world, rank, group
        if path.item_rows_pad % world:
            raise ValueError("item rows (padded to %d) do not split over %d ranks" % (path.item_rows_pad, world))
        self.k = kernels if kernels is not None else HipStepKernels(path)
        off = path.tab_off["item"]
        self.off_item = off
        self.shard_elems = path.item_rows_pad // world * D
        lo = off + rank * self.shard_elems
        self.lo, self.hi = lo, lo + self.shard_elems
        # the rank's rows that exist (the last rank's range ends in pad rows)
        self.hi_true = max(lo, min(self.hi, path.n_total))
        dev = path.flat_g.device
        self.sq = torch.zeros(1, dtype=torch.float64, device=dev)

    # ---- the two in-place collectives over the item region (the SAME calls under RCCL and under gloo: this build's
    # gloo has both, so the 2- and 4-rank CPU tests run exactly these lines)
    def _item_region(self, flat):
        item = flat[self.off_item:self.off_item + self.world * self.shard_elems]
        mine = flat[self.lo:self.hi]
        # the in-place contract of both collectives: this rank's piece IS piece `rank` of the whole buffer
        assert self.lo == self.off_item + self.rank * self.shard_elems and self.hi - self.lo == self.shard_elems
        assert mine.untyped_storage().data_ptr() == item.untyped_storage().data_ptr()
        assert mine.storage_offset() == item.storage_offset() + self.rank * self.shard_elems
        assert item.numel() == self.world * mine.numel() and item.is_contiguous() and mine.is_contiguous()
        return item, mine

    def reduce_scatter_item_gradient(self):
        item, mine = self._item_region(self.p.flat_g)
        dist.reduce_scatter_tensor(mine, item, op=dist.ReduceOp.SUM, group=self.group)

    def all_gather_item_rows(self):
        item, mine = self._item_region(self.p.flat_p)
        dist.all_gather_into_tensor(item, mine, group=self.group)

    def exchange(self):
        p = self.p
        dist.all_reduce(p.flat_g[:self.off_item], op=dist.ReduceOp.SUM, group=self.group)
        self.reduce_scatter_item_gradient()

    def apply(self, bt):
        p, k = self.p, self.k
        # squared norm of the summed gradient: own item rows everywhere, the replicated part counted once
        k.sq_sum(p.flat_g[self.lo:self.hi_true], 1.0, self.sq, False) if self.hi_true > self.lo else self.sq.zero_()
        if self.rank == 0:
            k.sq_sum(p.flat_g[:self.off_item], 1.0, self.sq, True)
        dist.all_reduce(self.sq, op=dist.ReduceOp.SUM, group=self.group)
        k.clip_scale(self.sq, p.clip, p.scale, bt.feed["lr"], p.adam_state)
        # the reported loss: reg * l2 (a sum over ranks) + mean cross entropy (each rank's sum / global batch)
        k.loss(bt, p.reg, 1.0 / p.gb(bt))
        dist.all_reduce(bt.loss, op=dist.ReduceOp.SUM, group=self.group)
        # replicated small part, then the owned item rows (IndexedSlices form of the update: tables)
        k.adam(p.flat_p[:self.off_item], p.flat_m[:self.off_item], p.flat_v[:self.off_item],
               p.flat_g[:self.off_item], p.scale, p.adam_state, p.n_dense)
        if self.hi_true > self.lo:
            k.adam(p.flat_p[self.lo:self.hi_true], p.flat_m[self.lo:self.hi_true], p.flat_v[self.lo:self.hi_true],
                   p.flat_g[self.lo:self.hi_true], p.scale, p.adam_state, 0)
        self.publish_item_rows()
        p.refresh_derived()

    def publish_item_rows(self):
        """After the update: every replica gets every owner's rows."""
        self.all_gather_item_rows()

    def exchange_and_apply(self, bt):
        if self.p.optimizer != "adam":
            raise NotImplementedError("the row-sharded exchange is built for Adam (the reference's default optimizer)")
        self.exchange()
        self.apply(bt)


class HipScoringKernels(HipStepKernels):
    """The scoring-side kernels ShardedScoringExchange drives, on a ROW RANGE of the item table (csrc/score32.hip,
    csrc/emb.hip through the C ABI); a CPU twin stands in for them in the gloo tests."""

    def __init__(self, path):
        super(HipScoringKernels, self).__init__(path)
        self._partials = {}

    def _bufs(self, Bg, V):
        key = (Bg, V)
        if key not in self._partials:
            dev = self.p.device
            self._partials[key] = (torch.zeros(self.ops.score32_partials(Bg, V), device=dev),
                                   torch.zeros(self.ops.score32_sq_partials(V), device=dev))
        return self._partials[key]

    def lse_range(self, E_rows, row0, pred_all, tgt_all, lse_part, tlogit):
        Bg, V = pred_all.shape[0], E_rows.shape[0]
        self.ops.score32_lse(E_rows, pred_all, tgt_all, Bg, V, self._bufs(Bg, V)[0], lse_part, tlogit, row0=row0)

    def bwd_range(self, E_rows, row0, pred_all, lse_all, tgt_all, scale, d_pred_all, dE_rows):
        Bg, V = pred_all.shape[0], E_rows.shape[0]
        d_pred_all.zero_()
        self.ops.score32_bwd(E_rows, pred_all, lse_all, tgt_all, Bg, V, scale, d_pred_all, dE_rows, None, row0=row0)

    def gather_owned(self, E_rows, row0, ids, out):
        self.ops.rows_gather_range(E_rows, row0, ids, out)

    def scatter_items(self, d_ic, ic, item_ids, seq_len, B, L, reg, g_item, rows):
        n = self.ops.emb_scatter_partials(B, L)
        if getattr(self, "_slot_sq", None) is None or self._slot_sq.numel() < n:
            self._slot_sq = torch.zeros(n, device=self.p.device)
        self.ops.emb_scatter_add_items_range(d_ic, ic, item_ids, seq_len, B, L, reg, g_item, self._slot_sq, rows)


class ShardedScoringExchange(ShardedItemExchange):
    """Data parallelism with the item table ROW-SHARDED FOR SCORING as well ("alternative worth
    measuring"; Model/base_model.py:309-322 is the product being distributed).

    Under ShardedItemExchange every rank scores the whole catalog for its own samples, so every rank holds a dense
    [V, 128] item gradient and the ranks exchange it (reduce-scatter: 4.5 GB per rank at 10 M items, then as much
    again for the all-gather of the updated rows).  Here a rank scores only the rows it OWNS, for the samples of
    EVERY rank:

      forward (graph)   lookups .. decoder -> pred [B, 128] of the rank's own samples
      all-gather        pred -> [G B, 128], target ids -> [G B]                                   (0.5 MB at G = 8)
      lse pass          own rows x all samples -> per-sample log-sum-exp over the range + target logit if owned
      all-reduce        max, then (sum of exp, target logit): the whole catalog's lse and the loss terms  (8 KB)
      backward pass     own rows x all samples: dE of the own rows -- COMPLETE, nothing to exchange -- and the range's
                        share of d_pred for all samples
      reduce-scatter    d_pred shares -> the rank's own samples                                    (0.5 MB)
      backward (graph)  decoder .. lookups from d_pred; the scatter-add applies the rank's history-row gradients to
                        its OWN item rows only
      all-gather        every rank's history slots (d[item|category] rows, looked-up rows, ids, lengths: 13 MB per
                        rank); each rank applies the other ranks' slots that fall into its rows
      then as ShardedItemExchange: all-reduce of the small gradients, clip from owned squares, Adam on the owned rows,
      all-gather of the updated rows (history lookups read a replicated table).

    Wire volume per rank at 10 M items, 8 ranks: 4.5 GB (the updated rows) instead of 9 GB; the dense item gradient is
    never moved.  Arithmetic is the replicated exchange's: the same products, each sample's softmax over the same
    rows -- sums in a different order (the lse combines per-rank partial sums), so equality is to fp32 rounding, not
    bit for bit.  No run with more than one GPU exists yet: 2- and 4-rank gloo tests on CPU tensors against the
    replicated update, a 1-rank RCCL test through the real kernels against the single-GPU step.
    """

    def __init__(self, path, world, rank, group=None, kernels=None, replicate_table=True):
        super(ShardedScoringExchange, self).__init__(path, world, rank, group,
                                                     kernels if kernels is not None else HipScoringKernels(path))
        # replicate_table=False ("sharded-table"): the updated rows are NOT all-gathered.  A rank's copy of the rows it
        # does not own goes stale; the embedding lookups take their item rows from the owners instead
        # (fetch_history_rows: all-gather of the ranks' history ids, every rank looks its own rows up for ALL of them,
        # a reduce-scatter of the [G B L, 128] result -- each row has one owner, so the sum is the row -- hands every
        # rank the rows of its own batch: 26 MB at 8 ranks of 128 x 50 against the 4.5 GB all-gather at 10 M items).
        # sync_item_table() brings a replica up to date for evaluation and checkpoints.
        self.replicate_table = bool(replicate_table)
        self.table_current = True
        self.rows_per_rank = path.item_rows_pad // world
        self.row_lo = rank * self.rows_per_rank
        self.row_hi = max(self.row_lo, min(self.row_lo + self.rows_per_rank, path.item_rows))   # rows that exist
        if self.row_hi <= self.row_lo:
            raise ValueError("rank %d owns no item row (%d rows over %d ranks)" % (rank, path.item_rows, world))
        self._b = {}

    def _buffers(self, bt):
        B = bt.B
        if B not in self._b:
            G, dev = self.world, self.p.flat_p.device
            f = lambda *shape: torch.zeros(shape, dtype=torch.float32, device=dev)
            R = B * self.p.L
            self._b[B] = dict(pred=f(G * B, D), tgt=torch.zeros(G * B, dtype=torch.int32, device=dev), lse_part=f(G * B),
                              stats=f(2, G * B), lse=f(G * B), m=f(G * B), d_pred=f(G * B, D),
                              d_ic=f(G, R, 2 * D), ic=f(G, R, 2 * D), rows_all=None if self.replicate_table else f(G * R, D),
                              ids=torch.zeros((G, R), dtype=torch.int32, device=dev),
                              sl=torch.zeros((G, B), dtype=torch.int32, device=dev))
        return self._b[B]

    # ---- before the forward ("sharded-table")
    def fetch_history_rows(self, bt):
        """bt.item_rows [B L, 128] <- the item rows of this rank's history ids, each from its owner."""
        p, k, w = self.p, self.k, self._buffers(bt)
        dist.all_gather_into_tensor(w["ids"].view(-1), bt.feed["item_list"].reshape(-1), group=self.group)
        k.gather_owned(p.tables["item"][self.row_lo:self.row_hi], self.row_lo, w["ids"].view(-1), w["rows_all"])
        dist.reduce_scatter_tensor(bt.item_rows, w["rows_all"], op=dist.ReduceOp.SUM, group=self.group)

    def sync_item_table(self):
        """All-gather the owners' rows into every replica ("sharded-table": before evaluation / a checkpoint)."""
        if not self.table_current:
            self.all_gather_item_rows()
            self.p.refresh_derived()
            self.table_current = True

    # ---- between the two halves of the step
    def score(self, bt):
        """bt.pred (own samples) -> bt.lse, bt.ce (own samples), bt.d_pred (own samples), dE of the own item rows."""
        p, k, w = self.p, self.k, self._buffers(bt)
        B, G, r = bt.B, self.world, self.rank
        dist.all_gather_into_tensor(w["pred"], bt.pred, group=self.group)
        dist.all_gather_into_tensor(w["tgt"], bt.feed["target_item_id"], group=self.group)
        E_rows = p.tables["item"][self.row_lo:self.row_hi]
        lse_part, tlogit = w["lse_part"], w["stats"][1]
        k.lse_range(E_rows, self.row_lo, w["pred"], w["tgt"], lse_part, tlogit)
        # the whole catalog's log-sum-exp: m = max_r lse_r, lse = m + log sum_r exp(lse_r - m); the target's logit
        # lives on exactly one rank
        m = w["m"]                          # (every buffer of the step is allocated once per batch size)
        m.copy_(lse_part)
        dist.all_reduce(m, op=dist.ReduceOp.MAX, group=self.group)
        torch.sub(lse_part, m, out=w["stats"][0])
        w["stats"][0].exp_()
        dist.all_reduce(w["stats"], op=dist.ReduceOp.SUM, group=self.group)
        torch.log(w["stats"][0], out=w["lse"])
        w["lse"].add_(m)
        own = slice(r * B, (r + 1) * B)
        bt.lse.copy_(w["lse"][own])
        torch.sub(w["lse"][own], w["stats"][1][own], out=bt.ce)
        dE_rows = p.g_tab["item"][self.row_lo:self.row_hi]
        k.bwd_range(E_rows, self.row_lo, w["pred"], w["lse"], w["tgt"], 1.0 / p.gb(bt), w["d_pred"], dE_rows)
        dist.reduce_scatter_tensor(bt.d_pred, w["d_pred"], op=dist.ReduceOp.SUM, group=self.group)

    # ---- after the backward
    def exchange(self, bt):
        p, k, w = self.p, self.k, self._buffers(bt)
        dist.all_reduce(p.flat_g[:self.off_item], op=dist.ReduceOp.SUM, group=self.group)
        # every rank's history slots; the other ranks' slots that fall into this rank's rows are added to them
        # (outputs as the concatenation along dim 0 of the ranks' inputs: the shape both backends accept)
        dist.all_gather_into_tensor(w["d_ic"].view(-1, 2 * D), bt.d_ic, group=self.group)
        dist.all_gather_into_tensor(w["ic"].view(-1, 2 * D), bt.ic, group=self.group)
        if self.replicate_table:            # ("sharded-table": fetch_history_rows has gathered the ids already)
            dist.all_gather_into_tensor(w["ids"].view(-1), bt.feed["item_list"].reshape(-1), group=self.group)
        dist.all_gather_into_tensor(w["sl"].view(-1), bt.feed["seq_length"], group=self.group)
        for src in range(self.world):
            if src != self.rank:
                k.scatter_items(w["d_ic"][src], w["ic"][src], w["ids"][src], w["sl"][src], bt.B, p.L, p.reg,
                                p.g_tab["item"], (self.row_lo, self.row_hi))

    def exchange_and_apply(self, bt):
        if self.p.optimizer != "adam":
            raise NotImplementedError("the row-sharded exchange is built for Adam (the reference's default optimizer)")
        self.exchange(bt)
        self.apply(bt)

    def publish_item_rows(self):
        if self.replicate_table:
            self.all_gather_item_rows()
        else:
            self.table_current = False      # "sharded-table": the replicas' foreign rows are stale from here on


def broadcast_parameters(path, src=0, group=None):
    """Replicas start identical (they would anyway with equal seeds; this makes it explicit)."""
    dist.broadcast(path.flat_p, src=src, group=group)
    path.refresh_derived()


def max_over_ranks(value, device):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
