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


def attach(path, world_size, group=None, force=False, shard_items=None):
    """Make ``path.train_kernels`` exchange gradients; call once after building the model.
    ``force`` installs the exchange even for one rank (tests the code path on one GPU).
    ``shard_items``: True / False, or None = by size -- the item table's gradient is exchanged as
    reduce-scatter + shard-owned update + all-gather (ShardedItemExchange) once it is at least
    MTAM_DP_SHARD_MIN_BYTES (default 64 MiB); below that one flat all-reduce of every gradient is cheaper."""
    if world_size <= 1 and not force:
        return
    path.tf_compat = False
    path.world_size = world_size
    if shard_items is None:
        shard_items = path.item_rows * D * 4 >= int(os.environ.get("MTAM_DP_SHARD_MIN_BYTES", str(64 << 20)))
    if shard_items:
        rank = dist.get_rank(group) if dist.is_initialized() else 0
        path.sharded = ShardedItemExchange(path, max(world_size, 1), rank, group)
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

    def _gloo(self):
        return dist.get_backend(self.group) == "gloo"

    def exchange(self):
        p = self.p
        dist.all_reduce(p.flat_g[:self.off_item], op=dist.ReduceOp.SUM, group=self.group)
        item = p.flat_g[self.off_item:self.off_item + self.world * self.shard_elems]
        if self._gloo():
            # the CPU test backend has no reduce-scatter: an all-reduce leaves the same sums on the owned rows
            dist.all_reduce(item, op=dist.ReduceOp.SUM, group=self.group)
        else:
            # in place (RCCL: recvbuff == sendbuff + rank * recvcount)
            dist.reduce_scatter_tensor(p.flat_g[self.lo:self.hi], item, op=dist.ReduceOp.SUM, group=self.group)

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
        item = p.flat_p[self.off_item:self.off_item + self.world * self.shard_elems]
        if self._gloo():
            mine = p.flat_p[self.lo:self.hi].clone()
            dist.all_gather([item[r * self.shard_elems:(r + 1) * self.shard_elems] for r in range(self.world)], mine,
                            group=self.group)
        else:
            dist.all_gather_into_tensor(item, p.flat_p[self.lo:self.hi], group=self.group)      # in place
        p.refresh_derived()

    def exchange_and_apply(self, bt):
        if self.p.optimizer != "adam":
            raise NotImplementedError("the row-sharded exchange is built for Adam (the reference's default optimizer)")
        self.exchange()
        self.apply(bt)


def broadcast_parameters(path, src=0, group=None):
    """Replicas start identical (they would anyway with equal seeds; this makes it explicit)."""
    dist.broadcast(path.flat_p, src=src, group=group)
    path.refresh_derived()


def max_over_ranks(value, device):
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
