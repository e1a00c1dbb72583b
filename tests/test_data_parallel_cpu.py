"""Data-parallel logic with two gloo ranks on the CPU: sharding, gradient exchange, and the
identity it relies on -- summed shard gradients (loss scaled by the GLOBAL batch) equal the
full-batch gradient, so replicas that all-reduce stay identical."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from mtamrecommender_amd import data_parallel
    from tests.test_oracle import REG, small_case
    from oracle import mtam_oracle as O
    feed, arrays = small_case("MTAM", B=6, L=8, D=16, NB=1, H=2, seed=9)
    B = len(feed["user_id"])
    rows = data_parallel.shard(list(range(B)), rank, world)
    local = {k: v[rows] for k, v in feed.items()}
    _, grads, _ = O.loss_and_grads("MTAM", arrays, local, 2, 1, REG, torch.float64, global_batch=B)
    names = sorted(k for k, g in grads.items() if g is not None)
    bufs = [torch.from_numpy(np.ascontiguousarray(grads[k])) for k in names]
    data_parallel.allreduce_gradients(bufs)
    if rank == 0:
        _, full, _ = O.loss_and_grads("MTAM", arrays, feed, 2, 1, REG, torch.float64)
        worst = max(float(np.abs(b.numpy() - full[k]).max()) for k, b in zip(names, bufs))
        np.save(os.path.join(out_dir, "worst.npy"), np.array([worst]))
    t = torch.tensor([float(rank + 1)])
    assert data_parallel.max_over_ranks(float(rank + 1), "cpu") == float(world)
    dist.barrier()
    dist.destroy_process_group()


def test_shard_is_a_partition():
    from mtamrecommender_amd import data_parallel
    recs = list(range(11))
    parts = [data_parallel.shard(recs, r, 4) for r in range(4)]
    assert sum(parts, []) == recs and max(map(len, parts)) - min(map(len, parts)) <= 1


def test_two_rank_gradient_exchange_matches_full_batch(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    worst = float(np.load(os.path.join(str(tmp_path), "worst.npy"))[0])
    assert worst < 1e-12


# ------------------------------------------------------------------ row-sharded item-table exchange
class _TorchStepKernels(object):
    """CPU twin of data_parallel.HipStepKernels: the same five operations with torch ops (Adam as csrc/optim.hip
    writes it: dense form m += (g - m)(1 - b1) before `sparse_begin`, IndexedSlices form m = m b1 + g (1 - b1) after)."""

    def sq_sum(self, g, weight, out, accumulate):
        t = weight * float((g.double() ** 2).sum())
        out[0] = out[0] + t if accumulate else t

    def clip_scale(self, sq_total, clip, scale, lr, adam_state):
        norm = np.float32(np.sqrt(np.float32(sq_total[0].item())))
        scale[0] = float(np.float32(clip) * min(np.float32(1.0) / norm, np.float32(1.0) / np.float32(clip)))
        scale[1] = float(norm)
        b1p, b2p = np.float32(adam_state[4].item()), np.float32(adam_state[5].item())
        adam_state[0] = float(np.float32(lr[0].item()) * np.sqrt(np.float32(1.0) - b2p) / (np.float32(1.0) - b1p))
        adam_state[4] = float(b1p * np.float32(adam_state[1].item()))
        adam_state[5] = float(b2p * np.float32(adam_state[2].item()))

    def loss(self, bt, reg, ce_scale):
        bt.loss[0] = reg * bt.l2 + bt.ce.sum() * ce_scale
        bt.loss[1] = bt.l2
        bt.loss[2] = bt.ce.sum() * ce_scale

    def adam(self, p, m, v, g, scale, hyper, sparse_begin):
        lr_t, b1, b2, eps = (hyper[i].clone() for i in range(4))
        gs = g * scale[0]
        n = p.numel()
        sb = min(int(sparse_begin), n)
        m[:sb] += (gs[:sb] - m[:sb]) * (1.0 - b1)
        m[sb:] = m[sb:] * b1 + gs[sb:] * (1.0 - b1)
        v[:sb] += (gs[:sb] * gs[:sb] - v[:sb]) * (1.0 - b2)
        v[sb:] = v[sb:] * b2 + gs[sb:] * gs[sb:] * (1.0 - b2)
        p -= lr_t * m / (torch.sqrt(v) + eps)


class _FakeBatch(object):
    def __init__(self, B, lr):
        self.B = B
        self.feed = {"lr": torch.tensor([lr], dtype=torch.float32)}
        self.loss = torch.zeros(3)
        self.l2, self.ce = 0.0, torch.zeros(B)


class _FakePath(object):
    """The attributes of TimeAwarePath the exchange touches, on CPU tensors."""
    optimizer = "adam"

    def __init__(self, n_dense, small_rows, item_rows, seed):
        D = 128
        g = torch.Generator().manual_seed(seed)
        self.n_dense = n_dense
        self.tab_off = {"item": n_dense + small_rows * D}
        self.item_rows, self.item_rows_pad = item_rows, (item_rows + 7) // 8 * 8
        self.n_total = self.tab_off["item"] + item_rows * D
        self.n_alloc = self.tab_off["item"] + self.item_rows_pad * D
        z = lambda: torch.zeros(self.n_alloc)
        self.flat_p, self.flat_g, self.flat_m, self.flat_v = z(), z(), z(), z()
        self.flat_p[:self.n_total] = torch.randn(self.n_total, generator=g) * 0.1
        self.flat_m[:self.n_total] = torch.randn(self.n_total, generator=g) * 0.01
        self.flat_v[:self.n_total] = torch.rand(self.n_total, generator=g) * 1e-3
        self.clip, self.reg = 1.0, 5e-5
        self.scale = torch.zeros(2)
        self.adam_state = torch.tensor([0.0, 0.9, 0.999, 1e-8, 0.9 ** 3, 0.999 ** 3, 0.0, 0.0])
        self.global_batch, self.world_size, self.device = None, 1, "cpu"
        self.refreshed = 0

    def gb(self, bt):
        return bt.B * self.world_size

    def refresh_derived(self):
        self.refreshed += 1


def _sharded_worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from mtamrecommender_amd import data_parallel
    D = 128
    # the same replica on every rank; every rank's OWN gradient (seeded by rank); item rows not a multiple of 8
    path = _FakePath(n_dense=4096, small_rows=37, item_rows=1003, seed=1)
    path.world_size = world
    gen = torch.Generator().manual_seed(100 + rank)
    # multiples of 2^-12: sums over ranks are exact in fp32, so the comparison does not depend on the order in
    # which a collective happens to add its operands (gloo's ring cuts buffers of different sizes differently)
    path.flat_g[:path.n_total] = torch.round(torch.randn(path.n_total, generator=gen) * 0.02 * 4096) / 4096
    bt = _FakeBatch(B=6, lr=1e-3)
    bt.l2, bt.ce = 3.0 + rank, torch.full((6,), 2.0 + rank)
    # ---- the replicated update on the all-reduced gradient (what the flat exchange computes), kept aside
    ref = _FakePath(n_dense=4096, small_rows=37, item_rows=1003, seed=1)
    ref.world_size = world
    g_all = path.flat_g.clone()
    dist.all_reduce(g_all)
    k = _TorchStepKernels()
    sq = torch.zeros(1, dtype=torch.float64)
    k.sq_sum(g_all[:ref.n_total], 1.0, sq, False)
    k.clip_scale(sq, ref.clip, ref.scale, bt.feed["lr"], ref.adam_state)
    k.adam(ref.flat_p[:ref.n_total], ref.flat_m[:ref.n_total], ref.flat_v[:ref.n_total], g_all[:ref.n_total], ref.scale,
           ref.adam_state, ref.n_dense)
    # ---- the row-sharded exchange
    ex = data_parallel.ShardedItemExchange(path, world, rank, kernels=k)
    assert ex.shard_elems * world == path.item_rows_pad * D and ex.lo == path.tab_off["item"] + rank * ex.shard_elems
    ex.exchange_and_apply(bt)
    rel = abs(float(path.scale[1]) - float(ref.scale[1])) / float(ref.scale[1])
    same = (torch.equal(path.flat_p, ref.flat_p) and torch.equal(path.adam_state, ref.adam_state))
    own = slice(ex.lo, ex.hi_true)
    same_slots = (torch.equal(path.flat_m[own], ref.flat_m[own]) and torch.equal(path.flat_v[own], ref.flat_v[own]) and
                  torch.equal(path.flat_m[:ex.off_item], ref.flat_m[:ex.off_item]))
    pad_zero = not bool(path.flat_p[path.n_total:].any()) and not bool(path.flat_g[path.n_total:].any())
    # loss = reg * (l2 summed over ranks) + (ce summed over ranks) / global batch
    want_loss = path.reg * sum(3.0 + r for r in range(world)) + sum(6 * (2.0 + r) for r in range(world)) / (6 * world)
    np.save(os.path.join(out_dir, "sharded_%d.npy" % rank),
            np.array([rel, float(same), float(same_slots), float(pad_zero), float(bt.loss[0]) - want_loss,
                      float(torch.equal(path.scale, ref.scale)), float(path.refreshed)]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_item_exchange_equals_the_replicated_update(tmp_path, world):
    """reduce-scatter by row range + shard-owned clip contribution and Adam + all-gather (gloo stand-ins for the
    two collectives it lacks) against dense Adam on the all-reduced gradient: parameters, both slots and the Adam
    state bit for bit on every rank, 1,003 item rows (pad rows untouched), loss summed over ranks."""
    port = 29500 + (os.getpid() % 2000) + 7 * world
    mp.spawn(_sharded_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    for rank in range(world):
        rel, same, same_slots, pad_zero, dloss, same_scale, refreshed = np.load(
            os.path.join(str(tmp_path), "sharded_%d.npy" % rank))
        assert rel < 1e-7 and same == 1.0 and same_slots == 1.0 and pad_zero == 1.0, (rank, rel, same, same_slots)
        assert abs(dloss) < 1e-5 and refreshed == 1.0
        assert same_scale == 1.0            # the clip scale itself came out bit-identical too


# ------------------------------------------------- item table row-sharded for SCORING as well (SURVEY.md 8e alternative)
class _TorchScoringKernels(_TorchStepKernels):
    """CPU twin of data_parallel.HipScoringKernels: the two scoring passes over a row range and the item-only
    scatter-add of another rank's slots, with torch ops."""

    def lse_range(self, E_rows, row0, pred_all, tgt_all, lse_part, tlogit):
        logits = pred_all @ E_rows.t()
        lse_part.copy_(torch.logsumexp(logits, 1))
        t = tgt_all.long() - row0
        owned = (t >= 0) & (t < E_rows.shape[0])
        tlogit.copy_(torch.where(owned, logits.gather(1, t.clamp(0, E_rows.shape[0] - 1)[:, None])[:, 0],
                                 torch.zeros_like(lse_part)))

    def bwd_range(self, E_rows, row0, pred_all, lse_all, tgt_all, scale, d_pred_all, dE_rows):
        G = torch.exp(pred_all @ E_rows.t() - lse_all[:, None])
        t = tgt_all.long() - row0
        owned = (t >= 0) & (t < E_rows.shape[0])
        G[torch.nonzero(owned)[:, 0], t[owned]] -= 1.0
        G *= scale
        d_pred_all.copy_(G @ E_rows)
        dE_rows.copy_(G.t() @ pred_all)

    def gather_owned(self, E_rows, row0, ids, out):
        t = ids.long() - row0
        owned = (t >= 0) & (t < E_rows.shape[0])
        out.zero_()
        out[owned] = E_rows[t[owned]]

    def scatter_items(self, d_ic, ic, item_ids, seq_len, B, L, reg, g_item, rows):
        D = 128
        live = (torch.arange(L)[None, :] < seq_len[:, None]).reshape(-1)
        g = torch.where(live[:, None], d_ic[:, :D], torch.zeros(1)) + reg * ic[:, :D]      # padded slots: the L2 term only
        ids = item_ids.long()
        mine = (ids >= rows[0]) & (ids < rows[1])
        g_item.index_add_(0, ids[mine], g[mine])


class _ToyBatch(object):
    pass


def _toy_inputs(rank, B, L, V, seed=7):
    g = torch.Generator().manual_seed(seed * 100 + rank)
    sl = torch.randint(1, L + 1, (B,), generator=g, dtype=torch.int32)
    ids = torch.randint(0, V, (B, L), generator=g, dtype=torch.int32)
    ids = torch.where(torch.arange(L)[None, :] < sl[:, None], ids, torch.zeros_like(ids))     # the feed pads with id 0
    ids[0, 0] = V - 1                                                                          # the last rank's last row
    tgt = torch.randint(0, V, (B,), generator=g, dtype=torch.int32)
    return ids, sl, tgt


def _toy_loss(E, W, ids, sl, tgt, reg, global_batch):
    """pred_b = tanh(sum_{t < len_b} E[id_bt] . W); cross entropy over the whole catalog (a mean over the GLOBAL
    batch) + reg * 1/2 sum over ALL slots of |E[id]|^2 -- the shape of Model/base_model.py:300-328."""
    L = ids.shape[1]
    rows = E[ids.long()]
    live = (torch.arange(L)[None, :] < sl[:, None]).to(E.dtype)
    pred = torch.tanh((rows * live[:, :, None]).sum(1) @ W)
    logits = pred @ E.t()
    ce = torch.logsumexp(logits, 1) - logits.gather(1, tgt.long()[:, None])[:, 0]
    return ce.sum() / global_batch + reg * 0.5 * (rows ** 2).sum(), pred, ce


def _scoring_worker(rank, world, port, out_dir, replicate_table, n_steps):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from mtamrecommender_amd import data_parallel
    D, B, L, V = 128, 5, 4, 203
    # the same replica on every rank: a [D, D] dense matrix at the head of the flat space, 37 small-table rows whose
    # gradient is random per rank (they only ride the small all-reduce), the item table
    path = _FakePath(n_dense=D * D, small_rows=37, item_rows=V, seed=1)
    path.world_size, path.L = world, L
    off = path.tab_off["item"]
    path.tables = {"item": path.flat_p[off:off + V * D].view(V, D)}
    path.g_tab = {"item": path.flat_g[off:off + V * D].view(V, D)}
    ref = _FakePath(n_dense=D * D, small_rows=37, item_rows=V, seed=1)
    ref.world_size = world
    small = slice(D * D, off)
    k = _TorchScoringKernels()
    lr = torch.tensor([1e-3])
    ex = data_parallel.ShardedScoringExchange(path, world, rank, kernels=k, replicate_table=replicate_table)
    assert ex.row_lo == rank * (path.item_rows_pad // world) and ex.row_hi <= V
    err = lambda a, b: float((a - b).abs().max() / (b.abs().max() + 1e-30))
    loss_err = ce_err = 0.0
    for step in range(n_steps):
        gens = [torch.Generator().manual_seed(500 + 10 * step + r) for r in range(world)]
        small_g = [torch.round(torch.randn(off - D * D, generator=g) * 0.02 * 4096) / 4096 for g in gens]
        every = [_toy_inputs(r, B, L, V, seed=7 + step) for r in range(world)]
        # ---- the replicated update: the whole batch on one replica
        E0 = ref.flat_p[off:off + V * D].view(V, D).clone().requires_grad_(True)
        W0 = ref.flat_p[:D * D].view(D, D).clone().requires_grad_(True)
        ids_all, sl_all, tgt_all = (torch.cat([e[i] for e in every]) for i in range(3))
        loss_ref, _, ce_ref = _toy_loss(E0, W0, ids_all, sl_all, tgt_all, ref.reg, world * B)
        loss_ref.backward()
        ref.flat_g.zero_()
        ref.flat_g[:D * D] = W0.grad.reshape(-1)
        ref.flat_g[small] = sum(small_g)
        ref.flat_g[off:off + V * D] = E0.grad.reshape(-1)
        sq = torch.zeros(1, dtype=torch.float64)
        k.sq_sum(ref.flat_g[:ref.n_total], 1.0, sq, False)
        k.clip_scale(sq, ref.clip, ref.scale, lr, ref.adam_state)
        k.adam(ref.flat_p[:ref.n_total], ref.flat_m[:ref.n_total], ref.flat_v[:ref.n_total], ref.flat_g[:ref.n_total],
               ref.scale, ref.adam_state, ref.n_dense)

        # ---- the row-sharded step on this rank's slice of the batch
        ids, sl, tgt = every[rank]
        bt = _ToyBatch()
        bt.B = B
        bt.feed = {"lr": lr, "target_item_id": tgt, "item_list": ids, "seq_length": sl}
        bt.loss, bt.lse, bt.ce, bt.d_pred = torch.zeros(3), torch.zeros(B), torch.zeros(B), torch.zeros(B, D)
        bt.item_rows = torch.zeros(B * L, D)
        path.flat_g.zero_()
        # the looked-up item rows: from this rank's replica, or ("sharded-table") from the owners of the rows
        if replicate_table:
            looked_up = path.tables["item"][ids.reshape(-1).long()].clone()
        else:
            ex.fetch_history_rows(bt)
            looked_up = bt.item_rows.clone()
        rows = looked_up.view(B, L, D).requires_grad_(True)
        W = path.flat_p[:D * D].view(D, D).clone().requires_grad_(True)
        live = (torch.arange(L)[None, :] < sl[:, None]).float()
        pred = torch.tanh((rows * live[:, :, None]).sum(1) @ W)           # forward to pred
        bt.pred = pred.detach().clone()
        bt.ic = torch.cat([rows.detach().reshape(B * L, D), torch.zeros(B * L, D)], 1).contiguous()
        ex.score(bt)
        pred.backward(bt.d_pred)                                           # backward from d_pred
        path.flat_g[:D * D] = W.grad.reshape(-1)
        path.flat_g[small] = small_g[rank]
        bt.d_ic = torch.cat([rows.grad.reshape(B * L, D), torch.zeros(B * L, D)], 1).contiguous()
        bt.l2 = 0.5 * float((rows.detach() ** 2).sum())            # tf.nn.l2_loss: half the sum of squares
        # the rank's own slots into its own rows, then the exchange applies the other ranks' slots
        k.scatter_items(bt.d_ic, bt.ic, ids.reshape(-1), sl, B, L, path.reg, path.g_tab["item"], (ex.row_lo, ex.row_hi))
        ex.exchange_and_apply(bt)
        loss_err = max(loss_err, abs(float(bt.loss[0]) - float(loss_ref.detach())) / float(loss_ref.detach()))
        ce_err = max(ce_err, err(bt.ce, ce_ref[rank * B:(rank + 1) * B].detach()))

    own = slice(ex.lo, ex.hi_true)
    foreign_stale = 0.0
    if not replicate_table:
        # the rows this rank does not own were never refreshed: its replica differs from the reference there ...
        mask = torch.ones(path.n_total, dtype=torch.bool)
        mask[:off] = False
        mask[own] = False
        foreign_stale = float((path.flat_p[:path.n_total][mask] != ref.flat_p[:ref.n_total][mask]).float().mean())
        assert not ex.table_current
        ex.sync_item_table()        # ... until the replica is brought up to date (evaluation, checkpoints)
        assert ex.table_current
    res = [err(path.flat_p[:path.n_total], ref.flat_p[:ref.n_total]), err(path.flat_m[own], ref.flat_m[own]),
           err(path.flat_v[own], ref.flat_v[own]), err(path.flat_m[:off], ref.flat_m[:off]),
           abs(float(path.scale[1]) - float(ref.scale[1])) / float(ref.scale[1]), loss_err, ce_err,
           float(not bool(path.flat_p[path.n_total:].any())), float(path.refreshed), foreign_stale]
    np.save(os.path.join(out_dir, "scoring_%d.npy" % rank), np.array(res))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 4])
@pytest.mark.parametrize("replicate_table", [True, False])
def test_sharded_scoring_exchange_equals_the_replicated_update(tmp_path, world, replicate_table):
    """Item table row-sharded for scoring: all-gather pred, every rank scores its own rows for the samples of every
    rank, the (max, sum-exp, target logit) triples and the d_pred shares are reduced, dE is born sharded, the other
    ranks' history slots are all-gathered and applied to the owned rows -- against ONE replica doing the whole batch
    (autograd over the full-catalog softmax + L2, clip, dense Adam), over TWO steps: parameters and both slots to fp32
    rounding (the sums run in a different order: per-rank partial log-sum-exps), the clip norm, the loss and the
    per-sample cross entropies; 203 item rows over 2 and 4 ranks (pad rows untouched), the same in-place collectives
    as RCCL.  replicate_table=False ("sharded-table"): the updated rows are NOT all-gathered -- the second step's
    lookups must get the rows the owners updated in the first (fetch_history_rows), the replicas' foreign rows are
    measurably stale before sync_item_table() and equal to the reference after it."""
    port = 29500 + (os.getpid() % 2000) + 11 * world + (3 if replicate_table else 0)
    mp.spawn(_scoring_worker, args=(world, port, str(tmp_path), replicate_table, 2), nprocs=world, join=True)
    for rank in range(world):
        p_err, m_err, v_err, ms_err, norm_err, loss_err, ce_err, pad_zero, refreshed, stale = np.load(
            os.path.join(str(tmp_path), "scoring_%d.npy" % rank))
        assert p_err < 4e-6 and m_err < 4e-5 and v_err < 4e-5 and ms_err < 4e-5, (rank, p_err, m_err, v_err, ms_err)
        assert norm_err < 1e-6 and loss_err < 1e-6 and ce_err < 1e-5, (rank, norm_err, loss_err, ce_err)
        assert pad_zero == 1.0 and refreshed == (2.0 if replicate_table else 3.0)
        assert (stale == 0.0) if replicate_table else (stale > 0.9)
