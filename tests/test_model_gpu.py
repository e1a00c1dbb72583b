"""End-to-end parity of the HIP training path against the CPU oracle (GPU box).

Same records, same injected weights into ``MTAM`` (HIP) and ``oracle.mtam_oracle``:
  * forward logits within the stated fp32 tolerance (5e-5 of max|logit|, vs the
    float64 oracle),
  * top-K index lists: bit-exact against the oracle's top_k of a k-ordered fmaf
    scoring of the HIP ``pred`` (the accumulation-order contract), and equal to
    the float64 oracle's lists wherever its K/K+1 margin exceeds the tolerance,
  * every gradient tensor, the TF-style global norm, and the parameters after a
    few Adam steps.
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

LOGIT_TOL = 5e-5      # relative to max |logit|
GRAD_TOL = 5e-4       # relative max-norm per gradient tensor (fp32 kernels vs fp64 oracle)


def build(tmp_path, B, L, NB, H, items=300, cats=17, users=40, seed=5, id_dist="zipf", model_name="MTAM",
          optimizer=None, score_dtype=None):
    from mtamrecommender_amd.config.model_parameter import model_parameter
    from mtamrecommender_amd.Embedding.Behavior_embedding_time_aware_attention import \
        Behavior_embedding_time_aware_attention
    from mtamrecommender_amd.Model.MTAMRec_model import MTAM
    from mtamrecommender_amd.Model.PISTRec_model import Time_Aware_self_Attention_model
    from mtamrecommender_amd.Model.base_model import Session
    from mtamrecommender_amd.data.synthetic import SyntheticCatalog, make_records
    FLAGS = model_parameter().get_parameter("MTAMb1_movielen").FLAGS
    FLAGS.num_blocks, FLAGS.num_heads, FLAGS.length_of_user_history = NB, H, L
    FLAGS.checkpoint_path_dir = str(tmp_path)
    if optimizer is not None:
        FLAGS.optimizer = optimizer
    if score_dtype is not None:
        FLAGS.score_dtype = score_dtype
    cat = SyntheticCatalog(items, cats, users, seed=seed)
    emb = Behavior_embedding_time_aware_attention(True, users, items, cats, L, seed=seed)
    from mtamrecommender_amd.Model import MTAMRec_model as family
    if model_name == "Time_Aware_Self_Attention_Model":
        from mtamrecommender_amd.Model.attention_baseline_models import Time_Aware_Self_Attention_Model as cls
    else:
        cls = Time_Aware_self_Attention_model if model_name == "PISTRec" else getattr(family, model_name)
    model = cls(FLAGS, emb, Session("cuda:0"))
    # make every bias / scale non-trivial so that all gradient paths are exercised
    rng = np.random.default_rng(seed)
    arrays = model.get_variables()
    for k, v in arrays.items():
        if v.ndim == 1 or v.shape[0] == 1:
            arrays[k] = (v + rng.normal(0, 0.05, v.shape)).astype(np.float32)
    model.set_variables(arrays)
    # the product's variable set against the ORACLE's own list (oracle/specs.py): same names, same shapes
    from oracle import specs as S
    spec_model = "PISTRec" if model_name in ("PISTRec", "Time_Aware_Self_Attention_Model") else model_name
    want = {v.name: tuple(v.shape) for v in S.model_vars(spec_model, users, items, cats, L, 128, NB)}
    assert {k: tuple(v.shape) for k, v in arrays.items()} == want
    records = make_records(cat, B, L, seed=seed + 1, id_dist=id_dist)
    return model, FLAGS, records


def rel(got, ref):
    ref = np.asarray(ref, np.float64)
    return float(np.abs(np.asarray(got, np.float64) - ref).max() / (np.abs(ref).max() + 1e-30))


@pytest.mark.parametrize("B,L,NB,H", [(6, 8, 1, 1), (33, 50, 2, 2), (128, 50, 1, 1)])
def test_forward_logits_and_topk(hip_lib, tmp_path, B, L, NB, H):
    import oracle.c_oracle as co
    import oracle.mtam_oracle as O
    model, FLAGS, records = build(tmp_path, B, L, NB, H)
    arrays = model.get_variables()
    feed = model.embedding.make_feed_dic_new(records)
    p = model.path
    bt = p.load_feed(feed)
    p.eval_kernels(bt, 50)
    logits = bt.logits.cpu().numpy()
    pred = bt.pred.cpu().numpy()
    top = bt.topk_idx.cpu().numpy()

    w = O.split_item_table(arrays, torch.float64, False)
    ref = O.forward("MTAM", w, O.feed_to_torch(feed, torch.float64), H, NB, FLAGS.regulation_rate)
    ref_logits = ref["logits"].numpy()
    assert rel(pred, ref["pred"].numpy()) < LOGIT_TOL
    assert rel(logits, ref_logits) < LOGIT_TOL

    # scoring contract: logits are a k-ordered fmaf chain of (pred, table) -> rankings bit-exact
    chain = co.score_fma(pred, arrays["embedding_layer/item"])
    assert np.array_equal(logits, chain)
    k = min(50, logits.shape[1])
    assert np.array_equal(top[:, :k], O.top_k(chain, k))

    # against the float64 oracle: identical lists wherever its ranking is not a near-tie
    ref_top = O.top_k(ref_logits, k)
    tol = LOGIT_TOL * np.abs(ref_logits).max() * 2
    srt = -np.sort(-ref_logits, axis=1)
    for b in range(B):
        gaps = srt[b, :k] - srt[b, 1:k + 1] if logits.shape[1] > k else srt[b, :k - 1] - srt[b, 1:k]
        safe = int(np.argmax(gaps < tol)) if np.any(gaps < tol) else len(gaps)
        assert np.array_equal(top[b, :safe], ref_top[b, :safe])

    # the reference's metric tuple through the public call
    got = model.metrics_topK(model.sess, records, 0, FLAGS.top_k)
    want = O.metrics_topK(chain, feed["target_item_id"])
    assert np.allclose(got, want, atol=1e-12)


@pytest.mark.parametrize("B,L,NB,H,tf_compat", [(6, 8, 1, 1, True), (33, 50, 2, 2, True), (128, 50, 1, 1, True),
                                                (33, 50, 2, 1, False)])
def test_train_step_gradients_and_update(hip_lib, tmp_path, B, L, NB, H, tf_compat):
    import oracle.mtam_oracle as O
    model, FLAGS, records = build(tmp_path, B, L, NB, H)
    model.use_graph = False
    p = model.path
    p.tf_compat = tf_compat
    arrays = {k: v.copy() for k, v in model.get_variables().items()}
    feed = model.embedding.make_feed_dic_new(records)

    out, grads, slot_sq = O.loss_and_grads("MTAM", arrays, feed, H, NB, FLAGS.regulation_rate, torch.float64)
    ref_norm = O.global_norm(grads, slot_sq, "MTAM", tf_compat)
    loss, summary = model.train(model.sess, records, 1e-3)
    assert abs(loss - float(out["loss"])) / abs(float(out["loss"])) < 2e-5
    assert abs(summary["l2_norm"] - float(out["l2"])) / float(out["l2"]) < 2e-5
    got = p.grads_tf()
    for name, g in grads.items():
        if g is None:
            continue
        assert rel(got[name], g) < GRAD_TOL, name
    assert abs(float(p.scale[1]) - ref_norm) / ref_norm < 1e-4
    c = FLAGS.max_gradient_norm
    assert abs(float(p.scale[0]) - c * min(1 / ref_norm, 1 / c)) < 1e-5


def test_three_adam_steps_track_the_oracle(hip_lib, tmp_path):
    import oracle.mtam_oracle as O
    B, L, NB, H = 48, 50, 2, 1
    model, FLAGS, records = build(tmp_path, B, L, NB, H)
    arrays = {k: v.copy() for k, v in model.get_variables().items()}
    state = O.AdamState(arrays)
    lr = 1e-3
    for step in range(3):
        batch = records[step * 16:(step + 1) * 16]
        feed = model.embedding.make_feed_dic_new(batch)
        ref = O.train_step("MTAM", arrays, state, feed, lr, H, NB, FLAGS.regulation_rate,
                           FLAGS.max_gradient_norm, True)
        loss, _ = model.train(model.sess, batch, lr)          # step 3 replays a captured hipGraph
        assert abs(loss - ref["loss"]) / abs(ref["loss"]) < 1e-4, step
    got = model.get_variables()
    for name, want in arrays.items():
        d = np.abs(got[name].astype(np.float64) - want)
        # Adam moves every touched weight by ~lr per step; a sign flip of a rounding-level
        # gradient is the only legitimate source of a visible difference.
        assert d.max() <= 2.1 * lr * 3, name
        assert (d > 2e-5).mean() < 2e-3, name


@pytest.mark.parametrize("switch", ["MTAM_CLIP_IN_ADAM", "MTAM_NORM_RIDER", "score32_fused"])
def test_step_switches_give_the_same_training(hip_lib, tmp_path, monkeypatch, switch):
    """The step's merged launches against the forms they replace, through model.train(): the clip scale formed inside
    the optimizer launch (MTAM_CLIP_IN_ADAM=0: ticket launch + Adam), its partial pass riding in the scatter-add launch
    (MTAM_NORM_RIDER=0: a launch of its own), training's scoring as one launch (off: lse + finish + backward).  Same
    records, same weights, five steps (eager, capture, replays): the same losses and parameters to fp32 rounding."""
    from mtamrecommender_amd import hip_ops as ops
    B, L = 128, 50
    runs = []
    for off in (False, True):
        if off and switch == "score32_fused":
            ops.score32_set_fused(False)
        elif off:
            monkeypatch.setenv(switch, "0")
        try:
            model, FLAGS, records = build(tmp_path, B, L, 1, 1, items=3706)
            assert ops.score32_train_is_fused(B, 3709) == (not (off and switch == "score32_fused"))
            losses = [model.train(model.sess, records, 1e-3)[0] for _ in range(5)]
            runs.append((losses, model.get_variables(), float(model.path.scale[1])))
        finally:
            ops.score32_set_fused(True)
            monkeypatch.delenv(switch, raising=False)
    (la, va, na), (lb, vb, nb) = runs
    assert all(abs(x - y) <= 2e-5 * abs(x) for x, y in zip(la, lb)), (la, lb)
    assert abs(na - nb) <= 1e-4 * na                  # the last step's gradient norm
    for name, want in va.items():
        d = np.abs(vb[name].astype(np.float64) - want)
        assert d.max() <= 2.1 * 1e-3 * 5 and (d > 2e-5).mean() < 2e-3, name


def test_loss_decreases_and_recall_rises(hip_lib, tmp_path):
    """Loss-curve smoke test: 60 steps on 512 synthetic records."""
    B, L = 64, 20
    model, FLAGS, records = build(tmp_path, 512, L, 1, 1, items=200, cats=11, users=60)
    first = last = None
    for epoch in range(8):
        for s in range(0, 512, B):
            loss, _ = model.train(model.sess, records[s:s + B], 3e-3)
            first = loss if first is None else first
            last = loss
    assert last < 0.8 * first
    assert model.recall_at(model.sess, records[:B], 20) > 0.2


def test_checkpoint_round_trip(hip_lib, tmp_path):
    model, FLAGS, records = build(tmp_path, 16, 8, 1, 1)
    model.train(model.sess, records, 1e-3)
    before = model.get_variables()
    model.save(model.sess, global_step=1)
    model.train(model.sess, records, 1e-3)
    model.restore(model.sess, str(tmp_path))
    after = model.get_variables()
    for k in before:
        assert np.array_equal(before[k], after[k]), k


def test_optimizer_state_is_saved_over_the_true_parameter_space(hip_lib, tmp_path):
    """The checkpoint holds Adam's slots over [:n_total] only -- not the item pad rows or the data-parallel loss tail
    behind them -- and restore() takes a state of that length as well as one written when the whole allocation was
    saved (longer: its head is the state)."""
    model, FLAGS, records = build(tmp_path, 16, 8, 1, 1)
    p = model.path
    model.train(model.sess, records, 1e-3)
    st = p.optimizer_state()
    assert st["flat_m"].numel() == p.n_total == st["flat_v"].numel() and p.n_alloc > p.n_total
    m, v = p.flat_m.clone(), p.flat_v.clone()
    model.save(model.sess, global_step=1)
    model.train(model.sess, records, 1e-3)
    p.flat_g[p.n_total:].fill_(7.0)             # (what an all-reduced loss tail leaves behind: never state)
    model.restore(model.sess, str(tmp_path))
    assert torch.equal(p.flat_m, m) and torch.equal(p.flat_v, v)
    old_style = {"flat_m": torch.cat([st["flat_m"], torch.full((p.n_alloc - p.n_total,), 3.0)]),
                 "flat_v": torch.cat([st["flat_v"], torch.full((p.n_alloc - p.n_total,), 3.0)]),
                 "adam_state": st["adam_state"]}
    p.load_optimizer_state(old_style)
    assert torch.equal(p.flat_m, m) and torch.equal(p.flat_v, v)        # the tail of an old file is not state
    with pytest.raises(ValueError):
        p.load_optimizer_state({"flat_m": st["flat_m"][:-1], "flat_v": st["flat_v"], "adam_state": st["adam_state"]})


@pytest.mark.parametrize("dp_mode", ["fused", "split"])
def test_data_parallel_code_path_single_rank(hip_lib, tmp_path, dp_mode):
    """world_size = 1 over RCCL: the data-parallel step -- one graph with the all-reduce captured inside
    ("fused") or graph A -> all-reduce -> graph B ("split") -- must equal the single-GPU one, up to the
    clip norm (data parallelism clips by the true norm, see data_parallel.py)."""
    import torch.distributed as dist
    from mtamrecommender_amd import data_parallel
    os_env = __import__("os").environ
    os_env.setdefault("MASTER_ADDR", "127.0.0.1")
    os_env["MASTER_PORT"] = "29617" if dp_mode == "fused" else "29618"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        model_a, FLAGS, records = build(tmp_path, 32, 50, 1, 1)
        model_b, _, _ = build(tmp_path, 32, 50, 1, 1)
        model_a.path.tf_compat = False
        model_b._dp_mode = dp_mode
        data_parallel.attach(model_b.path, 1, force=True)
        data_parallel.broadcast_parameters(model_b.path)
        for step in range(4):
            la, _ = model_a.train(model_a.sess, records, 1e-3)
            lb, _ = model_b.train(model_b.sess, records, 1e-3)   # (its loss travelled through the all-reduce's tail)
            assert abs(la - lb) <= 1e-6 * abs(la), step
        assert model_b.path.loss_in_tail and model_b.path.batch(32).loss.data_ptr() == model_b.path.loss_tail.data_ptr()
        assert model_b._dp_mode == dp_mode                  # "fused" did not fall back
        assert (("train_dp", 32, None) in model_b._graphs) == (dp_mode == "fused")
        va, vb = model_a.get_variables(), model_b.get_variables()
        for k in va:
            assert np.abs(va[k] - vb[k]).max() <= 2.1e-3 * 4, k
            assert (np.abs(va[k] - vb[k]) > 2e-5).mean() < 2e-3, k
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("dp_mode", ["fused", "split"])
def test_feed_ring_under_the_flat_exchange_single_rank(hip_lib, tmp_path, dp_mode):
    """The flat data-parallel step (one-rank RCCL group, both graph forms) fed from the HBM ring -- the update graph
    behind the all-reduce carries the hand-over of the next feed -- against the same step with the arena copied in
    front of it: the same loss at each of 9 steps over a ring of 3, the arena holding the next slot after every step,
    the same parameters afterwards.  The row-sharded exchanges refuse a ring."""
    import torch.distributed as dist
    from mtamrecommender_amd import data_parallel
    os_env = __import__("os").environ
    os_env.setdefault("MASTER_ADDR", "127.0.0.1")
    os_env["MASTER_PORT"] = "29621" if dp_mode == "fused" else "29622"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        B, L, n_slots = 32, 50, 3
        model_a, FLAGS, records = build(tmp_path, n_slots * B, L, 1, 1)
        model_b, _, _ = build(tmp_path, n_slots * B, L, 1, 1)
        for m in (model_a, model_b):
            m._dp_mode = dp_mode
            data_parallel.attach(m.path, 1, force=True, exchange="flat")
            data_parallel.broadcast_parameters(m.path)
        pa, pb = model_a.path, model_b.path
        feeds = [model_a.embedding.make_feed_dic_new(records[i * B:(i + 1) * B]) for i in range(n_slots)]
        staged = [pa.stage(f, 1e-3 * (1 + i)) for i, f in enumerate(feeds)]
        bta, btb = pa.batch(B), pb.batch(B)
        assert pb.ring_supported(btb)
        ring = pb.feed_ring(btb, n_slots)
        for i, st in enumerate(staged):
            ring.put(i, st)
        ring.prime(0)
        for k in range(9):
            bta.arena.copy_(staged[k % n_slots])
            model_a.step_train(bta)
            model_b.step_train(btb)
            la, lb = float(bta.loss[0].item()), float(btb.loss[0].item())
            assert abs(la - lb) <= 2e-5 * abs(la), (k, la, lb)
            assert torch.equal(btb.arena, ring.slots[(k + 1) % n_slots]), k
            assert int(ring.cursor.item()) == k + 2 and ring.consumed == k + 1
        assert model_b._dp_mode == dp_mode
        va, vb = model_a.get_variables(), model_b.get_variables()
        for k in va:
            assert np.abs(va[k] - vb[k]).max() <= 2e-4 * max(1.0, np.abs(va[k]).max()), k
        model_c, _, _ = build(tmp_path, B, L, 1, 1)
        data_parallel.attach(model_c.path, 1, force=True, exchange="sharded")
        assert not model_c.path.ring_supported(model_c.path.batch(B))
        with pytest.raises(RuntimeError):
            model_c.path.feed_ring(model_c.path.batch(B), 2)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("score_dtype", ["f32", "bf16"])
def test_sharded_item_exchange_single_rank(hip_lib, tmp_path, score_dtype):
    """data_parallel.ShardedItemExchange (reduce-scatter by row range, shard-owned clip share and Adam, all-gather)
    through the real kernels and a one-rank RCCL group: the same losses and parameters as the single-GPU step with
    the true clip norm.  1,003 item rows: the padded tail of the flat buffers stays zero."""
    import torch.distributed as dist
    from mtamrecommender_amd import data_parallel
    os_env = __import__("os").environ
    os_env.setdefault("MASTER_ADDR", "127.0.0.1")
    os_env["MASTER_PORT"] = "29619" if score_dtype == "f32" else "29620"
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        kw = dict(items=1000, score_dtype=score_dtype)
        model_a, FLAGS, records = build(tmp_path, 32, 50, 1, 1, **kw)
        model_b, _, _ = build(tmp_path, 32, 50, 1, 1, **kw)
        model_a.path.tf_compat = False
        data_parallel.attach(model_b.path, 1, force=True, shard_items=True)
        assert model_b.path.sharded is not None and model_b.path.item_rows_pad == 1008
        data_parallel.broadcast_parameters(model_b.path)
        for step in range(4):
            la, _ = model_a.train(model_a.sess, records, 1e-3)
            lb, _ = model_b.train(model_b.sess, records, 1e-3)
            # (bf16: an update that differs in the last fp32 bit can flip the rounding of a scoring-copy entry)
            assert abs(la - lb) <= (2e-6 if score_dtype == "f32" else 3e-5) * abs(la), step
        pa, pb = model_a.path, model_b.path
        assert abs(float(pa.scale[1]) - float(pb.scale[1])) <= 1e-5 * float(pa.scale[1])
        va, vb = model_a.get_variables(), model_b.get_variables()
        for k in va:
            assert np.abs(va[k] - vb[k]).max() <= 2.1e-3 * 4, k
            assert (np.abs(va[k] - vb[k]) > 2e-5).mean() < 2e-3, k
        for flat in (pb.flat_p, pb.flat_g, pb.flat_m, pb.flat_v):
            assert not bool(flat[pb.n_total:].any())
        if score_dtype == "bf16":
            assert torch.equal(pb.item16.view(torch.int16), pb.tables["item"].bfloat16().view(torch.int16))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("graph,exchange", [(True, "sharded-scoring"), (False, "sharded-scoring"), (True, "sharded-table"),
                                            (False, "sharded-table")])
def test_sharded_scoring_exchange_single_rank(hip_lib, tmp_path, graph, exchange):
    """data_parallel.ShardedScoringExchange (item table row-sharded for SCORING: all-gather pred, ranged lse /
    backward passes, reduced (max, sum-exp, logit) and d_pred, slot exchange, shard-owned update, all-gather -- or,
    "sharded-table", no all-gather and the lookups served from rows fetched from their owners) through
    the real kernels and a one-rank RCCL group -- forward-to-pred and backward-from-d_pred as two hipGraphs with the
    scoring passes and their collectives between them -- against the single-GPU step with the true clip norm."""
    import torch.distributed as dist
    from mtamrecommender_amd import data_parallel
    os_env = __import__("os").environ
    os_env.setdefault("MASTER_ADDR", "127.0.0.1")
    os_env["MASTER_PORT"] = str(29631 + int(graph) + 2 * int(exchange == "sharded-table"))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
    try:
        model_a, FLAGS, records = build(tmp_path, 32, 50, 1, 1, items=1000)
        model_b, _, _ = build(tmp_path, 32, 50, 1, 1, items=1000)
        model_a.path.tf_compat = False
        model_a.use_graph = model_b.use_graph = graph
        data_parallel.attach(model_b.path, 1, force=True, exchange=exchange)
        ex = model_b.path.sharded_scoring
        assert ex is not None and (ex.row_lo, ex.row_hi) == (0, 1003) and model_b.path.dp_exchange == exchange
        assert ex.replicate_table == (exchange == "sharded-scoring")      # "sharded-table": lookups from fetched rows
        data_parallel.broadcast_parameters(model_b.path)
        for step in range(4):
            la, _ = model_a.train(model_a.sess, records, 1e-3)
            lb, _ = model_b.train(model_b.sess, records, 1e-3)
            assert abs(la - lb) <= 2e-6 * abs(la), (step, la, lb)
        pa, pb = model_a.path, model_b.path
        assert abs(float(pa.scale[1]) - float(pb.scale[1])) <= 1e-5 * float(pa.scale[1])
        va, vb = model_a.get_variables(), model_b.get_variables()
        for k in va:
            assert np.abs(va[k] - vb[k]).max() <= 2.1e-3 * 4, k
            assert (np.abs(va[k] - vb[k]) > 2e-5).mean() < 2e-3, k
        for flat in (pb.flat_p, pb.flat_g, pb.flat_m, pb.flat_v):
            assert not bool(flat[pb.n_total:].any())
        # evaluation brings a "sharded-table" replica up to date first (a collective inside metrics_topK)
        assert model_b.recall_at(model_b.sess, records, 20) >= 0.0 and ex.table_current
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ PISTRec
@pytest.mark.parametrize("B,L,NB,H", [(5, 10, 1, 1), (24, 50, 2, 2), (16, 100, 1, 4)])
def test_pistrec_forward_and_gradients(hip_lib, tmp_path, B, L, NB, H):
    """Time_Aware_self_Attention_model: logits, loss (no user L2 term), every gradient, clip norm."""
    import oracle.c_oracle as co
    import oracle.mtam_oracle as O
    model, FLAGS, records = build(tmp_path, B, L, NB, H, model_name="PISTRec")
    model.use_graph = False
    p = model.path
    arrays = {k: v.copy() for k, v in model.get_variables().items()}
    feed = model.embedding.make_feed_dic_new(records)
    bt = p.load_feed(feed)
    p.eval_kernels(bt, 50)
    logits, pred = bt.logits.cpu().numpy(), bt.pred.cpu().numpy()
    out, grads, slot_sq = O.loss_and_grads("PISTRec", arrays, feed, H, NB, FLAGS.regulation_rate, torch.float64)
    assert rel(logits, out["logits"].detach().numpy()) < LOGIT_TOL
    chain = co.score_fma(pred, arrays["embedding_layer/item"])
    assert np.array_equal(logits, chain)
    k = min(50, logits.shape[1])
    assert np.array_equal(bt.topk_idx.cpu().numpy()[:, :k], O.top_k(chain, k))

    loss, summary = model.train(model.sess, records, 1e-3)
    ref_loss = float(out["loss"].detach())
    assert abs(loss - ref_loss) / abs(ref_loss) < 2e-5
    got = p.grads_tf()
    for name, g in grads.items():
        if g is None:
            assert name not in got or not np.any(got[name]), name      # user table: no gradient
            continue
        assert rel(got[name], g) < GRAD_TOL, name
    ref_norm = O.global_norm(grads, slot_sq, "PISTRec", True)
    assert abs(float(p.scale[1]) - ref_norm) / ref_norm < 1e-4


def test_time_aware_self_attention_model_has_the_user_l2_term(hip_lib, tmp_path):
    """experiment_type 'Time_Aware_Self_Attention_Model' (Model/attention_baseline_models.py:47-65): PISTRec's encoder
    under base_model.output() -- the loss and the clip norm include the user rows, the user table gets a gradient."""
    import oracle.mtam_oracle as O
    B, L, NB, H = 24, 50, 2, 2
    model, FLAGS, records = build(tmp_path, B, L, NB, H, model_name="Time_Aware_Self_Attention_Model")
    model.use_graph = False
    p = model.path
    arrays = {k: v.copy() for k, v in model.get_variables().items()}
    feed = model.embedding.make_feed_dic_new(records)
    out, grads, slot_sq = O.loss_and_grads(O.TASA, arrays, feed, H, NB, FLAGS.regulation_rate, torch.float64)
    plain, _, _ = O.loss_and_grads("PISTRec", arrays, feed, H, NB, FLAGS.regulation_rate, torch.float64)
    assert float(out["loss"].detach()) > float(plain["loss"].detach())
    loss, _ = model.train(model.sess, records, 1e-3)
    ref_loss = float(out["loss"].detach())
    assert abs(loss - ref_loss) / abs(ref_loss) < 2e-5
    got = p.grads_tf()
    assert grads["embedding_layer/user"] is not None and np.any(got["embedding_layer/user"])
    for name, g in grads.items():
        if g is not None:
            assert rel(got[name], g) < GRAD_TOL, name
    ref_norm = O.global_norm(grads, slot_sq, O.TASA, True)
    assert abs(float(p.scale[1]) - ref_norm) / ref_norm < 1e-4


def test_pistrec_adam_steps_track_the_oracle(hip_lib, tmp_path):
    import oracle.mtam_oracle as O
    B, L, NB, H = 32, 50, 2, 1
    model, FLAGS, records = build(tmp_path, B, L, NB, H, model_name="PISTRec")
    arrays = {k: v.copy() for k, v in model.get_variables().items()}
    state = O.AdamState(arrays)
    for step in range(3):
        feed = model.embedding.make_feed_dic_new(records)
        ref = O.train_step("PISTRec", arrays, state, feed, 1e-3, H, NB, FLAGS.regulation_rate,
                           FLAGS.max_gradient_norm, True)
        loss, _ = model.train(model.sess, records, 1e-3)
        assert abs(loss - ref["loss"]) / abs(ref["loss"]) < 1e-4, step
    got = model.get_variables()
    for name, want in arrays.items():
        dd = np.abs(got[name].astype(np.float64) - want)
        assert dd.max() <= 2.1e-3 * 3, name
        assert (dd > 2e-5).mean() < 2e-3, name


@pytest.mark.parametrize("optimizer", ["sgd", "adadelta", "rmsprop"])
def test_other_optimizers_track_the_oracle(hip_lib, tmp_path, optimizer):
    """Model/base_model.py:71-80's other branches: three steps vs the oracle's TF-formula restatement."""
    import oracle.mtam_oracle as O
    B, L, NB, H = 24, 20, 1, 1
    model, FLAGS, records = build(tmp_path, 3 * B, L, NB, H,
                                  optimizer=optimizer if optimizer != "sgd" else "anything_else_is_sgd")
    assert model.opt == optimizer and model.path.optimizer == optimizer
    arrays = {k: v.copy() for k, v in model.get_variables().items()}
    state = O.SlotState(optimizer, arrays)
    lr = {"sgd": 0.5, "adadelta": 1.0, "rmsprop": 1e-3}[optimizer]
    for step in range(3):
        batch = records[step * B:(step + 1) * B]
        feed = model.embedding.make_feed_dic_new(batch)
        ref = O.train_step("MTAM", arrays, state, feed, lr, H, NB, FLAGS.regulation_rate,
                           FLAGS.max_gradient_norm, True)
        loss, _ = model.train(model.sess, batch, lr)
        assert abs(loss - ref["loss"]) / abs(ref["loss"]) < 1e-4, step
    got = model.get_variables()
    for name, want in arrays.items():
        d = np.abs(got[name].astype(np.float64) - want)
        scale = np.abs(want).max()
        if optimizer == "rmsprop":
            # sign-like update (lr * g / sqrt(ms)): a rounding-level gradient may flip a step
            assert d.max() <= 3 * 3.2 * lr * 2 and (d > 2e-5).mean() < 2e-3, name
        else:
            assert d.max() <= 2e-4 * scale, name


def test_native_feed_equals_python_feed(hip_lib, tmp_path):
    """libmtam_host.so's packed arenas drive the same steps as make_feed_dic_new through train /
    metrics_topK / recall_at.  The arenas are bit-identical (tests/test_native_input.py); two runs of a
    training step agree only to float-atomic rounding (scatter-add and split-K sums are order dependent)."""
    from mtamrecommender_amd.DataHandle.native_input import BatchPacker, NativeDataInput, RecordSet
    B, L = 16, 20
    model_a, FLAGS, records = build(tmp_path, 3 * B, L, 1, 1)
    model_b, _, _ = build(tmp_path, 3 * B, L, 1, 1)
    rs = RecordSet.from_records(records)
    packer = BatchPacker(model_b.path, model_b.embedding)
    for step, packed in NativeDataInput(rs, B, packer):
        batch = records[(step - 1) * B:step * B]
        la, _ = model_a.train(model_a.sess, batch, 1e-3)
        lb, _ = model_b.train(model_b.sess, packed, 1e-3)
        assert abs(la - lb) <= 1e-5 * abs(la), step
    model_b.set_variables(model_a.get_variables())        # identical weights: eval is deterministic
    packed = packer.pack(rs, list(range(B)))
    assert model_a.metrics_topK(model_a.sess, records[:B], 0, 20) == model_b.metrics_topK(model_b.sess, packed, 0, 20)
    assert model_a.recall_at(model_a.sess, records[:B], 20) == model_b.recall_at(model_b.sess, packed, 20)


@pytest.mark.parametrize("async_loss", [False, True])
def test_feed_and_loss_copies_inside_the_graph(hip_lib, tmp_path, async_loss):
    """A PackedBatch step is ONE graph launch: the host -> device copy of the pinned feed arena is the graph's first
    node (one graph per arena of the stream's pool of three), the loss's device -> host copy its last under
    async_loss.  12 steps over 4 distinct batches (every arena and every loss slot replayed several times, with new
    contents each time) against a model fed Python lists with the copies outside the graph: the same loss at every
    step -- each exactly once under async_loss -- and the same parameters afterwards."""
    from mtamrecommender_amd.DataHandle.native_input import BatchPacker, NativeDataInput, RecordSet
    B, L, steps = 16, 20, 12
    model_a, FLAGS, records = build(tmp_path, 4 * B, L, 1, 1)
    model_b, _, _ = build(tmp_path, 4 * B, L, 1, 1)
    rs = RecordSet.from_records(records)
    packer = BatchPacker(model_b.path, model_b.embedding)
    model_b.async_loss = async_loss
    want, got, done = [], {}, 0
    while done < steps:
        for i, packed in NativeDataInput(rs, B, packer, consumer="t"):
            assert model_b._feed_in_graph(packed) and not model_a._feed_in_graph(records[:B])
            want.append(model_a.train(model_a.sess, records[(i - 1) * B:i * B], 1e-3 * (1 + done % 3), global_step=done)[0])
            loss, summary = model_b.train(model_b.sess, packed, 1e-3 * (1 + done % 3), global_step=done)
            if summary["loss_step"] is not None:
                assert summary["loss_step"] not in got
                got[summary["loss_step"]] = (loss, summary["Learning_rate"])
            done += 1
            if done == steps:
                break
    last = model_b.drain_loss()
    if async_loss:
        got[last[1]["loss_step"]] = (last[0], last[1]["Learning_rate"])
    else:
        assert last is None
    assert sorted(got) == list(range(steps))
    for s_ in range(steps):
        assert abs(got[s_][0] - want[s_]) <= 2e-5 * abs(want[s_]), (s_, got[s_], want[s_])
        assert abs(got[s_][1] - 1e-3 * (1 + s_ % 3)) < 1e-12            # the learning rate travelled with its step
    keys = [k for k in model_b._graphs if k[0] == "train_feed" and k[-1] != "warm"]
    assert len(keys) == 3 and len({k[3] for k in keys}) == 3            # one captured graph per pinned arena
    va, vb = model_a.get_variables(), model_b.get_variables()
    for k in va:
        assert np.abs(va[k] - vb[k]).max() <= 2e-4 * max(1.0, np.abs(va[k]).max()), k


@pytest.mark.parametrize("model_name", ["MTAM", "PISTRec", "MTAM_with_T_SeqRec"])
def test_feed_ring_steps_equal_steps_fed_by_copies(hip_lib, tmp_path, model_name):
    """A training step fed from a ring of HBM-resident packed feeds (path.feed_ring: the optimizer launch of step k
    copies slot k + 1 into the arena, nothing stands in front of the graph) against the same steps with the arena
    copied in front of each: 11 steps over a ring of 4 slots with a learning rate per slot -- the same loss at every
    step, the arena holding the NEXT slot after every step, the cursor in step with the host's count, ONE captured
    graph, the same parameters afterwards.  Refused without prime(), with another optimizer and under data parallelism."""
    B, L, n_slots, steps = 16, 20, 4, 11
    model_a, FLAGS, records = build(tmp_path, n_slots * B, L, 1, 1, model_name=model_name)
    model_b, _, _ = build(tmp_path, n_slots * B, L, 1, 1, model_name=model_name)
    pa, pb = model_a.path, model_b.path
    feeds = [model_a.embedding.make_feed_dic_new(records[i * B:(i + 1) * B]) for i in range(n_slots)]
    staged = [pa.stage(f, 1e-3 * (1 + i)) for i, f in enumerate(feeds)]
    bta, btb = pa.batch(B), pb.batch(B)
    ring = pb.feed_ring(btb, n_slots)
    with pytest.raises(RuntimeError):
        model_b.step_train(btb)                       # not primed
    for i, st in enumerate(staged):
        ring.put(i, st.to(btb.arena.device))
    ring.prime(0)
    for k in range(steps):
        bta.arena.copy_(staged[k % n_slots])
        model_a.step_train(bta)
        model_b.step_train(btb)
        la, lb = float(bta.loss[0].item()), float(btb.loss[0].item())
        assert abs(la - lb) <= 2e-5 * abs(la), (k, la, lb)
        assert torch.equal(btb.arena, ring.slots[(k + 1) % n_slots]), k
        assert int(ring.cursor.item()) == k + 2 and ring.consumed == k + 1
    assert len([k_ for k_ in model_b._graphs if k_[0] == "train" and k_[-1] != "warm"]) == 1
    va, vb = model_a.get_variables(), model_b.get_variables()
    for k in va:
        assert np.abs(va[k] - vb[k]).max() <= 2e-4 * max(1.0, np.abs(va[k]).max()), k
    # an ordinary step on the batch the ring is attached to (train() on a list of records puts its own feed into the
    # arena): it runs without the hand-over, and the ring's next step puts its slot back
    cursor = int(ring.cursor.item())
    model_b.train(model_b.sess, records[:B], 1e-3)
    assert not ring.primed and ring.taken and int(ring.cursor.item()) == cursor
    ring.prime(2)
    model_b.step_train(btb)
    assert torch.equal(btb.arena, ring.slots[3 % n_slots]) and int(ring.cursor.item()) == 4
    # detached again: the ordinary route, its own graph
    btb.feed_ring = None
    btb.arena.copy_(staged[0])
    model_b.step_train(btb)
    model_c, _, _ = build(tmp_path, B, L, 1, 1, optimizer="sgd")
    with pytest.raises(RuntimeError):
        model_c.path.feed_ring(model_c.path.batch(B), 2)


@pytest.mark.parametrize("native", [True, False])
def test_trainer_loop_runs_on_both_feeds(hip_lib, tmp_path, native):
    from mtamrecommender_amd.data.synthetic import SyntheticCatalog, make_records
    from mtamrecommender_amd.train_process import Train_main_process
    cat = SyntheticCatalog(120, 9, 30, seed=2)
    train, test = make_records(cat, 200, 20, seed=3), make_records(cat, 40, 20, seed=4)
    argv = ["--length_of_user_history", "20", "--train_batch_size", "32", "--test_batch_size", "16",
            "--max_epochs", "1", "--eval_freq", "4", "--checkpoint_path_dir", str(tmp_path),
            "--native_input", "true" if native else "false"]
    t = Train_main_process("MTAMb1_movielen", argv, train_set=train, test_set=test,
                           counts=dict(user_count=30, item_count=120, category_count=9))
    t.train(max_steps=6)
    assert t.global_step == 6


@pytest.mark.parametrize("test_batch", [16, 32])
def test_resident_epoch_trainer_equals_the_streamed_one(hip_lib, tmp_path, test_batch):
    """FLAGS.resident_epoch: every full batch of an epoch packed into HBM up front, the optimizer launch of step k
    handing step k + 1 its feed, against the same trainer fed batch by batch: two epochs of 6 full batches + a partial
    one (200 records / 32), evaluation every 4 steps -- with a test batch of the TRAINING batch's size the evaluation
    takes the ring's arena and the ring must put its slot back -- the same losses under the same steps, the same
    learning rates, the same parameters afterwards (float-atomic rounding apart)."""
    import random
    from mtamrecommender_amd.data.synthetic import SyntheticCatalog, make_records
    from mtamrecommender_amd.train_process import Train_main_process
    cat = SyntheticCatalog(120, 9, 30, seed=2)
    train, test = make_records(cat, 200, 20, seed=3), make_records(cat, 40, 20, seed=4)
    runs = []
    for resident in (False, True):
        d = tmp_path / ("r%d" % resident)
        d.mkdir()
        argv = ["--length_of_user_history", "20", "--train_batch_size", "32", "--test_batch_size", str(test_batch),
                "--max_epochs", "2", "--eval_freq", "4", "--checkpoint_path_dir", str(d), "--native_input", "true",
                "--resident_epoch", "true" if resident else "false"]
        t = Train_main_process("MTAMb1_movielen", argv, train_set=list(train), test_set=list(test),
                               counts=dict(user_count=30, item_count=120, category_count=9))
        logged = {}
        random.seed(5)
        t.build_model()
        built = t.model
        if runs:
            built.set_variables(start)          # both trainers start from the same parameters
        else:
            start = built.get_variables()
        t.build_model = lambda: None
        built.train_writer.add_summary = lambda summary, step, _l=logged: _l.setdefault(step, dict(summary)) \
            if "Training Loss" in summary else None
        t.train()
        assert t.global_step == 14 and t._resident_epoch_on() == resident
        runs.append((logged, built.get_variables(), built))
    (la, va, _), (lb, vb, mb) = runs
    assert sorted(la) == sorted(lb) == list(range(14))
    for step in la:
        assert abs(la[step]["normalized Training Loss"] - lb[step]["normalized Training Loss"]) <= \
            3e-5 * abs(la[step]["normalized Training Loss"]), step
        assert la[step]["Learning_rate"] == lb[step]["Learning_rate"]
    for k in va:
        assert np.abs(va[k] - vb[k]).max() <= 3e-4 * max(1.0, np.abs(va[k]).max()), k
    ring = mb.path.batch(32).feed_ring
    assert ring is not None and ring.n == 6
    # handles out of order (a skipped step): the slot goes in by a copy, the hand-over carries on from there
    handles = mb.load_resident_epoch(t._train_rs, np.arange(192, dtype=np.int64), 32, [1e-3] * 6, t._packer)
    for k in (0, 1, 4, 5, 2):
        mb.train(mb.sess, handles[k], 1e-3)
        assert ring.consumed == k + 1 and int(ring.cursor.item()) == k + 2
        assert torch.equal(ring.bt.arena, ring.slots[(k + 1) % 6])
    mb.drain_loss()
    keys = [k for k in mb._graphs if k[0] == "train_ring" and k[-1] != "warm"]
    assert 1 <= len(keys) <= 3          # one captured graph per pinned loss slot, none per batch


@pytest.mark.parametrize("model_name,NB", [("MTAM", 2), ("MTAM_via_T_GRU", 1), ("PISTRec", 2)])
def test_dead_variables_get_no_gradient_and_no_update(hip_lib, tmp_path, model_name, NB):
    """The variables the reference declares and never reads (oracle/specs.py ``live=False``: 6 GRU vectors,
    time_output_w3 per block) receive a None gradient in TF: clip_by_global_norm and
    apply_gradients skip them.  Here: they are not part of the gradient the step produces, the clip norm equals the
    norm over the LIVE variables, and three optimizer steps leave them bit-identical -- while every live variable
    that the oracle gives a non-zero gradient moves."""
    import oracle.mtam_oracle as O
    from oracle import specs as S
    model, FLAGS, records = build(tmp_path, 6, 8, NB, 1, model_name=model_name)
    vars_ = S.model_vars(model_name, 40, 300, 17, 8, 128, NB)
    dead, live = S.dead_names(vars_), S.live_names(vars_)
    assert len(dead) == (NB if model_name == "PISTRec" else 6 + NB)
    before = {k: v.copy() for k, v in model.get_variables().items()}
    feed = model.embedding.make_feed_dic_new(records)
    out, grads, slot_sq = O.loss_and_grads(model_name, before, feed, 1, NB, FLAGS.regulation_rate, torch.float64)
    assert sorted(k for k in dead) == sorted(k for k, g in grads.items() if g is None and k in dead)
    for _ in range(3):
        model.train(model.sess, records, 1e-3)
    got = model.path.grads_tf()
    assert not (set(got) & set(dead)), "a dead variable was given a gradient"
    after = model.get_variables()
    for k in dead:
        assert np.array_equal(before[k], after[k]), k
    for k in live:
        if grads.get(k) is not None and np.abs(grads[k]).max() > 0:
            assert not np.array_equal(before[k], after[k]), k


@pytest.mark.parametrize("optimizer", ["adam", "sgd"])
def test_weight_images_follow_the_weights(hip_lib, tmp_path, optimizer):
    """The bf16 operand images the forward's fused projection kernel reads (dense4emb/w, kv/w, gru/wx) are derived
    state: after optimizer steps (Adam re-writes them in its own launch; the other optimizers refresh them behind
    theirs), after set_variables() and after restore() they equal a fresh split of the current fp32 weights, bit
    for bit -- a stale image would train on last step's weights."""
    from mtamrecommender_amd import hip_ops as ops
    model, FLAGS, records = build(tmp_path, 16, 8, 2, 1, optimizer=optimizer)
    p = model.path
    assert p.wimg is not None and [n for n, *_ in p._wimg_parts] == ["dense4emb/w", "kv/w", "gru/wx"]

    def fresh():
        n_img = p.wimg.numel()
        buf = torch.zeros(2 * n_img, dtype=torch.bfloat16, device="cuda")
        n_x = p.layout.segments["gru/wx"].shape[1]
        for which, name in enumerate(("dense4emb/w", "kv/w", "gru/wx")):
            o = ops.seq_chain_image_offset(which, n_x)
            ops.split_weight_images(p.seg(name), buf[o:])
            ops.split_weight_rows(p.seg(name), buf[n_img + o:])
        return buf

    def fresh_gru():
        img = torch.zeros_like(p.gru_img)
        ops.gru_weight_image(p.seg("gru/wh_g"), p.seg("gru/wh_c"), img)
        return img

    whole = lambda: torch.cat([p.wimg, p.wimg_r])
    assert torch.equal(whole(), fresh()) and torch.equal(p.gru_img, fresh_gru())
    w0 = p.seg("gru/wx").clone()
    for _ in range(3):
        model.train(model.sess, records, 1e-3)
    assert not torch.equal(p.seg("gru/wx"), w0) and torch.equal(whole(), fresh()) and torch.equal(p.gru_img, fresh_gru())
    model.save(model.sess, global_step=1)
    arrays = model.get_variables()
    arrays["position_embedding/dense4emb/kernel"] = arrays["position_embedding/dense4emb/kernel"] * 1.5
    arrays[[k for k in arrays if k.endswith("candidate/kernel")][0]] *= 0.5
    model.set_variables(arrays)
    assert torch.equal(whole(), fresh()) and torch.equal(p.gru_img, fresh_gru())
    model.restore(model.sess, str(tmp_path))
    assert torch.equal(whole(), fresh()) and torch.equal(p.gru_img, fresh_gru())


def test_async_loss_is_logged_once_under_its_own_step(hip_lib, tmp_path):
    """async_loss: train() hands over the PREVIOUS step's loss together with the step it belongs to, nothing on the
    first call, and drain_loss() the most recent one -- so a loop sees every step's loss exactly once, equal to
    what the blocking form returns for that step (the reference's sess.run returns the loss of the step it ran,
    Model/base_model.py:159-167)."""
    model_a, FLAGS, records = build(tmp_path, 16, 8, 1, 1)
    model_b, _, _ = build(tmp_path, 16, 8, 1, 1)
    model_a.use_graph = model_b.use_graph = False     # (float atomics aside, the two models then walk the same path)
    want = [model_a.train(model_a.sess, records, 1e-3, global_step=s)[0] for s in range(5)]
    model_b.async_loss = True
    seen = {}
    for s in range(5):
        loss, summary = model_b.train(model_b.sess, records, 1e-3, global_step=s)
        if s in (0, 3):                                 # nothing finished yet / already handed over by the drain
            assert np.isnan(loss) and summary["loss_step"] is None and model_b.loss_step is None
        else:
            assert summary["loss_step"] == s - 1 == model_b.loss_step
            assert summary["loss_step"] not in seen
            seen[summary["loss_step"]] = loss
        if s == 2:                                      # e.g. an evaluation point: the window must be complete
            loss, summary = model_b.drain_loss()
            assert summary["loss_step"] == 2
            seen[2] = loss
            assert model_b.drain_loss() is None
    loss, summary = model_b.drain_loss()
    seen[summary["loss_step"]] = loss
    assert sorted(seen) == [0, 1, 2, 3, 4]
    for s in range(5):
        assert abs(seen[s] - want[s]) <= 2e-5 * abs(want[s]), (s, seen[s], want[s])
    assert want[4] < want[0]


def test_tf_named_npz_round_trip(hip_lib, tmp_path):
    """export_tf_npz / import_tf_npz: TF variable names incl. Adam slots; the restored model holds the same
    state bit for bit and continues the same way (up to float-atomic rounding inside a step)."""
    model_a, FLAGS, records = build(tmp_path, 16, 8, 1, 1)
    model_b, _, _ = build(tmp_path, 16, 8, 1, 1, seed=9)          # different weights
    model_a.train(model_a.sess, records, 1e-3)
    names = model_a.export_tf_npz(str(tmp_path / "tf_vars.npz"))
    assert "embedding_layer/item" in names and "embedding_layer/item/Adam_1" in names and "beta2_power" in names
    assert any(n.endswith("time_aware_gru_cell_decay_new/gates/kernel/Adam") for n in names)
    assert any(n.endswith("_time_history_b1") for n in names)     # a variable the reference never updates
    rec_b = records                                                 # same batch through both models
    model_b.import_tf_npz(str(tmp_path / "tf_vars.npz"))
    va, vb = model_a.get_variables(), model_b.get_variables()
    assert all(np.array_equal(va[k], vb[k]) for k in va)
    sa, sb = model_a.path.optimizer_state(), model_b.path.optimizer_state()
    assert torch.equal(sa["flat_m"], sb["flat_m"]) and torch.equal(sa["flat_v"], sb["flat_v"])
    assert torch.equal(sa["adam_state"][1:6], sb["adam_state"][1:6])      # [0] is lr_t, rewritten every step
    la, _ = model_a.train(model_a.sess, records, 1e-3)
    lb, _ = model_b.train(model_b.sess, rec_b, 1e-3)
    assert abs(la - lb) <= 1e-5 * abs(la)


def test_tf_checkpoint_bundle_round_trip(hip_lib, tmp_path):
    """export_tf_checkpoint / import_tf_checkpoint / restore(): the same state as a TensorFlow checkpoint bundle
    (index table + data shard + `checkpoint` state file, util/tf_bundle.py); a model pointed at the directory with
    the reference's restore() call picks it up.  The bundle format itself: tests/test_tf_bundle.py."""
    from mtamrecommender_amd.util import tf_bundle
    model_a, FLAGS, records = build(tmp_path, 16, 8, 1, 1)
    model_b, _, _ = build(tmp_path, 16, 8, 1, 1, seed=9)
    model_c, _, _ = build(tmp_path, 16, 8, 1, 1, seed=11)
    model_a.train(model_a.sess, records, 1e-3)
    ckpt_dir = tmp_path / "tf_ckpt"
    names = model_a.export_tf_checkpoint(str(ckpt_dir / "model.ckpt-1"))
    assert "embedding_layer/item/Adam" in names and "beta1_power" in names
    listed = tf_bundle.list_bundle(str(ckpt_dir / "model.ckpt-1"))
    assert listed["embedding_layer/item"] == (np.float32, tuple(model_a.get_variables()["embedding_layer/item"].shape))
    assert listed["beta1_power"] == (np.float32, ())
    assert model_b.import_tf_checkpoint(str(ckpt_dir)) == str(ckpt_dir / "model.ckpt-1")
    model_c.restore(model_c.sess, str(ckpt_dir))                  # the reference's call, on a TF-format directory
    va = model_a.get_variables()
    sa = model_a.path.optimizer_state()
    for other in (model_b, model_c):
        vo, so = other.get_variables(), other.path.optimizer_state()
        assert all(np.array_equal(va[k], vo[k]) for k in va)
        assert torch.equal(sa["flat_m"], so["flat_m"]) and torch.equal(sa["flat_v"], so["flat_v"])
        assert torch.equal(sa["adam_state"][1:6], so["adam_state"][1:6])
    la, _ = model_a.train(model_a.sess, records, 1e-3)
    lb, _ = model_b.train(model_b.sess, records, 1e-3)
    assert abs(la - lb) <= 1e-5 * abs(la)


@pytest.mark.parametrize("member", ["MTAM_only_time_aware_RNN", "MTAM_no_time_aware_rnn", "MTAM_via_T_GRU",
                                    "MTAM_via_rnn", "MTAM_with_T_SeqRec", "MTAM_hybird"])
@pytest.mark.parametrize("B,L,NB,H", [(6, 8, 1, 1), (33, 50, 2, 2)])
def test_mtam_family_forward_and_gradients(hip_lib, tmp_path, member, B, L, NB, H):
    """The ablation members of Model/MTAMRec_model.py:40-238 that run on the MTAM kernels: logits, loss,
    every gradient and the clip norm against the oracle."""
    import oracle.c_oracle as co
    import oracle.mtam_oracle as O
    model, FLAGS, records = build(tmp_path, B, L, NB, H, model_name=member)
    model.use_graph = False
    p = model.path
    arrays = {k: v.copy() for k, v in model.get_variables().items()}
    feed = model.embedding.make_feed_dic_new(records)
    bt = p.load_feed(feed)
    p.eval_kernels(bt, 50)
    logits, pred = bt.logits.cpu().numpy(), bt.pred.cpu().numpy()
    out, grads, slot_sq = O.loss_and_grads(member, arrays, feed, H, NB, FLAGS.regulation_rate, torch.float64)
    assert rel(logits, out["logits"].detach().numpy()) < LOGIT_TOL
    assert np.array_equal(logits, co.score_fma(pred, arrays["embedding_layer/item"]))
    loss, summary = model.train(model.sess, records, 1e-3)
    ref_loss = float(out["loss"].detach())
    assert abs(loss - ref_loss) / abs(ref_loss) < 2e-5
    got = p.grads_tf()
    for name, g in grads.items():
        if g is None:
            continue
        assert rel(got[name], g) < GRAD_TOL, name
    assert set(got) >= {k for k, g in grads.items() if g is not None}
    ref_norm = O.global_norm(grads, slot_sq, member, True)
    assert abs(float(p.scale[1]) - ref_norm) / ref_norm < 1e-4


@pytest.mark.parametrize("member", ["MTAM_via_T_GRU", "MTAM_with_T_SeqRec"])
def test_mtam_family_trains_through_the_graph(hip_lib, tmp_path, member):
    import oracle.mtam_oracle as O
    B, L, NB, H = 16, 20, 1, 1
    model, FLAGS, records = build(tmp_path, B, L, NB, H, model_name=member)
    arrays = {k: v.copy() for k, v in model.get_variables().items()}
    state = O.AdamState(arrays)
    for step in range(4):                                   # step 3+ replays the captured hipGraph
        feed = model.embedding.make_feed_dic_new(records)
        ref = O.train_step(member, arrays, state, feed, 1e-3, H, NB, FLAGS.regulation_rate,
                           FLAGS.max_gradient_norm, True)
        loss, _ = model.train(model.sess, records, 1e-3)
        assert abs(loss - ref["loss"]) / abs(ref["loss"]) < 1e-4, step


# ------------------------------------------------------------------ bf16 scoring (BASELINE.json configs[4])
@pytest.mark.parametrize("model_name,B,L,items", [("MTAM", 128, 50, 3706), ("MTAM", 37, 20, 300),
                                                   ("PISTRec", 64, 30, 1000)])
def test_bf16_scoring_mode(hip_lib, tmp_path, model_name, B, L, items):
    """FLAGS.score_dtype = 'bf16': logits-free bf16-MFMA scoring (csrc/score16.hip) inside the full step.
    The oracle rounds both scoring operands to bf16 the same way and keeps everything else in float64.
    Tolerances: the kernel's scores against float64 products of ITS OWN bf16 operands 1e-5 (summation
    order only); against the oracle's scores 2e-3 of max |logit| (an element of pred that sits on a bf16
    rounding boundary may round the other way: 2^-9 of one product); gradients 1e-2 of each tensor's largest
    entry (G is rounded to bf16 before the two backward products); loss 1e-4."""
    import oracle.mtam_oracle as O
    NB, H = 1, 1
    model, FLAGS, records = build(tmp_path, B, L, NB, H, items=items, cats=31, users=200, model_name=model_name,
                                  score_dtype="bf16")
    p = model.path
    assert p.score_dtype == "bf16" and p.item16 is not None
    arrays = {k: v.copy() for k, v in model.get_variables().items()}
    feed = model.embedding.make_feed_dic_new(records)
    assert torch.equal(p.item16.view(torch.int16), p.tables["item"].bfloat16().view(torch.int16))

    # ---- evaluation forward: logits and top-K
    bt = p.load_feed(feed)
    p.eval_kernels(bt, 50)
    out, grads, slot_sq = O.loss_and_grads(model_name, arrays, feed, H, NB, FLAGS.regulation_rate, torch.float64,
                                           score_dtype="bf16")
    pred = bt.pred.cpu().numpy()
    assert rel(pred, out["pred"].detach().numpy()) < 2e-5
    own = bt.pred.bfloat16().double() @ p.item16.double().T
    assert float((bt.logits.double() - own).abs().max()) < 1e-5 * float(own.abs().max())
    assert rel(bt.logits.cpu().numpy(), out["logits"].detach().numpy()) < 2e-3
    assert np.array_equal(bt.topk_idx.cpu().numpy(), O.top_k(bt.logits.cpu().numpy(), 50))

    # ---- one training step without the graph: loss, every gradient, the TF-style global norm
    model.use_graph = False
    loss, _ = model.train(model.sess, records, 1e-3)
    ref_loss = float(out["loss"].detach())
    assert abs(loss - ref_loss) / abs(ref_loss) < 1e-4
    got = p.grads_tf()
    for name, g in grads.items():
        if g is not None:
            assert rel(got[name], g) < 1e-2, name
    ref_norm = O.global_norm(grads, slot_sq, model_name, True)
    assert abs(float(p.scale[1]) - ref_norm) / ref_norm < 2e-3
    # the scoring copy follows the master weights
    assert torch.equal(p.item16.view(torch.int16), p.tables["item"].bfloat16().view(torch.int16))

    # ---- through the hipGraph: the loss falls, the copy stays in step
    model.use_graph = True
    losses = [model.train(model.sess, records, 1e-3)[0] for _ in range(6)]
    assert np.isfinite(losses).all() and losses[-1] < loss
    torch.cuda.synchronize()
    assert torch.equal(p.item16.view(torch.int16), p.tables["item"].bfloat16().view(torch.int16))
    hr = model.metrics_topK(model.sess, records, 0, [1, 5, 10, 30, 50])
    assert len(hr) == 10 and all(0.0 <= x <= 1.0 for x in hr)
