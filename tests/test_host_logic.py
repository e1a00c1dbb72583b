"""Host-side logic: feed padding, batch iterator, flags, parameter layout, trainer rules (CPU only)."""
import numpy as np
import pytest

from mtamrecommender_amd.config.model_parameter import model_parameter
from mtamrecommender_amd.data.synthetic import ML1M, SyntheticCatalog, make_records
from mtamrecommender_amd.DataHandle.get_input_data import DataInput
from mtamrecommender_amd.Embedding.feed import pad_batch
from mtamrecommender_amd.Model.param_layout import DenseLayout
from mtamrecommender_amd.Model.variables import init_variables, mtam_dense_specs, pistrec_dense_specs


def toy_records():
    """Seven hand-written records in the reference's 9-tuple shape."""
    recs = []
    for i in range(7):
        n = 1 + i % 4
        items = list(range(10 + i, 10 + i + n)) + [101]
        recs.append((i, items, [3] * n + [8], list(range(1000, 1000 + n)) + [2000], [0] + [1] * (n - 1) + [0],
                     [5] * n + [0], list(range(n)) + [n], [40 + i, 2, 2000], n + 1))
    return recs


def test_pad_batch_pads_at_the_end_with_zeros():
    feed = pad_batch(toy_records(), 6)
    assert feed["item_list"].dtype == np.int32 and feed["time_list"].dtype == np.float32
    assert feed["item_list"].shape == (7, 6)
    assert feed["item_list"][0].tolist() == [10, 101, 0, 0, 0, 0]
    assert feed["item_list"][3].tolist() == [13, 14, 15, 16, 101, 0]
    assert feed["position_list"][3].tolist() == [0, 1, 2, 3, 4, 0]
    assert feed["time_list"][1].tolist() == [1000.0, 1001.0, 2000.0, 0.0, 0.0, 0.0]
    assert feed["seq_length"].tolist() == [2, 3, 4, 5, 2, 3, 4]
    assert feed["target_item_id"].tolist() == [40, 41, 42, 43, 44, 45, 46]
    assert feed["target_item_time"][0] == 2000.0
    with pytest.raises(ValueError):
        pad_batch(toy_records(), 4)                       # a record longer than length_of_user_history


def test_data_input_sequential_batches_with_short_tail():
    got = [(i, [r[0] for r in b]) for i, b in DataInput(toy_records(), 3)]
    assert got == [(1, [0, 1, 2]), (2, [3, 4, 5]), (3, [6])]
    assert list(DataInput([], 3)) == []


def test_flags_defaults_and_presets():
    mp = model_parameter()
    f = mp.get_parameter("no_such_preset").FLAGS
    assert (f.num_units, f.num_heads, f.num_blocks, f.train_batch_size, f.length_of_user_history) == (128, 8, 6, 256, 50)
    assert f.regulation_rate == 5e-5 and f.decay_rate == 0.001 and f.max_gradient_norm == 1.0 and f.top_k == 20
    f = model_parameter().get_parameter("MTAMb7_elec").FLAGS
    assert (f.experiment_type, f.num_blocks, f.num_heads, f.decay_rate, f.test_batch_size) == ("MTAM", 7, 1, 0.995, 2048)
    assert f.checkpoint_path_dir is None and f.version == "MTAMb7_elec"
    assert "num_units" in f.flag_values_dict()
    with pytest.raises(AttributeError):
        f.not_a_flag = 1
    f = model_parameter().get_parameter("MTAMb1_movielen").parse_argv(["--train_batch_size", "64", "--num_heads", "2"]).FLAGS
    assert f.train_batch_size == 64 and f.num_heads == 2


@pytest.mark.parametrize("model,specs_fn", [("MTAM", mtam_dense_specs), ("PISTRec", pistrec_dense_specs)] + [
    (m, None) for m in ("MTAM_only_time_aware_RNN", "MTAM_no_time_aware_rnn", "MTAM_via_T_GRU", "MTAM_via_rnn",
                        "MTAM_with_T_SeqRec", "MTAM_hybird")])
def test_dense_layout_round_trip(model, specs_fn):
    specs = specs_fn(128, 50, 3) if specs_fn else mtam_dense_specs(128, 50, 3, model)
    values = init_variables(specs, seed=1)
    live = {s.name: values[s.name] for s in specs if s.trainable_grad}
    lay = DenseLayout(model, 128, 50, 3)
    flat = lay.pack(live)
    assert flat.dtype == np.float32 and flat.size == lay.total and lay.total % 4 == 0
    assert all(s.offset % 4 == 0 for s in lay.segments.values())
    back = lay.unpack(flat)
    assert set(back) == set(live)
    assert all(np.array_equal(back[k], live[k]) for k in live)
    assert sum(int(np.prod(v.shape)) for v in live.values()) <= lay.total
    assert set(lay.dead_names()) == {s.name for s in specs if not s.trainable_grad}


def test_parameter_counts_match_the_survey():
    """SURVEY.md Appendix B counts every created variable, dead ones included."""
    count = lambda specs: sum(int(np.prod(s.shape)) for s in specs)
    assert count(mtam_dense_specs(128, 50, 1)) == 199980
    assert count(mtam_dense_specs(128, 50, 6)) == 532360
    assert count(pistrec_dense_specs(128, 100, 1)) == 159200
    live = sum(int(np.prod(s.shape)) for s in mtam_dense_specs(128, 50, 1) if s.trainable_grad)
    assert live == 199980 - 6 * 128 - 50            # 6 dead GRU vectors + time_output_w3


def test_synthetic_records_follow_the_reference_format():
    cat = SyntheticCatalog(seed=1234, **ML1M)
    recs = make_records(cat, 200, 50, seed=1)
    for r in recs:
        n = r[8] - 1
        assert 1 <= n <= 49 and len(r[1]) == len(r[2]) == len(r[3]) == len(r[4]) == len(r[5]) == len(r[6]) == n + 1
        assert r[1][-1] == cat.item_count + 1 and r[2][-1] == cat.category_count + 1
        assert r[4][0] == 0 and r[4][-1] == 0 and r[5][-1] == 0
        assert all(r[3][i + 1] - r[3][i] == r[4][i + 1] for i in range(n - 1))
        assert all(r[7][2] - r[3][i] == r[5][i] for i in range(n)) and r[3][-1] == r[7][2]
        assert r[6][:n] == list(range(n)) and r[6][-1] <= 49
        assert all(0 <= x < cat.item_count for x in r[1][:-1]) and 0 <= r[7][0] < cat.item_count
    assert make_records(cat, 3, 50, seed=1) == recs[:3]            # deterministic


def test_learning_rate_rule():
    from mtamrecommender_amd.train_process import average_metrics, exponential_decay, next_learning_rate
    assert exponential_decay(0.001, 250, 100, 0.995) == pytest.approx(0.001 * 0.995 ** 2, rel=1e-6)
    # default lr 1e-3 is not > 1e-3: the decay_rate branch is taken from the first step on
    assert next_learning_rate(0.001, 0.001, 0.995, 0) == pytest.approx(0.001)
    assert next_learning_rate(0.001, 0.001, 0.995, 199) == pytest.approx(0.001 * 0.995, rel=1e-6)
    # a larger configured lr decays by 0.99 per 100 steps until it drops to <= 1e-3
    assert next_learning_rate(0.01, 0.01, 0.995, 300) == pytest.approx(0.01 * 0.99 ** 3, rel=1e-6)
    # the flag default decay_rate = 0.001 collapses the rate after 100 steps
    assert next_learning_rate(0.001, 0.001, 0.001, 100) == pytest.approx(1e-6, rel=1e-5)
    m = average_metrics([(1.0,) * 10, (0.0,) * 10, (0.5,) * 10])
    assert m == tuple([0.5] * 10) and average_metrics([]) == tuple([0.0] * 10)


def test_embedding_feed_and_validation():
    from mtamrecommender_amd.Embedding.Behavior_embedding_time_aware_attention import \
        Behavior_embedding_time_aware_attention
    emb = Behavior_embedding_time_aware_attention(True, user_count=10, item_count=120, category_count=9, max_length_seq=6)
    assert emb.position_count == 6
    tup = emb.get_embedding(128)
    assert len(tup) == 10 and emb.item_emb_lookup_table.shape == (123, 128) and emb.position_emb_lookup_table.shape == (9, 128)
    r = float(np.sqrt(np.float32(6.0 / 128)))
    assert np.abs(emb.item_emb_lookup_table).max() <= r
    feed = emb.make_feed_dic_new(toy_records())
    assert set(feed) == {"user_id", "item_list", "category_list", "time_list", "timelast_list", "timenow_list",
                         "position_list", "target_item_id", "target_item_category", "target_item_time", "seq_length"}
    emb.validate_ids(feed)
    bad = dict(feed)
    bad["item_list"] = feed["item_list"].copy()
    bad["item_list"][0, 0] = 123
    with pytest.raises(IndexError):
        emb.validate_ids(bad)


def test_feed_arena_layout_is_16_byte_granular():
    """Every field of the packed feed arena starts on a 16-byte boundary and the arena is a whole number of 16-byte
    pieces -- what the feed ring's hand-over (mtam_adam_images_clip_feed: feed_words % 4 == 0, 16-byte copies) and the
    native packer rely on -- for ragged batch sizes and lengths too; the fields do not overlap."""
    from mtamrecommender_amd.Model.time_aware_path import arena_layout
    for B, L in [(1, 1), (3, 7), (16, 20), (33, 50), (128, 50), (127, 51), (2048, 200)]:
        offsets, words = arena_layout(B, L)
        assert words % 4 == 0
        spans = sorted((o, o + n) for o, n, _, _ in offsets.values())
        assert all(o % 4 == 0 for o, _ in spans)
        assert all(a_end <= b_start for (_, a_end), (b_start, _) in zip(spans, spans[1:]))
        assert spans[-1][1] <= words
        assert offsets["lr"][1] == 4 and offsets["item_list"][1] == B * L


def test_resident_epoch_plan_is_the_loops_schedule_and_order():
    """FLAGS.resident_epoch bakes every step's learning rate into its feed slot ahead of time (Train_main_process.
    _resident_plan): the list must be exactly what the per-step loop computes -- next_learning_rate restarts from
    FLAGS.learning_rate at the epoch's first step and is a function of the running value and the global step
    (the reference's train_process.py:330-337) -- across a decay boundary too; the full batches and the partial last
    one partition the order random.shuffle produces, as DataInput cuts it."""
    import random
    import types
    from mtamrecommender_amd.train_process import Train_main_process, next_learning_rate
    for flags_lr, step0, n_rec, B in [(1e-3, 0, 200, 32), (5e-3, 95, 1000, 7), (1e-3, 1234, 64, 32)]:
        me = types.SimpleNamespace(FLAGS=types.SimpleNamespace(train_batch_size=B, learning_rate=flags_lr, decay_rate=0.99),
                                   _order=list(range(n_rec)))
        random.seed(3)
        plan = Train_main_process._resident_plan(me, step0)
        random.seed(3)
        want = list(range(n_rec))
        random.shuffle(want)
        n_full = n_rec // B
        assert list(plan["index"]) == want[:n_full * B] and plan["tail"] == want[n_full * B:]
        lr, loop = flags_lr, []
        for k in range(n_full):                  # the loop of Train_main_process.train, step by step
            lr = next_learning_rate(lr, flags_lr, 0.99, step0 + k)
            loop.append(lr)
        assert plan["lrs"] == loop and plan["step0"] == step0 and plan["B"] == B
    assert len(set(loop)) == 1 and len(set(Train_main_process._resident_plan(
        types.SimpleNamespace(FLAGS=types.SimpleNamespace(train_batch_size=1, learning_rate=1e-3, decay_rate=0.99),
                              _order=list(range(250))), 0)["lrs"])) > 1      # the schedule moves inside a long epoch
