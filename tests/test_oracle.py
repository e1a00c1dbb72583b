"""Pins the CPU oracle (CPU only, no GPU).

The reference ships no fixtures (parity unpinned), so the oracle is
held in place by: two independent restatements agreeing, hand-derived
known-answer cases, float64 finite differences, structural invariants of the
reference graph, and the committed golden vectors (regression pin).
"""
import glob
import math
import os

import numpy as np
import pytest
import torch

from oracle import feed_ref, mtam_oracle as O, numpy_ref as N, records as R, specs as S
from oracle.family import CELL_SCOPE

GRU_SCOPE, TSR_SCOPE = CELL_SCOPE["decay_new"], CELL_SCOPE["sigmoid"]

HERE = os.path.dirname(os.path.abspath(__file__))
REG = 5e-5


def small_case(model, B=5, L=8, D=16, NB=2, H=2, seed=3):
    """Inputs and weights from the oracle's own generators and variable list (oracle/records.py, feed_ref.py,
    specs.py): nothing of the product takes part."""
    feed = feed_ref.make_feed_dic_new(R.make_records(60, 7, 20, B, L, seed=seed), L)
    arrays = S.init_arrays(S.model_vars(model, 20, 60, 7, L, D, NB), seed=seed + 4, jitter=0.1)
    return feed, arrays


@pytest.mark.parametrize("model", ["MTAM", "PISTRec"])
@pytest.mark.parametrize("H", [1, 2, 4])
def test_two_restatements_agree(model, H):
    feed, arrays = small_case(model, H=H)
    w = O.split_item_table(arrays, torch.float64, False)
    out = O.forward(model, w, O.feed_to_torch(feed, torch.float64), H, 2, REG)
    ref = N.forward(model, arrays, feed, H, 2, REG)
    assert np.abs(out["logits"].numpy() - ref["logits"]).max() < 1e-10
    assert abs(float(out["loss"]) - ref["loss"]) < 1e-10
    assert abs(float(out["l2"]) - ref["l2"]) < 1e-9


def test_float32_oracle_close_to_float64():
    feed, arrays = small_case("MTAM")
    w32 = O.split_item_table(arrays, torch.float32, False)
    w64 = O.split_item_table(arrays, torch.float64, False)
    a = O.forward("MTAM", w32, O.feed_to_torch(feed, torch.float32), 2, 2, REG)["logits"].numpy()
    b = O.forward("MTAM", w64, O.feed_to_torch(feed, torch.float64), 2, 2, REG)["logits"].numpy()
    assert np.abs(a - b).max() / np.abs(b).max() < 1e-5      # the 1e-5 agreement SURVEY.md 8(c) asks for


# ------------------------------------------------------------ known answers
def test_known_answer_gru_single_step():
    """seq_len = 2: the GRU runs one step from h = 0, so r drops out and
    h1 = (1 - u) * tanh(x Wc_x + bc) * T with u = sigmoid(x Wg_x[:, D:] + bg[D:])."""
    D = 4
    rng = np.random.default_rng(0)
    w = {GRU_SCOPE + "gates/kernel": rng.normal(size=(2 * D, 2 * D)), GRU_SCOPE + "gates/bias": rng.normal(size=2 * D),
         GRU_SCOPE + "candidate/kernel": rng.normal(size=(2 * D, D)), GRU_SCOPE + "candidate/bias": rng.normal(size=D)}
    for n in ("_time_kernel_w1", "_time_kernel_b1", "_time_history_w1", "_time_w1", "_time_b1", "_time_kernel_w2",
              "_time_w12", "_time_b12"):
        w[GRU_SCOPE + n] = rng.normal(size=D)
    wt = {k: torch.tensor(v) for k, v in w.items()}
    x = rng.normal(size=(1, 3, D))
    dl = np.array([[0.0, 7.0, 0.0]])
    hs = O.time_aware_gru(wt, torch.tensor(x), torch.tensor(dl), torch.tensor([1]))
    sig = lambda z: 1 / (1 + np.exp(-z))
    x0 = x[0, 0]
    u = sig(x0 @ w[GRU_SCOPE + "gates/kernel"][:D, D:] + w[GRU_SCOPE + "gates/bias"][D:])
    c = np.tanh(x0 @ w[GRU_SCOPE + "candidate/kernel"][:D] + w[GRU_SCOPE + "candidate/bias"])
    tw = np.maximum(x0 * w[GRU_SCOPE + "_time_kernel_w1"] + w[GRU_SCOPE + "_time_kernel_b1"], 0)
    ts = np.maximum(w[GRU_SCOPE + "_time_w1"] * 0.0 + w[GRU_SCOPE + "_time_b1"], 0)
    T = sig(w[GRU_SCOPE + "_time_kernel_w2"] * tw + w[GRU_SCOPE + "_time_w12"] * ts + w[GRU_SCOPE + "_time_b12"])
    want = (1 - u) * c * T
    assert np.allclose(hs[0, 0].numpy(), want, atol=1e-12)
    assert np.all(hs[0, 1:].numpy() == 0)                        # dynamic_rnn zero-fills past sequence_length
    assert np.allclose(O.gather_indexes(hs, torch.tensor([0]))[0].numpy(), want, atol=1e-12)


def test_known_answer_attention_two_keys():
    """All-padding-but-two sequence with identical keys: softmax over two equal scores = 1/2 each,
    so the block output is LN(V_row + q)."""
    D, L = 8, 5
    rng = np.random.default_rng(1)
    scope, inner = "s/", "vanilla_attention"
    w = {}
    for layer in ("dense", "dense_1", "dense_2"):
        w[scope + layer + "/kernel"] = torch.tensor(rng.normal(size=(D, D)))
        w[scope + layer + "/bias"] = torch.tensor(rng.normal(size=D))
    s = scope + inner + "/"
    w[s + "_time_input_w"] = torch.tensor(rng.normal(size=(D, D)))
    for n in ("_time_input_w1", "_time_input_b1", "time_output_w1", "time_output_w2", "time_output_b"):
        w[s + n] = torch.tensor(np.full((1, L), rng.normal()))        # same gate parameters at every key
    w[s + "ln/Variable"] = torch.zeros(D, dtype=torch.float64)
    w[s + "ln/Variable_1"] = torch.ones(D, dtype=torch.float64)
    key = rng.normal(size=D)
    k = np.tile(key, (1, L, 1))
    k[0, 2:] = rng.normal(size=(L - 2, D)) * 100                      # garbage in the padded slots
    q = rng.normal(size=(1, 1, D))
    tk = np.array([[5.0, 5.0, 1e6, -3.0, 0.0]])
    out, att = O.time_aware_multihead_attention(w, scope, inner, torch.tensor(q), torch.tensor(k),
                                                torch.tensor([2]), torch.tensor([1]),
                                                torch.tensor([[9.0]]), torch.tensor(tk), 1)
    a = att[0, 0].numpy()
    assert np.allclose(a[:2], 0.5, atol=1e-12) and np.all(a[2:] == 0.0)   # masked keys: exactly zero weight
    V = np.maximum(key @ w[scope + "dense_2/kernel"].numpy() + w[scope + "dense_2/bias"].numpy(), 0)
    y = V + q[0, 0]
    want = (y - y.mean()) / np.sqrt(y.var() + 1e-8)
    assert np.allclose(out[0, 0].numpy(), want, atol=1e-10)


def test_known_answer_delta_t_zero():
    """A key whose time equals the query time has decay input log(0 + 1) = 0: tanh(b1)."""
    D, L = 4, 3
    scope, inner = "s/", "vanilla_attention"
    w = {}
    for layer in ("dense", "dense_1", "dense_2"):
        w[scope + layer + "/kernel"] = torch.zeros(D, D, dtype=torch.float64)
        w[scope + layer + "/bias"] = torch.ones(D, dtype=torch.float64)
    s = scope + inner + "/"
    w[s + "_time_input_w"] = torch.zeros(D, D, dtype=torch.float64)           # tanh(q Wt k^T) = 0
    w[s + "_time_input_w1"] = torch.full((1, L), 3.0, dtype=torch.float64)
    w[s + "_time_input_b1"] = torch.tensor([[0.3, 0.3, 0.3]], dtype=torch.float64)
    w[s + "time_output_w1"] = torch.ones(1, L, dtype=torch.float64)
    w[s + "time_output_w2"] = torch.ones(1, L, dtype=torch.float64)
    w[s + "time_output_b"] = torch.zeros(1, L, dtype=torch.float64)
    w[s + "ln/Variable"] = torch.zeros(D, dtype=torch.float64)
    w[s + "ln/Variable_1"] = torch.ones(D, dtype=torch.float64)
    q = torch.zeros(1, 1, D, dtype=torch.float64)
    k = torch.zeros(1, L, D, dtype=torch.float64)
    _, att = O.time_aware_multihead_attention(w, scope, inner, q, k, torch.tensor([2]), torch.tensor([1]),
                                              torch.tensor([[7.0]], dtype=torch.float64),
                                              torch.tensor([[7.0, 7.0 - (math.e - 1), 0.0]], dtype=torch.float64), 1)
    # Q.K = D for both keys; gates: sigmoid(tanh(0.3)) and sigmoid(tanh(3*1 + 0.3)); scale 1/sqrt(D)
    sig = lambda z: 1 / (1 + math.exp(-z))
    s0 = D * sig(math.tanh(0.3)) / math.sqrt(D)
    s1 = D * sig(math.tanh(3.3)) / math.sqrt(D)
    want0 = math.exp(s0) / (math.exp(s0) + math.exp(s1))
    assert abs(float(att[0, 0, 0]) - want0) < 1e-12


# ---------------------------------------------------------------- invariants
def test_loss_ignores_padded_slots_and_dead_timenow():
    feed, arrays = small_case("MTAM")
    base = O.loss_and_grads("MTAM", arrays, feed, 2, 2, REG, torch.float64)[0]["logits"].detach().numpy()
    f2 = {k: v.copy() for k, v in feed.items()}
    for b, sl in enumerate(feed["seq_length"]):
        f2["time_list"][b, sl:] = 12345.0             # padded time slots
        f2["timelast_list"][b, sl - 1:] = 77.0        # the GRU runs sl-1 steps only
    f2["timenow_list"][:] = 999.0                     # sliced off and unused by the 'new' cell
    other = O.loss_and_grads("MTAM", arrays, f2, 2, 2, REG, torch.float64)[0]["logits"].detach().numpy()
    assert np.array_equal(base, other)


def test_padded_slots_have_exactly_zero_upstream_gradient():
    """What mtam_emb_scatter_add_bwd relies on: d loss / d X at t >= seq_len comes from the L2 term only."""
    feed, arrays = small_case("MTAM")
    out, grads, _ = O.loss_and_grads("MTAM", arrays, feed, 2, 2, 0.0, torch.float64)   # reg = 0
    gi = out["item"].grad.numpy()
    for b, sl in enumerate(feed["seq_length"]):
        assert np.all(gi[b, sl:] == 0)
        assert np.any(gi[b, sl - 1] != 0)             # the mask-token slot is an attention key


FAMILY = ["MTAM", "MTAM_only_time_aware_RNN", "MTAM_no_time_aware_rnn", "MTAM_via_T_GRU", "MTAM_via_rnn",
          "MTAM_with_T_SeqRec", "MTAM_hybird"]


@pytest.mark.parametrize("model", FAMILY)
def test_finite_differences_float64(model):
    feed, arrays = small_case(model, B=3, L=6, D=8, NB=1, H=2)
    _, grads, _ = O.loss_and_grads(model, arrays, feed, 2, 1, REG, torch.float64)
    rng = np.random.default_rng(0)

    def loss_of(a):
        w = O.split_item_table(a, torch.float64, False)
        return float(O.forward(model, w, O.feed_to_torch(feed, torch.float64), 2, 1, REG)["loss"])

    names = [k for k, g in grads.items() if g is not None]
    for name in names:
        g = grads[name]
        for _ in range(2):
            idx = tuple(rng.integers(0, s) for s in g.shape)
            if abs(g[idx]) < 1e-9:
                continue
            eps = 1e-5
            a1 = {k: v.astype(np.float64).copy() for k, v in arrays.items()}
            a2 = {k: v.astype(np.float64).copy() for k, v in arrays.items()}
            a1[name][idx] += eps
            a2[name][idx] -= eps
            fd = (loss_of(a1) - loss_of(a2)) / (2 * eps)
            assert abs(fd - g[idx]) <= 1e-6 * max(1.0, abs(g[idx])) + 1e-8, (name, idx, fd, g[idx])


def test_dead_variables_get_no_gradient():
    feed, arrays = small_case("MTAM")
    _, grads, _ = O.loss_and_grads("MTAM", arrays, feed, 2, 2, REG, torch.float64)
    dead = [k for k, g in grads.items() if g is None]
    assert len(dead) == 6 + 2                                   # 6 GRU vectors + time_output_w3 per block
    assert all(("time_output_w3" in k) or k.startswith(GRU_SCOPE) for k in dead)


# -------------------------------------------------------- clip, Adam, top-K
def test_tf_global_norm_exceeds_true_norm_by_duplicates():
    feed, arrays = small_case("MTAM")
    _, grads, slot_sq = O.loss_and_grads("MTAM", arrays, feed, 2, 2, REG, torch.float64)
    tf_norm = O.global_norm(grads, slot_sq, "MTAM", True)
    true_norm = O.global_norm(grads, slot_sq, "MTAM", False)
    assert tf_norm != true_norm and abs(tf_norm - true_norm) / true_norm < 0.5


def test_train_step_is_dense_adam_with_tf_formulas():
    feed, arrays = small_case("MTAM")
    before = {k: v.copy() for k, v in arrays.items()}
    state = O.AdamState(arrays)
    res = O.train_step("MTAM", arrays, state, feed, 1e-3, 2, 2, REG, 1.0, True)
    assert res["scale"] == pytest.approx(min(1.0 / res["global_norm"], 1.0), rel=1e-6)
    lr_t = 1e-3 * math.sqrt(1 - 0.999) / (1 - 0.9)
    k = "NextItemDecoder/LayerNorm/gamma"
    g = res["grads"][k] * res["scale"]
    m = 0.1 * g
    v = 0.001 * g * g
    assert np.allclose(arrays[k], before[k] - lr_t * m / (np.sqrt(v) + 1e-8), rtol=1e-5, atol=1e-9)
    dead = GRU_SCOPE + "_time_w2"
    assert np.array_equal(arrays[dead], before[dead])            # None gradient: never updated
    assert state.beta1_power == np.float32(0.9) * np.float32(0.9)


def test_top_k_ties_prefer_lower_index():
    s = np.array([[1.0, 3.0, 3.0, 2.0, 3.0], [0.0, -0.0, 0.0, -1.0, 5.0]], np.float32)
    assert O.top_k(s, 3).tolist() == [[1, 2, 4], [4, 0, 1]]
    hr, ndcg = O.calculate_topK(O.top_k(s, 3), [2, 3])
    assert hr == 0.5 and ndcg == pytest.approx(math.log(2) / math.log(3) / 2)


def test_c_oracle_matches_numpy_matmul():
    import oracle.c_oracle as co
    rng = np.random.default_rng(4)
    a = rng.standard_normal((9, 128)).astype(np.float32)
    b = rng.standard_normal((33, 128)).astype(np.float32)
    ref = a.astype(np.float64) @ b.T.astype(np.float64)
    assert np.abs(co.score_fma(a, b) - ref).max() < 2e-5
    one = np.zeros((1, 128), np.float32)
    one[0, 5] = 1.0
    assert np.array_equal(co.score_fma(one, b)[0], b[:, 5])


# ------------------------------------------------------------------- golden
@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "*.npz"))))
def test_oracle_reproduces_golden(path):
    from tests.golden.make_golden import CASES, make_case
    name = os.path.splitext(os.path.basename(path))[0]
    model, B, L, D, NB, H, items, cats, users, seed = CASES[name]
    records, feed, arrays = make_case(*CASES[name])
    gold = np.load(path)
    for k, v in feed.items():
        assert np.array_equal(gold["feed_" + k], v), k
    chk = float(sum(np.abs(v.astype(np.float64)).sum() for v in arrays.values()))
    assert abs(chk - gold["weights_checksum"][0]) / chk < 1e-12
    out, grads, slot_sq = O.loss_and_grads(model, arrays, feed, H, NB, REG, torch.float64)
    assert np.allclose(out["logits"].detach().numpy(), gold["logits"], rtol=0, atol=1e-11)
    assert abs(float(out["loss"].detach()) - gold["loss"][0]) < 1e-12
    assert np.array_equal(O.top_k(out["logits"].detach().numpy(), gold["top50"].shape[1]), gold["top50"])
    assert O.global_norm(grads, slot_sq, model, True) == pytest.approx(gold["global_norm_tf"][0], rel=1e-10)
    for k in gold.files:
        if k.startswith("grad/"):
            assert np.allclose(grads[k[5:]], gold[k], rtol=1e-9, atol=1e-13), k


def test_slot_optimizers_known_answers():
    """One scalar step of each TF 1.14 formula, worked by hand (g = 2, scale = 0.5 -> 1, lr = 0.1)."""
    import oracle.mtam_oracle as O
    for kind, want_p, want_s1, want_s2 in (
            ("sgd", 1.0 - 0.1 * 1.0, None, None),
            # accum = .05; update = sqrt(1e-8) / sqrt(.05 + 1e-8) * 1; accum_update = .05 * update^2
            ("adadelta", 1.0 - 0.1 * (1e-4 / np.sqrt(0.05 + 1e-8)), 0.05, 0.05 * (1e-4 / np.sqrt(0.05 + 1e-8)) ** 2),
            # dense form: ms = 1 + (1 - 1) * .1 = 1; mom = .1 * 1 / sqrt(1 + 1e-10)
            ("rmsprop", 1.0 - 0.1 / np.sqrt(1.0 + 1e-10), 1.0, 0.1 / np.sqrt(1.0 + 1e-10))):
        arrays = {"w": np.array([1.0], np.float32)}
        st = O.SlotState(kind, arrays)
        O.apply_slot_optimizer(arrays, st, {"w": np.array([2.0], np.float32)}, np.float32(0.5), 0.1, {})
        assert abs(arrays["w"][0] - want_p) < 1e-6, kind
        if want_s1 is not None:
            assert abs(st.s1["w"][0] - want_s1) < 1e-7 and abs(st.s2["w"][0] - want_s2) < 1e-8, kind
    # row-sparse tables: rows outside the batch keep parameters AND slots; duplicates are summed first
    arrays = {"embedding_layer/category": np.ones((4, 2), np.float32)}
    st = O.SlotState("rmsprop", arrays)
    st.s1["embedding_layer/category"][:] = 0.5
    g = np.zeros((4, 2), np.float32)
    g[2] = 3.0
    O.apply_slot_optimizer(arrays, st, {"embedding_layer/category": g}, np.float32(1.0), 0.1, {"category_list": [2, 2]})
    assert np.all(st.s1["embedding_layer/category"][[0, 1, 3]] == 0.5)
    assert np.all(arrays["embedding_layer/category"][[0, 1, 3]] == 1.0)
    ms = 0.5 * 0.9 + 9.0 * 0.1          # sparse form
    assert np.allclose(st.s1["embedding_layer/category"][2], ms, rtol=1e-6)
    assert np.allclose(arrays["embedding_layer/category"][2], 1.0 - 0.3 / np.sqrt(ms + 1e-10), rtol=1e-6)


def test_family_members_differ_where_the_reference_says():
    """Known structure of the ablation members (Model/MTAMRec_model.py:40-238): the plain GRUCell is the
    time-aware cell with T = 1; without a decoder the prediction is layer_norm(short-term intent); with the
    GRU outputs as keys, the key at the mask-token slot (t = len-1, a dead GRU step) is the zero vector."""
    feed, arrays = small_case("MTAM_via_rnn", B=4, L=8, D=16, NB=1, H=2)
    f = O.feed_to_torch(feed, torch.float64)
    w = O.split_item_table(arrays, torch.float64, False)
    out = O.forward("MTAM_via_rnn", w, f, 2, 1, REG)
    hs = out["hs"].numpy()
    for b, sl in enumerate(feed["seq_length"]):
        assert np.all(hs[b, sl - 1:] == 0) and np.any(hs[b, sl - 2] != 0)
    # plain GRU == time-aware GRU whose time gate is forced to 1 (huge bias b12, zero weights)
    feed2, arr_t = small_case("MTAM", B=4, L=8, D=16, NB=1, H=2)
    arr_p = {k.replace("time_aware_gru_cell_decay_new", "gru_cell"): v for k, v in arr_t.items() if "_time_" not in k
             or "vanilla_attention" in k}
    for k in list(arr_t):
        if k.startswith(O.GRU_SCOPE + "_time_"):
            arr_t[k] = np.zeros_like(arr_t[k])
    arr_t[O.GRU_SCOPE + "_time_b12"] = np.full_like(arr_t[O.GRU_SCOPE + "_time_b12"], 1e4)
    f2 = O.feed_to_torch(feed2, torch.float64)
    a = O.forward("MTAM", O.split_item_table(arr_t, torch.float64, False), f2, 2, 1, REG)
    b = O.forward("MTAM_no_time_aware_rnn", O.split_item_table(arr_p, torch.float64, False), f2, 2, 1, REG)
    assert np.abs(a["logits"].numpy() - b["logits"].numpy()).max() < 1e-12
    # no decoder: pred = layer_norm(short)
    feed3, arr3 = small_case("MTAM_only_time_aware_RNN", B=4, L=8, D=16, NB=1, H=2)
    f3 = O.feed_to_torch(feed3, torch.float64)
    w3 = O.split_item_table(arr3, torch.float64, False)
    o3 = O.forward("MTAM_only_time_aware_RNN", w3, f3, 2, 1, REG)
    short = O.gather_indexes(o3["hs"], f3["seq_length"] - 2)
    want = O.layer_norm(short, w3["ShortTermIntentEncoder/LayerNorm/beta"], w3["ShortTermIntentEncoder/LayerNorm/gamma"])
    assert np.abs(o3["pred"].numpy() - want.numpy()).max() < 1e-14


def test_seqrec_cell_known_answer():
    """TimeAwareGRUCell_sigmoid by hand (Model/Modules/time_aware_rnn.py:113-130): with every kernel zero the
    gates are sigmoid(bias): r = u = sigmoid(0) = 1/2, c = tanh(atanh(1/2)) = 1/2, sigmoid(now) = sigmoid(ln 3)
    = 3/4, sigmoid(last) = sigmoid(0) = 1/2:  h1 = (1-u) c s_last = 1/8,  h2 = u h1 s_now + 1/8 = 11/64; the
    third step is past the sequence length: output 0, state kept."""
    D = 4
    w = {v.name: torch.zeros(v.shape, dtype=torch.float64) for v in S.cell_vars("sigmoid", D)}
    w[TSR_SCOPE + "candidate/bias"] += float(np.arctanh(0.5))
    w[TSR_SCOPE + "_time_bias1"] += float(np.log(3.0))
    x = torch.ones((1, 3, D), dtype=torch.float64)
    t = torch.tensor([[5.0, 7.0, 9.0]], dtype=torch.float64)
    hs = O.seqrec_gru(w, x, t, t, torch.tensor([2]))
    assert torch.allclose(hs[0, 0], torch.full((D,), 1 / 8, dtype=torch.float64), atol=1e-15)
    assert torch.allclose(hs[0, 1], torch.full((D,), 11 / 64, dtype=torch.float64), atol=1e-15)
    assert torch.equal(hs[0, 2], torch.zeros(D, dtype=torch.float64))
    # the time inputs reach the state only through their [D, D] kernels: tanh(t w + b) with w = b = 0 is 0
    w[TSR_SCOPE + "_time_kernel_t2"] += 1.0
    w[TSR_SCOPE + "_time_input_bias2"] += float(np.arctanh(0.25))
    hs2 = O.seqrec_gru(w, x, t, t, torch.tensor([2]))
    s_last = 1.0 / (1.0 + np.exp(-D * 0.25))                 # last = sum_d tanh(b2) * 1 = D / 4
    assert torch.allclose(hs2[0, 0], torch.full((D,), 0.25 * s_last, dtype=torch.float64), atol=1e-15)
