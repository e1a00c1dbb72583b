"""CPU oracle of the time-aware training path.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors
 and its arithmetic lives in tensorflow==1.14.0 / keras, which
are not installed here -- this file restates the reference's
graph op for op and is pinned only by (1) agreement with the independent numpy
fp64 restatement in ``numpy_ref.py``, (2) hand-derived known-answer cases and
(3) finite-difference gradient checks (tests/test_oracle.py).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline
leg may import this module.  The product path never does.

Deliberately unfused and written with plain torch-CPU ops, one per TF op, so
that timing it is a fair "CPU restatement of the TF1.14 graph" baseline
(BASELINE.md section 3).  Each function cites the reference lines it follows.
TF 1.14 behaviours not visible in the tree are marked [TF1.14]
(SURVEY.md Appendix D).
"""
import math

import numpy as np
import torch

# the family wiring and the TF scope strings are the oracle's OWN restatement (oracle/family.py): nothing is
# imported from the product package, so a mis-wired member cannot pass by sharing a table with its checker
try:
    from .family import CELL_SCOPE, FAMILY, RUNNABLE, SHORT_LN_SCOPE, TIME_GATE_VARS
except ImportError:          # imported as a top-level module
    from family import CELL_SCOPE, FAMILY, RUNNABLE, SHORT_LN_SCOPE, TIME_GATE_VARS

GRU_SCOPE, TSR_SCOPE, PLAIN_GRU_SCOPE = CELL_SCOPE["decay_new"], CELL_SCOPE["sigmoid"], CELL_SCOPE["gru"]
SHORT_LN, TIME_GATE = SHORT_LN_SCOPE, TIME_GATE_VARS

# Model/attention_baseline_models.py:47-65 (experiment_type 'Time_Aware_Self_Attention_Model', train_process.py:209-210):
# PISTRec's encoder with base_model.output() as the loss.  Every other non-family name is PISTRec (:38-74)
TASA = "Time_Aware_Self_Attention_Model"
MASK_VALUE = float(-2 ** 32 + 1)      # time_aware_attention.py:392


def to_torch(arrays, dtype):
    out = {}
    for k, v in arrays.items():
        t = torch.from_numpy(np.ascontiguousarray(v))
        out[k] = t.to(dtype) if t.is_floating_point() else t
    return out


def feed_to_torch(feed, dtype):
    out = {}
    for k, v in feed.items():
        t = torch.from_numpy(np.ascontiguousarray(v))
        out[k] = t.to(dtype) if t.is_floating_point() else t.long()
    return out


# ---------------------------------------------------------------- embedding
def get_embedding(w, feed, item_dtype="f32"):
    """Embedding/Behavior_embedding_time_aware_attention.py:62-114.  item_dtype "bf16" (this build's mixed
    precision): the looked-up item rows are the bf16-rounded ones."""
    user = w["embedding_layer/user"][feed["user_id"]]
    item_table = w["embedding_layer/item_gather"]
    if item_dtype == "bf16":
        item_table = _bf16_straight_through(item_table)
    item = item_table[feed["item_list"]]
    cat = w["embedding_layer/category"][feed["category_list"]]
    pos = w["embedding_layer/position"][feed["position_list"]]
    concat = torch.cat([item, cat], dim=2)
    dense = torch.relu(torch.matmul(concat, w["position_embedding/dense4emb/kernel"]))
    dense = dense + pos
    return user, dense, item, cat, pos


# ---------------------------------------------------------------- GRU
def time_aware_gru(w, x, timelast, seq_len_m1):
    """dynamic_rnn(TimeAwareGRUCell_decay_new): Model/Modules/gru.py:69-77,
    Model/Modules/time_aware_rnn.py:186-269.  Outputs past ``sequence_length``
    are zero and the state is carried through [TF1.14]."""
    B, L, D = x.shape
    p = lambda n: w[GRU_SCOPE + n]
    h = torch.zeros(B, D, dtype=x.dtype)
    outs = []
    for t in range(L):
        xt = x[:, t, :]
        dlast = timelast[:, t:t + 1]
        tw = torch.relu(xt * p("_time_kernel_w1") + p("_time_kernel_b1") + h * p("_time_history_w1"))
        ts = torch.relu(p("_time_w1") * dlast + p("_time_b1"))
        tgate = torch.sigmoid(p("_time_kernel_w2") * tw + p("_time_w12") * ts + p("_time_b12"))
        gates = torch.sigmoid(torch.matmul(torch.cat([xt, h], 1), p("gates/kernel")) + p("gates/bias"))
        r, u = gates[:, :D], gates[:, D:]
        c = torch.tanh(torch.matmul(torch.cat([xt, r * h], 1), p("candidate/kernel"))
                       + p("candidate/bias"))
        new_h = u * h + (1 - u) * c * tgate
        live = (t < seq_len_m1).unsqueeze(1)
        outs.append(torch.where(live, new_h, torch.zeros_like(new_h)))
        h = torch.where(live, new_h, h)
    return torch.stack(outs, dim=1)


def seqrec_gru(w, x, timelast, timenow, lengths):
    """dynamic_rnn over TimeAwareGRUCell_sigmoid (Model/Modules/time_aware_rnn.py:73-131; the T-SeqRec cell
    of MTAM_with_T_SeqRec, Model/MTAMRec_model.py:275-306): a GRU step whose old-state term is gated by
    sigmoid(time_now_state) and whose candidate term by sigmoid(time_last_state) (:129); both states are
    x W + tanh(t w + b) T + bias (:113-121) with the RAW time scores (the log forms are commented out, :111-112)."""
    S = TSR_SCOPE
    B, L, D = x.shape
    h = torch.zeros((B, D), dtype=x.dtype)
    outs = []
    for t in range(L):
        xt = x[:, t]
        tn_in = torch.tanh(timenow[:, t:t + 1] * w[S + "_time_input_w1"] + w[S + "_time_input_bias1"])
        tl_in = torch.tanh(timelast[:, t:t + 1] * w[S + "_time_input_w2"] + w[S + "_time_input_bias2"])
        now = xt @ w[S + "_time_kernel_w1"] + tn_in @ w[S + "_time_kernel_t1"] + w[S + "_time_bias1"]
        last = xt @ w[S + "_time_kernel_w2"] + tl_in @ w[S + "_time_kernel_t2"] + w[S + "_time_bias2"]
        g = torch.sigmoid(torch.cat([xt, h], 1) @ w[S + "gates/kernel"] + w[S + "gates/bias"])
        r, u = g[:, :D], g[:, D:]
        c = torch.tanh(torch.cat([xt, r * h], 1) @ w[S + "candidate/kernel"] + w[S + "candidate/bias"])
        hn = u * h * torch.sigmoid(now) + (1 - u) * c * torch.sigmoid(last)
        alive = (t < lengths).to(x.dtype).unsqueeze(1)
        h = alive * hn + (1 - alive) * h
        outs.append(alive * hn)
    return torch.stack(outs, 1)


def plain_gru(w, x, seq_len_m1):
    """dynamic_rnn(MultiRNNCell([tf GRUCell])): Model/Modules/gru.py:13-39,60-67 [TF1.14 GRUCell:
    gates = sigmoid([x, h] Wg + bg), c = tanh([x, r*h] Wc + bc), h' = u*h + (1-u)*c]."""
    B, L, D = x.shape
    p = lambda n: w[PLAIN_GRU_SCOPE + n]
    h = torch.zeros(B, D, dtype=x.dtype)
    outs = []
    for t in range(L):
        xt = x[:, t, :]
        gates = torch.sigmoid(torch.matmul(torch.cat([xt, h], 1), p("gates/kernel")) + p("gates/bias"))
        r, u = gates[:, :D], gates[:, D:]
        c = torch.tanh(torch.matmul(torch.cat([xt, r * h], 1), p("candidate/kernel")) + p("candidate/bias"))
        new_h = u * h + (1 - u) * c
        live = (t < seq_len_m1).unsqueeze(1)
        outs.append(torch.where(live, new_h, torch.zeros_like(new_h)))
        h = torch.where(live, new_h, h)
    return torch.stack(outs, dim=1)


def gather_indexes(seq, positions):
    """Model/Modules/net_utils.py:82-92 (flat gather at b*L + pos)."""
    B, L, D = seq.shape
    flat = seq.reshape(B * L, D)
    return flat[torch.arange(B) * L + positions]


# ---------------------------------------------------------------- attention
def normalize(x, beta, gamma, eps=1e-8):
    """Time_Aware_Attention.normalize, time_aware_attention.py:7-34."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    return gamma * ((x - mean) / ((var + eps) ** 0.5)) + beta


def layer_norm(x, beta, gamma, eps=1e-12):
    """tf.contrib.layers.layer_norm via nn.batch_normalization [TF1.14]
    (net_utils.py:229-232): x*inv + (beta - mean*inv), inv = rsqrt(var+eps)*gamma."""
    mean = x.mean(dim=-1, keepdim=True)
    var = ((x - mean) ** 2).mean(dim=-1, keepdim=True)
    inv = torch.rsqrt(var + eps) * gamma
    return x * inv + (beta - mean * inv)


def time_aware_multihead_attention(w, scope, inner, q, k, key_len, query_len, tq, tk, num_heads):
    """time_aware_attention.py:215-456.  q:[B,Tq,D] k:[B,Tk,D] tq:[B,Tq] tk:[B,Tk]."""
    B, Tq, D = q.shape
    Tk = k.shape[1]
    g = lambda n: w[scope + n]
    Q = torch.relu(torch.matmul(q, g("dense/kernel")) + g("dense/bias"))
    K = torch.relu(torch.matmul(k, g("dense_1/kernel")) + g("dense_1/bias"))
    V = torch.relu(torch.matmul(k, g("dense_2/kernel")) + g("dense_2/bias"))
    s = scope + inner + "/"
    tqk = torch.tanh(torch.matmul(torch.matmul(q, w[s + "_time_input_w"]), k.transpose(1, 2)))
    decay = torch.log(torch.abs(tq.unsqueeze(2) - tk.unsqueeze(1)) + 1)
    decay = torch.tanh(decay * w[s + "_time_input_w1"] + w[s + "_time_input_b1"])
    gate = w[s + "time_output_w1"] * decay + w[s + "time_output_w2"] * tqk + w[s + "time_output_b"]
    d = D // num_heads
    Q_ = torch.cat(torch.split(Q, d, dim=2), dim=0)
    K_ = torch.cat(torch.split(K, d, dim=2), dim=0)
    V_ = torch.cat(torch.split(V, d, dim=2), dim=0)
    gate_ = torch.cat([gate] * num_heads, dim=0)
    out = torch.matmul(Q_, K_.transpose(1, 2))
    out = out * torch.sigmoid(gate_)
    out = out / (d ** 0.5)
    key_mask = (torch.arange(Tk).unsqueeze(0) < key_len.unsqueeze(1))           # [B,Tk]
    key_mask = key_mask.repeat(num_heads, 1).unsqueeze(1).expand(-1, Tq, -1)
    out = torch.where(key_mask, out, torch.full_like(out, MASK_VALUE))
    out = torch.softmax(out, dim=-1)
    q_mask = (torch.arange(Tq).unsqueeze(0) < query_len.unsqueeze(1)).to(out.dtype)
    out = out * q_mask.repeat(num_heads, 1).unsqueeze(2)
    att = out
    out = torch.matmul(out, V_)
    out = torch.cat(torch.split(out, B, dim=0), dim=2)
    out = out + q
    out = normalize(out, w[s + "ln/Variable"], w[s + "ln/Variable_1"])
    return out, att


def vanilla_attention(w, enc, dec, key_len, tq, tk, num_heads, num_blocks):
    """Decoder stack, time_aware_attention.py:524-556 (enc fixed, dec fed forward)."""
    B, _, D = dec.shape
    qlen = torch.ones(B, dtype=torch.long)
    for i in range(num_blocks):
        dec, _ = time_aware_multihead_attention(
            w, "NextItemDecoder/decoder/num_blocks_%d/" % i, "vanilla_attention",
            dec, enc, key_len, qlen, tq, tk, num_heads)
    return dec.reshape(-1, D)


def self_attention(w, enc, seq_len, t, num_heads, num_blocks):
    """Encoder stack, time_aware_attention.py:459-490."""
    for i in range(num_blocks):
        enc, _ = time_aware_multihead_attention(
            w, "UserHistoryEncoder/encoder/num_blocks_%d/" % i, "self_attention",
            enc, enc, seq_len, seq_len, t, t, num_heads)
    return enc


# ---------------------------------------------------------------- models
def _bf16_straight_through(t):
    """Value rounded to bf16 (round to nearest even), gradient of the identity: what a bf16 copy of an fp32
    master weight (or an operand cast in front of a bf16 matrix product) means for training."""
    return t + (t.to(torch.bfloat16).to(t.dtype) - t).detach()


def forward(model, w, feed, num_heads, num_blocks, regulation_rate, global_batch=None, score_dtype="f32"):
    """MTAM.build_model (Model/MTAMRec_model.py:61-92) or
    Time_Aware_self_Attention_model.build_model (Model/PISTRec_model.py:38-74),
    followed by base_model.output (Model/base_model.py:300-328).

    ``w`` holds the item table twice (``item_gather`` / ``item_score``) so that
    the gather gradient and the dense scoring gradient stay separable, as TF's
    IndexedSlices aggregation keeps them.
    ``global_batch``: mean divisor for data-parallel shards (default: local B).
    ``score_dtype`` "bf16" (an option of this build, BASELINE.json configs[4]; not in the reference): the
    item table is read in bf16 everywhere in the forward (history gathers and scoring), the scoring vector is
    rounded to bf16 in front of the catalog product; products and sums stay in the working precision.
    """
    user, x, item, cat, pos = get_embedding(w, feed, score_dtype)
    sl = feed["seq_length"]
    B = x.shape[0]
    if model in FAMILY:
        # the MTAM family, Model/MTAMRec_model.py:40-306, wired by the oracle's own table (oracle/family.py)
        cfg = FAMILY[model]
        if cfg["decoder"] == "plain":
            raise NotImplementedError("%s: live dropout (SURVEY.md F8), no parity statement possible" % model)
        if cfg["cell"] == "decay_new":
            hs = time_aware_gru(w, x, feed["timelast_list"], sl - 1)
        elif cfg["cell"] == "sigmoid":
            hs = seqrec_gru(w, x, feed["timelast_list"], feed["timenow_list"], sl - 1)
        else:
            hs = plain_gru(w, x, sl - 1)
        short = gather_indexes(hs, sl - 2)
        if cfg["short_ln"]:
            short = layer_norm(short, w[SHORT_LN + "beta"], w[SHORT_LN + "gamma"])
        last = short
        if cfg["decoder"] == "time_aware":
            user_history = hs if cfg["keys"] == "rnn" else x
            last = vanilla_attention(w, user_history, short.unsqueeze(1), sl,
                                     feed["target_item_time"].unsqueeze(1), feed["time_list"], num_heads,
                                     num_blocks)
        head = cfg["head_ln"]
        pred = layer_norm(last, w[head + "beta"], w[head + "gamma"])
        if cfg["output"] == "output_concat":
            # MTAM_hybird (Model/MTAMRec_model.py:272) + base_model.output_concat (Model/base_model.py:340-346)
            pred = torch.matmul(torch.cat([short, pred], 1), w["output_w"])
        l2 = 0.5 * (item ** 2).sum() + 0.5 * (cat ** 2).sum() + 0.5 * (pos ** 2).sum() \
            + 0.5 * (user ** 2).sum()
    else:
        enc = self_attention(w, x, sl, feed["time_list"], num_heads, num_blocks)
        long_term = gather_indexes(enc, sl - 1)
        pred = layer_norm(long_term, w["UserHistoryEncoder/LayerNorm/beta"],
                          w["UserHistoryEncoder/LayerNorm/gamma"])
        l2 = 0.5 * (item ** 2).sum() + 0.5 * (cat ** 2).sum() + 0.5 * (pos ** 2).sum()
        if model == TASA:
            # Model/attention_baseline_models.py:47-65: the same graph finished by base_model.output()
            # (Model/base_model.py:300-307), whose L2 sum includes the user embedding
            l2 = l2 + 0.5 * (user ** 2).sum()
    if score_dtype == "bf16":
        logits = torch.matmul(_bf16_straight_through(pred),
                              _bf16_straight_through(w["embedding_layer/item_score"]).t())
    else:
        logits = torch.matmul(pred, w["embedding_layer/item_score"].t())
    log_probs = torch.log_softmax(logits, dim=-1)
    one_hot = torch.nn.functional.one_hot(feed["target_item_id"], logits.shape[1]).to(logits.dtype)
    ce = -(log_probs * one_hot).sum(dim=-1)
    denom = B if global_batch is None else global_batch
    loss = regulation_rate * l2 + ce.sum() / denom
    return dict(pred=pred, logits=logits, ce=ce, l2=l2, loss=loss, x=x,
                user=user, item=item, cat=cat, pos=pos,
                hs=hs if model in FAMILY else None)


def split_item_table(arrays, dtype, requires_grad=True):
    """numpy name->array  =>  torch leaves with the item table duplicated."""
    w = {}
    for k, v in arrays.items():
        t = torch.from_numpy(np.ascontiguousarray(v)).to(dtype)
        if k == "embedding_layer/item":
            w["embedding_layer/item_gather"] = t.clone().requires_grad_(requires_grad)
            w["embedding_layer/item_score"] = t.clone().requires_grad_(requires_grad)
        else:
            w[k] = t.clone().requires_grad_(requires_grad)
    return w


def loss_and_grads(model, arrays, feed, num_heads, num_blocks, regulation_rate, dtype=torch.float32,
                   global_batch=None, score_dtype="f32"):
    """tf.gradients(loss, trainable) -- Model/base_model.py:292.

    Returns (out, grads, slot_sq) where grads[name] is the dense (summed)
    gradient per variable (None for variables the loss does not reach) and
    slot_sq is the sum over un-deduplicated IndexedSlices rows of ||row||^2
    for the four tables (what tf.global_norm sees, App D-5).
    """
    w = split_item_table(arrays, dtype)
    f = feed_to_torch(feed, dtype)
    out = forward(model, w, f, num_heads, num_blocks, regulation_rate, global_batch, score_dtype)
    for key in ("user", "item", "cat", "pos"):
        out[key].retain_grad()
    out["loss"].backward()
    grads = {}
    for k, t in w.items():
        if k in ("embedding_layer/item_gather", "embedding_layer/item_score"):
            continue
        grads[k] = None if t.grad is None else t.grad.detach().numpy()
    grads["embedding_layer/item"] = (w["embedding_layer/item_gather"].grad
                                     + w["embedding_layer/item_score"].grad).detach().numpy()
    slot_sq = {}
    for key in ("item", "cat", "pos"):
        slot_sq[key] = float((out[key].grad.double() ** 2).sum())
    slot_sq["user"] = float((out["user"].grad.double() ** 2).sum()) if out["user"].grad is not None else 0.0
    slot_sq["item_dense"] = float((w["embedding_layer/item_score"].grad.double() ** 2).sum())
    return out, grads, slot_sq


def global_norm(grads, slot_sq, model, tf_compat=True):
    """tf.global_norm over the gradient list (Model/base_model.py:294).
    tf_compat: tables contribute the un-deduplicated IndexedSlices norm."""
    total = 0.0
    for k, g in grads.items():
        if g is None:
            continue
        if tf_compat and k.startswith("embedding_layer/"):
            continue
        total += float((g.astype(np.float64) ** 2).sum())
    if tf_compat:
        total += slot_sq["item"] + slot_sq["item_dense"] + slot_sq["cat"] + slot_sq["pos"]
        if model in FAMILY or model == TASA:
            total += slot_sq["user"]
    return math.sqrt(total)


class AdamState(object):
    """tf.train.AdamOptimizer state [TF1.14]: m, v per variable and the two
    beta-power accumulators kept in float32."""

    def __init__(self, arrays):
        self.m = {k: np.zeros_like(v) for k, v in arrays.items()}
        self.v = {k: np.zeros_like(v) for k, v in arrays.items()}
        self.beta1_power = np.float32(0.9)
        self.beta2_power = np.float32(0.999)


class SlotState(object):
    """Slots of the reference's other optimizers (Model/base_model.py:71-80) [TF1.14]:
    adadelta: accum, accum_update (zeros); rmsprop: rms (ONES), momentum (zeros); sgd: none."""

    def __init__(self, kind, arrays):
        assert kind in ("sgd", "adadelta", "rmsprop")
        self.kind = kind
        one = kind == "rmsprop"
        self.s1 = {k: (np.ones_like(v) if one else np.zeros_like(v)) for k, v in arrays.items()}
        self.s2 = {k: np.zeros_like(v) for k, v in arrays.items()}


ROW_SPARSE = {"embedding_layer/category": "category_list", "embedding_layer/position": "position_list",
              "embedding_layer/user": "user_id"}


def apply_slot_optimizer(arrays, state, grads, scale, lr, feed):
    """GradientDescent / Adadelta (rho .95, eps 1e-8) / RMSProp (decay .9, momentum 0, eps 1e-10)
    .apply_gradients [TF1.14 training_ops].  The looked-up tables get IndexedSlices gradients: the
    sparse kernels run on the rows present in the batch only (after summing duplicates); the item
    table's IndexedSlices holds every row (dense scoring gradient concatenated with the lookups)."""
    f = np.float32
    lr = f(lr)
    for k, g in grads.items():
        if g is None:
            continue
        g = (g * scale).astype(np.float32)
        if k in ROW_SPARSE:
            rows = np.unique(np.asarray(feed[ROW_SPARSE[k]]).ravel())
        else:
            rows = slice(None)
        p, a, b, gg = arrays[k][rows], state.s1[k][rows], state.s2[k][rows], g[rows]
        if state.kind == "sgd":
            p = p - lr * gg
        elif state.kind == "adadelta":
            rho, eps = f(0.95), f(1e-8)
            a = a * rho + (gg * gg) * (f(1) - rho)
            upd = np.sqrt(b + eps) * (f(1) / np.sqrt(a + eps)) * gg
            p = p - upd * lr
            b = b * rho + (upd * upd) * (f(1) - rho)
        else:
            rho, eps = f(0.9), f(1e-10)
            if k.startswith("embedding_layer/"):
                a = a * rho + (gg * gg) * (f(1) - rho)            # SparseApplyRMSProp form
            else:
                a = a + (gg * gg - a) * (f(1) - rho)              # ApplyRMSProp form
            b = (gg * lr) / np.sqrt(a + eps)                      # momentum = 0
            p = p - b
        arrays[k][rows] = p.astype(np.float32)
        state.s1[k][rows] = a.astype(np.float32)
        state.s2[k][rows] = b.astype(np.float32)


def train_step(model, arrays, state, feed, lr, num_heads, num_blocks, regulation_rate,
               max_gradient_norm=1.0, tf_compat_norm=True, global_batch=None, score_dtype="f32"):
    """One ``sess.run([loss, merged, train_op])`` (Model/base_model.py:150-167,290-297):
    gradients -> clip_by_global_norm -> optimizer (Adam for an AdamState, else the SlotState's kind).
    Updates ``arrays``/``state`` in place."""
    out, grads, slot_sq = loss_and_grads(model, arrays, feed, num_heads, num_blocks,
                                         regulation_rate, torch.float32, global_batch, score_dtype)
    norm = np.float32(global_norm(grads, slot_sq, model, tf_compat_norm))
    c = np.float32(max_gradient_norm)
    scale = c * min(np.float32(1.0) / norm, np.float32(1.0) / c)          # clip_by_global_norm [TF1.14]
    if isinstance(state, SlotState):
        apply_slot_optimizer(arrays, state, grads, scale, lr, feed)
        return dict(loss=float(out["loss"].detach()), ce_mean=float(out["ce"].detach().mean()),
                    l2=float(out["l2"].detach()), global_norm=float(norm), scale=float(scale),
                    logits=out["logits"].detach().numpy(), pred=out["pred"].detach().numpy(), grads=grads)
    b1, b2, eps = np.float32(0.9), np.float32(0.999), np.float32(1e-8)
    lr32 = np.float32(lr)                                                 # lr placeholder is f64, cast to var dtype
    lr_t = lr32 * np.sqrt(np.float32(1) - state.beta2_power) / (np.float32(1) - state.beta1_power)
    for k, g in grads.items():
        if g is None:
            continue
        g = (g * scale).astype(np.float32)
        if k.startswith("embedding_layer/"):          # IndexedSlices path, _apply_sparse_shared
            state.m[k] = state.m[k] * b1 + g * (np.float32(1) - b1)
            state.v[k] = state.v[k] * b2 + (g * g) * (np.float32(1) - b2)
        else:                                         # ApplyAdam kernel
            state.m[k] = state.m[k] + (g - state.m[k]) * (np.float32(1) - b1)
            state.v[k] = state.v[k] + (g * g - state.v[k]) * (np.float32(1) - b2)
        arrays[k] = (arrays[k] - lr_t * state.m[k] / (np.sqrt(state.v[k]) + eps)).astype(np.float32)
    state.beta1_power = np.float32(state.beta1_power * b1)
    state.beta2_power = np.float32(state.beta2_power * b2)
    return dict(loss=float(out["loss"].detach()), ce_mean=float(out["ce"].detach().mean()), l2=float(out["l2"].detach()),
                global_norm=float(norm), scale=float(scale), logits=out["logits"].detach().numpy(),
                pred=out["pred"].detach().numpy(), grads=grads)


# ---------------------------------------------------------------- eval
def top_k(scores, k):
    """tf.nn.top_k [TF1.14]: descending, equal values -> lower index first."""
    scores = np.asarray(scores)
    order = np.argsort(-scores, axis=1, kind="stable")
    return order[:, :k].astype(np.int32)


def calculate_topK(indices, targets):
    """Model/base_model.py:215-242: recall and NDCG of one batch."""
    hits = 0
    ndcg = 0.0
    for row, tgt in zip(indices, targets):
        row = list(row)
        if tgt in row:
            hits += 1
            ndcg += math.log(2) / math.log(row.index(tgt) + 2)
    n = len(targets)
    return hits / n, ndcg / n


def metrics_topK(scores, targets, ks=(1, 5, 10, 30, 50)):
    """Model/base_model.py:188-213 -> (hr1, ndcg1, hr5, ndcg5, ...)."""
    out = []
    top = top_k(scores, max(ks))
    for k in ks:
        hr, nd = calculate_topK(top[:, :k], targets)
        out += [hr, nd]
    return tuple(out)
