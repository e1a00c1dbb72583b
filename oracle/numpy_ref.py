"""Second, independent restatement of the forward pass: numpy float64, one
sample at a time, written from SURVEY.md Appendix A rather than from
``mtam_oracle.py``.  TEST INFRASTRUCTURE ONLY (see mtam_oracle.py header;
PARITY UNPINNED applies equally).  Its only job is to disagree with the torch
restatement when one of the two mis-reads the reference.
"""
import numpy as np

try:
    from .family import CELL_SCOPE
except ImportError:
    from family import CELL_SCOPE

GRU_SCOPE = CELL_SCOPE["decay_new"]


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def _ln(x, beta, gamma, eps):
    mu = x.mean()
    var = ((x - mu) ** 2).mean()
    return (x - mu) / np.sqrt(var + eps) * gamma + beta


def _block(w, scope, inner, q, k, key_len, query_len, tq, tk, H):
    """Appendix A-3 for one sample. q:[Tq,D] k:[Tk,D]."""
    Tq, D = q.shape
    Tk = k.shape[0]
    d = D // H
    Q = np.maximum(q @ w[scope + "dense/kernel"] + w[scope + "dense/bias"], 0)
    K = np.maximum(k @ w[scope + "dense_1/kernel"] + w[scope + "dense_1/bias"], 0)
    V = np.maximum(k @ w[scope + "dense_2/kernel"] + w[scope + "dense_2/bias"], 0)
    s = scope + inner + "/"
    A = np.tanh((q @ w[s + "_time_input_w"]) @ k.T)
    delta = np.log(np.abs(tq[:, None] - tk[None, :]) + 1.0)
    Dk = np.tanh(delta * w[s + "_time_input_w1"] + w[s + "_time_input_b1"])
    G = w[s + "time_output_w1"] * Dk + w[s + "time_output_w2"] * A + w[s + "time_output_b"]
    out = np.zeros((Tq, D))
    for h in range(H):
        sl = slice(h * d, (h + 1) * d)
        S = (Q[:, sl] @ K[:, sl].T) * _sigmoid(G) / np.sqrt(d)
        S[:, key_len:] = -2.0 ** 32 + 1
        S = S - S.max(axis=1, keepdims=True)
        W = np.exp(S)
        W = W / W.sum(axis=1, keepdims=True)
        W[query_len:, :] = 0.0
        out[:, sl] = W @ V[:, sl]
    out = out + q
    beta, gamma = w[s + "ln/Variable"], w[s + "ln/Variable_1"]
    return np.stack([_ln(out[i], beta, gamma, 1e-8) for i in range(Tq)])


def forward(model, arrays, feed, num_heads, num_blocks, regulation_rate):
    w = {k: np.asarray(v, dtype=np.float64) for k, v in arrays.items()}
    B, L = feed["item_list"].shape
    E_i = w["embedding_layer/item"]
    D = E_i.shape[1]
    preds, l2 = [], 0.0
    g = lambda n: w[GRU_SCOPE + n]
    for b in range(B):
        I = E_i[feed["item_list"][b]]
        C = w["embedding_layer/category"][feed["category_list"][b]]
        P = w["embedding_layer/position"][feed["position_list"][b]]
        u = w["embedding_layer/user"][feed["user_id"][b]]
        X = np.maximum(np.concatenate([I, C], axis=1) @ w["position_embedding/dense4emb/kernel"], 0) + P
        l2 += 0.5 * ((I ** 2).sum() + (C ** 2).sum() + (P ** 2).sum())
        sl = int(feed["seq_length"][b])
        tl = np.asarray(feed["time_list"][b], dtype=np.float64)
        if model == "MTAM":
            l2 += 0.5 * (u ** 2).sum()
            h = np.zeros(D)
            for t in range(sl - 1):
                x = X[t]
                dl = float(feed["timelast_list"][b][t])
                tw = np.maximum(x * g("_time_kernel_w1") + g("_time_kernel_b1") + h * g("_time_history_w1"), 0)
                ts = np.maximum(g("_time_w1") * dl + g("_time_b1"), 0)
                T = _sigmoid(g("_time_kernel_w2") * tw + g("_time_w12") * ts + g("_time_b12"))
                ru = _sigmoid(np.concatenate([x, h]) @ g("gates/kernel") + g("gates/bias"))
                r, uu = ru[:D], ru[D:]
                c = np.tanh(np.concatenate([x, r * h]) @ g("candidate/kernel") + g("candidate/bias"))
                h = uu * h + (1 - uu) * c * T
            dec = h[None, :]                       # outputs[b, sl-2] == state after sl-1 steps
            tq = np.array([float(feed["target_item_time"][b])])
            for i in range(num_blocks):
                dec = _block(w, "NextItemDecoder/decoder/num_blocks_%d/" % i, "vanilla_attention",
                             dec, X, sl, 1, tq, tl, num_heads)
            v = dec[0]
            beta, gamma = w["NextItemDecoder/LayerNorm/beta"], w["NextItemDecoder/LayerNorm/gamma"]
        else:
            enc = X
            for i in range(num_blocks):
                enc = _block(w, "UserHistoryEncoder/encoder/num_blocks_%d/" % i, "self_attention",
                             enc, enc, sl, sl, tl, tl, num_heads)
            v = enc[sl - 1]
            beta, gamma = w["UserHistoryEncoder/LayerNorm/beta"], w["UserHistoryEncoder/LayerNorm/gamma"]
        preds.append(_ln(v, beta, gamma, 1e-12))
    pred = np.stack(preds)
    logits = pred @ E_i.T
    m = logits.max(axis=1, keepdims=True)
    lse = m[:, 0] + np.log(np.exp(logits - m).sum(axis=1))
    ce = lse - logits[np.arange(B), feed["target_item_id"]]
    loss = regulation_rate * l2 + ce.mean()
    return dict(pred=pred, logits=logits, ce=ce, l2=l2, loss=loss)
