"""Every variable the reference's graph creates on this path -- the ORACLE's own list.  TEST INFRASTRUCTURE ONLY.

One entry per ``tf.get_variable`` / ``add_variable`` / ``tf.layers.dense`` / ``tf.Variable`` /
``contrib.layers.layer_norm`` the reference executes for MTAM, its family members and the time-aware
self-attention model, restated from the reference TEXT with the file:line of each declaration, its shape, its
initialiser and whether the graph ever READS it ("live") or only declares it ("dead": ``tf.gradients`` returns None
for it, ``clip_by_global_norm`` and ``apply_gradients`` skip it, it is never updated -- SURVEY.md App D-7).

Independent of the product's table (``mtamrecommender_amd/Model/variables.py``): golden fixtures and kernel tests
take names, shapes and initial values from HERE; ``tests/test_variable_specs.py`` walks the reference files as text
(``get_variable(`` / ``add_variable(`` / ``tf.layers.dense(`` / ``tf.Variable(``) and asserts that this table and
the product's both say what the text says -- same names, same shape expressions, the same 7 dead variables.

Default initialiser (none given) = glorot-uniform, limit sqrt(6 / (fan_in + fan_out)); a 1-D shape [n] has
fan_in = fan_out = n [TF1.14 variable_scope default + init_ops._compute_fans].  TF's seeded streams cannot be
replayed outside TF, so initial VALUES are drawn from a numpy PCG64 stream (``init_arrays``).
"""
import collections
import math

import numpy as np

try:
    from .family import CELL_SCOPE, DECODER_LN_SCOPE, FAMILY, SHORT_LN_SCOPE
except ImportError:                                     # imported as a top-level module by some tools
    from family import CELL_SCOPE, DECODER_LN_SCOPE, FAMILY, SHORT_LN_SCOPE

Var = collections.namedtuple("Var", "name shape init live cite")
# init: ("glorot", limit) | ("uniform", limit) | ("const", value)

RNN = "Model/Modules/time_aware_rnn.py"
ATT = "Model/Modules/time_aware_attention.py"
EMB = "Embedding/Behavior_embedding_time_aware_attention.py"


def glorot(shape):
    fan_in, fan_out = (shape[0], shape[0]) if len(shape) == 1 else (shape[0], shape[1])
    return ("glorot", math.sqrt(6.0 / (fan_in + fan_out)))


# ---------------------------------------------------------------------------------------------------- tables
def table_vars(user_count, item_count, category_count, position_count, D):
    """``init_embedding_lookup_table`` (Embedding/base_embedding.py:46-60): ``[count + 3, D]`` rows (the + 3 at
    Embedding/Behavior_embedding_time_aware_attention.py:64,71,78,86), U(-r, r) with r = sqrt(float32(6 / D))."""
    r = ("uniform", float(np.sqrt(np.float32(6.0 / D))))
    cite = "Embedding/base_embedding.py:56"
    return [Var("embedding_layer/user", (user_count + 3, D), r, True, cite + "; " + EMB + ":64"),
            Var("embedding_layer/item", (item_count + 3, D), r, True, cite + "; " + EMB + ":71"),
            Var("embedding_layer/category", (category_count + 3, D), r, True, cite + "; " + EMB + ":78"),
            Var("embedding_layer/position", (position_count + 3, D), r, True, cite + "; " + EMB + ":86")]


def dense4emb_vars(D):
    """``tf.layers.dense(concat(item, category), num_units, relu, use_bias=False, name='dense4emb')`` inside
    ``variable_scope("position_embedding")`` (:93-101): one kernel, no bias."""
    return [Var("position_embedding/dense4emb/kernel", (2 * D, D), glorot((2 * D, D)), True, EMB + ":98")]


# ------------------------------------------------------------------------------------------------ recurrent cells
# TimeAwareGRUCell_decay_new.call declares 14 [num_units] vectors (:197-225) and reads 8 of them (:227,235-236);
# the other 6 appear only in commented-out lines.
DECAY_NEW_LIVE = (("_time_kernel_w1", 197), ("_time_kernel_b1", 199), ("_time_history_w1", 201), ("_time_w1", 205),
                  ("_time_w12", 207), ("_time_b1", 209), ("_time_b12", 211), ("_time_kernel_w2", 214))
DECAY_NEW_DEAD = (("_time_history_b1", 203), ("_time_kernel_b2", 216), ("_time_history_w2", 218),
                  ("_time_history_b2", 220), ("_time_w2", 222), ("_time_b2", 224))
# TimeAwareGRUCell_sigmoid.call (:81-100): four [num_units] time-input vectors, per time gate an input kernel
# [input_size, num_units], a time kernel [num_units, num_units] and a bias; all ten are read (:103-116)
SIGMOID_VEC = (("_time_input_w1", 81), ("_time_input_bias1", 83), ("_time_input_w2", 85), ("_time_input_bias2", 87),
               ("_time_bias1", 93), ("_time_bias2", 99))
SIGMOID_MAT = (("_time_kernel_w1", 89), ("_time_kernel_t1", 91), ("_time_kernel_w2", 95), ("_time_kernel_t2", 97))


def gru_kernel_vars(scope, D, cite):
    """``build()`` of all three cells: gates [input_depth + units, 2 units] + bias (init 1.0), candidate
    [input_depth + units, units] + bias (init 0); input_depth = (D + 2) - 2 for the two time-aware cells
    (time_aware_rnn.py:49,165), D for tf's GRUCell."""
    return [Var(scope + "gates/kernel", (2 * D, 2 * D), glorot((2 * D, 2 * D)), True, cite[0]),
            Var(scope + "gates/bias", (2 * D,), ("const", 1.0), True, cite[1]),
            Var(scope + "candidate/kernel", (2 * D, D), glorot((2 * D, D)), True, cite[2]),
            Var(scope + "candidate/bias", (D,), ("const", 0.0), True, cite[3])]


def cell_vars(cell, D):
    S = CELL_SCOPE[cell]
    if cell == "decay_new":
        out = gru_kernel_vars(S, D, [RNN + ":166", RNN + ":170", RNN + ":176", RNN + ":180"])
        out += [Var(S + n, (D,), glorot((D,)), True, "%s:%d" % (RNN, ln)) for n, ln in DECAY_NEW_LIVE]
        out += [Var(S + n, (D,), glorot((D,)), False, "%s:%d" % (RNN, ln)) for n, ln in DECAY_NEW_DEAD]
        return out
    if cell == "sigmoid":
        out = gru_kernel_vars(S, D, [RNN + ":50", RNN + ":54", RNN + ":60", RNN + ":64"])
        out += [Var(S + n, (D,), glorot((D,)), True, "%s:%d" % (RNN, ln)) for n, ln in SIGMOID_VEC]
        out += [Var(S + n, (D, D), glorot((D, D)), True, "%s:%d" % (RNN, ln)) for n, ln in SIGMOID_MAT]
        return out
    if cell == "gru":                       # tensorflow GRUCell, Model/Modules/gru.py:2,21 [TF1.14 rnn_cell_impl]
        return gru_kernel_vars(S, D, ["Model/Modules/gru.py:21"] * 4)
    raise ValueError(cell)


# --------------------------------------------------------------------------------------------------- attention
# the [Tq, Tk] tensors of one time_aware_multihead_attention block in declaration order (:295-312);
# time_output_w3 (:307) is declared and never read (:350 uses w1, w2, b only)
GATE_LIVE = (("_time_input_w1", 295), ("_time_input_b1", 298), ("time_output_w1", 301), ("time_output_w2", 304),
             ("time_output_b", 310))
GATE_DEAD = (("time_output_w3", 307),)


def attention_block_vars(scope, inner, D, Tq, Tk):
    """``time_aware_multihead_attention`` (:215-456).  The three ``tf.layers.dense`` calls (:249,251,253) sit
    OUTSIDE ``variable_scope(scope)`` (:257), so TF names them dense, dense_1, dense_2 directly under the block
    scope; everything else lives under ``scope`` = "vanilla_attention" / "self_attention"; ``normalize`` (:451)
    creates two unnamed ``tf.Variable`` in scope "ln" (:29-30): Variable = beta, Variable_1 = gamma."""
    out = []
    for layer, ln in (("dense", 249), ("dense_1", 251), ("dense_2", 253)):
        out.append(Var(scope + layer + "/kernel", (D, D), glorot((D, D)), True, "%s:%d" % (ATT, ln)))
        out.append(Var(scope + layer + "/bias", (D,), ("const", 0.0), True, "%s:%d" % (ATT, ln)))
    s = scope + inner + "/"
    out.append(Var(s + "_time_input_w", (D, D), glorot((D, D)), True, ATT + ":269"))
    decl = sorted(GATE_LIVE + GATE_DEAD, key=lambda x: x[1])
    dead = {n for n, _ in GATE_DEAD}
    out += [Var(s + n, (Tq, Tk), glorot((Tq, Tk)), n not in dead, "%s:%d" % (ATT, ln)) for n, ln in decl]
    out.append(Var(s + "ln/Variable", (D,), ("const", 0.0), True, ATT + ":29"))
    out.append(Var(s + "ln/Variable_1", (D,), ("const", 1.0), True, ATT + ":30"))
    return out


def layer_norm_vars(scope, D, cite):
    """``tf.contrib.layers.layer_norm`` (Model/Modules/net_utils.py:229-232): beta 0, gamma 1 in scope LayerNorm."""
    return [Var(scope + "beta", (D,), ("const", 0.0), True, cite), Var(scope + "gamma", (D,), ("const", 1.0), True, cite)]


# ------------------------------------------------------------------------------------------------------ models
# the line of Model/MTAMRec_model.py at which each member applies layer_norm to produce predict_behavior_emb
HEAD_LN_LINE = {"MTAM_only_time_aware_RNN": 58, "MTAM": 91, "MTAM_no_time_aware_rnn": 126, "MTAM_via_T_GRU": 198,
                "MTAM_via_rnn": 232, "MTAM_hybird": 272, "MTAM_with_T_SeqRec": 305}
SHORT_LN_LINE = {"MTAM_via_T_GRU": 186, "MTAM_via_rnn": 220}


def dense_vars(model, L, D, num_blocks):
    """Non-table variables in graph-construction order."""
    out = dense4emb_vars(D)
    if model == "PISTRec":                  # Time_Aware_self_Attention_model, Model/PISTRec_model.py:38-74
        for i in range(num_blocks):
            out += attention_block_vars("UserHistoryEncoder/encoder/num_blocks_%d/" % i, "self_attention", D, L, L)
        return out + layer_norm_vars("UserHistoryEncoder/LayerNorm/", D, "Model/PISTRec_model.py:53")
    f = FAMILY[model]
    out += cell_vars(f["cell"], D)
    if f["short_ln"]:
        out += layer_norm_vars(SHORT_LN_SCOPE, D, "Model/MTAMRec_model.py:%d" % SHORT_LN_LINE[model])
    if f["decoder"] == "time_aware":
        for i in range(num_blocks):
            out += attention_block_vars("NextItemDecoder/decoder/num_blocks_%d/" % i, "vanilla_attention", D, 1, L)
    elif f["decoder"] is not None:
        raise ValueError("%s: the plain attention decoder is not restated (live dropout, SURVEY.md F8)" % model)
    if f["head_ln"] is not None:
        out += layer_norm_vars(f["head_ln"], D, "Model/MTAMRec_model.py:%d" % HEAD_LN_LINE[model])
    if f["output"] == "output_concat":
        out.append(Var("output_w", (2 * D, D), glorot((2 * D, D)), True, "Model/base_model.py:340-342"))
    return out


def model_vars(model, user_count, item_count, category_count, L, D, num_blocks):
    return table_vars(user_count, item_count, category_count, L, D) + dense_vars(model, L, D, num_blocks)


def live_names(vars_):
    return [v.name for v in vars_ if v.live]


def dead_names(vars_):
    return [v.name for v in vars_ if not v.live]


def init_arrays(vars_, seed=1234, jitter=0.0):
    """name -> float32 array, one PCG64 stream in table order.  ``jitter``: N(0, jitter) added to every bias-like
    tensor (1-D, or a [1, Tk] gate row) so that zeros / ones initialisers do not hide a sign or an ordering error."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = collections.OrderedDict()
    for v in vars_:
        kind, value = v.init
        if kind == "const":
            out[v.name] = np.full(v.shape, value, dtype=np.float32)
        else:
            out[v.name] = rng.uniform(-value, value, size=v.shape).astype(np.float32)
    if jitter:
        for k, a in out.items():
            if a.ndim == 1 or a.shape[0] == 1:
                out[k] = (a + rng.normal(0, jitter, a.shape)).astype(np.float32)
    return out
