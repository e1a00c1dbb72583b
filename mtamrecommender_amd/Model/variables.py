"""Trainable variables of the time-aware models: names, shapes, initialisers.

One entry per ``tf.get_variable`` / ``add_variable`` / ``tf.layers`` kernel the
reference creates on this path.  Names are the TF 1.14
scope strings (best effort; they matter only for checkpoint interchange):

* tables        Embedding/base_embedding.py:46-60, U(+-sqrt(6/D)), ``count+3`` rows
* dense4emb     Embedding/Behavior_embedding_time_aware_attention.py:93-101
* GRU cell      Model/Modules/time_aware_rnn.py:160-225 (gate bias 1.0, 6 dead vectors)
* attention     Model/Modules/time_aware_attention.py:249-253,269-312,29-30
* head LN       Model/Modules/net_utils.py:229-232

Initialiser with none given = glorot-uniform, limit sqrt(6/(fan_in+fan_out));
a 1-D shape [n] has fan_in = fan_out = n.  TF's seeded streams cannot be
replayed outside TF, so values come from a numpy PCG64 stream; parity tests
inject the same arrays into the oracle and into the HIP path.
"""
import collections
import math

import numpy as np

GRU_SCOPE = "ShortTermIntentEncoder/rnn/multi_rnn_cell/cell_0/time_aware_gru_cell_decay_new/"
PLAIN_GRU_SCOPE = "ShortTermIntentEncoder/rnn/multi_rnn_cell/cell_0/gru_cell/"      # tf GRUCell (gru.py:13-39)
TSR_SCOPE = "ShortTermIntentEncoder/rnn/multi_rnn_cell/cell_0/time_aware_gru_cell_sigmoid/"   # time_aware_rnn.py:19-131
SHORT_LN = "ShortTermIntentEncoder/LayerNorm/"

# The MTAM family (Model/MTAMRec_model.py:40-306): which recurrent cell encodes the short-term intent,
# what the decoder attends over, and where layer norms sit.  The three members that need other
# kernels is not built: MTAM_no_time_aware_att (non-time-aware attention with live dropout).
# head="concat" (MTAM_hybird): predict = concat(short-term intent, layer_norm(decoder)) . output_w before the
# catalog product (base_model.output_concat, Model/base_model.py:329-357).
MTAM_VARIANTS = {
    # name: (gru cell, attention keys, layer_norm on the short-term intent, attention decoder)
    "MTAM": dict(gru="time", keys="x", short_ln=False, attention=True),                       # :61-92
    # type='T-SeqRec' = TimeAwareGRUCell_sigmoid (:51, Model/Modules/gru.py:70-71); experiment_type 'T_GRU'
    "MTAM_only_time_aware_RNN": dict(gru="seqrec", keys=None, short_ln=False, attention=False),  # :40-59
    "MTAM_no_time_aware_rnn": dict(gru="plain", keys="x", short_ln=False, attention=True),     # :93-127
    "MTAM_via_T_GRU": dict(gru="time", keys="gru", short_ln=True, attention=True),             # :167-204
    "MTAM_via_rnn": dict(gru="plain", keys="gru", short_ln=True, attention=True),              # :206-238
    "MTAM_with_T_SeqRec": dict(gru="seqrec", keys="x", short_ln=False, attention=True),        # :275-306
    "MTAM_hybird": dict(gru="time", keys="x", short_ln=False, attention=True, head="concat"),  # :240-273
}


def gru_scope(variant):
    return {"time": GRU_SCOPE, "plain": PLAIN_GRU_SCOPE, "seqrec": TSR_SCOPE}[MTAM_VARIANTS[variant]["gru"]]


def head_ln_scope(variant):
    """The layer_norm that produces predict_behavior_emb: inside NextItemDecoder for the members with a
    decoder, inside ShortTermIntentEncoder for MTAM_only_time_aware_RNN (:58)."""
    return "NextItemDecoder/LayerNorm/" if MTAM_VARIANTS[variant]["attention"] else SHORT_LN
GRU_USED = ("_time_kernel_w1", "_time_kernel_b1", "_time_history_w1", "_time_w1",
            "_time_b1", "_time_kernel_w2", "_time_w12", "_time_b12")
GRU_DEAD = ("_time_history_b1", "_time_kernel_b2", "_time_history_w2", "_time_history_b2",
            "_time_w2", "_time_b2")
# TimeAwareGRUCell_sigmoid (time_aware_rnn.py:79-106): [D] vectors of the two time inputs, then per gate
# an input kernel, a time kernel (both [D, D]) and a bias
TSR_VEC = ("_time_input_w1", "_time_input_bias1", "_time_input_w2", "_time_input_bias2")
TSR_MAT = ("_time_kernel_w1", "_time_kernel_t1", "_time_kernel_w2", "_time_kernel_t2")
TSR_BIAS = ("_time_bias1", "_time_bias2")
TIME_GATE = ("_time_input_w1", "_time_input_b1", "time_output_w1", "time_output_w2",
             "time_output_b")

VarSpec = collections.namedtuple("VarSpec", "name shape init trainable_grad")
# init: ("uniform", limit) | ("const", value); trainable_grad False = variable
# exists but receives a None gradient in the reference (never updated).


def _glorot(shape):
    if len(shape) == 1:
        fan_in = fan_out = shape[0]
    else:
        fan_in, fan_out = shape[0], shape[1]
    return ("uniform", math.sqrt(6.0 / (fan_in + fan_out)))


def table_specs(user_count, item_count, category_count, position_count, D):
    r = ("uniform", float(np.sqrt(np.float32(6.0 / D))))
    return [VarSpec("embedding_layer/user", (user_count + 3, D), r, True),
            VarSpec("embedding_layer/item", (item_count + 3, D), r, True),
            VarSpec("embedding_layer/category", (category_count + 3, D), r, True),
            VarSpec("embedding_layer/position", (position_count + 3, D), r, True)]


def attention_block_specs(scope, inner, D, Tq, Tk):
    """One ``time_aware_multihead_attention`` block (time_aware_attention.py:215-456)."""
    specs = []
    for layer in ("dense", "dense_1", "dense_2"):           # Q, K, V projections
        specs.append(VarSpec(scope + layer + "/kernel", (D, D), _glorot((D, D)), True))
        specs.append(VarSpec(scope + layer + "/bias", (D,), ("const", 0.0), True))
    specs.append(VarSpec(scope + inner + "/_time_input_w", (D, D), _glorot((D, D)), True))
    for name in TIME_GATE:
        specs.append(VarSpec(scope + inner + "/" + name, (Tq, Tk), _glorot((Tq, Tk)), True))
    specs.append(VarSpec(scope + inner + "/time_output_w3", (Tq, Tk), _glorot((Tq, Tk)), False))
    specs.append(VarSpec(scope + inner + "/ln/Variable", (D,), ("const", 0.0), True))    # beta
    specs.append(VarSpec(scope + inner + "/ln/Variable_1", (D,), ("const", 1.0), True))  # gamma
    return specs


def mtam_dense_specs(D, L, num_blocks, variant="MTAM"):
    cfg = MTAM_VARIANTS[variant]
    G = gru_scope(variant)
    specs = [VarSpec("position_embedding/dense4emb/kernel", (2 * D, D), _glorot((2 * D, D)), True),
             VarSpec(G + "gates/kernel", (2 * D, 2 * D), _glorot((2 * D, 2 * D)), True),
             VarSpec(G + "gates/bias", (2 * D,), ("const", 1.0), True),
             VarSpec(G + "candidate/kernel", (2 * D, D), _glorot((2 * D, D)), True),
             VarSpec(G + "candidate/bias", (D,), ("const", 0.0), True)]
    if cfg["gru"] == "time":
        for name in GRU_USED:
            specs.append(VarSpec(G + name, (D,), _glorot((D,)), True))
        for name in GRU_DEAD:
            specs.append(VarSpec(G + name, (D,), _glorot((D,)), False))
    if cfg["gru"] == "seqrec":
        for name in TSR_VEC + TSR_BIAS:
            specs.append(VarSpec(G + name, (D,), _glorot((D,)), True))
        for name in TSR_MAT:
            specs.append(VarSpec(G + name, (D, D), _glorot((D, D)), True))
    if cfg["short_ln"]:
        specs.append(VarSpec(SHORT_LN + "beta", (D,), ("const", 0.0), True))
        specs.append(VarSpec(SHORT_LN + "gamma", (D,), ("const", 1.0), True))
    if cfg["attention"]:
        for i in range(num_blocks):
            specs += attention_block_specs("NextItemDecoder/decoder/num_blocks_%d/" % i,
                                           "vanilla_attention", D, 1, L)
    head = head_ln_scope(variant)
    specs.append(VarSpec(head + "beta", (D,), ("const", 0.0), True))
    specs.append(VarSpec(head + "gamma", (D,), ("const", 1.0), True))
    if cfg.get("head") == "concat":
        specs.append(VarSpec("output_w", (2 * D, D), _glorot((2 * D, D)), True))       # base_model.py:340-342
    return specs


def pistrec_dense_specs(D, L, num_blocks):
    specs = [VarSpec("position_embedding/dense4emb/kernel", (2 * D, D), _glorot((2 * D, D)), True)]
    for i in range(num_blocks):
        specs += attention_block_specs("UserHistoryEncoder/encoder/num_blocks_%d/" % i,
                                       "self_attention", D, L, L)
    specs.append(VarSpec("UserHistoryEncoder/LayerNorm/beta", (D,), ("const", 0.0), True))
    specs.append(VarSpec("UserHistoryEncoder/LayerNorm/gamma", (D,), ("const", 1.0), True))
    return specs


def model_specs(model, user_count, item_count, category_count, L, D, num_blocks):
    tables = table_specs(user_count, item_count, category_count, L, D)
    if model in MTAM_VARIANTS:
        return tables + mtam_dense_specs(D, L, num_blocks, model)
    if model == "PISTRec":
        return tables + pistrec_dense_specs(D, L, num_blocks)
    raise ValueError("unknown model family: %s" % model)


def init_variables(specs, seed=1234):
    """name -> float32 array, drawn from one PCG64 stream in spec order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    out = collections.OrderedDict()
    for spec in specs:
        kind, value = spec.init
        if kind == "uniform":
            out[spec.name] = rng.uniform(-value, value, size=spec.shape).astype(np.float32)
        else:
            out[spec.name] = np.full(spec.shape, value, dtype=np.float32)
    return out
