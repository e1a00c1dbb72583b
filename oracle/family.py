"""The MTAM family as the reference wires it -- the ORACLE's own table.  TEST INFRASTRUCTURE ONLY.

Restated from the text of ``Model/MTAMRec_model.py`` (one entry per ``class ...(MTAMRec_model)``) and
``Model/Modules/gru.py:56-77`` (which cell a ``type=`` string builds), independently of the product's table in
``mtamrecommender_amd/Model/variables.py``.  ``tests/test_family_table.py`` walks the reference file as text and
asserts that BOTH tables say what it says, so a mis-wired member can no longer hide behind a shared table.

Fields
  cell       recurrent cell of the short-term intent encoder:
               "decay_new" = TimeAwareGRUCell_decay_new  (time_aware_gru_net(type='new'),      gru.py:72-73, :90-92)
               "sigmoid"   = TimeAwareGRUCell_sigmoid    (time_aware_gru_net(type='T-SeqRec'), gru.py:70-71, :85-87)
               "gru"       = tf GRUCell                  (gru_net(...),                        gru.py:56-63)
  keys       what ``user_history`` (the decoder's keys) is: "x" = behavior_list_embedding_dense,
             "rnn" = the recurrent net's outputs (``user_history = self.short_term_intent_temp``), None = no decoder
  short_ln   layer_norm applied to the gathered short-term intent before the decoder
  decoder    "time_aware" (Time_Aware_Attention.vanilla_attention), "plain" (Attention.vanilla_attention) or None
  head_ln    variable scope that holds the layer_norm producing predict_behavior_emb (None = no layer_norm)
  output     "output" (base_model.output, Model/base_model.py:300-328) or
             "output_concat" (Model/base_model.py:329-357)
"""

# TF 1.14 variable scopes of the three cells below ShortTermIntentEncoder (MultiRNNCell([cell]) under dynamic_rnn:
# rnn/multi_rnn_cell/cell_0/<snake-cased class name>/) [TF1.14 naming, best effort]
CELL_SCOPE = {
    "decay_new": "ShortTermIntentEncoder/rnn/multi_rnn_cell/cell_0/time_aware_gru_cell_decay_new/",   # time_aware_rnn.py:133
    "sigmoid": "ShortTermIntentEncoder/rnn/multi_rnn_cell/cell_0/time_aware_gru_cell_sigmoid/",       # time_aware_rnn.py:19
    "gru": "ShortTermIntentEncoder/rnn/multi_rnn_cell/cell_0/gru_cell/",                              # gru.py:13-39
}
SHORT_LN_SCOPE = "ShortTermIntentEncoder/LayerNorm/"
DECODER_LN_SCOPE = "NextItemDecoder/LayerNorm/"
# the five [Tq, Tk] gate tensors a time-aware attention block uses (time_aware_attention.py:295-312), in the order
# the oracle reads them
TIME_GATE_VARS = ("_time_input_w1", "_time_input_b1", "time_output_w1", "time_output_w2", "time_output_b")

FAMILY = {
    # Model/MTAMRec_model.py:40-59 -- type='T-SeqRec' (:51), no decoder, layer_norm inside ShortTermIntentEncoder (:58)
    "MTAM_only_time_aware_RNN": dict(cell="sigmoid", keys=None, short_ln=False, decoder=None,
                                     head_ln=SHORT_LN_SCOPE, output="output"),
    # :61-92 -- type='new' (:74), user_history = behavior_list_embedding_dense (:66), layer_norm in NextItemDecoder (:91)
    "MTAM": dict(cell="decay_new", keys="x", short_ln=False, decoder="time_aware",
                 head_ln=DECODER_LN_SCOPE, output="output"),
    # :93-127 -- gru_net (:101), keys x (:98)
    "MTAM_no_time_aware_rnn": dict(cell="gru", keys="x", short_ln=False, decoder="time_aware",
                                   head_ln=DECODER_LN_SCOPE, output="output"),
    # :128-165 -- type='new' (:142), Attention.vanilla_attention (:153), predict = hybird_preference with the layer_norm commented out (:157-158)
    "MTAM_no_time_aware_att": dict(cell="decay_new", keys="x", short_ln=False, decoder="plain",
                                   head_ln=None, output="output"),
    # :167-204 -- type='new' (:179), user_history = short_term_intent_temp (:180), layer_norm on the intent (:186)
    "MTAM_via_T_GRU": dict(cell="decay_new", keys="rnn", short_ln=True, decoder="time_aware",
                           head_ln=DECODER_LN_SCOPE, output="output"),
    # :206-238 -- gru_net (:211), user_history = short_term_intent_temp (:214), layer_norm on the intent (:220)
    "MTAM_via_rnn": dict(cell="gru", keys="rnn", short_ln=True, decoder="time_aware",
                         head_ln=DECODER_LN_SCOPE, output="output"),
    # :240-273 -- type='new' (:254), concat(short_term_intent, layer_norm(hybird_preference)) (:272), output_concat (:273)
    "MTAM_hybird": dict(cell="decay_new", keys="x", short_ln=False, decoder="time_aware",
                        head_ln=DECODER_LN_SCOPE, output="output_concat"),
    # :275-306 -- type='T-SeqRec' (:288), keys x (:280)
    "MTAM_with_T_SeqRec": dict(cell="sigmoid", keys="x", short_ln=False, decoder="time_aware",
                               head_ln=DECODER_LN_SCOPE, output="output"),
}

# members the oracle (and the product) can run: MTAM_no_time_aware_att applies tf.layers.dropout(training=True) in
# training AND evaluation (Model/Modules/multihead_attention.py) -- TF's random stream cannot be replayed
RUNNABLE = tuple(k for k, v in FAMILY.items() if v["decoder"] != "plain")
