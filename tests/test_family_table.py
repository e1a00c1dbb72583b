"""The MTAM family's wiring: the oracle's table (oracle/family.py) and the product's
(mtamrecommender_amd/Model/variables.py) are two independent restatements of
Model/MTAMRec_model.py:40-306.  Both are checked

* against each other (always), and
* against the reference file itself, read as TEXT (only where /root/reference exists -- the build container;
  nothing is imported or executed from it), so that a mis-wired member fails here instead of passing
  because product and checker share one table (round 1: MTAM_only_time_aware_RNN ran the decay_new cell).
"""
import os
import re

import pytest

from mtamrecommender_amd.Model import variables as V
from oracle import family as F

REF = "/root/reference/Model/MTAMRec_model.py"
REF_GRU = "/root/reference/Model/Modules/gru.py"

CELL_OF_PRODUCT = {"time": "decay_new", "seqrec": "sigmoid", "plain": "gru"}
KEYS_OF_PRODUCT = {"x": "x", "gru": "rnn", None: None}


def product_as_family(name):
    """The product's entry, translated into the oracle table's vocabulary."""
    cfg = V.MTAM_VARIANTS[name]
    return dict(cell=CELL_OF_PRODUCT[cfg["gru"]], keys=KEYS_OF_PRODUCT[cfg["keys"]], short_ln=cfg["short_ln"],
                decoder="time_aware" if cfg["attention"] else None, head_ln=V.head_ln_scope(name),
                output="output_concat" if cfg.get("head") == "concat" else "output")


def test_product_table_equals_oracle_table():
    assert set(V.MTAM_VARIANTS) == set(F.RUNNABLE)
    for name in F.RUNNABLE:
        assert product_as_family(name) == F.FAMILY[name], name
        assert V.gru_scope(name) == F.CELL_SCOPE[F.FAMILY[name]["cell"]], name
    assert V.SHORT_LN == F.SHORT_LN_SCOPE
    assert tuple(V.TIME_GATE) == tuple(F.TIME_GATE_VARS)


def walk_reference(text):
    """class name -> wiring, read off the text of Model/MTAMRec_model.py.  Comment lines are dropped first
    (the file keeps several commented-out alternatives)."""
    lines = [ln for ln in text.splitlines() if not ln.lstrip().startswith("#")]
    body, out, name = {}, {}, None
    for ln in lines:
        m = re.match(r"class\s+(\w+)\(MTAMRec_model\)", ln)
        if m:
            name = m.group(1)
            body[name] = []
        elif re.match(r"class\s+\w+", ln):
            name = None
        elif name:
            body[name].append(ln)
    for name, blines in body.items():
        src = "\n".join(blines)
        types = re.findall(r"type='([^']+)'", src)
        if "time_aware_gru_net(" in src:
            assert len(types) == 1, (name, types)
            cell = {"new": "decay_new", "T-SeqRec": "sigmoid"}[types[0]]
        else:
            assert re.search(r"\.gru_net\(", src), name
            cell = "gru"
        if "time_aware_attention.vanilla_attention(" in src:
            decoder = "time_aware"
        elif re.search(r"\battention\.vanilla_attention\(", src):
            decoder = "plain"
        else:
            decoder = None
        keys = None
        if decoder:
            m = re.search(r"user_history\s*=\s*self\.(\w+)", src)
            keys = {"behavior_list_embedding_dense": "x", "short_term_intent_temp": "rnn"}[m.group(1)]
        short_ln = bool(re.search(r"self\.short_term_intent\s*=\s*layer_norm\(self\.short_term_intent\)", src))
        # which variable scope is open where predict_behavior_emb's layer_norm is created
        scope, head_ln = None, None
        for ln in blines:
            m = re.search(r"with tf\.variable_scope\(['\"](\w+)['\"]\)", ln)
            if m:
                scope = m.group(1)
            if "self.predict_behavior_emb" in ln and "layer_norm(" in ln:
                head_ln = scope + "/LayerNorm/"
        output = "output_concat" if "self.output_concat()" in src else "output"
        assert ("self.output()" in src) != (output == "output_concat"), name
        out[name] = dict(cell=cell, keys=keys, short_ln=short_ln, decoder=decoder, head_ln=head_ln, output=output)
    return out


@pytest.mark.skipif(not os.path.exists(REF), reason="reference tree not present (GPU box)")
def test_both_tables_agree_with_the_reference_text():
    with open(REF, encoding="utf-8") as f:
        ref = walk_reference(f.read())
    assert set(ref) == set(F.FAMILY), (sorted(ref), sorted(F.FAMILY))
    for name, wiring in ref.items():
        assert F.FAMILY[name] == wiring, ("oracle table", name, F.FAMILY[name], wiring)
        if name in V.MTAM_VARIANTS:
            assert product_as_family(name) == wiring, ("product table", name, product_as_family(name), wiring)
    # the member round 1 had wrong, spelled out
    assert ref["MTAM_only_time_aware_RNN"]["cell"] == "sigmoid"
    assert V.MTAM_VARIANTS["MTAM_only_time_aware_RNN"]["gru"] == "seqrec"


@pytest.mark.skipif(not os.path.exists(REF_GRU), reason="reference tree not present (GPU box)")
def test_type_strings_build_the_cells_the_tables_assume():
    """Model/Modules/gru.py:69-92: 'T-SeqRec' -> TimeAwareGRUCell_sigmoid, 'new' -> TimeAwareGRUCell_decay_new."""
    with open(REF_GRU, encoding="utf-8") as f:
        src = f.read()
    m = re.search(r"if type\s*==\s*'T-SeqRec':\s*\n\s*cell = self\.(\w+)\(", src)
    assert m and re.search(r"def %s\(self, hidden_units\):\s*\n\s*cell = TimeAwareGRUCell_sigmoid\(" % m.group(1), src)
    m = re.search(r"elif type\s*==\s*'new':\s*\n\s*cell = self\.(\w+)\(", src)
    assert m and re.search(r"def %s\(self, hidden_units\):\s*\n\s*cell = TimeAwareGRUCell_decay_new\(" % m.group(1), src)


def test_walker_catches_the_round1_mistake():
    """The walk fails on a table that gives MTAM_only_time_aware_RNN the decay_new cell."""
    sample = '''
class MTAM_only_time_aware_RNN(MTAMRec_model):
    def build_model(self):
        with tf.variable_scope('ShortTermIntentEncoder'):
            self.short_term_intent_temp = self.gru_net_ins.time_aware_gru_net(hidden_units=self.num_units,
                                                                              type='T-SeqRec')
            self.predict_behavior_emb = layer_norm(self.short_term_intent)
        self.output()
'''
    w = walk_reference(sample)["MTAM_only_time_aware_RNN"]
    assert w["cell"] == "sigmoid" and w["head_ln"] == "ShortTermIntentEncoder/LayerNorm/"
    old_entry = dict(gru="time", keys=None, short_ln=False, attention=False)
    assert CELL_OF_PRODUCT[old_entry["gru"]] != w["cell"]
