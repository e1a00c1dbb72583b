"""Data preparation, record files and the batch feed against fixtures written BY THE REFERENCE ITSELF
(tests/golden/prepare/*, made by tests/golden/make_prepare_golden.py from /root/reference/Prepare in the build
container; only the fixtures travel).  Reference: Prepare/prepare_data_base.py:115-339, Prepare/mask_data_process.py:
28-71,158-202,244-262, Embedding/Behavior_embedding_time_aware_attention.py:146-192.  No GPU needed."""
import ast
import glob
import json
import os
import random
import types

import numpy as np
import pandas as pd
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = sorted(os.path.basename(p) for p in glob.glob(os.path.join(HERE, "golden", "prepare", "*")))


def _case(name):
    d = os.path.join(HERE, "golden", "prepare", name)
    flags = json.load(open(os.path.join(d, "flags.json")))
    params = json.load(open(os.path.join(d, "parameters.json")))
    origin = pd.read_csv(os.path.join(d, "origin.csv"))
    return d, flags, params, origin


def test_fixtures_exist():
    assert CASES == ["l50_keepdup", "l50_userlimit", "l6_dedup"]


@pytest.mark.parametrize("name", CASES)
def test_prepare_reproduces_the_reference_files_byte_for_byte(hip_lib, tmp_path, name):
    from mtamrecommender_amd.Prepare.prepare_data_base import prepare_data_base
    d, flags, params, origin = _case(name)
    seed = flags.pop("shuffle_seed")
    p = prepare_data_base(types.SimpleNamespace(**flags), origin, root=str(tmp_path))
    assert (p.item_count, p.user_count, p.category_count) == (params["item_count"], params["user_count"],
                                                              params["category_count"])
    assert {str(k): int(v) for k, v in p.item_category_dic.items()} == params["item_category"]
    assert [float(g) for g in p.gap] == params["gap"]
    random.seed(seed)
    train, test = p.get_train_test()
    assert (len(train), len(test)) == (params["n_train"], params["n_test"])
    for mine, ref in ((p.dataset_class_train, "train_data.txt"), (p.dataset_class_test, "test_data.txt")):
        assert open(mine, "rb").read() == open(os.path.join(d, ref), "rb").read(), ref


@pytest.mark.parametrize("name", CASES)
def test_prepare_is_seed_independent_as_a_set(hip_lib, tmp_path, name):
    """Another shuffle seed permutes the lines and changes nothing else."""
    from mtamrecommender_amd.Prepare.prepare_data_base import prepare_data_base
    d, flags, params, origin = _case(name)
    flags.pop("shuffle_seed")
    p = prepare_data_base(types.SimpleNamespace(**flags), origin, root=str(tmp_path))
    random.seed(12345)
    p.get_train_test()
    for mine, ref in ((p.dataset_class_train, "train_data.txt"), (p.dataset_class_test, "test_data.txt")):
        assert sorted(open(mine).read().splitlines()) == sorted(open(os.path.join(d, ref)).read().splitlines())


@pytest.mark.parametrize("name", CASES)
def test_native_parser_reads_the_reference_files(hip_lib, name):
    """libmtam_host.so on the reference-written text == the reference's own eval(line) (here: ast.literal_eval)."""
    from mtamrecommender_amd.DataHandle.native_input import RecordSet
    d, flags, params, origin = _case(name)
    for fn in ("train_data.txt", "test_data.txt"):
        lines = open(os.path.join(d, fn)).read().splitlines()
        want = [ast.literal_eval(ln) for ln in lines]
        rs = RecordSet.from_file(os.path.join(d, fn))
        assert len(rs) == len(want)
        for i, w in enumerate(want):
            g = rs.record(i)
            assert (g[0], g[1], g[2], g[6], g[8]) == (w[0], w[1], w[2], w[6], w[8])
            assert g[3] == [float(x) for x in w[3]] and g[4] == [float(x) for x in w[4]]
            assert g[5] == [float(x) for x in w[5]] and g[7] == [w[7][0], w[7][1], float(w[7][2])]


@pytest.mark.parametrize("name", CASES)
def test_both_feeds_equal_the_reference_padding(hip_lib, name):
    """make_feed_dic_new (Python) and the native packer, on the reference's records, against the oracle's
    np.pad restatement of Embedding/Behavior_embedding_time_aware_attention.py:146-192: bit-identical."""
    from mtamrecommender_amd.DataHandle.native_input import BatchPacker, RecordSet
    from mtamrecommender_amd.Embedding.Behavior_embedding_time_aware_attention import \
        Behavior_embedding_time_aware_attention
    from oracle import feed_ref
    d, flags, params, origin = _case(name)
    L = flags["length_of_user_history"]
    records = [ast.literal_eval(ln) for ln in open(os.path.join(d, "train_data.txt")).read().splitlines()]
    emb = Behavior_embedding_time_aware_attention(True, params["user_count"], params["item_count"],
                                                  params["category_count"], L)
    emb.init_placeholders()
    rs = RecordSet.from_file(os.path.join(d, "train_data.txt"))
    packer = BatchPacker(L, emb)
    rng = np.random.default_rng(5)
    # with L = 6 the reference still writes the target's position as min(index, 49) (prepare_data_base.py:295-298),
    # past the position table's L + 3 rows: TF's CPU gather raises on such a batch, and so does the packer
    fits = [i for i, r in enumerate(records) if max(r[6]) < L + 3]
    over = [i for i, r in enumerate(records) if max(r[6]) >= L + 3]
    assert (name == "l6_dedup") == bool(over)
    if over:
        with pytest.raises(IndexError):
            packer.pack(rs, over[:1], lr=0.5)
    for B in (1, 7, len(fits)):
        idx = [int(i) for i in rng.permutation(fits)[:B]]
        batch = [records[i] for i in idx]
        want = feed_ref.make_feed_dic_new(batch, L)
        got = emb.make_feed_dic_new(batch)
        packed = packer.pack(rs, idx, lr=0.5)
        for k, w in want.items():
            assert got[k].dtype == w.dtype and np.array_equal(got[k], w), k
            if k != "target_item_category":              # not part of the device arena (the model never reads it)
                pk = packed.field(k)
                assert pk.dtype == w.dtype and np.array_equal(pk.reshape(w.shape), w), k
    # every history is right-padded with zeros and positions clamp at 49 (prepare_data_base.py:295-298)
    full = feed_ref.make_feed_dic_new(records, L)
    for b, r in enumerate(records):
        assert not full["item_list"][b, r[8]:].any() and full["item_list"][b, r[8] - 1] == params["item_count"] + 1
        assert full["position_list"][b, r[8] - 1] <= 49
