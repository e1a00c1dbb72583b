"""libmtam_host.so (include/mtam_host.h): record parsing, batch packing and the prefetching batch iterator
against the Python route (Embedding.make_feed_dic_new + DataInput).  No GPU needed."""
import ctypes
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host(hip_lib):
    from mtamrecommender_amd import _host_lib
    return _host_lib.load()


def _records(n=37, L=12, seed=3):
    from mtamrecommender_amd.data.synthetic import SyntheticCatalog, make_records
    cat = SyntheticCatalog(50, 7, 20, seed=seed)
    return cat, make_records(cat, n, L, seed=seed + 1)


def test_header_and_binding_table_agree(host):
    from mtamrecommender_amd import _host_lib
    text = open(os.path.join(ROOT, "include", "mtam_host.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(mtam_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(_host_lib.SIGNATURES)
    lib = ctypes.CDLL(_host_lib.LIB_PATH)
    for name in declared:
        assert hasattr(lib, name), name
    assert host.mtam_host_version() >= 1


def test_text_round_trip_matches_eval(host, tmp_path):
    """The reference reads train_data.txt with eval(line) (Prepare/prepare_data_base.py:79-92) after writing
    str(tuple) per line (:335-337): the native parser returns the same records."""
    from mtamrecommender_amd.DataHandle.native_input import RecordSet
    cat, records = _records()
    path = tmp_path / "train_data.txt"
    path.write_text("".join(str(r) + "\n" for r in records))
    rs = RecordSet.from_file(path)
    assert len(rs) == len(records) and rs.max_length == max(len(r[1]) for r in records)
    for i, want in enumerate(records):
        got = rs.record(i)
        assert got[0] == want[0] and got[1] == want[1] and got[2] == want[2] and got[6] == want[6]
        assert got[3] == [float(x) for x in want[3]] and got[4] == [float(x) for x in want[4]]
        assert got[7] == [want[7][0], want[7][1], float(want[7][2])] and got[8] == want[8]
    # the in-memory constructor agrees with the parser
    rs2 = RecordSet.from_records(records)
    assert all(rs.record(i) == rs2.record(i) for i in range(len(records)))


@pytest.mark.parametrize("bad", ["(1, [1, 2], [1], [1, 2], [0, 1], [1, 0], [0, 1], [3, 1, 5], 3)",
                                 "(1, [1, 2], [1, 1], [1, 2], [0, 1], [1, 0], [0, 1], [3, 1], 3)",
                                 "1, [1]", "(1.5, [1], [1], [1], [0], [0], [0], [3, 1, 5], 2)"])
def test_malformed_lines_are_reported(host, bad):
    from mtamrecommender_amd.DataHandle.native_input import RecordSet
    with pytest.raises(ValueError) as e:
        RecordSet.from_text("(1, [1], [1], [1], [0], [0], [0], [3, 1, 5], 2)\n" + bad + "\n")
    assert "line 2" in str(e.value)


def test_pack_matches_make_feed_dic_new(host):
    from mtamrecommender_amd.DataHandle.native_input import BatchPacker, RecordSet
    from mtamrecommender_amd.Embedding.Behavior_embedding_time_aware_attention import \
        Behavior_embedding_time_aware_attention
    L = 12
    cat, records = _records(L=L)
    emb = Behavior_embedding_time_aware_attention(True, cat.user_count, cat.item_count, cat.category_count, L)
    emb.init_placeholders()
    rs = RecordSet.from_records(records)
    packer = BatchPacker(L, emb)
    idx = [5, 0, 36, 7, 7, 19]
    packed = packer.pack(rs, idx, lr=0.125)
    feed = emb.make_feed_dic_new([records[i] for i in idx])
    for name in ("user_id", "item_list", "category_list", "position_list", "target_item_id", "seq_length",
                 "time_list", "timelast_list", "timenow_list", "target_item_time"):
        assert np.array_equal(packed.field(name), feed[name]), name
    assert packed.field("lr")[0] == np.float32(0.125)
    assert len(packed) == 6 and packed.records()[2] == rs.record(36)


def test_pack_rejects_what_the_python_route_rejects(host):
    from mtamrecommender_amd.DataHandle.native_input import BatchPacker, RecordSet
    from mtamrecommender_amd.Embedding.Behavior_embedding_time_aware_attention import \
        Behavior_embedding_time_aware_attention
    cat, records = _records(L=12)
    emb = Behavior_embedding_time_aware_attention(True, cat.user_count, cat.item_count, cat.category_count, 12)
    rs = RecordSet.from_records(records)
    with pytest.raises(ValueError):                       # a record longer than length_of_user_history
        BatchPacker(6, emb).pack(rs, [int(np.argmax([r[8] for r in records]))])
    bad = list(records[0])
    bad[1] = list(bad[1])
    bad[1][0] = cat.item_count + 3                        # first id past the table (count + 3 rows)
    with pytest.raises(IndexError):
        BatchPacker(12, emb).pack(RecordSet.from_records([tuple(bad)]), [0])
    with pytest.raises(ValueError):
        BatchPacker(12, emb).pack(rs, [len(records)])     # record index out of range


def test_native_data_input_slices_like_data_input(host):
    from mtamrecommender_amd.DataHandle.get_input_data import DataInput
    from mtamrecommender_amd.DataHandle.native_input import BatchPacker, NativeDataInput, RecordSet, shuffled_index
    from mtamrecommender_amd.Embedding.Behavior_embedding_time_aware_attention import \
        Behavior_embedding_time_aware_attention
    cat, records = _records(n=37, L=12)
    emb = Behavior_embedding_time_aware_attention(True, cat.user_count, cat.item_count, cat.category_count, 12)
    emb.init_placeholders()
    rs = RecordSet.from_records(records)
    packer = BatchPacker(12, emb)
    for prefetch in (True, False):
        steps, sizes = [], []
        # a PackedBatch lives in a rotating pool of pinned arenas: it is consumed before the next ones are packed
        for (gs, gb), (ws, wb) in zip(NativeDataInput(rs, 8, packer, prefetch=prefetch), DataInput(records, 8)):
            assert gs == ws
            steps.append(gs)
            sizes.append(len(gb))
            assert np.array_equal(gb.field("item_list"), emb.make_feed_dic_new(wb)["item_list"])
            assert np.array_equal(gb.field("seq_length"), emb.make_feed_dic_new(wb)["seq_length"])
        assert steps == [1, 2, 3, 4, 5] and sizes == [8, 8, 8, 8, 5]
    # an epoch order: every record exactly once, reproducible from the seed
    order = shuffled_index(len(rs), 99)
    assert sorted(order.tolist()) == list(range(37)) and np.array_equal(order, shuffled_index(37, 99))
    assert not np.array_equal(order, np.arange(37))
    seen = np.concatenate([b.index for _, b in NativeDataInput(rs, 8, packer, index=order)])
    assert np.array_equal(seen, order)


def test_eval_between_two_train_batches_leaves_the_prefetched_train_batch_alone(host):
    """Equal train / test batch sizes (round-1 advice): a >= 3-batch evaluation pass between two train batches must
    not touch the arena that holds the already prefetched train batch -- every batch stream has its own pool."""
    from mtamrecommender_amd.DataHandle.native_input import BatchPacker, NativeDataInput, RecordSet
    from mtamrecommender_amd.Embedding.Behavior_embedding_time_aware_attention import \
        Behavior_embedding_time_aware_attention
    cat, records = _records(n=64, L=12)
    emb = Behavior_embedding_time_aware_attention(True, cat.user_count, cat.item_count, cat.category_count, 12)
    emb.init_placeholders()
    rs = RecordSet.from_records(records)
    packer = BatchPacker(12, emb)
    train = NativeDataInput(rs, 8, packer, consumer="train")
    step, first = next(train)                       # batch 2 is now being prefetched into the train pool
    pending = train.peek_prefetched()
    snapshot = pending.arena.clone()
    seen = 0
    for _, test_batch in NativeDataInput(rs, 8, packer, index=np.arange(63, -1, -1), consumer="eval"):
        assert test_batch.arena.data_ptr() != pending.arena.data_ptr()
        seen += 1
    assert seen == 8
    assert torch_equal(pending.arena, snapshot)
    step2, second = next(train)
    assert step2 == 2 and np.array_equal(second.field("item_list"),
                                         emb.make_feed_dic_new(records[8:16])["item_list"])
    # two anonymous iterators never share a pool either
    a, b = NativeDataInput(rs, 8, packer), NativeDataInput(rs, 8, packer)
    assert a.consumer != b.consumer and next(a)[1].arena.data_ptr() != next(b)[1].arena.data_ptr()


def torch_equal(x, y):
    import torch
    return torch.equal(x, y)


def test_sharded_iterators_partition_every_global_batch(host):
    """Data parallel: rank r's iterator packs its contiguous slice of each global batch (data_parallel.shard) and
    reports the global size; the slices of all ranks are a partition, in order."""
    from mtamrecommender_amd import data_parallel
    from mtamrecommender_amd.DataHandle.native_input import BatchPacker, NativeDataInput, RecordSet, shuffled_index
    from mtamrecommender_amd.Embedding.Behavior_embedding_time_aware_attention import \
        Behavior_embedding_time_aware_attention
    cat, records = _records(n=37, L=12)
    emb = Behavior_embedding_time_aware_attention(True, cat.user_count, cat.item_count, cat.category_count, 12)
    emb.init_placeholders()
    rs = RecordSet.from_records(records)
    packer = BatchPacker(12, emb)
    order = shuffled_index(len(rs), 5)
    world = 3
    its = [NativeDataInput(rs, 10, packer, index=order, shard=(r, world)) for r in range(world)]
    whole = NativeDataInput(rs, 10, packer, index=order)
    for (step, g), *parts in zip(whole, *its):
        assert all(s == step for s, _ in parts)
        assert all(b.global_size == len(g) for _, b in parts)
        assert np.array_equal(np.concatenate([b.index for _, b in parts]), g.index)
        want = [data_parallel.shard(list(g.index), r, world) for r in range(world)]
        assert [list(b.index) for _, b in parts] == want


def test_a_last_global_batch_smaller_than_the_world_is_dropped_on_every_rank(host):
    """21 records, global batch 10, 2 ranks: the last global batch has ONE record -- rank 0's slice would be empty
    while rank 1 waits inside the gradient exchange.  Decided on the global size, so both ranks drop it: both
    iterators yield the same number of steps and no empty batch (and so does the Python-list route)."""
    from mtamrecommender_amd import data_parallel
    from mtamrecommender_amd.DataHandle.get_input_data import DataInput
    from mtamrecommender_amd.DataHandle.native_input import BatchPacker, NativeDataInput, RecordSet
    from mtamrecommender_amd.Embedding.Behavior_embedding_time_aware_attention import \
        Behavior_embedding_time_aware_attention
    cat, records = _records(n=21, L=12)
    emb = Behavior_embedding_time_aware_attention(True, cat.user_count, cat.item_count, cat.category_count, 12)
    emb.init_placeholders()
    rs = RecordSet.from_records(records)
    packer = BatchPacker(12, emb)
    world = 2
    steps = [[(s, b.B, b.global_size) for s, b in NativeDataInput(rs, 10, packer, shard=(r, world))]
             for r in range(world)]
    assert [len(x) for x in steps] == [2, 2]
    assert all(B > 0 for x in steps for _, B, _ in x)
    assert [g for _, _, g in steps[0]] == [g for _, _, g in steps[1]] == [10, 10]
    # one rank keeps the partial batch, as the reference does
    assert [b.B for _, b in NativeDataInput(rs, 10, packer)] == [10, 10, 1]
    # the list route: the same rule, stated once
    kept = [len(b) for _, b in DataInput(records, 10) if data_parallel.keep_global_batch(len(b), world)]
    assert kept == [10, 10]
    # a partial batch that still gives every rank a sample stays (sizes differ by one)
    cat, records = _records(n=23, L=12)
    rs = RecordSet.from_records(records)
    sizes = [[b.B for _, b in NativeDataInput(rs, 10, packer, shard=(r, world))] for r in range(world)]
    assert sizes == [[5, 5, 1], [5, 5, 2]]
