"""Data preparation (Prepare/prepare_data_base.py + mask_data_process.py mirror): a seven-event toy log ->
expected records, file round trip through the native parser.  Expected values are worked by hand from the
reference's rules (prepare_data_base.py:252-314, mask_data_process.py:158-202,244-262)."""
import random

import pandas as pd


def _flags(tmp, L=4, **kw):
    from mtamrecommender_amd.config.model_parameter import model_parameter
    FLAGS = model_parameter().get_parameter("MTAMb1_movielen").FLAGS
    FLAGS.type, FLAGS.pos_embedding, FLAGS.experiment_data_type, FLAGS.causality = "toy", "time", "item_based", "unidirection"
    FLAGS.length_of_user_history, FLAGS.init_train_data, FLAGS.remove_duplicate = L, True, False
    FLAGS.user_count_limit = 1000
    for k, v in kw.items():
        setattr(FLAGS, k, v)
    return FLAGS


TOY = pd.DataFrame({
    # user "b" has 4 events (out of time order), user "a" has 3
    "user_id": ["b", "a", "b", "a", "b", "a", "b"],
    "item_id": [30, 10, 10, 20, 20, 30, 40],
    "cat_id": ["y", "x", "x", "x", "x", "y", "z"],
    "time_stamp": [7200 * 3, 3600 * 1, 3600 * 2, 3600 * 5, 3600 * 4 + 59, 3600 * 9, 3600 * 30],
})


def test_toy_log_gives_the_expected_records(hip_lib, tmp_path):
    from mtamrecommender_amd.Prepare.prepare_data_base import prepare_data_base
    random.seed(0)
    p = prepare_data_base(_flags(tmp_path), TOY, root=str(tmp_path))
    assert (p.user_count, p.item_count, p.category_count) == (2, 4, 3)
    assert p.item_category_dic == {0: 0, 1: 0, 2: 1, 3: 2}            # items 10,20,30,40 -> 0..3; cats x,y,z -> 0..2
    train, test = p.get_train_test()
    by_key = {(r[0], r[7][0], r[8]): r for r in train + test}
    # user a (id 0): events (item 0 @1h), (item 1 @5h), (item 2 @9h)
    #   target index 1 -> history [item 0], mask token item_count + 1 = 5, category_count + 1 = 4
    assert by_key[(0, 1, 2)] == (0, [0, 5], [0, 4], [1, 5], [0, 0], [4, 0], [0, 1], [1, 0, 5], 2)
    #   target index 2 (the last event -> test set)
    assert by_key[(0, 2, 3)] == (0, [0, 1, 5], [0, 0, 4], [1, 5, 9], [0, 4, 0], [8, 4, 0], [0, 1, 2], [2, 1, 9], 3)
    assert by_key[(0, 2, 3)] in test and by_key[(0, 1, 2)] in train
    # user b (id 1), sorted by time: (item 0 @2h), (item 1 @4h), (item 2 @6h), (item 3 @30h)
    #   L = 4: the last target keeps the L - 1 = 3 events before it
    assert by_key[(1, 3, 4)] == (1, [0, 1, 2, 5], [0, 0, 1, 4], [2, 4, 6, 30], [0, 2, 2, 0], [28, 26, 24, 0],
                                 [0, 1, 2, 3], [3, 2, 30], 4)
    assert len(train) == 3 and len(test) == 2


def test_history_is_cut_to_the_last_L_minus_1_events(hip_lib, tmp_path):
    from mtamrecommender_amd.Prepare.prepare_data_base import prepare_data_base
    p = prepare_data_base(_flags(tmp_path, L=3), TOY, root=str(tmp_path))
    train, test = p.get_train_test()
    last_b = [r for r in test if r[0] == 1][0]
    assert last_b[1] == [1, 2, 5] and last_b[6] == [0, 1, 3] and last_b[8] == 3      # position of the target: its index


def test_files_round_trip(hip_lib, tmp_path):
    from mtamrecommender_amd.Prepare.prepare_data_base import prepare_data_base
    FLAGS = _flags(tmp_path)
    p = prepare_data_base(FLAGS, TOY, root=str(tmp_path))
    train, test = p.get_train_test()
    FLAGS.init_train_data = False
    q = prepare_data_base(FLAGS, None, root=str(tmp_path))
    norm = lambda r: (r[0], r[1], r[2], [float(x) for x in r[3]], [float(x) for x in r[4]], [float(x) for x in r[5]],
                      r[6], [r[7][0], r[7][1], float(r[7][2])], r[8])
    assert [norm(r) for r in train] == q.train_set and [norm(r) for r in test] == q.test_set
    assert (q.item_count, q.user_count, q.category_count) == (4, 2, 3) and q.item_category_dic == p.item_category_dic
    assert list(q.gap) == [60, 3600, 86400, 172800, 345600]
