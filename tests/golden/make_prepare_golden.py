"""Runs the REFERENCE's own data preparation (imported from /root/reference, build container only) on small
synthetic interaction logs and commits what it wrote as fixtures under tests/golden/prepare/<case>/:

    origin.csv        the input interactions (user_id, item_id, cat_id, time_stamp) -- made here, seeded
    flags.json        the FLAGS fields the reference's Prepare code reads (a stub namespace: the reference's
                      config/model_parameter.py needs tf.app.flags, TensorFlow is not installed)
    train_data.txt    |  byte for byte what prepare_data_base.get_train_test() saved
    test_data.txt     |  (Prepare/prepare_data_base.py:161-216,334-339)
    parameters.json   item_count / user_count / category_count / gap / item_category read off the reference object
                      (the reference pickles them, :204-211; a pickle is not a fixture)

What runs from the reference: Prepare/prepare_data_base.py (prepare_data_base.__init__ -> get_gap_list, map_process;
get_train_test -> data_handle_process) and Prepare/mask_data_process.py, plus util/model_log.create_log for their
logger.  Only the fixtures travel; tests/test_prepare_golden.py compares this repo's Prepare/ mirror, the native
record parser (libmtam_host.so) and both feed packers with them.

Environment notes (recorded in DESIGN.md section 6):
* numpy 2.x prints numpy scalars inside containers as ``np.int64(3)``; the reference's ``str(tuple)`` lines carry one
  numpy scalar (target_category, looked up in a dict of numpy ints, prepare_data_base.py:136-138,300) and its own
  loader ``eval(line)`` cannot read that form.  ``np.set_printoptions(legacy="1.25")`` -- numpy's documented switch
  back to the 1.x scalar repr the reference was written against -- is set before the run.
* ``random.seed`` is set before get_train_test() so that the two ``random.shuffle`` calls (:191-192) are replayable.

    python tests/golden/make_prepare_golden.py
"""
import json
import os
import random
import shutil
import sys
import tempfile
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = "/root/reference"

CASES = {
    # name: (seed, L = length_of_user_history, remove_duplicate, user_count_limit, shuffle seed)
    "l50_keepdup": (101, 50, False, 100000, 7),
    "l6_dedup": (202, 6, True, 100000, 8),
    "l50_userlimit": (303, 50, False, 3, 9),
}


def make_origin(seed):
    """A small log with the cases the reference's rules branch on: a user with ONE event (no sample), users with
    2 events, a user with more than 50 events (position clamp at 49, history cut to L - 1), exact duplicate rows,
    out-of-order rows, two events of one user in the same second, string and integer raw ids."""
    rng = np.random.default_rng(seed)
    items = ["i%03d" % i for i in range(40)]
    cat_of = {it: "c%d" % (int(it[1:]) % 7) for it in items}
    rows = []
    t0 = 1_000_000_000
    lengths = {"u_one": 1, "u_two": 2, "u_twob": 2, "u_long": 63, "u_mid": 9, "u_dup": 6, "u_tie": 5}
    for u, n in lengths.items():
        t = t0 + int(rng.integers(0, 10 ** 6))
        for _ in range(n):
            it = items[int(rng.integers(0, len(items)))]
            t += int(rng.integers(30, 5 * 86400))
            rows.append((u, it, cat_of[it], t))
    for k in range(3):                                       # exact duplicates of rows of u_dup (and one of u_mid)
        src = [r for r in rows if r[0] == "u_dup"][k + 1]
        rows.append(src)
    rows.append([r for r in rows if r[0] == "u_mid"][4])
    tie = [r for r in rows if r[0] == "u_tie"][2]            # a second event in the same second
    rows.append(("u_tie", items[3], cat_of[items[3]], tie[3]))
    order = rng.permutation(len(rows))
    rows = [rows[i] for i in order]
    return pd.DataFrame(rows, columns=["user_id", "item_id", "cat_id", "time_stamp"])


def stub_flags(name, L, remove_duplicate, user_count_limit):
    return dict(type="golden", pos_embedding="time", experiment_data_type="item_based", causality="unidirection",
                experiment_type="MTAM", version="golden_" + name, user_count_limit=user_count_limit, test_frac=5,
                neg_sample_ratio=20, mask_rate=0.2, gap_num=6, init_train_data=True,
                remove_duplicate=remove_duplicate, length_of_user_history=L)


def main():
    if not os.path.isdir(REFERENCE):
        raise SystemExit("the reference tree is not here: fixtures can only be made in the build container")
    np.set_printoptions(legacy="1.25")
    sys.path.insert(0, REFERENCE)
    cwd = os.getcwd()
    work = tempfile.mkdtemp(prefix="prepare_golden_")
    try:
        os.chdir(work)                                       # the reference writes below ./data/
        os.makedirs("data/log_data")
        os.makedirs("data/training_testing_data")
        from util.model_log import create_log               # noqa: E402  (reference)
        create_log("golden", "MTAM", "fixtures")            # the singleton the Prepare classes fetch their logger from
        from Prepare.prepare_data_base import prepare_data_base  # noqa: E402  (reference)
        for name, (seed, L, dedup, limit, shuffle_seed) in CASES.items():
            origin = make_origin(seed)
            flags = stub_flags(name, L, dedup, limit)
            shutil.rmtree("data/training_testing_data", ignore_errors=True)
            os.makedirs("data/training_testing_data")
            prep = prepare_data_base(types.SimpleNamespace(**flags), origin.copy())
            random.seed(shuffle_seed)
            train, test = prep.get_train_test()
            out = os.path.join(HERE, "prepare", name)
            os.makedirs(out, exist_ok=True)
            origin.to_csv(os.path.join(out, "origin.csv"), index=False)
            flags["shuffle_seed"] = shuffle_seed
            with open(os.path.join(out, "flags.json"), "w") as f:
                json.dump(flags, f, indent=1, sort_keys=True)
            shutil.copy(prep.dataset_class_train, os.path.join(out, "train_data.txt"))
            shutil.copy(prep.dataset_class_test, os.path.join(out, "test_data.txt"))
            with open(os.path.join(out, "parameters.json"), "w") as f:
                json.dump({"item_count": int(prep.item_count), "user_count": int(prep.user_count),
                           "category_count": int(prep.category_count), "gap": [float(g) for g in prep.gap],
                           "item_category": {str(int(k)): int(v) for k, v in sorted(prep.item_category_dic.items())},
                           "n_train": len(train), "n_test": len(test)}, f, indent=1, sort_keys=True)
            print(name, "train", len(train), "test", len(test), "users", prep.user_count, "items", prep.item_count)
    finally:
        os.chdir(cwd)
        shutil.rmtree(work, ignore_errors=True)


if __name__ == "__main__":
    main()
