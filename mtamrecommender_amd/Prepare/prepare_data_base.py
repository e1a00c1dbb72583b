"""Origin interactions -> training / testing records and their on-disk form.

Mirror of the reference's Prepare/prepare_data_base.py:13-341 for the time-aware path
(SURVEY.md section 8(f) rank 3, App C).  Input: a DataFrame with columns
``user_id, item_id, cat_id, time_stamp`` (seconds), as the ``DataHandle`` adapters produce
(DataHandle/get_origin_data_ml.py:33-46).  Steps kept from the reference:

* ``map_process`` (:115-154): ids label-encoded in sorted order (sklearn's LabelEncoder), the
  item -> category dictionary, rows sorted by (user, time);
* ``get_train_test`` / ``data_handle_process`` (:161-312): users in id order up to
  ``user_count_limit`` (+1: the reference's check is ``>``), per user optionally
  ``drop_duplicates(keep='last')``, events sorted by time; every event but the first is a target;
  history = the at most L-1 events before it; mask tokens ``item_count + 1`` / ``category_count + 1``
  in the last slot; times in whole hours; ``timelast`` / ``timenow`` with 0 at the mask slot;
  positions 0..n-1 then ``min(index, 49)``; the record whose target is the user's LAST event goes to
  the test set, all others to the training set; both shuffled, the test set capped at 20,000;
* files (:55-62,204-216,334-339): ``data/training_testing_data/<type>_<pos_embedding>_
  <experiment_data_type>_<causality>/{train_data.txt,test_data.txt,parameters.pkl}`` -- one
  ``str(tuple)`` per line and a pickled dict of counts, gaps and the item -> category map.
  ``parameters.json`` is written next to the pickle and preferred on load (nothing is executed
  from it); the text files are read by libmtam_host.so instead of ``eval(line)``.
"""
import json
import os
import pickle
import random

import numpy as np

from ..util.model_log import create_log
from .mask_data_process import mask_data_process

np.random.seed(1234)


def _label_encode(values):
    """sklearn.preprocessing.LabelEncoder().fit_transform: index into the sorted unique values."""
    classes, codes = np.unique(np.asarray(values), return_inverse=True)
    return codes.astype(np.int64), classes


class prepare_data_base(object):

    def __init__(self, FLAGS, origin_data=None, root="data/training_testing_data"):
        self.FLAGS = FLAGS
        self.type = FLAGS.type
        self.user_count_limit = FLAGS.user_count_limit
        self.test_frac = FLAGS.test_frac
        self.experiment_type = FLAGS.experiment_type
        self.origin_data = origin_data
        self.use_action = False
        self.data_type_error = 0
        self.data_too_short = 0
        self.dataset_path = os.path.join(root, "%s_%s_%s_%s" % (self.type, FLAGS.pos_embedding,
                                                                FLAGS.experiment_data_type, FLAGS.causality))
        os.makedirs(self.dataset_path, exist_ok=True)
        self.dataset_class_pkl = os.path.join(self.dataset_path, "parameters.pkl")
        self.dataset_class_json = os.path.join(self.dataset_path, "parameters.json")
        self.dataset_class_train = os.path.join(self.dataset_path, "train_data.txt")
        self.dataset_class_test = os.path.join(self.dataset_path, "test_data.txt")
        self.mask_rate = FLAGS.mask_rate
        self.logger = create_log().logger
        self.init_train_data = bool(FLAGS.init_train_data)
        if self.init_train_data:
            if origin_data is None:
                raise ValueError("init_train_data needs the origin interactions")
            self.get_gap_list(FLAGS.gap_num)
            self.map_process()
            self.filter_repetition()
        else:
            self.load()

    # ------------------------------------------------------------------ load
    def load(self):
        from ..DataHandle.native_input import RecordSet
        self.train_records = RecordSet.from_file(self.dataset_class_train)
        self.test_records = RecordSet.from_file(self.dataset_class_test)
        self.train_set = [self.train_records.record(i) for i in range(len(self.train_records))]
        self.test_set = [self.test_records.record(i) for i in range(len(self.test_records))]
        if os.path.exists(self.dataset_class_json):
            data_dic = json.load(open(self.dataset_class_json))
            data_dic["item_category"] = {int(k): v for k, v in data_dic["item_category"].items()}
        elif getattr(self.FLAGS, "allow_pickle_parameters", False):
            # a parameters.pkl written by the reference's own prepare step, loaded the way the reference
            # loads it (:99-101).  Explicit opt-in: unpickling executes what the file says
            with open(self.dataset_class_pkl, "rb") as f:
                data_dic = pickle.load(f)
        else:
            raise FileNotFoundError(
                "%s not found.  A directory prepared by the reference holds only parameters.pkl; set "
                "FLAGS.allow_pickle_parameters = True to unpickle it (only for files you produced yourself), or "
                "re-run the prepare step of this build, which writes parameters.json" % self.dataset_class_json)
        self.item_count = data_dic["item_count"]
        self.user_count = data_dic["user_count"]
        self.category_count = data_dic["category_count"]
        self.gap = np.asarray(data_dic["gap"])
        self.item_category_dic = data_dic["item_category"]
        self.logger.info("load data finish")
        self.logger.info("Size of training set is " + str(len(self.train_set)))
        self.logger.info("Size of testing set is " + str(len(self.test_set)))

    # ------------------------------------------------------------------ build
    def map_process(self):
        d = self.origin_data.copy()
        item_id, item_classes = _label_encode(d["item_id"].tolist())
        user_id, user_classes = _label_encode(d["user_id"].tolist())
        cat_id, cat_classes = _label_encode(d["cat_id"].tolist())
        self.item_count, self.user_count, self.category_count = len(item_classes), len(user_classes), len(cat_classes)
        self.item_category_dic = {}
        for i, c in zip(item_id.tolist(), cat_id.tolist()):
            self.item_category_dic[i] = c                    # last occurrence wins, as in the reference loop
        self.logger.warning("item Count :" + str(self.item_count))
        self.logger.info("user count is " + str(self.user_count))
        self.logger.info("category count is " + str(self.category_count))
        d["item_id"], d["user_id"], d["cat_id"] = item_id, user_id, cat_id
        self.origin_data = d.sort_values(["user_id", "time_stamp"]).reset_index(drop=True)
        return self.user_count, self.item_count

    def filter_repetition(self):
        pass

    def get_train_test(self):
        if not self.init_train_data:
            return self.train_set, self.test_set
        self.train_set, self.test_set = [], []
        self.now_count = 0
        for _, group in self.origin_data.groupby("user_id", sort=True):
            self.data_handle_process(group)
        random.shuffle(self.train_set)
        random.shuffle(self.test_set)
        if len(self.test_set) > 20000:
            self.test_set = random.sample(self.test_set, 20000)
        self.logger.info("Size of training set is " + str(len(self.train_set)))
        self.logger.info("Size of testing set is " + str(len(self.test_set)))
        data_dic = {"item_count": self.item_count, "user_count": self.user_count,
                    "category_count": self.category_count, "gap": self.gap,
                    "item_category": self.item_category_dic}
        with open(self.dataset_class_pkl, "wb") as f:
            pickle.dump(data_dic, f, pickle.HIGHEST_PROTOCOL)
        with open(self.dataset_class_json, "w") as f:
            json.dump({"item_count": int(self.item_count), "user_count": int(self.user_count),
                       "category_count": int(self.category_count), "gap": [float(g) for g in self.gap],
                       "item_category": {str(k): int(v) for k, v in self.item_category_dic.items()}}, f)
        self.save(self.train_set, self.dataset_class_train)
        self.save(self.test_set, self.dataset_class_test)
        return self.train_set, self.test_set

    def data_handle_process_base(self, x):
        behavior_seq = x.copy()
        if self.FLAGS.remove_duplicate:
            behavior_seq = behavior_seq.drop_duplicates(keep="last")
        behavior_seq = behavior_seq.sort_values(by=["time_stamp"], na_position="first").reset_index(drop=True)
        if "user_id" not in behavior_seq.columns:
            self.data_type_error += 1
            return None
        if self.now_count > self.user_count_limit:
            return None
        self.now_count += 1
        return behavior_seq

    def data_handle_process(self, x):
        behavior_seq = self.data_handle_process_base(x)
        if behavior_seq is None:
            return
        L = self.FLAGS.length_of_user_history
        m = mask_data_process(behavior_seq=behavior_seq, use_action=self.use_action, mask_rate=self.mask_rate)
        m.get_mask_index_list_behaivor()
        for index in m.mask_index_list:
            user_id, item_seq_temp, factor_list = m.mask_process_unidirectional(
                self.FLAGS.causality, index=index, time_window=24 * 3600 * 35, lengeth_limit=L)
            cat_list = factor_list[0]
            time_list = [int(t / 3600) for t in factor_list[1]]                 # whole hours
            target_time = int(m.time_stamp_seq[index] / 3600)
            item_seq_temp.append(self.item_count + 1)                           # the mask tokens
            cat_list.append(self.category_count + 1)
            timelast_list, timenow_list = m.pro_time_method(time_list, target_time)
            position_list = m.proc_pos_emb(time_list)
            time_list.append(target_time)
            timelast_list.append(0)
            timenow_list.append(0)
            position_list.append(49 if index > 49 else index)
            target_id = m.item_seq[index]
            target_category = self.item_category_dic[target_id]
            record = (int(user_id), [int(v) for v in item_seq_temp], [int(v) for v in cat_list], time_list,
                      timelast_list, timenow_list, position_list,
                      [int(target_id), int(target_category), target_time], len(item_seq_temp))
            if index == len(m.mask_index_list):          # the user's last event
                self.test_set.append(record)
            else:
                self.train_set.append(record)

    def get_gap_list(self, gapnum):
        gap = []
        for i in range(1, gapnum):
            if i == 1:
                gap.append(60)
            elif i == 2:
                gap.append(60 * 60)
            else:
                gap.append(3600 * 24 * np.power(2, i - 3))
        self.gap = np.array(gap)

    def save(self, data_list, file_path):
        with open(file_path, "w+") as fp:
            for i in data_list:
                fp.write(str(i) + "\n")
