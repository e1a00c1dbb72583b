"""Batch records -> padded feed arrays (host side, numpy only).

What ``make_feed_dic_new`` does in the reference
(Embedding/Behavior_embedding_time_aware_attention.py:146-192): every list of a
record is right-padded with 0 to ``position_count`` (= length_of_user_history),
integers feed int32 placeholders and times feed float32 placeholders (:26-46).
``normalize_time`` is defined there but never called, so times stay raw hours.
The reference pads one example at a time with six ``np.pad`` calls; here one
pre-zeroed array per field is filled row by row.
"""
import numpy as np

INT_FIELDS = ("user_id", "item_list", "category_list", "position_list",
              "target_item_id", "target_item_category", "seq_length")
FLOAT_FIELDS = ("time_list", "timelast_list", "timenow_list", "target_item_time")
FEED_FIELDS = ("user_id", "item_list", "category_list", "time_list", "timelast_list",
               "timenow_list", "position_list", "target_item_id", "target_item_category",
               "target_item_time", "seq_length")


def pad_batch(batch_data, max_len):
    n = len(batch_data)
    feed = {
        "user_id": np.zeros(n, np.int32),
        "item_list": np.zeros((n, max_len), np.int32),
        "category_list": np.zeros((n, max_len), np.int32),
        "time_list": np.zeros((n, max_len), np.float32),
        "timelast_list": np.zeros((n, max_len), np.float32),
        "timenow_list": np.zeros((n, max_len), np.float32),
        "position_list": np.zeros((n, max_len), np.int32),
        "target_item_id": np.zeros(n, np.int32),
        "target_item_category": np.zeros(n, np.int32),
        "target_item_time": np.zeros(n, np.float32),
        "seq_length": np.zeros(n, np.int32),
    }
    for i, ex in enumerate(batch_data):
        length = int(ex[8])
        if length > max_len:
            # np.pad with a negative width raises in the reference as well
            raise ValueError("record length %d exceeds length_of_user_history %d" % (length, max_len))
        feed["user_id"][i] = ex[0]
        feed["item_list"][i, :len(ex[1])] = ex[1]
        feed["category_list"][i, :len(ex[2])] = ex[2]
        feed["time_list"][i, :len(ex[3])] = ex[3]
        feed["timelast_list"][i, :len(ex[4])] = ex[4]
        feed["timenow_list"][i, :len(ex[5])] = ex[5]
        feed["position_list"][i, :len(ex[6])] = ex[6]
        feed["target_item_id"][i] = ex[7][0]
        feed["target_item_category"][i] = ex[7][1]
        feed["target_item_time"][i] = ex[7][2]
        feed["seq_length"][i] = length
    return feed
