"""Oracle restatement of the reference's batch feed.  TEST INFRASTRUCTURE ONLY (see mtam_oracle.py header).

``make_feed_dic_new`` (Embedding/Behavior_embedding_time_aware_attention.py:146-192) pads every list of every
example with ``np.pad(list, [0, position_count - length], 'constant')`` and hands python lists to placeholders
declared int32 (ids, lengths) and float32 (times) at :26-46.  Written with the same six np.pad calls per example;
the product's feed (one pre-zeroed array per field, and the native packer) is checked against this.
"""
import numpy as np

INT32 = ("user_id", "item_list", "category_list", "position_list", "target_item_id", "target_item_category",
         "seq_length")
FLOAT32 = ("time_list", "timelast_list", "timenow_list", "target_item_time")


def make_feed_dic_new(batch_data, position_count):
    cols = {k: [] for k in INT32 + FLOAT32}
    for example in batch_data:
        padding_size = [0, int(position_count - example[8])]                       # :167
        cols["user_id"].append(example[0])
        cols["item_list"].append(np.pad(example[1], padding_size, "constant"))      # :169
        cols["category_list"].append(np.pad(example[2], padding_size, "constant"))
        cols["time_list"].append(np.pad(example[3], padding_size, "constant"))
        cols["timelast_list"].append(np.pad(example[4], padding_size, "constant"))
        cols["timenow_list"].append(np.pad(example[5], padding_size, "constant"))
        cols["position_list"].append(np.pad(example[6], padding_size, "constant"))
        cols["target_item_id"].append(example[7][0])
        cols["target_item_category"].append(example[7][1])
        cols["target_item_time"].append(example[7][2])
        cols["seq_length"].append(example[8])
    # what sess.run's feed conversion does with the placeholder dtypes (:26-46)
    out = {k: np.asarray(cols[k], dtype=np.int32) for k in INT32}
    out.update({k: np.asarray(cols[k], dtype=np.float32) for k in FLOAT32})
    return out
