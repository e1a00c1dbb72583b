"""One user's behaviour sequence -> leave-one-out histories (the part of the reference's
Prepare/mask_data_process.py:28-71,158-202,244-262 that the time-aware path's data prep uses).

``mask_process_unidirectional`` returns, for a target at ``index``, the at most
``lengeth_limit - 1`` events strictly before the cut point; the caller appends the mask token so
that the record has at most ``lengeth_limit`` slots.  Time features: ``pro_time_method`` gives
``timelast[i] = t[i] - t[i-1]`` (first 0) and ``timenow[i] = t_target - t[i]``;
``proc_pos_emb`` gives positions 0..n-1.
"""
import random


class mask_data_process(object):

    def __init__(self, behavior_seq, use_action=False, mask_rate=0.2):
        self.user_seq = behavior_seq["user_id"].tolist()
        self.item_seq = behavior_seq["item_id"].tolist()
        self.category_seq = behavior_seq["cat_id"].tolist()
        self.time_stamp_seq = behavior_seq["time_stamp"].tolist()
        if use_action:
            raise NotImplementedError("action-typed data sets (use_action) are not part of the time-aware path")
        self.use_action = False
        self.length = behavior_seq.shape[0]
        self.mask_rate = mask_rate

    def get_mask_index_list_behaivor(self, only_last=False):
        """Every event but the first is a prediction target (reference :58-71)."""
        self.mask_index_list = [self.length - 1] if only_last else list(range(1, self.length))

    def mask_process_unidirectional(self, type, index, time_window=24 * 3600 * 35, lengeth_limit=50):
        if type == "unidirection":
            temp_index = index
        elif type == "random":
            where = self.mask_index_list.index(index)
            start = self.mask_index_list[where - 1] if where - 1 >= 0 else 0
            temp_index = random.randint(start + 1, index)
        elif type == "time_window":
            target_time = self.time_stamp_seq[index]
            temp_index = index
            for i in range(0, index + 1):
                if target_time - self.time_stamp_seq[i] <= time_window:
                    temp_index = i
                    break
        else:
            raise ValueError("unknown causality %r" % (type,))
        start = max(0, temp_index - lengeth_limit + 1)
        keep = range(start, min(temp_index, self.length))
        user_seq_temp = [self.user_seq[i] for i in keep]
        item_seq_temp = [self.item_seq[i] for i in keep]
        category_seq_temp = [self.category_seq[i] for i in keep]
        time_stamp_seq_temp = [self.time_stamp_seq[i] for i in keep]
        user = user_seq_temp[0] if user_seq_temp else self.user_seq[0]
        return user, item_seq_temp, [category_seq_temp, time_stamp_seq_temp]

    def proc_pos_emb(self, time_stamp_seq):
        return list(range(len(time_stamp_seq)))

    def pro_time_method(self, time_stamp_seq, mask_time):
        timelast_list = [time_stamp_seq[i + 1] - time_stamp_seq[i] for i in range(len(time_stamp_seq) - 1)]
        timelast_list.insert(0, 0)
        timenow_list = [mask_time - t for t in time_stamp_seq]
        return timelast_list, timenow_list
