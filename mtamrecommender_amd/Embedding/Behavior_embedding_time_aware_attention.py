"""Embedding layer of the time-aware models.

Mirror of Embedding/Behavior_embedding_time_aware_attention.py:10-192: same
constructor, same attribute names, ``get_embedding(num_units)`` creates the four
lookup tables (``count + 3`` rows each) and returns the 10-tuple the models
unpack, ``make_feed_dic_new(batch)`` pads a list of records into the 11 feeds.
The reference's placeholders are graph tensors used as feed-dict keys; here
they are plain strings with the same attribute names, so
``feed[emb.target_item_id]`` keeps working.  The lookups themselves run inside
the model's step (mtam_emb_gather_fwd); the tuple holds named slots that the
model resolves to device buffers.
"""
import numpy as np

from .base_embedding import Base_embedding
from .feed import pad_batch


class Slot(object):
    """Name of a tensor produced inside the device step."""

    def __init__(self, name):
        self.name = name

    def __repr__(self):
        return "Slot(%s)" % self.name


class Behavior_embedding_time_aware_attention(Base_embedding):

    def __init__(self, is_training=True, user_count=0, item_count=0, category_count=0, max_length_seq=0,
                 seed=1234):
        super(Behavior_embedding_time_aware_attention, self).__init__(is_training)
        self.user_count = user_count
        self.item_count = item_count
        self.category_count = category_count
        self.position_count = max_length_seq
        self.seed = seed

    def init_placeholders(self):
        for name in ("user_id", "item_list", "category_list", "time_list", "timelast_list", "timenow_list",
                     "position_list", "target_item_id", "target_item_category", "target_item_time",
                     "seq_length"):
            setattr(self, name, name)

    def get_embedding(self, num_units):
        rng = np.random.Generator(np.random.PCG64(self.seed))
        self.user_emb_lookup_table = self.init_embedding_lookup_table(
            "user", self.user_count + 3, num_units, self.is_training, rng)
        self.item_emb_lookup_table = self.init_embedding_lookup_table(
            "item", self.item_count + 3, num_units, self.is_training, rng)
        self.category_emb_lookup_table = self.init_embedding_lookup_table(
            "category", self.category_count + 3, num_units, self.is_training, rng)
        self.position_emb_lookup_table = self.init_embedding_lookup_table(
            "position", self.position_count + 3, num_units, self.is_training, rng)
        return (Slot("user_embedding"), Slot("behavior_list_embedding_dense"), Slot("item_list_embedding"),
                Slot("category_list_embedding"), Slot("position_list_embedding"), self.time_list,
                self.timelast_list, self.timenow_list,
                [self.target_item_id, self.target_item_category, self.target_item_time], self.seq_length)

    def tables(self):
        return {"user": self.user_emb_lookup_table, "item": self.item_emb_lookup_table,
                "category": self.category_emb_lookup_table, "position": self.position_emb_lookup_table}

    def validate_ids(self, feed):
        """TF's CPU gather raises on an out-of-range id; the HIP gather clamps, so check here."""
        checks = (("item_list", self.item_count + 3), ("category_list", self.category_count + 3),
                  ("position_list", self.position_count + 3), ("user_id", self.user_count + 3),
                  ("target_item_id", self.item_count + 3))
        for key, rows in checks:
            a = feed[key]
            if a.size and (a.min() < 0 or a.max() >= rows):
                raise IndexError("%s: id out of range [0, %d)" % (key, rows))
        sl = feed["seq_length"]
        if sl.size and (sl.min() < 2 or sl.max() > self.position_count):
            raise ValueError("seq_length must be in [2, %d]" % self.position_count)

    def make_feed_dic_new(self, batch_data):
        feed = pad_batch(batch_data, self.position_count)
        return {getattr(self, k): v for k, v in feed.items()}
