"""Synthetic interaction records of the reference's on-disk shape.

The reference's data files are stripped, so every measurement
and every parity case runs on synthetic records.  One record is the 9-tuple
that Prepare/prepare_data_base.py:252-314 writes and
Embedding/Behavior_embedding_time_aware_attention.py:166-190 reads:

    (user_id, item_seq, cat_list, time_list, timelast_list, timenow_list,
     position_list, [target_id, target_category, target_hours], length)

with ``length = n + 1`` (n history events + the mask-token slot), the mask
token ``item_count + 1`` / ``category_count + 1`` in the last slot, times in
integer hours, ``timelast[i] = t[i] - t[i-1]`` (first 0), ``timenow[i] =
t_target - t[i]`` and both 0 at the mask slot (mask_data_process.py:250-255,
prepare_data_base.py:283-298).  Generator spec: SURVEY.md section 8(d); one
deviation: Zipf draws above ``item_count`` are folded back by modulo instead
of clipped (a clip would put ~40 % of all events on one id).
"""
import numpy as np


class SyntheticCatalog(object):
    """Fixed item->category map plus the table sizes of one synthetic dataset."""

    def __init__(self, item_count, category_count, user_count, seed=1234):
        self.item_count = int(item_count)
        self.category_count = int(category_count)
        self.user_count = int(user_count)
        rng = np.random.Generator(np.random.PCG64(seed))
        # mirrors item_category_dic (prepare_data_base.py:136-138)
        self.item_category = rng.integers(0, self.category_count, size=self.item_count,
                                          dtype=np.int64)


ML1M = dict(item_count=3706, category_count=301, user_count=4832)


def _draw_items(rng, n, item_count, id_dist):
    if id_dist == "uniform":
        return rng.integers(0, item_count, size=n, dtype=np.int64)
    z = rng.zipf(1.1, size=n).astype(np.int64) - 1
    return np.mod(z, item_count)


def make_records(catalog, num, max_len, seed=1234, id_dist="zipf"):
    """``num`` records with ``length ~ U{2..max_len}``."""
    rng = np.random.Generator(np.random.PCG64(seed))
    records = []
    for _ in range(num):
        length = int(rng.integers(2, max_len + 1))
        n = length - 1
        items = _draw_items(rng, n, catalog.item_count, id_dist)
        cats = catalog.item_category[items]
        t0 = int(rng.integers(240000, 270000))
        gaps = np.floor(rng.exponential(24.0, size=n)).astype(np.int64)
        gaps[0] = 0
        times = t0 + np.cumsum(gaps)
        target_time = int(times[-1] + int(np.floor(rng.exponential(24.0))))
        target_id = int(_draw_items(rng, 1, catalog.item_count, id_dist)[0])
        target_cat = int(catalog.item_category[target_id])
        timelast = np.concatenate([[0], np.diff(times)])
        timenow = target_time - times
        index = n + int(rng.integers(0, 20))          # position of the target in the full history
        position = list(range(n)) + [min(index, 49, max_len - 1)]   # 49: prepare_data_base.py:295-298
        records.append((
            int(rng.integers(0, catalog.user_count)),
            [int(x) for x in items] + [catalog.item_count + 1],
            [int(x) for x in cats] + [catalog.category_count + 1],
            [int(x) for x in times] + [target_time],
            [int(x) for x in timelast] + [0],
            [int(x) for x in timenow] + [0],
            position,
            [target_id, target_cat, target_time],
            length))
    return records


def make_id_batch(B, L, item_rows, category_rows, user_rows, id_dist="zipf", seed=1234):
    """Padded id arrays of one batch, drawn directly (no record tuples): what the embedding kernels' roofline
    legs at 512 / 2,048 / 8,192 sequences per launch are fed (bench.py ``roofline_at_scale``,
    tools/emb_roofline.py).  Same shape rules as ``make_records``: length ~ U{2..L}, Zipf(1.1) items folded by
    modulo (or uniform), a fixed item -> category map, positions 0..len-1, zeros past the length."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sl = rng.integers(2, L + 1, size=B)
    live = np.arange(L)[None, :] < sl[:, None]
    if id_dist == "zipf":
        items = np.mod(rng.zipf(1.1, size=(B, L)).astype(np.int64) - 1, item_rows - 3)
    else:
        items = rng.integers(0, item_rows - 3, size=(B, L))
    cmap = np.random.Generator(np.random.PCG64(4321)).integers(0, category_rows - 3, size=item_rows)
    cats = cmap[items]
    pos = np.tile(np.arange(L), (B, 1))
    i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
    return dict(item_list=i32(items * live), category_list=i32(cats * live), position_list=i32(pos * live),
                user_id=i32(rng.integers(0, user_rows, size=B)), seq_length=i32(sl),
                live_rows=int(live.sum()) * 3 + B)
