"""Generates tests/golden/mtam_*.npz from the CPU oracle.

PARITY UNPINNED: the reference has no golden vectors and TensorFlow 1.14 cannot
run here, so these fixtures come from this repo's own oracle
(oracle/mtam_oracle.py, float64), cross-checked against oracle/numpy_ref.py when
they are made.  They pin the oracle against regressions and travel to the GPU
box, where the HIP path is compared with them.  Records, the padded feed, the
variable list and the initial values are the oracle's own (oracle/records.py,
feed_ref.py, specs.py); the GPU test rebuilds the records from the committed feed
arrays and the weights from oracle/specs.py.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import feed_ref, mtam_oracle as O, numpy_ref as N, records as R, specs as S  # noqa: E402

CASES = {
    # name: (model, B, L, D, NB, H, items, cats, users, seed)
    "mtam_b6_l8_nb2_h2": ("MTAM", 6, 8, 128, 2, 2, 90, 7, 25, 11),
    "mtam_b16_l50_nb1_h1": ("MTAM", 16, 50, 128, 1, 1, 300, 17, 40, 12),
    "pistrec_b5_l10_nb2_h1": ("PISTRec", 5, 10, 128, 2, 1, 80, 6, 20, 13),
    # family members with kernels of their own (the T-SeqRec cell, the output_concat head)
    "mtam_with_t_seqrec_b6_l8_nb1_h2": ("MTAM_with_T_SeqRec", 6, 8, 128, 1, 2, 90, 7, 25, 14),
    "mtam_hybird_b6_l8_nb1_h1": ("MTAM_hybird", 6, 8, 128, 1, 1, 90, 7, 25, 15),
    # experiment_type 'T_GRU': the T-SeqRec cell with no decoder (Model/MTAMRec_model.py:40-59; round 1 wired
    # the decay_new cell here)
    "mtam_only_time_aware_rnn_b6_l8": ("MTAM_only_time_aware_RNN", 6, 8, 128, 1, 1, 90, 7, 25, 16),
}
REG = 5e-5


def make_case(model, B, L, D, NB, H, items, cats, users, seed):
    """Inputs, weights and the variable list all come from oracle/ (records.py, feed_ref.py, specs.py): nothing of
    the product takes part in making a fixture.  jitter: non-trivial biases / LN scales."""
    records = R.make_records(items, cats, users, B, L, seed=seed + 1)
    feed = feed_ref.make_feed_dic_new(records, L)
    arrays = S.init_arrays(S.model_vars(model, users, items, cats, L, D, NB), seed=seed + 2, jitter=0.05)
    return records, feed, arrays


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    only = sys.argv[1:]
    for name, (model, B, L, D, NB, H, items, cats, users, seed) in CASES.items():
        if only and name not in only:
            continue
        records, feed, arrays = make_case(model, B, L, D, NB, H, items, cats, users, seed)
        out, grads, slot_sq = O.loss_and_grads(model, arrays, feed, H, NB, REG, torch.float64)
        if model in ("MTAM", "PISTRec"):             # the second restatement covers the two main models
            ref = N.forward(model, arrays, feed, H, NB, REG)
            assert np.abs(out["logits"].detach().numpy() - ref["logits"]).max() < 1e-10
        logits = out["logits"].detach().numpy()
        payload = {"feed_" + k: v for k, v in feed.items()}
        payload.update({
            "weights_checksum": np.array([float(sum(np.abs(v.astype(np.float64)).sum() for v in arrays.values()))]),
            "logits": logits.astype(np.float64),
            "pred": out["pred"].detach().numpy(),
            "loss": np.array([float(out["loss"].detach())]),
            "l2": np.array([float(out["l2"].detach())]),
            "ce": out["ce"].detach().numpy(),
            "top50": O.top_k(logits, min(50, logits.shape[1])),
            "global_norm_tf": np.array([O.global_norm(grads, slot_sq, model, True)]),
            "global_norm_true": np.array([O.global_norm(grads, slot_sq, model, False)]),
        })
        for k, g in grads.items():
            if g is not None and g.size <= 4096:                     # small tensors whole, large ones by norm
                payload["grad/" + k] = g
            elif g is not None:
                payload["gradnorm/" + k] = np.array([np.sqrt((g.astype(np.float64) ** 2).sum())])
        np.savez_compressed(os.path.join(here, name + ".npz"), **payload)
        print(name, "loss", payload["loss"][0], "bytes", os.path.getsize(os.path.join(here, name + ".npz")))


if __name__ == "__main__":
    main()
