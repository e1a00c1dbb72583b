"""__graft_entry__.smoke(): one small MTAM training step on cuda:0, checked against the oracle."""
import tempfile

import numpy as np
import torch


def run_smoke():
    import oracle.mtam_oracle as O
    from mtamrecommender_amd import _lib
    from tests.test_model_gpu import build, rel
    _lib.load()                                          # fails loudly if the HIP library is missing
    B, L, NB, H = 16, 12, 1, 1
    with tempfile.TemporaryDirectory() as tmp:
        model, FLAGS, records = build(tmp, B, L, NB, H, items=120, cats=9, users=30)
        arrays = {k: v.copy() for k, v in model.get_variables().items()}
        feed = model.embedding.make_feed_dic_new(records)
        out, grads, slot_sq = O.loss_and_grads("MTAM", arrays, feed, H, NB, FLAGS.regulation_rate, torch.float64)
        loss, _ = model.train(model.sess, records, 1e-3)
        ref = float(out["loss"].detach())
        assert abs(loss - ref) / abs(ref) < 2e-5, (loss, ref)
        got = model.path.grads_tf()
        worst = max(rel(got[k], g) for k, g in grads.items() if g is not None)
        assert worst < 5e-4, worst
        hr = model.metrics_topK(model.sess, records, 0, 20)
        assert len(hr) == 10 and all(np.isfinite(hr))
    print("smoke ok: loss %.6f (oracle %.6f), worst gradient rel err %.2e" % (loss, ref, worst))
