"""HIP path against the committed golden vectors (tests/golden/*.npz, made by the oracle)."""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize("path", sorted(glob.glob(os.path.join(HERE, "golden", "*.npz"))))
def test_hip_matches_golden(hip_lib, tmp_path, path):
    from mtamrecommender_amd.config.model_parameter import model_parameter
    from mtamrecommender_amd.Embedding.Behavior_embedding_time_aware_attention import \
        Behavior_embedding_time_aware_attention
    from mtamrecommender_amd.Model.base_model import Session
    from mtamrecommender_amd.Model import MTAMRec_model as family
    from mtamrecommender_amd.Model.PISTRec_model import Time_Aware_self_Attention_model
    from tests.golden.make_golden import CASES, make_case
    name = os.path.splitext(os.path.basename(path))[0]
    model_name, B, L, D, NB, H, items, cats, users, seed = CASES[name]
    from oracle import records as R
    gold = np.load(path)
    # inputs are the COMMITTED arrays (the records rebuilt from them); weights are re-drawn from the oracle's
    # variable list and must reproduce the committed checksum
    feed = {k[5:]: gold[k] for k in gold.files if k.startswith("feed_")}
    records = R.records_from_feed(feed)
    _, feed_again, arrays = make_case(*CASES[name])
    assert all(np.array_equal(feed[k], feed_again[k]) for k in feed)
    checksum = float(sum(np.abs(v.astype(np.float64)).sum() for v in arrays.values()))
    assert abs(checksum - gold["weights_checksum"][0]) <= 1e-9 * checksum
    FLAGS = model_parameter().get_parameter("MTAMb1_movielen").FLAGS
    FLAGS.num_blocks, FLAGS.num_heads, FLAGS.length_of_user_history = NB, H, L
    FLAGS.checkpoint_path_dir = str(tmp_path)
    emb = Behavior_embedding_time_aware_attention(True, users, items, cats, L)
    cls = Time_Aware_self_Attention_model if model_name == "PISTRec" else getattr(family, model_name)
    model = cls(FLAGS, emb, Session("cuda:0"))
    model.use_graph = False
    model.set_variables(arrays)
    p = model.path
    bt = p.load_feed(feed)
    p.eval_kernels(bt, 50)
    logits = bt.logits.cpu().numpy()
    scale = np.abs(gold["logits"]).max()
    assert np.abs(logits - gold["logits"]).max() / scale < 5e-5          # stated fp32 tolerance
    k = gold["top50"].shape[1]
    top = bt.topk_idx.cpu().numpy()[:, :k]
    srt = -np.sort(-gold["logits"], axis=1)
    for b in range(B):
        gaps = srt[b, :k - 1] - srt[b, 1:k]
        tight = np.nonzero(gaps < 1e-4 * scale)[0]
        safe = int(tight[0]) if tight.size else k - 1
        assert np.array_equal(top[b, :safe], gold["top50"][b, :safe])
    loss, summary = model.train(model.sess, records, 1e-3)
    assert abs(loss - gold["loss"][0]) / gold["loss"][0] < 2e-5
    assert abs(float(p.scale[1]) - gold["global_norm_tf"][0]) / gold["global_norm_tf"][0] < 1e-4
    got = p.grads_tf()
    for key in gold.files:
        if key.startswith("grad/"):
            ref = gold[key]
            assert np.abs(got[key[5:]] - ref).max() <= 5e-4 * np.abs(ref).max() + 1e-12, key
        elif key.startswith("gradnorm/"):
            n = np.sqrt((got[key[9:]].astype(np.float64) ** 2).sum())
            assert abs(n - gold[key][0]) / gold[key][0] < 1e-4, key
