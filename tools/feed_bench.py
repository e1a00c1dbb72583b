"""Host-inclusive training rate: records -> feed -> H2D -> step -> loss read-back, per step.
Python route (make_feed_dic_new per step, the reference's way) vs the native packer on a worker thread.
This is the PCIe-inclusive figure DESIGN.md quotes next to bench.py's HBM-resident `value`."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

entry.build()
from mtamrecommender_amd.config.model_parameter import model_parameter  # noqa: E402
from mtamrecommender_amd.data.synthetic import ML1M, SyntheticCatalog, make_records  # noqa: E402
from mtamrecommender_amd.DataHandle.get_input_data import DataInput  # noqa: E402
from mtamrecommender_amd.DataHandle.native_input import BatchPacker, NativeDataInput, RecordSet  # noqa: E402
from mtamrecommender_amd.Embedding.Behavior_embedding_time_aware_attention import \
    Behavior_embedding_time_aware_attention  # noqa: E402
from mtamrecommender_amd.Model.base_model import Session  # noqa: E402
from mtamrecommender_amd.Model.MTAMRec_model import MTAM  # noqa: E402

B, L, STEPS = 128, 50, 300
FLAGS = model_parameter().get_parameter("MTAMb1_movielen").FLAGS
FLAGS.num_blocks, FLAGS.num_heads, FLAGS.length_of_user_history = 1, 1, L
FLAGS.checkpoint_path_dir = "/tmp/mtam_feed_bench"
cat = SyntheticCatalog(seed=1234, **ML1M)
emb = Behavior_embedding_time_aware_attention(True, cat.user_count, cat.item_count, cat.category_count, L, seed=1234)
model = MTAM(FLAGS, emb, Session("cuda:0"))
records = make_records(cat, B * STEPS, L, seed=7)
for _, batch in DataInput(records[:B * 5], B):          # warm up, capture the graph
    model.train(model.sess, batch, 1e-3)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _, batch in DataInput(records, B):
    model.train(model.sess, batch, 1e-3)
torch.cuda.synchronize()
t_py = time.perf_counter() - t0
t0 = time.perf_counter()
rs = RecordSet.from_records(records)
t_build = time.perf_counter() - t0
packer = BatchPacker(model.path, emb)
t0 = time.perf_counter()
for _, packed in NativeDataInput(rs, B, packer):
    model.train(model.sess, packed, 1e-3)
torch.cuda.synchronize()
t_nat = time.perf_counter() - t0
print(json.dumps({"steps": STEPS, "batch": B,
                  "python_feed_seq_per_s": B * STEPS / t_py, "python_feed_ms_per_step": t_py / STEPS * 1e3,
                  "native_feed_seq_per_s": B * STEPS / t_nat, "native_feed_ms_per_step": t_nat / STEPS * 1e3,
                  "recordset_build_s": t_build}))
