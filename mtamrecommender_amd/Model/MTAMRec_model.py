"""MTAM: time-aware GRU -> time-aware attention decoder -> full-catalog softmax.
Mirror of Model/MTAMRec_model.py:12-38 (MTAMRec_model), :61-92 (MTAM) and the ablation members that
run on the same kernels (:40-59, :93-127, :167-238)."""
import numpy as np

from .base_model import base_model
from .time_aware_path import TimeAwarePath
from .variables import init_variables, mtam_dense_specs


class MTAMRec_model(base_model):

    def __init__(self, FLAGS, Embeding, sess):
        super(MTAMRec_model, self).__init__(FLAGS, Embeding)
        self.sess = sess
        self.now_bacth_data_size = "batch_size"          # (sic) placeholder name, reference :17
        self.num_units = self.FLAGS.num_units
        if int(self.num_units) != 128:       # MTAM_D of include/mtam_hip.h; also checked by model_parameter.validate()
            raise ValueError("num_units = %s: this build's kernels are compiled for num_units = 128 only"
                             % (self.num_units,))
        self.num_heads = self.FLAGS.num_heads
        self.num_blocks = self.FLAGS.num_blocks
        self.dropout_rate = self.FLAGS.dropout           # unused on this path
        self.regulation_rate = self.FLAGS.regulation_rate
        self.user_embedding, self.behavior_list_embedding_dense, self.item_list_emb, \
            self.category_list_emb, self.position_list_emb, self.time_list, self.timelast_list, \
            self.timenow_list, self.target, self.seq_length = self.embedding.get_embedding(self.num_units)
        self.max_len = self.FLAGS.length_of_user_history
        self.build_model()
        self.init_variables(sess, self.checkpoint_path_dir)


class MTAM(MTAMRec_model):
    """Multi-hop Time-aware Attentive Memory network."""
    VARIANT = "MTAM"

    def build_model(self, seed=1234):
        D, L, NB = self.num_units, self.max_len, self.num_blocks
        if self.embedding.position_count != L:
            raise ValueError("embedding max_length_seq %d != length_of_user_history %d"
                             % (self.embedding.position_count, L))
        specs = mtam_dense_specs(D, L, NB, self.VARIANT)
        values = init_variables(specs, seed=seed + 1)
        live = {s.name: values[s.name] for s in specs if s.trainable_grad}
        self.dead_variables = {s.name: values[s.name] for s in specs if not s.trainable_grad}
        device = getattr(self.sess, "device", "cuda:0")
        self.path = TimeAwarePath(self.embedding.tables(), live, L, self.num_heads, NB,
                                  self.regulation_rate, self.FLAGS.max_gradient_norm,
                                  tf_compat_global_norm=self.FLAGS.tf_compat_global_norm, device=device,
                                  optimizer=self.opt, variant=self.VARIANT,
                                  score_dtype=getattr(self.FLAGS, "score_dtype", "f32"))
        self.summery()

    # weight injection for parity tests / checkpoint interchange (TF names)
    def set_variables(self, arrays):
        import torch
        p = self.path
        dense = p.dense_tf()
        for k, v in arrays.items():
            if k.startswith("embedding_layer/"):
                p.tables[k.split("/")[1]].copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)))
            elif k in dense:
                dense[k] = np.asarray(v, np.float32)
            elif k in self.dead_variables:
                self.dead_variables[k] = np.asarray(v, np.float32)
            else:
                raise KeyError(k)
        p.params.copy_(torch.from_numpy(p.layout.pack(dense)))
        p.refresh_derived()

    def get_variables(self):
        out = dict(self.path.dense_tf())
        for k, v in self.path.tables_numpy().items():
            out["embedding_layer/" + k] = v
        out.update(self.dead_variables)
        return out


class MTAM_only_time_aware_RNN(MTAM):
    """Model/MTAMRec_model.py:40-59: time-aware GRU -> layer_norm -> scoring (no decoder)."""
    VARIANT = "MTAM_only_time_aware_RNN"


class MTAM_no_time_aware_rnn(MTAM):
    """Model/MTAMRec_model.py:93-127: the plain tf GRUCell encodes the short-term intent."""
    VARIANT = "MTAM_no_time_aware_rnn"


class MTAM_via_T_GRU(MTAM):
    """Model/MTAMRec_model.py:167-204: the decoder attends over the time-aware GRU's outputs; the
    short-term intent is layer-normed before it enters the decoder."""
    VARIANT = "MTAM_via_T_GRU"


class MTAM_via_rnn(MTAM):
    """Model/MTAMRec_model.py:206-238: as MTAM_via_T_GRU with the plain GRUCell."""
    VARIANT = "MTAM_via_rnn"


class MTAM_with_T_SeqRec(MTAM):
    """Model/MTAMRec_model.py:275-306: MTAM with the T-SeqRec cell (TimeAwareGRUCell_sigmoid) as the
    short-term encoder."""
    VARIANT = "MTAM_with_T_SeqRec"


class MTAM_hybird(MTAM):
    """Model/MTAMRec_model.py:240-273 (sic): MTAM whose scoring vector is
    concat(short-term intent, layer_norm(decoder output)) . output_w (base_model.output_concat,
    Model/base_model.py:329-357).  The reference imports it in train_process.py:26-27 and never dispatches it."""
    VARIANT = "MTAM_hybird"
