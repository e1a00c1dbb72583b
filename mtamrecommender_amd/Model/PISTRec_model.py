"""Time-aware self-attention recommender (PISTRec).  Mirror of Model/PISTRec_model.py:11-35
(PISTRec_model) and :38-74 (Time_Aware_self_Attention_model); the reference's three hybrid classes
(:76-459) cannot execute as written and are not mirrored."""
from .base_model import base_model
from .MTAMRec_model import MTAM
from .self_attention_path import SelfAttentionPath
from .variables import init_variables, pistrec_dense_specs


class PISTRec_model(base_model):

    def __init__(self, FLAGS, Embeding, sess):
        super(PISTRec_model, self).__init__(FLAGS, Embeding)
        self.sess = sess
        self.now_bacth_data_size = "batch_size"
        self.num_units = self.FLAGS.num_units
        self.num_heads = self.FLAGS.num_heads
        self.num_blocks = self.FLAGS.num_blocks
        self.dropout_rate = self.FLAGS.dropout
        self.user_embedding, self.behavior_list_embedding_dense, self.item_list_emb, \
            self.category_list_emb, self.position_list_emb, self.time_list, self.timelast_list, \
            self.timenow_list, self.target, self.seq_length = self.embedding.get_embedding(self.num_units)
        self.max_len = self.FLAGS.length_of_user_history
        self.build_model()
        self.init_variables(sess, self.checkpoint_path_dir)


class Time_Aware_self_Attention_model(PISTRec_model):
    PATH_CLASS = SelfAttentionPath
    ORACLE_NAME = "PISTRec"

    def build_model(self, seed=1234):
        D, L, NB = self.num_units, self.max_len, self.num_blocks
        if D != 128:
            raise ValueError("the reference hard-codes num_units=128 for this model (PISTRec_model.py:42)")
        specs = pistrec_dense_specs(D, L, NB)
        values = init_variables(specs, seed=seed + 1)
        live = {s.name: values[s.name] for s in specs if s.trainable_grad}
        self.dead_variables = {s.name: values[s.name] for s in specs if not s.trainable_grad}
        device = getattr(self.sess, "device", "cuda:0")
        self.path = self.PATH_CLASS(self.embedding.tables(), live, L, self.num_heads, NB,
                                      self.FLAGS.regulation_rate, self.FLAGS.max_gradient_norm,
                                      tf_compat_global_norm=self.FLAGS.tf_compat_global_norm, device=device,
                                      optimizer=self.opt, score_dtype=getattr(self.FLAGS, "score_dtype", "f32"))
        self.summery()

    set_variables = MTAM.set_variables
    get_variables = MTAM.get_variables
