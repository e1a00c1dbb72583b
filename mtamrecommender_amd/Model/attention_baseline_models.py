"""``Time_Aware_Self_Attention_Model`` (Model/attention_baseline_models.py:7-31,47-65), the class the reference's trainer
dispatches for experiment_type 'Time_Aware_Self_Attention_Model' (train_process.py:209-210).  Its graph is
PISTRec's ``Time_Aware_self_Attention_model`` (Model/PISTRec_model.py:38-74) -- same encoder, same variable
scopes -- finished by ``base_model.output()`` (Model/base_model.py:300-328), whose L2 sum includes the user
embedding; PISTRec's hand-written loss (:52-69) leaves it out.  The other classes of that file (Self_Attention_Model,
Ti_Self_Attention_Model, ...) use non-time-aware attention with live dropout and are out of scope."""
from .PISTRec_model import Time_Aware_self_Attention_model
from .self_attention_path import UserL2SelfAttentionPath


class Attention_Baseline_Model(Time_Aware_self_Attention_model):
    PATH_CLASS = UserL2SelfAttentionPath
    ORACLE_NAME = "Time_Aware_Self_Attention_Model"


class Time_Aware_Self_Attention_Model(Attention_Baseline_Model):
    pass
