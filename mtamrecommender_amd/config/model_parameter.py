"""Flag surface of the time-aware training path, without TensorFlow.

Mirrors the reference's ``model_parameter`` (config/model_parameter.py:4-72:
one ``tf.flags`` definition per hyper-parameter, and ``get_parameter(name)``
at :75-396 that mutates the global FLAGS object per named preset and returns
``self`` so callers read ``.FLAGS``).  Only the flags the hot path or its
trainer read are kept; names, types and defaults are the
reference's.  ``FLAGS`` is a plain attribute bag with ``flag_values_dict()``
(train_process.py:40-42,51 logs that dict).
"""
import argparse


class _Flags(object):
    """Attribute bag standing in for ``tf.flags.FLAGS``."""

    def __init__(self):
        object.__setattr__(self, "_values", {})
        object.__setattr__(self, "_help", {})

    def _define(self, name, default, help_text):
        self._values[name] = default
        self._help[name] = help_text

    def __getattr__(self, name):
        values = object.__getattribute__(self, "_values")
        if name in values:
            return values[name]
        raise AttributeError("unknown flag: %s" % name)

    def __setattr__(self, name, value):
        if name not in self._values:
            raise AttributeError("unknown flag: %s" % name)
        self._values[name] = value

    def flag_values_dict(self):
        return dict(self._values)


class _FlagsModule(object):
    """Just enough of the ``tf.flags`` module API for the reference's idiom."""

    def __init__(self):
        self.FLAGS = _Flags()

    def DEFINE_string(self, name, default, help_text):
        self.FLAGS._define(name, default, help_text)

    def DEFINE_integer(self, name, default, help_text):
        self.FLAGS._define(name, None if default is None else int(default), help_text)

    def DEFINE_float(self, name, default, help_text):
        self.FLAGS._define(name, None if default is None else float(default), help_text)

    def DEFINE_boolean(self, name, default, help_text):
        self.FLAGS._define(name, bool(default), help_text)


# Presets of the MTAM / time-aware self-attention experiments
# (config/model_parameter.py:171-396).  Every preset uses num_heads=1,
# train_batch_size=256, length_of_user_history=50.
_COMMON = dict(causality="unidirection", num_heads=1, learning_rate=0.001, decay_rate=0.995,
               regulation_rate=0.00005, checkpoint_path_dir=None, user_count_limit=1000000,
               init_train_data=False, init_origin_data=False, max_epochs=200,
               load_type="from_scratch", train_batch_size=256, test_batch_size=2048,
               eval_freq=500, dropout=0.5, cuda_visible_devices="0", length_of_user_history=50)

_PRESETS = {
    "MTAMb7_elec": dict(type="elec", num_blocks=7, experiment_type="MTAM"),
    "MTAMb8_elec": dict(type="elec", num_blocks=8, experiment_type="MTAM"),
    "Time_Aware_Self_Attention_Modelb3_yoochoose": dict(
        type="yoochoose", num_blocks=3, experiment_type="Time_Aware_Self_Attention_Model"),
    "Time_Aware_Self_Attention_Modelb3_music": dict(
        type="music", num_blocks=3, experiment_type="Time_Aware_Self_Attention_Model"),
    "Time_Aware_Self_Attention_Modelb2_elec": dict(
        type="elec", num_blocks=2, experiment_type="Time_Aware_Self_Attention_Model"),
    "Time_Aware_Self_Attention_Modelb1_elec": dict(
        type="elec", num_blocks=1, experiment_type="Time_Aware_Self_Attention_Model"),
    # Not in the reference (it has no movielen preset): the
    # BASELINE.json configuration, ml-1m shapes, batch 128.
    "MTAMb1_movielen": dict(type="movielen", num_blocks=1, experiment_type="MTAM",
                            train_batch_size=128),
}


class model_parameter(object):

    def __init__(self):
        self.flags = _FlagsModule()
        f = self.flags
        f.DEFINE_string('version', 'bpr', 'model version')
        f.DEFINE_string('checkpoint_path_dir', None, 'directory of save model')
        f.DEFINE_integer('hidden_units', 128, 'Number of hidden units in each layer')
        f.DEFINE_integer('num_blocks', 6, 'Number of blocks in each attention')
        f.DEFINE_integer('num_heads', 8, 'Number of heads in each attention')
        f.DEFINE_integer('num_units', 128, 'Number of units in each attention')
        f.DEFINE_float('dropout', 0.5, 'Dropout probability (unused on the time-aware path)')
        f.DEFINE_float('regulation_rate', 0.00005, 'L2 regulation rate')
        f.DEFINE_string('optimizer', 'adam', 'Optimizer for training: (adadelta, adam, rmsprop, sgd*)')
        f.DEFINE_float('learning_rate', 0.001, 'Learning rate')
        f.DEFINE_float('decay_rate', 0.001, 'decay rate')
        f.DEFINE_float('max_gradient_norm', 1.0, 'Clip gradients to this norm')
        f.DEFINE_integer('train_batch_size', 256, 'Training Batch size')
        f.DEFINE_integer('test_batch_size', 100, 'Testing Batch size')
        f.DEFINE_integer('max_epochs', 200, 'Maximum # of training epochs')
        f.DEFINE_integer('display_freq', 10, 'Display training status every this iteration')
        f.DEFINE_integer('eval_freq', 200, 'Evaluate every this iteration')
        f.DEFINE_integer('max_len', 150, 'max len of attention')
        f.DEFINE_integer('global_step', 100, 'global_step to summery AUC')
        f.DEFINE_string('cuda_visible_devices', '2', 'Choice which GPU to use')
        f.DEFINE_float('per_process_gpu_memory_fraction', 0.8, 'kept for drop-in; unused')
        f.DEFINE_integer('gap_num', 6, 'sequence gap')
        f.DEFINE_boolean('is_training', True, 'train of inference')
        f.DEFINE_string('type', "yoochoose", 'raw date type')
        f.DEFINE_string('experiment_type', "pistrec", 'experiment date type')
        f.DEFINE_integer('length_of_user_history', 50, 'the maximum length of user history')
        f.DEFINE_integer('length_of_item_history', 50, 'the maximum length of item history')
        f.DEFINE_integer('max_length_seq', 50, 'the length of the seq')
        f.DEFINE_boolean('init_origin_data', False, 'whether to initialize the raw data')
        f.DEFINE_boolean('init_train_data', False, 'whether to initialize the origin data')
        f.DEFINE_integer('user_count_limit', 10000, "the limit of user")
        f.DEFINE_string('causality', "unidirection", "the mask method")
        f.DEFINE_string('pos_embedding', "time", "the method to embed pos")
        f.DEFINE_integer('test_frac', 5, "train test radio")
        f.DEFINE_float('mask_rate', 0.2, 'mask rate')
        f.DEFINE_float('neg_sample_ratio', 20, 'negetive sample ratio')
        f.DEFINE_boolean('remove_duplicate', True, 'whether to remove duplicate entries')
        f.DEFINE_string('experiment_data_type', 'item_based', 'item_based, dual')
        f.DEFINE_string('fine_tune_load_path', None, 'the check point path for the fine tune mode')
        f.DEFINE_string('load_type', "from_scratch", "the type of loading data")
        f.DEFINE_boolean('draw_pic', False, "whether to draw picture")
        f.DEFINE_integer('top_k', 20, "evaluate recall ndcg for k users")
        f.DEFINE_string('experiment_name', "data_init", "the experiment")
        # Additions of this build (not reference flags).
        f.DEFINE_boolean('tf_compat_global_norm', True,
                         'clip with TF1.14 IndexedSlices norm (SURVEY.md App D-5)')
        f.DEFINE_string('score_dtype', 'f32',
                        "'f32', or 'bf16': score the catalog from a bf16 copy of the item table with bf16 MFMA, "
                        "fp32 accumulation and no stored logits (BASELINE.json configs[4]; csrc/score16.hip)")
        f.DEFINE_boolean('native_input', True,
                         'pack batches with libmtam_host.so on a worker thread (DataHandle/native_input.py) instead of '
                         'make_feed_dic_new per step')
        f.DEFINE_boolean('swallow_step_errors', False,
                         'log-and-continue on a failed step like train_process.py:369-371')
        f.DEFINE_boolean('async_loss', True,
                         'Train_main_process: model.train() returns the loss one step late instead of waiting for the '
                         'step it has just launched (the logged averages cover the same steps, shifted by one)')
        f.DEFINE_integer('resident_epoch_max_bytes', 4 << 30,
                         'resident_epoch only while an epoch of packed feeds (4 x (10 B L + 3 B + 4) bytes a batch: '
                         '128 KB at 128 x 50) fits this many bytes of HBM; larger epochs are streamed batch by batch')
        f.DEFINE_boolean('resident_epoch', True,
                         'Train_main_process (native_input, one GPU, adam): pack every full batch of an epoch into HBM '
                         'up front (a ring of feed arenas: 128 KB per batch of 128 x 50) and let the optimizer launch of '
                         'step k hand step k + 1 its feed (Model/time_aware_path.py FeedRing): no host -> device copy '
                         'per step.  The learning rate of every step is baked into its slot (the schedule is a function '
                         'of the global step); a last partial batch takes the ordinary route')
        f.DEFINE_boolean('allow_pickle_parameters', False,
                         'load a parameters.pkl written by the reference (Prepare/prepare_data_base.py:99-101) when no '
                         'parameters.json sits next to it; unpickling executes what the file says, so opt in only for '
                         'directories you produced yourself')

    SUPPORTED_NUM_UNITS = 128       # MTAM_D of include/mtam_hip.h: every kernel is built for 512-byte rows

    def validate(self):
        """Reject flag values the HIP path has no kernels for -- here, not deep inside a launch."""
        F = self.flags.FLAGS
        if int(F.num_units) != self.SUPPORTED_NUM_UNITS:
            raise ValueError("num_units = %s: this build's kernels are compiled for num_units = %d only "
                             "(MTAM_D in include/mtam_hip.h; every reference preset uses 128)"
                             % (F.num_units, self.SUPPORTED_NUM_UNITS))
        if F.score_dtype not in ("f32", "bf16"):
            raise ValueError("score_dtype must be 'f32' or 'bf16' (got %r)" % (F.score_dtype,))
        return self

    def get_parameter(self, type):
        preset = _PRESETS.get(type)
        if preset is not None:
            merged = dict(_COMMON)
            merged.update(preset)
            merged.setdefault("version", type)
            for key, value in merged.items():
                setattr(self.flags.FLAGS, key, value)
        self.FLAGS = self.flags.FLAGS
        return self.validate()

    def parse_argv(self, argv):
        """CLI overrides (``--flag value``), as absl would apply them."""
        parser = argparse.ArgumentParser(add_help=True)
        values = self.flags.FLAGS.flag_values_dict()
        for name, default in values.items():
            if isinstance(default, bool):
                parser.add_argument("--" + name, type=lambda s: s.lower() in ("1", "true", "yes"),
                                    default=default)
            elif isinstance(default, int):
                parser.add_argument("--" + name, type=int, default=default)
            elif isinstance(default, float):
                parser.add_argument("--" + name, type=float, default=default)
            else:
                parser.add_argument("--" + name, type=str, default=default)
        ns = parser.parse_args(argv)
        for name in values:
            setattr(self.flags.FLAGS, name, getattr(ns, name))
        self.FLAGS = self.flags.FLAGS
        return self.validate()
