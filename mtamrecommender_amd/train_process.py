"""Trainer: batch loop, learning-rate schedule, evaluation cadence.

Mirror of the reference's ``Train_main_process`` (train_process.py:34-472) for the
experiment types the HIP path implements.  Kept semantics:

* learning rate (train_process.py:154-159,324,333-336): every epoch starts from
  ``FLAGS.learning_rate``; while the current value is > 1e-3 the next one is
  ``FLAGS.learning_rate * 0.99 ** floor(step / 100)``, otherwise
  ``1e-3 * FLAGS.decay_rate ** floor(step / 100)`` (tf.train.exponential_decay,
  staircase) -- with the default learning_rate 1e-3 the second rule always applies;
* ``random.shuffle(train_set)`` per epoch, sequential ``DataInput`` batches with a
  short final batch (:321,326);
* evaluation before training, every ``eval_freq`` global steps and at the end of
  every epoch (:308,350-362,379); per-batch metrics are averaged UNWEIGHTED over
  batches (:257-277); best-so-far per K only when both HR and NDCG improve (:279-288);
* checkpoint at every evaluation point and at the end (:364,403).

Differences: the reference swallows every per-step exception and continues
(:329-371); here that is opt-in (``FLAGS.swallow_step_errors``).  The reference's
data adapters are missing from its tree, so records come from the
synthetic generator unless ``train_set`` / ``test_set`` are handed in.
"""
import math
import os
import random
import time

import numpy as np

from .config.model_parameter import model_parameter
from .DataHandle.get_input_data import DataInput
from .util.model_log import create_log

random.seed(1234)
np.random.seed(1234)

KS = (1, 5, 10, 30, 50)


def exponential_decay(learning_rate, global_step, decay_steps, decay_rate):
    """tf.train.exponential_decay(staircase=True) [TF1.14]: lr * rate ** floor(step / decay_steps),
    evaluated in float32 like the TF op."""
    p = np.float32(math.floor(global_step / decay_steps))
    return float(np.float32(learning_rate) * np.power(np.float32(decay_rate), p))


def next_learning_rate(current, flags_learning_rate, decay_rate, global_step):
    """One evaluation of the reference's lr1 / lr2 choice (train_process.py:333-336)."""
    if current > 0.001:
        return exponential_decay(flags_learning_rate, global_step, 100, 0.99)
    return exponential_decay(0.001, global_step, 100, decay_rate)


def average_metrics(per_batch):
    """Unweighted mean over batches of the 10-tuples from metrics_topK (train_process.py:257-277)."""
    if not per_batch:
        return tuple([0.0] * 10)
    return tuple(float(np.mean([m[i] for m in per_batch])) for i in range(10))


class _GlobalBatch(list):
    """A rank's slice of a global batch (a list of records) that remembers the size of the whole batch."""

    def __init__(self, records, global_size):
        super(_GlobalBatch, self).__init__(records)
        self.global_size = global_size


class Train_main_process(object):

    def __init__(self, experiment_name="MTAMb1_movielen", argv=None, train_set=None, test_set=None,
                 counts=None, device="cuda:0"):
        start_time = time.time()
        model_parameter_ins = model_parameter()
        model_parameter_ins.get_parameter(experiment_name)
        if argv:
            model_parameter_ins.parse_argv(argv)
        self.FLAGS = model_parameter_ins.FLAGS
        self.logger = create_log(type=self.FLAGS.type, experiment_type=self.FLAGS.experiment_type,
                                 version=self.FLAGS.version).logger
        for k, v in self.FLAGS.flag_values_dict().items():
            self.logger.info("%s: %s" % (k, v))
        L = self.FLAGS.length_of_user_history
        if train_set is None:
            from .data.synthetic import ML1M, SyntheticCatalog, make_records
            cat = SyntheticCatalog(seed=1234, **ML1M)
            train_set = make_records(cat, 20000, L, seed=1234)
            test_set = make_records(cat, 2000, L, seed=4321)
            counts = dict(user_count=cat.user_count, item_count=cat.item_count,
                          category_count=cat.category_count)
        self.train_set, self.test_set = list(train_set), list(test_set)
        self.user_count, self.item_count, self.category_count = \
            counts["user_count"], counts["item_count"], counts["category_count"]
        self.logger.info('Init data finish.\tCost time: %.2fs' % (time.time() - start_time))
        from .Embedding.Behavior_embedding_time_aware_attention import Behavior_embedding_time_aware_attention
        self.emb = Behavior_embedding_time_aware_attention(
            is_training=self.FLAGS.is_training, user_count=self.user_count, item_count=self.item_count,
            category_count=self.category_count, max_length_seq=L)
        # Data parallel (one process per GPU, torch.distributed.run sets WORLD_SIZE / RANK / LOCAL_RANK): every rank
        # walks the same global batches (same shuffle seed) and trains on its slice; see data_parallel.py
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        if self.world > 1:
            import torch
            import torch.distributed as dist
            device = "cuda:%d" % int(os.environ.get("LOCAL_RANK", "0"))
            torch.cuda.set_device(device)
            if not dist.is_initialized():
                os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
                dist.init_process_group(backend="nccl", device_id=torch.device(device))
        self.device = device
        self.global_step = 0
        self.one_epoch_step = 0
        self.now_epoch = 0
        self.best = {k: (0.0, 0.0) for k in KS}

    # ------------------------------------------------------------------ pieces
    def build_model(self):
        from .Model.base_model import Session
        from .Model import MTAMRec_model as family
        self.sess = Session(self.device)
        # train_process.py:188-204 dispatches the MTAM family by experiment_type ('T_GRU' is
        # MTAM_only_time_aware_RNN there)
        members = {"MTAM": family.MTAM, "T_GRU": family.MTAM_only_time_aware_RNN,
                   "MTAM_no_time_aware_rnn": family.MTAM_no_time_aware_rnn,
                   "MTAM_via_T_GRU": family.MTAM_via_T_GRU, "MTAM_via_rnn": family.MTAM_via_rnn,
                   "MTAM_with_T_SeqRec": family.MTAM_with_T_SeqRec,          # train_process.py:217-218
                   "MTAM_hybird": family.MTAM_hybird}      # imported by the reference's trainer, reachable only here
        if self.FLAGS.experiment_type in members:
            self.model = members[self.FLAGS.experiment_type](self.FLAGS, self.emb, self.sess)
        elif self.FLAGS.experiment_type == "Time_Aware_Self_Attention_Model":
            # train_process.py:209-210 -> Model/attention_baseline_models.py:47-65: PISTRec's encoder under
            # base_model.output() (user embedding inside the L2 sum)
            from .Model.attention_baseline_models import Time_Aware_Self_Attention_Model
            self.model = Time_Aware_Self_Attention_Model(self.FLAGS, self.emb, self.sess)
        elif self.FLAGS.experiment_type == "PISTRec":
            # Model/PISTRec_model.py:38-74 (its own loss, no user L2); not dispatched by the reference's trainer
            from .Model.PISTRec_model import Time_Aware_self_Attention_model
            self.model = Time_Aware_self_Attention_model(self.FLAGS, self.emb, self.sess)
        else:
            raise NotImplementedError("experiment_type %r has no HIP path (MTAM and the time-aware "
                                      "self-attention model only)" % self.FLAGS.experiment_type)
        # the trainer only averages and logs losses: let train() return them one step late instead of blocking
        self.model.async_loss = bool(getattr(self.FLAGS, "async_loss", True))
        if self.world > 1:
            from . import data_parallel
            data_parallel.attach(self.model.path, self.world)
            data_parallel.broadcast_parameters(self.model.path)
        return self.model

    # ------------------------------------------------------------------- feeds
    def _init_feeds(self):
        """Native feed (FLAGS.native_input): records move into libmtam_host.so once; batches are padded and
        range-checked in C++ one step ahead of the device.  Otherwise the reference's route: Python lists
        through DataInput and make_feed_dic_new."""
        self._native = bool(getattr(self.FLAGS, "native_input", False))
        if self._native:
            from .DataHandle.native_input import BatchPacker, RecordSet
            self._packer = BatchPacker(self.model.path, self.emb)
            self._train_rs = RecordSet.from_records(self.train_set)
            self._test_rs = RecordSet.from_records(self.test_set)
            self._order = list(range(len(self.train_set)))

    def _train_batches(self):
        """(step, batch) over the epoch; data parallel: ``train_batch_size`` is the GLOBAL batch and every rank gets
        its contiguous slice of each (sizes differ by at most one; the loss stays the mean over the global batch)."""
        shard = (self.rank, self.world) if self.world > 1 else None
        if not self._native:
            random.shuffle(self.train_set)
            batches = DataInput(self.train_set, self.FLAGS.train_batch_size)
            if shard is None:
                return batches
            from .data_parallel import keep_global_batch, shard as cut
            return ((i, _GlobalBatch(cut(b, self.rank, self.world), len(b))) for i, b in batches
                    if keep_global_batch(len(b), self.world))
        if self._resident_epoch_on():
            return self._resident_batches()
        from .DataHandle.native_input import NativeDataInput
        random.shuffle(self._order)          # the same permutation random.shuffle(train_set) would apply
        return NativeDataInput(self._train_rs, self.FLAGS.train_batch_size, self._packer, index=self._order,
                               consumer="train", shard=shard)

    def _resident_epoch_on(self):
        """FLAGS.resident_epoch: the epoch's full batches live in HBM and no feed is copied per step
        (base_model.load_resident_epoch) -- native feed, one GPU, Adam, captured steps."""
        if not (bool(getattr(self.FLAGS, "resident_epoch", False)) and self._native and self.world == 1 and
                self.model.path.optimizer == "adam" and self.model.use_graph):
            return False
        from .Model.time_aware_path import arena_layout
        B = int(self.FLAGS.train_batch_size)
        n_full = len(self.train_set) // B
        if n_full < 1:
            return False
        epoch_bytes = n_full * arena_layout(B, self.FLAGS.length_of_user_history)[1] * 4
        if epoch_bytes > int(getattr(self.FLAGS, "resident_epoch_max_bytes", 4 << 30)):
            return False
        return self.model.path.ring_supported(self.model.path.batch(B))

    def _resident_plan(self, global_step0):
        """Shuffle the order for one more epoch and describe it: (global step it starts at, record order of its full
        batches, their learning rates -- next_learning_rate from FLAGS.learning_rate at the epoch's first step, as the
        loop below computes them --, the records of the last partial batch)."""
        B = int(self.FLAGS.train_batch_size)
        random.shuffle(self._order)          # the same permutation random.shuffle(train_set) would apply
        n_full = len(self._order) // B
        lrs, lr = [], self.FLAGS.learning_rate
        for k in range(n_full):
            lr = next_learning_rate(lr, self.FLAGS.learning_rate, self.FLAGS.decay_rate, global_step0 + k)
            lrs.append(lr)
        return dict(step0=global_step0, index=np.asarray(self._order[:n_full * B], np.int64), lrs=lrs,
                    tail=list(self._order[n_full * B:]), B=B, prepared=None)

    def _resident_prepare(self, plan):
        if plan["lrs"]:
            plan["prepared"] = self.model.prepare_resident_epoch(self._train_rs, plan["index"], plan["B"], plan["lrs"],
                                                                 self._packer)
        return plan

    def _resident_batches(self):
        """The same (step, batch) stream as NativeDataInput over the same shuffled order: every full batch as a
        ResidentBatch (its learning rate -- a function of the global step -- baked into its slot), then the last
        partial batch, if any, the ordinary way (the reference keeps it: DataInput, input.py:12-19).  The NEXT
        epoch is shuffled and packed on a worker thread as soon as this one is on the device."""
        plan = getattr(self, "_resident_next", None)
        self._resident_next = None
        if plan is not None and plan["step0"] != self.global_step:
            # (a swallowed step error moved the global step: the baked learning rates would be another step's.
            # The order stays -- it is the epoch's shuffle -- the rates are recomputed.)
            B, lrs, lr = plan["B"], [], self.FLAGS.learning_rate
            for k in range(len(plan["lrs"])):
                lr = next_learning_rate(lr, self.FLAGS.learning_rate, self.FLAGS.decay_rate, self.global_step + k)
                lrs.append(lr)
            if plan["prepared"] is not None:
                plan["prepared"].thread.join()
            plan.update(step0=self.global_step, lrs=lrs, prepared=None)
        if plan is None:
            plan = self._resident_plan(self.global_step)
        if plan["prepared"] is None:
            self._resident_prepare(plan)
        handles = self.model.load_resident_epoch(prepared=plan["prepared"]) if plan["lrs"] else []
        if self.now_epoch + 1 < self.FLAGS.max_epochs:
            n_steps = len(plan["lrs"]) + (1 if plan["tail"] else 0)
            self._resident_next = self._resident_prepare(self._resident_plan(self.global_step + n_steps))
        for k, h in enumerate(handles):
            yield k + 1, h
        if plan["tail"]:
            yield len(handles) + 1, self._packer.pack(self._train_rs, plan["tail"], consumer="train")

    def _test_batches(self):
        if not self._native:
            return DataInput(self.test_set, self.FLAGS.test_batch_size)
        from .DataHandle.native_input import NativeDataInput
        # its own arena pool: an evaluation pass runs while the next TRAIN batch is already prefetched
        return NativeDataInput(self._test_rs, self.FLAGS.test_batch_size, self._packer, consumer="eval")

    def eval_topk(self):
        per_batch = []
        for _, batch_data in self._test_batches():
            per_batch.append(self.model.metrics_topK(sess=self.sess, batch_data=batch_data,
                                                     global_step=self.global_step, topk=self.FLAGS.top_k))
        avg = average_metrics(per_batch)
        for i, k in enumerate(KS):
            hr, ndcg = avg[2 * i], avg[2 * i + 1]
            if hr > self.best[k][0] and ndcg > self.best[k][1]:
                self.best[k] = (hr, ndcg)
            self.model.train_writer.add_summary({"recall@%d" % k: hr, "ndgc@%d" % k: ndcg}, self.global_step)
            self.logger.info('Test recall rate @ %d : %.4f   ndcg @ %d: %.4f' % (k, hr, k, ndcg))
        return avg

    def save_model(self):
        self.model._current_table()      # (data-parallel "sharded-table": a collective, every rank takes part)
        if self.rank == 0:
            self.model.save(self.sess, self.global_step)

    # -------------------------------------------------------------------- loop
    def train(self, max_steps=None):
        start_time = time.time()
        self.build_model()
        self._init_feeds()
        self.logger.info('Init finish.\tCost time: %.2fs' % (time.time() - start_time))
        test_start = time.time()
        self.eval_topk()
        self.logger.info('End test. \tTest Cost time: %.2fs' % (time.time() - test_start))
        self.logger.info('Training....\tmax_epochs:%d\tepoch_size:%d'
                         % (self.FLAGS.max_epochs, self.FLAGS.train_batch_size))
        start_time, avg_loss, step_loss = time.time(), 0.0, float("nan")
        for epoch in range(self.FLAGS.max_epochs):
            batches = self._train_batches()
            self.logger.info('tain_set:%d' % len(self.train_set))
            epoch_start_time = time.time()
            learning_rate = self.FLAGS.learning_rate
            for step_i, train_batch_data in batches:
                try:
                    learning_rate = next_learning_rate(learning_rate, self.FLAGS.learning_rate,
                                                       self.FLAGS.decay_rate, self.global_step)
                    add_summary = bool(self.global_step % self.FLAGS.display_freq == 0)
                    if self.world > 1:      # the loss is a mean over the whole batch, not over this rank's slice
                        self.model.path.global_batch = train_batch_data.global_size
                    got = self.model.train(self.sess, train_batch_data, learning_rate,
                                           add_summary, self.global_step, epoch)
                    self.global_step = self.global_step + 1
                    self.one_epoch_step = self.one_epoch_step + 1
                    at_eval = self.global_step % self.FLAGS.eval_freq == 0
                    # async_loss: `got` is the PREVIOUS step's loss (nothing on the first call); every loss is
                    # logged exactly once under the step it belongs to, and the window average is complete
                    # because the most recent step's is drained before it is printed
                    for loss, merge in filter(None, (got, self.model.drain_loss() if at_eval else None)):
                        if merge.get("loss_step") is not None:
                            step_loss = loss
                            self.model.train_writer.add_summary(merge, merge["loss_step"])
                            avg_loss = avg_loss + loss
                    if at_eval:
                        self.logger.info("Epoch step is " + str(self.one_epoch_step))
                        self.logger.info("Global step is " + str(self.global_step))
                        self.logger.info("Train_loss is " + str(avg_loss / self.FLAGS.eval_freq))
                        self.eval_topk()
                        avg_loss = 0
                        self.save_model()
                except Exception as e:
                    if not self.FLAGS.swallow_step_errors:
                        raise
                    self.logger.info("Error in training step")
                    self.logger.info(e)
                if max_steps is not None and self.global_step >= max_steps:
                    break
            last = self.model.drain_loss()
            if last is not None:
                step_loss = last[0]
                self.model.train_writer.add_summary(last[1], last[1]["loss_step"])
                avg_loss = avg_loss + step_loss
            self.logger.info('one epoch Cost time: %.2f' % (time.time() - epoch_start_time))
            self.logger.info("Global step is " + str(self.global_step))
            self.logger.info("Train_loss is " + str(step_loss))
            self.eval_topk()
            for k in KS:
                self.logger.info('Max recall rate @ %d: %.4f   ndcg @ %d: %.4f' % (k, self.best[k][0], k, self.best[k][1]))
            self.one_epoch_step = 0
            self.logger.info('Epoch %d DONE\tCost time: %.2f' % (self.now_epoch, time.time() - start_time))
            self.now_epoch = self.now_epoch + 1
            if max_steps is not None and self.global_step >= max_steps:
                break
        self.save_model()
        self.logger.info('Finished')


if __name__ == '__main__':
    import sys
    name = "MTAMb1_movielen"
    argv = sys.argv[1:]
    if argv and not argv[0].startswith("--"):
        name, argv = argv[0], argv[1:]
    main_process = Train_main_process(name, argv)
    main_process.train()
