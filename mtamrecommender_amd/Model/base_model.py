"""Base class of the time-aware models: optimizer choice, train / eval entry
points, checkpointing, metrics.  Mirror of Model/base_model.py:18-357 with the
TensorFlow session replaced by the HIP step of ``time_aware_path.py``.

Kept from the reference: constructor signature ``(FLAGS, Embedding)``,
checkpoint directory rule (:33-39), ``train`` returning ``(loss, summary)``
(:150-167), ``metrics_topK`` returning the ten floats for K in {1,5,10,30,50}
(:188-213), ``calculate_topK`` (:215-242), ``save`` / ``restore`` (:124-147).
"""
import glob
import math
import os
import time

import numpy as np
import torch

from ..util.model_log import create_log


class Session(object):
    """Stand-in for ``tf.Session``: names the device the step runs on."""

    def __init__(self, device="cuda:0"):
        self.device = device

    def as_default(self):
        return self

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


class SummaryWriter(object):
    """``tf.summary.FileWriter`` stand-in: scalars appended to a CSV file."""

    def __init__(self, path=None):
        self.path = path
        self.rows = []

    def add_summary(self, summary, global_step=None):
        if not summary:
            return
        for tag, value in summary.items():
            if tag == "loss_step":           # which step the values belong to (async_loss): the row key, not a value
                continue
            self.rows.append((global_step, tag, float(value)))
        if self.path is not None and len(self.rows) >= 256:
            self.flush()

    def flush(self):
        if self.path is None or not self.rows:
            return
        os.makedirs(os.path.dirname(self.path), exist_ok=True)
        with open(self.path, "a") as f:
            for step, tag, value in self.rows:
                f.write("%s,%s,%.8g\n" % (step, tag, value))
        self.rows = []


class ResidentBatch(object):
    """Handle of one batch of a device-resident epoch (``base_model.load_resident_epoch``): slot ``k`` of the feed
    ring, the learning rate baked into it and the batch's target items (a view of the pinned staging buffer)."""

    def __init__(self, ring, k, B, lr, targets):
        self.ring, self.k, self.B, self.lr, self._targets = ring, k, B, lr, targets

    def __len__(self):
        return self.B

    def field(self, name):
        if name != "target_item_id":
            raise KeyError(name)
        return self._targets.numpy()


class _ResidentPrep(object):
    """An epoch being packed into pinned staging buffer ``j`` (base_model.prepare_resident_epoch)."""

    def __init__(self, j, stage, n, B, lrs):
        self.j, self.stage, self.n, self.B, self.lrs = j, stage, n, B, lrs
        self.thread, self.error = None, None


class base_model(object):

    def __init__(self, FLAGS, Embedding):
        self.FLAGS = FLAGS
        self.version = self.FLAGS.version
        self.learning_rate = "learning_rate"      # placeholder name (float64 scalar in the reference, :25)
        if self.FLAGS.checkpoint_path_dir is not None:
            self.checkpoint_path_dir = self.FLAGS.checkpoint_path_dir
        else:
            self.checkpoint_path_dir = "data/check_point/" + self.FLAGS.type + "_" + \
                self.FLAGS.experiment_type + "_" + self.version
            if not os.path.exists(self.checkpoint_path_dir):
                os.makedirs(self.checkpoint_path_dir)
        self.init_optimizer()
        self.embedding = Embedding
        self.logger = create_log().logger
        self.path = None            # TimeAwarePath, built by build_model()
        self.use_graph = os.environ.get("MTAM_HIP_GRAPH", "1") != "0"
        self._dp_mode = os.environ.get("MTAM_DP_GRAPH")       # None: decided at the first data-parallel step
        self._graphs = {}
        # load_resident_epoch: two pinned [batches, words] staging buffers (+ the event behind the copy that last read
        # each), used in turn; the side stream of the host -> device copies
        self._resident_stages, self._resident_flip, self._resident_stream = [None, None], 0, None
        self._feed_refs = {}        # pinned feed arenas whose address a captured step reads (kept alive with the graphs)
        # async_loss: train() hands back the loss of the PREVIOUS step (with the global_step it belongs to; nothing
        # on the first call) instead of blocking on this step's -- the reference's sess.run blocks (:159-164), and
        # with a blocking read-back the host cannot prepare batch t + 1 while the device runs step t.  Off by
        # default (drop-in semantics); Train_main_process, which only averages and logs the losses, turns it on
        # and calls drain_loss() wherever it needs the most recent step's.
        self.async_loss = False
        self._loss_ring, self._loss_slot, self._loss_unread, self.loss_step = None, 0, False, None

    # ------------------------------------------------------------ life cycle
    def init_variables(self, sess, path, var_list=None):
        if self.FLAGS.load_type == "full":
            self.restore(sess, path=path)
        elif self.FLAGS.load_type == "fine_tune":
            self.restore(sess, path=self.FLAGS.fine_tune_load_path, variable_list=var_list)
        elif self.FLAGS.load_type == "from_scratch":
            pass

    def init_optimizer(self):
        # Model/base_model.py:71-80: 'adadelta', 'adam', 'rmsprop', anything else -> plain SGD.
        # Each has a HIP update kernel (mtam_adam / mtam_opt_update).
        name = self.FLAGS.optimizer
        self.opt = name if name in ("adadelta", "adam", "rmsprop") else "sgd"

    def build_model(self):
        pass

    def summery(self):
        stamp = time.strftime("%Y-%m-%d--%H:%M:%S", time.localtime(time.time()))
        name = "data/tensorboard_result/%s_%s_%s_%s" % (self.FLAGS.type, self.FLAGS.experiment_type,
                                                        self.FLAGS.version, stamp)
        write = os.environ.get("MTAM_WRITE_SUMMARIES", "0") == "1"
        self.train_writer = SummaryWriter(name + "/tensorboard_train.csv" if write else None)
        self.eval_writer = SummaryWriter(name + "/tensorboard_eval.csv" if write else None)

    # -------------------------------------------------------------- checkpoint
    def save(self, sess, global_step=None, path=None, variable_list=None):
        if path is None:
            path = self.checkpoint_path_dir
        os.makedirs(path, exist_ok=True)
        save_path = os.path.join(path, "model.ckpt-%s.pt" % global_step)
        p = self.path
        state = {"global_step": global_step,
                 "tables": {k: v.detach().cpu() for k, v in p.tables.items()},
                 "dense": {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in p.dense_tf().items()},
                 "dead": {k: torch.from_numpy(v) for k, v in self.dead_variables.items()},
                 "adam": p.optimizer_state()}
        if variable_list is not None:
            state["dense"] = {k: v for k, v in state["dense"].items() if k in variable_list}
        torch.save(state, save_path)
        self.logger.info('model saved at %s' % save_path)
        return save_path

    def restore(self, sess, path, variable_list=None, graph_path=None):
        files = sorted(glob.glob(os.path.join(path, "model.ckpt-*.pt")), key=os.path.getmtime)
        if not files:
            from ..util import tf_bundle
            if variable_list is None and os.path.isdir(path) and tf_bundle.latest_checkpoint(path):
                # a directory the REFERENCE saved into: its TensorFlow bundle, read without TensorFlow
                prefix = self.import_tf_checkpoint(path)
                self.logger.info('model restored from the TensorFlow checkpoint %s' % prefix)
                return
            raise FileNotFoundError("no checkpoint under %s" % path)
        state = torch.load(files[-1], weights_only=True)
        p = self.path
        for k, v in state["tables"].items():
            p.tables[k].copy_(v)
        dense = p.dense_tf()
        for k, v in state["dense"].items():
            if variable_list is None or k in variable_list:
                dense[k] = v.numpy()
        p.params.copy_(torch.from_numpy(p.layout.pack(dense)))
        p.refresh_derived()
        if variable_list is None and "adam" in state:
            p.load_optimizer_state(state["adam"])
        self.logger.info('model restored from %s' % path)

    # ------------------------------------------------- TF-named interchange
    def tf_named_arrays(self):
        """Every variable a TF 1.14 ``tf.train.Saver`` would hold for this graph, keyed by its TF name
        (SURVEY.md App B): trainable variables, the never-updated ones, and Adam's slots
        ``<name>/Adam`` (m), ``<name>/Adam_1`` (v), ``beta1_power``, ``beta2_power``."""
        p = self.path
        out = dict(self.get_variables())
        if p.optimizer == "adam":
            m, v = p.layout.unpack(p.m.detach().cpu().numpy()), p.layout.unpack(p.v.detach().cpu().numpy())
            for k in m:
                out[k + "/Adam"], out[k + "/Adam_1"] = m[k], v[k]
            for k in p.tables:
                out["embedding_layer/%s/Adam" % k] = p.tm[k].detach().cpu().numpy()
                out["embedding_layer/%s/Adam_1" % k] = p.tv[k].detach().cpu().numpy()
            st = p.adam_state.cpu().numpy()
            out["beta1_power"], out["beta2_power"] = np.float32(st[4]), np.float32(st[5])
        return out

    def export_tf_npz(self, file_path):
        """``tf_named_arrays()`` as one .npz file.  The other direction of ``import_tf_npz``; a reference checkpoint
        becomes such a file with ``{n: r.get_tensor(n) for n in r.get_variable_to_shape_map()}`` on a machine that
        has TF -- or is read directly by ``import_tf_checkpoint``."""
        out = self.tf_named_arrays()
        np.savez(file_path, **{k.replace("/", "__"): v for k, v in out.items()})
        return sorted(out)

    def export_tf_checkpoint(self, prefix):
        """``tf_named_arrays()`` as a TensorFlow checkpoint bundle (``<prefix>.index``, ``.data-00000-of-00001`` and
        the directory's ``checkpoint`` state file): what ``tf.train.Saver().save`` leaves (Model/base_model.py:331-337
        of the reference), written by ``util/tf_bundle.py`` without TensorFlow."""
        from ..util import tf_bundle
        names = tf_bundle.write_bundle(prefix, {k: np.asarray(v) for k, v in self.tf_named_arrays().items()})
        tf_bundle.write_checkpoint_state(os.path.dirname(os.path.abspath(prefix)), prefix)
        return names

    def import_tf_npz(self, file_path):
        """Load variables (and, when present, Adam slots) from an ``export_tf_npz``-style file.  Keys may
        carry TF's ``:0`` suffix; shapes are checked; a trainable variable missing from the file raises."""
        raw = np.load(file_path, allow_pickle=False)
        self._import_tf_arrays({k.replace("__", "/").split(":")[0]: raw[k] for k in raw.files}, file_path)

    def import_tf_checkpoint(self, prefix_or_dir):
        """Load a TensorFlow checkpoint bundle -- a reference ``model.ckpt-<step>`` prefix, or a directory holding a
        ``checkpoint`` state file (``tf.train.latest_checkpoint``) -- read by ``util/tf_bundle.py`` without TensorFlow
        (``saver.restore``, Model/base_model.py:339-343 of the reference).  Variables the model does not have
        (``global_step``, other optimizers' slots) are ignored."""
        from ..util import tf_bundle
        prefix = prefix_or_dir
        if os.path.isdir(prefix_or_dir):
            prefix = tf_bundle.latest_checkpoint(prefix_or_dir)
            if prefix is None:
                raise FileNotFoundError("no TensorFlow checkpoint bundle under %s" % prefix_or_dir)
        self._import_tf_arrays(tf_bundle.read_bundle(prefix), prefix)
        return prefix

    def _import_tf_arrays(self, arrays, origin):
        p = self.path
        mine = self.get_variables()
        for k, v in mine.items():
            if k not in arrays:
                if k in self.dead_variables:
                    continue
                raise KeyError("variable %s missing from %s" % (k, origin))
            if tuple(arrays[k].shape) != tuple(v.shape):
                raise ValueError("%s: shape %s in the file, %s in the model" % (k, arrays[k].shape, v.shape))
        self.set_variables({k: arrays[k] for k in mine if k in arrays})
        if p.optimizer == "adam" and "beta1_power" in arrays:
            dense = p.dense_tf()
            m = {k: arrays.get(k + "/Adam", np.zeros_like(v)) for k, v in dense.items()}
            v_ = {k: arrays.get(k + "/Adam_1", np.zeros_like(v)) for k, v in dense.items()}
            p.m.copy_(torch.from_numpy(p.layout.pack(m)))
            p.v.copy_(torch.from_numpy(p.layout.pack(v_)))
            for k in p.tables:
                if "embedding_layer/%s/Adam" % k in arrays:
                    p.tm[k].copy_(torch.from_numpy(np.ascontiguousarray(arrays["embedding_layer/%s/Adam" % k])))
                    p.tv[k].copy_(torch.from_numpy(np.ascontiguousarray(arrays["embedding_layer/%s/Adam_1" % k])))
            st = p.adam_state.cpu()
            st[4], st[5] = float(arrays["beta1_power"]), float(arrays["beta2_power"])
            p.adam_state.copy_(st)

    # ------------------------------------------------------------------ step
    def _run(self, kind, bt, fn, key_extra=(), **graph_kw):
        """Eager on first use of a batch size, then one hipGraph replay per step.  ``key_extra``: whatever else the
        captured launches bake in (the pinned arena a feed copy reads, the pinned slot a loss copy writes)."""
        if not self.use_graph:
            fn(bt)
            return
        # the global batch (data parallel: 1 / gb is a kernel argument) is part of what a captured step bakes in
        key = (kind, bt.B, getattr(self.path, "global_batch", None)) + tuple(key_extra)
        g = self._graphs.get(key)
        if g is None:
            if self._graphs.get(key + ("warm",)) is None:
                fn(bt)                                   # first call: plain launches (also warms caches)
                self._graphs[key + ("warm",)] = True
                return
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            if torch.distributed.is_available() and torch.distributed.is_initialized():
                # a process group's watchdog thread may query events while this thread captures: keep the
                # capture's error checking local to this thread
                graph_kw.setdefault("capture_error_mode", "thread_local")
            with torch.cuda.graph(g, **graph_kw):        # records the launches, executes nothing
                fn(bt)
            self._graphs[key] = g
        g.replay()

    def step_train(self, bt):
        """One training step on the feed already in ``bt.arena``.

        Data parallel: the RCCL all-reduce sits between backward and update.  ``MTAM_DP_GRAPH``:
        ``fused`` captures forward, backward, the collective and the update into ONE hipGraph (RCCL
        enqueues into the capturing stream; thread-local capture mode keeps the process group's watchdog
        thread from invalidating it); if that capture raises, or with ``split``, the step is two graphs
        with the collective launched eagerly between them.  Default: ``split`` for more than one rank --
        the fused form has only been run with a one-rank group (this build's GPU box has one GPU), and a
        collective that misbehaves inside a replayed graph cannot be recovered from -- ``fused`` otherwise."""
        p = self.path
        ring = getattr(bt, "feed_ring", None)
        if ring is not None and (p.sharded is not None or getattr(p, "sharded_scoring", None) is not None):
            raise RuntimeError("a feed ring is attached to this batch but the row-sharded exchanges update through "
                               "their own launches: detach it (bt.feed_ring = None)")
        if ring is not None and not ring.primed:
            if not ring.taken:
                raise RuntimeError("feed ring: prime() it before the first step")
            ring = None        # the caller has just put its own feed into the arena: an ordinary step, not the ring's
        if getattr(p, "sharded_scoring", None) is not None:
            # scoring row-sharded over the ranks: forward to pred (graph), the two scoring passes with their small
            # collectives (eager), backward from d_pred (graph), exchange + update (eager)
            if not p.sharded_scoring.replicate_table:       # "sharded-table": the batch's item rows from their owners
                p.row_iota(bt)
                p.sharded_scoring.fetch_history_rows(bt)
            self._run("train_pre", bt, p.forward_to_pred_kernels)
            p.sharded_scoring.score(bt)
            self._run("train_post", bt, p.backward_from_pred_kernels)
            p.sharded_scoring.exchange_and_apply(bt)
            return
        if getattr(p, "sharded", None) is not None:
            # large catalogs: forward + backward as one graph, then the row-sharded exchange and update eagerly
            # (a handful of launches between collectives; the step is milliseconds long at these sizes)
            self._run("train_fb", bt, p.forward_backward_kernels)
            p.sharded.exchange_and_apply(bt)
            return
        # ring: the feed is already in the arena (prime(), then every step's optimizer launch brings the next one) --
        # nothing is copied in front of the graph
        rkey = ("ring", ring.serial) if ring is not None else ()
        if p.allreduce_fn is None:
            self._run("train", bt, p.ring_train_kernels if ring is not None else p.train_kernels, key_extra=rkey)
            if ring is not None:
                ring.consumed += 1
            return
        if self._dp_mode is None:
            self._dp_mode = "fused" if p.world_size == 1 else "split"
        if self._dp_mode == "fused":
            try:
                self._run("train_dp", bt, p.ring_train_kernels if ring is not None else p.train_kernels,
                          key_extra=rkey, capture_error_mode="thread_local")
                if ring is not None:
                    ring.consumed += 1
                return
            except Exception as e:                       # capture of the collective not supported here
                self.logger.info("fused data-parallel graph unavailable (%s): using split graphs" % (e,))
                self._dp_mode = "split"
                self._graphs.pop(("train_dp", bt.B, getattr(p, "global_batch", None)) + rkey, None)
                torch.cuda.synchronize()
        self._run("train_fb", bt, p.forward_backward_kernels)
        p.allreduce_fn(p, bt)
        self._run("train_up", bt, p.ring_clip_and_apply if ring is not None else p.clip_and_apply, key_extra=rkey)
        if ring is not None:
            ring.consumed += 1

    def _load(self, batch_data, learning_rate=None):
        """Feed -> device arena.  A list of record tuples goes through make_feed_dic_new (the
        reference's route, :151); a PackedBatch (DataHandle/native_input.py) is already the arena."""
        from ..DataHandle.native_input import PackedBatch
        p = self.path
        if isinstance(batch_data, PackedBatch):
            bt = p.batch(batch_data.B)
            self._arena_taken(bt)
            lr_off = bt.offsets["lr"][0]
            batch_data.arena[lr_off:lr_off + 1].view(torch.float32)[0] = \
                float(np.float32(learning_rate)) if learning_rate is not None else 0.0
            bt.arena.copy_(batch_data.arena, non_blocking=True)
            return bt, batch_data.field("target_item_id")
        input_dic = self.embedding.make_feed_dic_new(batch_data=batch_data)
        self.embedding.validate_ids(input_dic)
        self._arena_taken(p.batch(len(input_dic[self.embedding.target_item_id])))
        return p.load_feed(input_dic, learning_rate), input_dic[self.embedding.target_item_id]

    @staticmethod
    def _arena_taken(bt):
        """Someone else's feed goes into ``bt.arena`` (an evaluation batch of the training batch's size, a step fed
        the ordinary way): a feed ring attached to it must put its next slot back before its next step.  The steps
        in between run WITHOUT the feed role -- they are not the ring's."""
        ring = getattr(bt, "feed_ring", None)
        if ring is not None:
            ring.primed, ring.taken = False, True

    RESIDENT_HEAD = 4         # slots copied (and waited for) before the epoch's first step; the rest lands behind it

    def prepare_resident_epoch(self, recordset, index, batch_size, learning_rates, packer):
        """Start packing an epoch (``len(learning_rates)`` full batches of ``batch_size`` records in the order
        ``index``) into one of two pinned staging buffers ON A WORKER THREAD and return at once; hand the result to
        ``load_resident_epoch(prepared=...)``.  A trainer calls this for epoch e + 1 right after loading epoch e, so
        the packing (host only: libmtam_host.so, no HIP call on the worker) runs beside the steps of epoch e."""
        import threading
        p = self.path
        n, B = len(learning_rates), int(batch_size)
        if n < 1 or len(index) < n * B:
            raise ValueError("resident epoch: %d batches of %d need %d records, got %d" % (n, B, n * B, len(index)))
        words = p.batch(B).arena.numel()
        j = self._resident_flip
        self._resident_flip ^= 1
        slot = self._resident_stages[j]
        if slot is None or tuple(slot[0].shape) != (n, words):        # (pinned memory: allocated on THIS thread)
            slot = self._resident_stages[j] = [torch.zeros((n, words), dtype=torch.int32).pin_memory(), None]
        elif slot[1] is not None:
            slot[1].synchronize()             # the copy that last read this buffer (two epochs ago) is through
        stage = slot[0]
        index = np.array(index[:n * B], dtype=np.int64)               # the caller may reshuffle its list meanwhile
        lrs = [float(np.float32(x)) for x in learning_rates]
        packer.layout_only(B)                 # (what the worker needs of the packer, built on this thread)
        prep = _ResidentPrep(j, stage, n, B, lrs)

        def work():
            try:
                for k in range(n):
                    packer.pack(recordset, index[k * B:(k + 1) * B], lr=lrs[k], into=stage[k])
            except Exception as e:                    # re-raised by load_resident_epoch
                prep.error = e

        prep.thread = threading.Thread(target=work, daemon=True)
        prep.thread.start()
        return prep

    def load_resident_epoch(self, recordset=None, index=None, batch_size=None, learning_rates=None, packer=None,
                            prepared=None):
        """Put an epoch's full batches (``len(learning_rates)`` of ``batch_size`` records, ``index`` = the epoch's
        record order; or an epoch ``prepare_resident_epoch`` has packed meanwhile) into the ring of feed arenas in HBM
        and return the ``ResidentBatch`` handles ``train()`` takes in order.  Step k's learning rate travels in its
        slot.  The optimizer launch of every step hands the next one its feed (Model/time_aware_path.py FeedRing): no
        copy per step (reference: a feed_dict per sess.run, :150-164).  The host -> device copy runs on a side stream
        in two pieces -- the first RESIDENT_HEAD slots, which the first steps wait for, and the rest, which lands
        behind them.  Single-GPU Adam steps through the graph only."""
        p = self.path
        if not self.use_graph:
            raise RuntimeError("resident epochs replay the captured step: use_graph is off")
        if p.allreduce_fn is not None or p.sharded is not None or p.sharded_scoring is not None:
            raise RuntimeError("resident epochs: single-GPU steps only (a data-parallel step is not one graph)")
        if prepared is None:
            prepared = self.prepare_resident_epoch(recordset, index, batch_size, learning_rates, packer)
        prepared.thread.join()
        if prepared.error is not None:
            raise prepared.error
        n, B, stage = prepared.n, prepared.B, prepared.stage
        bt = p.batch(B)
        ring = getattr(bt, "feed_ring", None)
        if ring is None or ring.n != n:
            if ring is not None:             # graphs captured on the old ring's addresses go with it
                torch.cuda.synchronize()
                self._graphs = {k: g for k, g in self._graphs.items()
                                if not (k[0] == "train_ring" and k[3] == ring.serial)}
            ring = p.feed_ring(bt, n)
        main = torch.cuda.current_stream()
        if self._resident_stream is None:
            self._resident_stream = torch.cuda.Stream()
        side = self._resident_stream
        # the previous epoch's last steps (queued on the main stream) still read the ring: the copy starts behind them
        behind = torch.cuda.Event()
        behind.record(main)
        head = min(n, self.RESIDENT_HEAD)
        ev_head, ev_rest = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.stream(side):
            side.wait_event(behind)
            ring.slots[:head].copy_(stage[:head], non_blocking=True)
            ev_head.record(side)
            if n > head:
                ring.slots[head:].copy_(stage[head:], non_blocking=True)
            ev_rest.record(side)
        self._resident_stages[prepared.j][1] = ev_rest
        main.wait_event(ev_head)
        ring.gate = (head, ev_rest) if n > head else None     # slots >= head: wait for ev_rest before the first use
        ring.primed = False
        targets = stage[:, bt.offsets["target_item_id"][0]:][:, :B]
        return [ResidentBatch(ring, k, B, prepared.lrs[k], targets[k]) for k in range(n)]

    def _train_resident(self, rb, learning_rate, global_step):
        ring, p = rb.ring, self.path
        bt = ring.bt
        if bt.feed_ring is not ring:
            raise RuntimeError("resident batch of a ring that is no longer attached")
        if abs(float(np.float32(learning_rate)) - rb.lr) > 1e-12:
            raise ValueError("resident batch %d was packed with learning rate %r, train() got %r"
                             % (rb.k, rb.lr, learning_rate))
        gate = getattr(ring, "gate", None)
        if gate is not None and rb.k + 1 >= gate[0]:          # this step's optimizer launch reads slot k + 1
            torch.cuda.current_stream().wait_event(gate[1])
            ring.gate = None
        if not ring.primed or ring.consumed != rb.k:
            # first step of the epoch; or the arena was used by someone else since; or a handle out of order (a step
            # that failed and was skipped, train_process's swallow_step_errors): the slot goes in by an ordinary copy
            ring.prime(rb.k)
        cur = self._loss_begin() if self.async_loss else -1
        host = self._loss_ring[cur][0] if cur >= 0 else None

        def fn(bt_):
            p.ring_train_kernels(bt_)
            if host is not None:
                host.copy_(bt_.loss, non_blocking=True)

        self._run("train_ring", bt, fn, key_extra=(ring.serial, cur))
        ring.consumed += 1
        if cur >= 0:
            return (bt,) + self._loss_end(cur, global_step, learning_rate)
        return bt, bt.loss.cpu().numpy(), global_step, learning_rate

    def train(self, sess, batch_data, learning_rate, add_summary=False, global_step=0, epoch=0):
        """One optimizer step on one batch -> (loss, summary) (reference :150-167).

        With ``async_loss`` the pair belongs to the PREVIOUS step (``summary["loss_step"]`` / ``self.loss_step`` say
        which ``global_step``); the first call returns ``(nan, {"loss_step": None})`` -- nothing has finished yet --
        and ``drain_loss()`` hands over the most recent step's once the loop ends (or before a checkpoint)."""
        if isinstance(batch_data, ResidentBatch):
            bt, loss, step, lr = self._train_resident(batch_data, learning_rate, global_step)
        elif self._feed_in_graph(batch_data):
            bt, loss, step, lr = self._train_feed_in_graph(batch_data, learning_rate, global_step)
        else:
            bt, _ = self._load(batch_data, learning_rate)
            self.step_train(bt)
            if self.async_loss:
                cur = self._loss_begin()
                self._loss_ring[cur][0].copy_(bt.loss, non_blocking=True)
                loss, step, lr = self._loss_end(cur, global_step, learning_rate)
            else:
                loss, step, lr = bt.loss.cpu().numpy(), global_step, learning_rate
        if loss is None:
            self.loss_step = None
            return float("nan"), {"loss_step": None}
        self.loss_step = step
        return float(loss[0]), self._summary(loss, lr, step)

    # ---- the feed copy (and the loss copy) INSIDE the step's hipGraph
    def _feed_in_graph(self, batch_data):
        """A PackedBatch sits in one of its stream's three pinned arenas (DataHandle/native_input.py): the host ->
        device copy of the feed can be the FIRST NODE of the captured step -- one graph per arena address -- and the
        device -> host copy of the loss its last (``async_loss``), so that a step is ONE hipGraphLaunch instead of
        copy + launch + copy back to back on the stream (profiles/r02_host_loop_cprofile.txt: ~25 us of copies and
        gaps per step).  Single-GPU steps only; MTAM_FEED_IN_GRAPH=0 keeps the copies outside."""
        from ..DataHandle.native_input import PackedBatch
        p = self.path
        return (self.use_graph and isinstance(batch_data, PackedBatch) and p.allreduce_fn is None and
                p.sharded is None and p.sharded_scoring is None and
                os.environ.get("MTAM_FEED_IN_GRAPH", "1") != "0")

    def _train_feed_in_graph(self, batch_data, learning_rate, global_step):
        p = self.path
        bt = p.batch(batch_data.B)
        self._arena_taken(bt)
        lr_off = bt.offsets["lr"][0]
        batch_data.arena[lr_off:lr_off + 1].view(torch.float32)[0] = float(np.float32(learning_rate))
        src = batch_data.arena
        # a captured graph reads this address at every replay: the arena must outlive the graph (a freed pinned block
        # could come back as someone else's memory under the same address)
        self._feed_refs[src.data_ptr()] = src
        cur = self._loss_begin() if self.async_loss else -1
        host = self._loss_ring[cur][0] if cur >= 0 else None

        def fn(bt_):
            bt_.arena.copy_(src, non_blocking=True)
            p.train_kernels(bt_)
            if host is not None:
                host.copy_(bt_.loss, non_blocking=True)

        self._run("train_feed", bt, fn, key_extra=(src.data_ptr(), cur))
        if cur >= 0:
            return (bt,) + self._loss_end(cur, global_step, learning_rate)
        return bt, bt.loss.cpu().numpy(), global_step, learning_rate

    @staticmethod
    def _summary(loss, learning_rate, step):
        return {"normalized Training Loss": float(loss[0]), "l2_norm": float(loss[1]),
                "Training Loss": float(loss[2]), "Learning_rate": float(learning_rate), "loss_step": step}

    # ---- the loss, one step late: N_LOSS_SLOTS pinned slots in rotation, one event each
    N_LOSS_SLOTS = 3          # as many as a batch stream has pinned arenas: (arena, slot) pairs repeat with period 3

    def _loss_begin(self):
        """The slot this step's loss goes to (the caller queues the device -> pinned-host copy behind the step)."""
        if self._loss_ring is None:
            self._loss_ring = [[torch.zeros(3).pin_memory(), torch.cuda.Event(), None, None]
                               for _ in range(self.N_LOSS_SLOTS)]
        return self._loss_slot

    def _loss_end(self, cur, global_step, learning_rate):
        """Mark slot ``cur`` as in flight and return the previous step's (loss[3], its global_step, its learning
        rate) -- or (None, None, None) when nothing is waiting (first call, or drained): the host never waits for
        the step it has just launched."""
        n = len(self._loss_ring)
        slot = self._loss_ring[cur]
        slot[1].record()
        slot[2], slot[3] = global_step, learning_rate
        self._loss_slot = (cur + 1) % n
        self._loss_unread = True
        prev = self._loss_ring[(cur - 1) % n]
        if prev[2] is None:
            return None, None, None
        prev[1].synchronize()
        out = (prev[0].numpy().copy(), prev[2], prev[3])
        prev[2] = None
        return out

    def _latest_slot(self):
        return self._loss_ring[(self._loss_slot - 1) % len(self._loss_ring)]

    def drain_loss(self):
        """(loss, summary) of the most recent step when train() has not handed it over yet (``async_loss``), else
        None; waits for that step.  The trainer calls it before it averages, evaluates, saves or ends an epoch."""
        if self._loss_ring is None or not self._loss_unread:
            return None
        slot = self._latest_slot()
        host, ev, step, lr = slot
        ev.synchronize()
        self._loss_unread = False
        slot[2] = None                         # the next train() call must not hand this one over again
        self.loss_step = step
        loss = host.numpy().copy()
        return float(loss[0]), self._summary(loss, lr, step)

    def last_loss(self):
        """The loss of the most recent step (waits for it)."""
        if self._loss_ring is None:
            return None
        host, ev = self._latest_slot()[:2]
        ev.synchronize()
        return float(host[0])

    def _current_table(self):
        """Data-parallel "sharded-table": bring this replica's item rows up to date (a collective: every rank calls it
        at the same point -- evaluation and checkpoints are replicated / rank-0 operations of the same loop)."""
        ex = getattr(self.path, "sharded_scoring", None)
        if ex is not None and not ex.replicate_table:
            ex.sync_item_table()

    def metrics_topK(self, sess, batch_data, global_step, topk):
        """hr/ndcg @ 1, 5, 10, 30, 50 over the full catalog (reference :188-213;
        the ``topk`` argument is accepted and ignored there too)."""
        self._current_table()
        p = self.path
        bt, result_item = self._load(batch_data)
        self._run("eval", bt, p.eval_kernels)
        top = bt.topk_idx.cpu().numpy()
        length = len(batch_data)
        out = []
        for k in (1, 5, 10, 30, 50):
            hr, ndcg = self.calculate_topK(k, top[:, :k], result_item, global_step, length)
            out += [hr, ndcg]
        self.predict_behavior_emb = bt.pred
        return tuple(out)

    def recall_at(self, sess, batch_data, k=20):
        """Recall@k per batch (BASELINE.json's metric; the reference never computes K=20)."""
        self._current_table()
        p = self.path
        bt, tgt = self._load(batch_data)
        self._run("eval", bt, p.eval_kernels)
        top = bt.topk_idx.cpu().numpy()[:, :k]
        return float((top == np.asarray(tgt)[:, None]).any(axis=1).mean())

    def calculate_topK(self, k, indices_result, result_item, global_step, length):
        total_count = 0
        recall_count = 0
        ndcg_value_list = []
        for one_user_data in indices_result:
            one_user_data = list(one_user_data)
            if result_item[total_count] in one_user_data:
                recall_count = recall_count + 1
                i = one_user_data.index(result_item[total_count])
                ndcg_value_list.append(math.log(2) / math.log(i + 2))
            total_count = total_count + 1
        recall_rate = recall_count / total_count
        avg_ndcg = float(sum(ndcg_value_list)) / length if len(ndcg_value_list) > 0 else 0
        return recall_rate, avg_ndcg
