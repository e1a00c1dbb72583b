"""Native batch feed: records held in libmtam_host.so, batches packed straight into pinned feed arenas.

Drop-in for the trainer's ``for step_i, batch in DataInput(data, batch_size)`` loop
(train_process.py:240,326): ``NativeDataInput`` yields ``(step_i, PackedBatch)`` with the same
slicing (sequential, non-overlapping, short final batch); ``model.train`` / ``model.metrics_topK``
accept a ``PackedBatch`` wherever they accept a list of record tuples.  Packing (pad to
length_of_user_history with zeros at the end, id range checks -- what
Embedding.make_feed_dic_new and TF's gather do, Embedding/Behavior_embedding_time_aware_attention.py:146-192)
runs in C++ on a worker thread one batch ahead of the device.
"""
import ctypes
import queue
import threading
from collections import deque

import numpy as np
import torch

from .. import _host_lib

ERR_LEN = 512


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


class RecordSet(object):
    """Records in structure-of-arrays form inside the native library."""

    def __init__(self, handle):
        self._lib = _host_lib.load()
        self._h = handle

    @classmethod
    def from_file(cls, path):
        lib = _host_lib.load()
        err = ctypes.create_string_buffer(ERR_LEN)
        h = lib.mtam_records_parse_file(str(path).encode(), err, ERR_LEN)
        if not h:
            raise ValueError("%s: %s" % (path, err.value.decode()))
        return cls(h)

    @classmethod
    def from_text(cls, text):
        lib = _host_lib.load()
        raw = text.encode()
        err = ctypes.create_string_buffer(ERR_LEN)
        h = lib.mtam_records_parse_text(raw, len(raw), err, ERR_LEN)
        if not h:
            raise ValueError(err.value.decode())
        return cls(h)

    @classmethod
    def from_records(cls, records):
        """From the reference's in-memory form: a list of 9-tuples."""
        lib = _host_lib.load()
        n = len(records)
        lens = np.fromiter((len(r[1]) for r in records), dtype=np.int64, count=n)
        for r in records:
            if not (len(r[1]) == len(r[2]) == len(r[3]) == len(r[4]) == len(r[5]) == len(r[6])):
                raise ValueError("the six lists of a record must have one length")
        offsets = np.zeros(n + 1, np.int64)
        np.cumsum(lens, out=offsets[1:])
        cat = lambda k, dt: np.ascontiguousarray(
            np.concatenate([np.asarray(r[k], dtype=dt) for r in records]) if n else np.zeros(0, dt), dtype=dt)
        item, category, position = cat(1, np.int32), cat(2, np.int32), cat(6, np.int32)
        time, timelast, timenow = cat(3, np.float32), cat(4, np.float32), cat(5, np.float32)
        col = lambda f, dt: np.ascontiguousarray(np.fromiter((f(r) for r in records), dtype=dt, count=n))
        user = col(lambda r: r[0], np.int32)
        tid, tcat = col(lambda r: r[7][0], np.int32), col(lambda r: r[7][1], np.int32)
        ttime = col(lambda r: r[7][2], np.float32)
        length = col(lambda r: r[8], np.int32)
        err = ctypes.create_string_buffer(ERR_LEN)
        h = lib.mtam_records_from_arrays(n, _ptr(offsets), _ptr(user), _ptr(item), _ptr(category), _ptr(time),
                                         _ptr(timelast), _ptr(timenow), _ptr(position), _ptr(tid), _ptr(tcat),
                                         _ptr(ttime), _ptr(length), err, ERR_LEN)
        if not h:
            raise ValueError(err.value.decode())
        return cls(h)

    def __len__(self):
        return int(self._lib.mtam_records_count(self._h))

    @property
    def max_length(self):
        return int(self._lib.mtam_records_max_length(self._h))

    def record(self, i):
        """Record i back as the reference's 9-tuple (times as floats)."""
        cap = max(1, self.max_length)
        ii = lambda: np.zeros(cap, np.int32)
        ff = lambda: np.zeros(cap, np.float32)
        item, category, position, time, timelast, timenow = ii(), ii(), ii(), ff(), ff(), ff()
        u, tid, tcat, length = (ctypes.c_int32() for _ in range(4))
        ttime = ctypes.c_float()
        n = self._lib.mtam_records_get(self._h, i, cap, ctypes.byref(u), _ptr(item), _ptr(category), _ptr(time),
                                       _ptr(timelast), _ptr(timenow), _ptr(position), ctypes.byref(tid),
                                       ctypes.byref(tcat), ctypes.byref(ttime), ctypes.byref(length))
        if n < 0:
            raise IndexError(i)
        return (u.value, item[:n].tolist(), category[:n].tolist(), time[:n].tolist(), timelast[:n].tolist(),
                timenow[:n].tolist(), position[:n].tolist(), [tid.value, tcat.value, ttime.value], length.value)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            self._lib.mtam_records_free(h)


def shuffled_index(n, seed):
    idx = np.zeros(n, np.int64)
    _host_lib.load().mtam_shuffle_index(_ptr(idx), n, seed & 0xFFFFFFFFFFFFFFFF)
    return idx


class PackedBatch(object):
    """One batch already laid out as the device feed arena (pinned host memory).  The arena belongs to a
    rotating pool of 3 buffers owned by ONE consumer (see BatchPacker.pack): consume the batch before two
    further ones of the same consumer are packed."""

    def __init__(self, arena, B, index, recordset, layout):
        self.arena, self.B, self.index, self.recordset, self.layout = arena, B, index, recordset, layout

    def __len__(self):
        return self.B

    def field(self, name):
        """numpy view of one feed field of the packed arena (e.g. 'target_item_id')."""
        o, n, shape, dt = self.layout[name]
        v = self.arena[o:o + n]
        return (v.view(torch.float32) if dt == torch.float32 else v).view(*shape).numpy()

    def records(self):
        return [self.recordset.record(int(i)) for i in self.index]


class BatchPacker(object):
    """Packs batches of a RecordSet for one model: owns the arena layout and the table row limits."""

    def __init__(self, path, embedding, n_buffers=3):
        """``path``: the model's TimeAwarePath, or just length_of_user_history (host-only use)."""
        self._lib = _host_lib.load()
        self.L = path if isinstance(path, int) else path.L
        self.rows = _host_lib.TableRows(embedding.item_count + 3, embedding.category_count + 3,
                                        embedding.position_count + 3, embedding.user_count + 3)
        self.n_buffers = n_buffers
        self._layouts = {}
        self._pools = {}
        self._lock = threading.Lock()

    def _layout(self, B, consumer=None):
        """(arena layout, field offsets, the rotating arena pool of ``consumer`` for batch size B).
        A pool belongs to ONE consumer (one batch stream): the train stream's prefetched batch must not sit
        in a buffer that an evaluation pass of the same batch size rotates through (round-1 advice: a shared
        pool let the third test batch overwrite the pending train batch when the batch sizes were equal)."""
        lay, offsets = self.layout_only(B)
        with self._lock:
            if (B, consumer) not in self._pools:
                pin = torch.cuda.is_available()
                self._pools[(B, consumer)] = deque(
                    (torch.zeros(lay.words, dtype=torch.int32).pin_memory() if pin
                     else torch.zeros(lay.words, dtype=torch.int32)) for _ in range(self.n_buffers))
            return lay, offsets, self._pools[(B, consumer)]

    def layout_only(self, B):
        """(C layout struct, field offsets) of batch size B without touching any arena pool."""
        with self._lock:
            if B not in self._layouts:
                from ..Model.time_aware_path import arena_layout
                offsets, words = arena_layout(B, self.L)
                lay = _host_lib.ArenaLayout()
                for k in ("user_id", "item_list", "category_list", "position_list", "target_item_id", "seq_length",
                          "time_list", "timelast_list", "target_item_time", "lr", "timenow_list"):
                    setattr(lay, k, offsets[k][0])
                lay.words = words
                self._layouts[B] = (lay, offsets)
            return self._layouts[B]

    def pack(self, recordset, index, lr=0.0, consumer=None, into=None):
        """``consumer``: name of the batch stream this batch belongs to (None = the packer's own default
        pool).  Every stream rotates through its own 3 arenas.  ``into``: a caller-owned int32 host tensor of the
        layout's size instead of a pool arena (resident epochs: the rows of one staging buffer)."""
        index = np.ascontiguousarray(index, dtype=np.int64)
        B = len(index)
        if into is not None:         # no pool: nothing is allocated (a worker thread must not make HIP calls)
            lay, offsets = self.layout_only(B)
        else:
            lay, offsets, pool = self._layout(B, consumer)
        if into is not None:
            if into.dtype != torch.int32 or into.numel() != lay.words or not into.is_contiguous() or into.is_cuda:
                raise ValueError("pack(into=...): a contiguous int32 host tensor of %d words" % lay.words)
            arena = into
        else:
            with self._lock:
                arena = pool[0]
                pool.rotate(-1)
        err = ctypes.create_string_buffer(ERR_LEN)
        rc = self._lib.mtam_pack_batch(recordset._h, _ptr(index), B, self.L, ctypes.byref(lay),
                                       ctypes.byref(self.rows), float(lr), ctypes.c_void_p(arena.data_ptr()), err,
                                       ERR_LEN)
        if rc == -4:
            raise IndexError(err.value.decode())
        if rc != 0:
            raise ValueError(err.value.decode())
        return PackedBatch(arena, B, index, recordset, offsets)


class NativeDataInput(object):
    """``DataInput`` over a RecordSet: yields (step_i, PackedBatch), packing one batch ahead on a thread.
    ``index`` (optional) is the epoch's record order, e.g. ``shuffled_index(len(rs), seed)``."""

    _streams = 0

    def __init__(self, recordset, batch_size, packer, index=None, prefetch=True, consumer=None, shard=None):
        """``consumer``: name of this batch stream's arena pool inside the packer.  Default: a pool of its
        own per iterator; a trainer that builds one iterator per epoch passes a fixed name ("train", "eval")
        so that the pinned arenas are allocated once."""
        self.rs, self.batch_size, self.packer = recordset, int(batch_size), packer
        # shard = (rank, world): data parallel -- this iterator packs only its rank's contiguous slice of every
        # global batch (data_parallel.shard); ``batch.global_size`` is the size of the whole batch
        self.shard = shard
        if consumer is None:
            NativeDataInput._streams += 1
            consumer = "stream%d" % NativeDataInput._streams
        self.consumer = consumer
        n = len(recordset)
        self.index = np.arange(n, dtype=np.int64) if index is None else np.ascontiguousarray(index, np.int64)
        n = len(self.index)
        self.epoch_size = (n + self.batch_size - 1) // self.batch_size
        self.i = 0
        self.prefetch = prefetch
        self._pending = False
        self._worker = None
        self._ahead = None
        # Pinned arenas are allocated HERE, on the caller's thread: the worker thread must not make HIP
        # calls (hipHostMalloc while the main thread captures a hipGraph invalidates the capture).
        sizes = {min(self.batch_size, n), n % self.batch_size}
        if shard is not None:
            rank, world = shard
            # A last global batch smaller than the world size would leave some ranks with NO samples: they would
            # skip (or fail) the step while the others wait inside the gradient exchange.  The decision is taken
            # on the GLOBAL batch, so every rank takes the same branch: such a batch is dropped everywhere
            # (data_parallel.keep_global_batch; at most world - 1 records per epoch).
            tail = n % self.batch_size
            if 0 < tail < world:
                self.epoch_size -= 1
                sizes.discard(tail)
            sizes = {(g * (rank + 1)) // world - (g * rank) // world for g in sizes}
        for size in sizes:
            if size > 0:
                packer._layout(size, self.consumer)

    def __iter__(self):
        return self

    def _global_slice(self, i):
        return self.index[i * self.batch_size:min((i + 1) * self.batch_size, len(self.index))]

    def _slice(self, i):
        g = self._global_slice(i)
        if self.shard is None:
            return g
        rank, world = self.shard
        return g[(len(g) * rank) // world:(len(g) * (rank + 1)) // world]

    # One worker thread per iterator, fed batch numbers through a queue (a thread per batch cost the consumer
    # ~90 us a step: creation plus the start handshake).  It packs exactly one batch ahead of the consumer -- the
    # pool's three arenas are the batch in use, the one packed ahead and the one the previous H2D copy may still read.
    def _work(self):
        while True:
            i = self._req.get()
            if i is None:
                return
            try:
                self._res.put(("ok", self.packer.pack(self.rs, self._slice(i), consumer=self.consumer)))
            except Exception as e:                     # re-raised on the consumer side
                self._res.put(("error", e))

    def _request(self, i):
        if self._worker is None:
            self._req, self._res = queue.Queue(), queue.Queue()
            self._worker = threading.Thread(target=self._work, daemon=True)
            self._worker.start()
        self._req.put(i)

    def peek_prefetched(self):
        """The batch packed ahead (waits for it; None when there is none).  It stays the next batch of the iterator."""
        if self._pending and self._ahead is None:
            self._ahead = self._res.get()
        return self._ahead[1] if self._ahead is not None and self._ahead[0] == "ok" else None

    def close(self):
        if self._worker is not None:
            self._req.put(None)
            self._worker = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __next__(self):
        if self.i == self.epoch_size:
            self.close()
            raise StopIteration
        if not self.prefetch:
            batch = self.packer.pack(self.rs, self._slice(self.i), consumer=self.consumer)
        else:
            if not self._pending:
                self._request(self.i)
            kind, batch = self._ahead if self._ahead is not None else self._res.get()
            self._ahead = None
            self._pending = self.i + 1 < self.epoch_size
            if self._pending:
                self._request(self.i + 1)
            if kind == "error":
                self.close()
                raise batch
        batch.global_size = len(self._global_slice(self.i))
        self.i += 1
        return self.i, batch

    next = __next__
