"""Reader and writer for TensorFlow's checkpoint "tensor bundle" files (what tf.train.Saver(write_version=V2)
leaves under the reference's ``checkpoint_path_dir``, Model/base_model.py:331-343): ``<prefix>.index`` +
``<prefix>.data-00000-of-00001`` + the ``checkpoint`` state file.  Pure Python / numpy: TensorFlow is not needed.

PARITY UNPINNED: no TensorFlow-written bundle is available in this environment (the reference ships none and TF 1.14
cannot run here).  The code follows the published formats --
  * the index is a LevelDB-format table (tensorflow/core/lib/io/{table_builder,block_builder,format}.cc): data
    blocks of prefix-compressed (key, value) entries with restart points, each followed by a 1-byte compression
    type and a masked CRC-32C; an index block of (separator key -> block handle); a 48-byte footer ending in the
    magic 0xdb4775248b80fb57.  BundleWriter writes it uncompressed (tensor_bundle.cc); a snappy block raises;
  * key "" holds a BundleHeaderProto, every other key a BundleEntryProto (tensor_bundle.proto): dtype, shape,
    shard, offset, size, masked CRC-32C of the tensor bytes;
  * tensors lie back to back in the data shard(s), little endian
-- and is tested against RFC 3720's CRC-32C vectors, a table assembled by hand from that description, and its own
round trip (tests/test_tf_bundle.py)."""
import os
import re
import struct

import numpy as np

TABLE_MAGIC = 0xDB4775248B80FB57
FOOTER_LEN = 48
MASK_DELTA = 0xA282EAD8

# tensorflow/core/framework/types.proto
DTYPES = {1: np.float32, 2: np.float64, 3: np.int32, 4: np.uint8, 5: np.int16, 6: np.int8, 9: np.int64, 10: np.bool_,
          17: np.uint16, 19: np.float16, 22: np.uint32, 23: np.uint64}
DT_OF = {np.dtype(v): k for k, v in DTYPES.items()}

_CRC_TABLE = None


def _crc32c_python(data, crc=0):
    global _CRC_TABLE
    if _CRC_TABLE is None:
        t = np.arange(256, dtype=np.uint32)
        for _ in range(8):
            t = np.where(t & 1, (t >> 1) ^ np.uint32(0x82F63B78), t >> 1).astype(np.uint32)
        _CRC_TABLE = t.tolist()
    tl = _CRC_TABLE
    c = (crc ^ 0xFFFFFFFF) & 0xFFFFFFFF
    for b in bytes(data):
        c = tl[(c ^ b) & 0xFF] ^ (c >> 8)
    return c ^ 0xFFFFFFFF


def crc32c(data, crc=0):
    """CRC-32C (Castagnoli, reflected polynomial 0x82F63B78), as tensorflow/core/lib/hash/crc32c.h.  Through
    libmtam_host.so's mtam_crc32c when the host library is built (catalog-sized tensors), else a byte loop."""
    data = bytes(data) if not isinstance(data, bytes) else data
    try:
        from .. import _host_lib
        lib = _host_lib.load()
    except Exception:
        return _crc32c_python(data, crc)
    return int(lib.mtam_crc32c(data, len(data), crc))


def mask_crc(crc):
    """crc32c::Mask: stored CRCs are rotated and offset so that a CRC of data that contains CRCs stays sound."""
    return (((crc >> 15) | (crc << 17)) + MASK_DELTA) & 0xFFFFFFFF


def unmask_crc(masked):
    rot = (masked - MASK_DELTA) & 0xFFFFFFFF
    return ((rot >> 17) | (rot << 15)) & 0xFFFFFFFF


# ------------------------------------------------------------------ varints and the few protobuf messages
def _put_varint(n):
    n &= (1 << 64) - 1
    out = bytearray()
    while True:
        b = n & 0x7F
        n >>= 7
        out.append(b | (0x80 if n else 0))
        if not n:
            return bytes(out)


def _get_varint(buf, pos):
    shift = result = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7
        if shift > 63:
            raise ValueError("varint too long")


def _fields(buf):
    """(field number, wire type, value) of a serialized protobuf message; value: int or bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        tag, pos = _get_varint(buf, pos)
        num, wt = tag >> 3, tag & 7
        if wt == 0:
            v, pos = _get_varint(buf, pos)
        elif wt == 1:
            v, pos = struct.unpack_from("<Q", buf, pos)[0], pos + 8
        elif wt == 2:
            ln, pos = _get_varint(buf, pos)
            v, pos = bytes(buf[pos:pos + ln]), pos + ln
        elif wt == 5:
            v, pos = struct.unpack_from("<I", buf, pos)[0], pos + 4
        else:
            raise ValueError("unsupported protobuf wire type %d" % wt)
        yield num, wt, v


def _signed64(v):
    return v - (1 << 64) if v >= 1 << 63 else v


def _parse_shape(buf):
    dims = []
    for num, _, v in _fields(buf):
        if num == 2:                                   # Dim
            size = 0
            for n2, _, v2 in _fields(v):
                if n2 == 1:
                    size = _signed64(v2)
            dims.append(size)
        elif num == 3 and v:
            raise ValueError("tensor of unknown rank in the bundle")
    return tuple(dims)


def _parse_entry(buf):
    e = dict(dtype=0, shape=(), shard_id=0, offset=0, size=0, crc32c=None, sliced=False)
    for num, _, v in _fields(buf):
        if num == 1:
            e["dtype"] = v
        elif num == 2:
            e["shape"] = _parse_shape(v)
        elif num == 3:
            e["shard_id"] = v
        elif num == 4:
            e["offset"] = _signed64(v)
        elif num == 5:
            e["size"] = _signed64(v)
        elif num == 6:
            e["crc32c"] = v
        elif num == 7:
            e["sliced"] = True
    return e


def _parse_header(buf):
    h = dict(num_shards=1, endianness=0, producer=0)
    for num, _, v in _fields(buf):
        if num == 1:
            h["num_shards"] = v
        elif num == 2:
            h["endianness"] = v
        elif num == 3:
            for n2, _, v2 in _fields(v):
                if n2 == 1:
                    h["producer"] = v2
    return h


def _msg(*parts):
    return b"".join(parts)


def _f_varint(num, v):
    return _put_varint(num << 3) + _put_varint(v)


def _f_bytes(num, b):
    return _put_varint((num << 3) | 2) + _put_varint(len(b)) + b


def _f_fixed32(num, v):
    return _put_varint((num << 3) | 5) + struct.pack("<I", v)


def _entry_bytes(dtype, shape, offset, size, crc_masked):
    shape_msg = _msg(*[_f_bytes(2, _f_varint(1, d)) for d in shape])
    parts = [_f_varint(1, dtype), _f_bytes(2, shape_msg)]
    if offset:
        parts.append(_f_varint(4, offset))
    parts += [_f_varint(5, size), _f_fixed32(6, crc_masked)]
    return _msg(*parts)


# ------------------------------------------------------------------ the LevelDB-format table
def _read_block(buf, offset, size, verify=True):
    """Entries of the block at (offset, size): list of (key bytes, value bytes)."""
    contents, trailer = buf[offset:offset + size], buf[offset + size:offset + size + 5]
    if len(trailer) != 5:
        raise ValueError("truncated table block")
    if verify:
        want = unmask_crc(struct.unpack("<I", trailer[1:])[0])
        if crc32c(bytes(contents) + bytes(trailer[:1])) != want:
            raise ValueError("table block checksum mismatch")
    if trailer[0] == 1:
        raise NotImplementedError("snappy-compressed table block (TensorFlow's BundleWriter writes none)")
    if trailer[0] != 0:
        raise ValueError("unknown block compression type %d" % trailer[0])
    n_restarts = struct.unpack_from("<I", contents, len(contents) - 4)[0]
    limit = len(contents) - 4 - 4 * n_restarts
    out, pos, key = [], 0, b""
    while pos < limit:
        shared, pos = _get_varint(contents, pos)
        non_shared, pos = _get_varint(contents, pos)
        vlen, pos = _get_varint(contents, pos)
        key = key[:shared] + bytes(contents[pos:pos + non_shared])
        pos += non_shared
        out.append((key, bytes(contents[pos:pos + vlen])))
        pos += vlen
    return out


def read_table(path, verify=True):
    """All (key, value) pairs of a LevelDB-format table file, in key order."""
    with open(path, "rb") as f:
        buf = f.read()
    if len(buf) < FOOTER_LEN or struct.unpack("<Q", buf[-8:])[0] != TABLE_MAGIC:
        raise ValueError("%s is not a table file (bad magic)" % path)
    footer = buf[-FOOTER_LEN:]
    _, pos = _get_varint(footer, 0)            # metaindex handle: offset
    _, pos = _get_varint(footer, pos)          # metaindex handle: size
    ioff, pos = _get_varint(footer, pos)
    isize, pos = _get_varint(footer, pos)
    out = []
    for _, handle in _read_block(buf, ioff, isize, verify):
        off, p = _get_varint(handle, 0)
        size, _ = _get_varint(handle, p)
        out += _read_block(buf, off, size, verify)
    return out


class _BlockBuilder(object):
    def __init__(self, restart_interval=16):
        self.buf, self.restarts, self.count, self.last, self.interval = bytearray(), [0], 0, b"", restart_interval

    def add(self, key, value):
        shared = 0
        if self.count < self.interval:
            n = min(len(self.last), len(key))
            while shared < n and self.last[shared] == key[shared]:
                shared += 1
        else:
            self.restarts.append(len(self.buf))
            self.count = 0
        self.buf += _put_varint(shared) + _put_varint(len(key) - shared) + _put_varint(len(value))
        self.buf += key[shared:] + value
        self.last, self.count = key, self.count + 1

    def finish(self):
        return bytes(self.buf) + b"".join(struct.pack("<I", r) for r in self.restarts) + \
            struct.pack("<I", len(self.restarts))

    def size(self):
        return len(self.buf) + 4 * len(self.restarts) + 4

    def empty(self):
        return not self.buf


def write_table(path, items, block_size=262144):
    """``items``: (key bytes, value bytes) in strictly increasing key order -> an uncompressed table file."""
    out = bytearray()
    index = _BlockBuilder(restart_interval=1)
    block, last_key = _BlockBuilder(), None

    def flush():
        nonlocal block
        if block.empty():
            return
        contents = block.finish()
        off = len(out)
        out.extend(contents + b"\x00" + struct.pack("<I", mask_crc(crc32c(contents + b"\x00"))))
        index.add(last_key, _put_varint(off) + _put_varint(len(contents)))      # (the last key is a valid separator)
        block = _BlockBuilder()

    prev = None
    for key, value in items:
        if prev is not None and key <= prev:
            raise ValueError("table keys must be strictly increasing")
        block.add(key, value)
        prev = last_key = key
        if block.size() >= block_size:
            flush()
    flush()
    handles = []
    for contents in (_BlockBuilder().finish(), index.finish()):                  # metaindex (empty), index
        handles.append((len(out), len(contents)))
        out.extend(contents + b"\x00" + struct.pack("<I", mask_crc(crc32c(contents + b"\x00"))))
    footer = b"".join(_put_varint(v) for h in handles for v in h)
    out.extend(footer + b"\x00" * (FOOTER_LEN - 8 - len(footer)) + struct.pack("<Q", TABLE_MAGIC))
    with open(path, "wb") as f:
        f.write(bytes(out))


# ------------------------------------------------------------------ the bundle
def _shard_path(prefix, shard, num_shards):
    return "%s.data-%05d-of-%05d" % (prefix, shard, num_shards)


def list_bundle(prefix):
    """{tensor name: (numpy dtype, shape)} without reading the data shards."""
    out = {}
    for key, value in read_table(prefix + ".index"):
        if key:
            e = _parse_entry(value)
            out[key.decode()] = (DTYPES.get(e["dtype"]), e["shape"])
    return out


def read_bundle(prefix, names=None, verify=True):
    """{tensor name: numpy array} of the checkpoint ``prefix`` (e.g. ``.../model.ckpt-1200``).
    ``names``: only these.  ``verify``: check the table's block checksums and every tensor's CRC-32C."""
    items = read_table(prefix + ".index", verify)
    if not items or items[0][0] != b"":
        raise ValueError("%s.index has no bundle header entry" % prefix)
    header = _parse_header(items[0][1])
    if header["endianness"] != 0:
        raise NotImplementedError("big-endian tensor bundle")
    shards, out = {}, {}
    try:
        for key, value in items[1:]:
            name = key.decode()
            if names is not None and name not in names:
                continue
            e = _parse_entry(value)
            if e["sliced"]:
                raise NotImplementedError("%s is a partitioned variable (tensor slices); not supported" % name)
            if e["dtype"] not in DTYPES:
                raise NotImplementedError("%s: tensor dtype enum %d is not supported" % (name, e["dtype"]))
            if e["shard_id"] not in shards:
                shards[e["shard_id"]] = open(_shard_path(prefix, e["shard_id"], header["num_shards"]), "rb")
            f = shards[e["shard_id"]]
            f.seek(e["offset"])
            raw = f.read(e["size"])
            dt = np.dtype(DTYPES[e["dtype"]])
            count = int(np.prod(e["shape"], dtype=np.int64)) if e["shape"] else 1
            if len(raw) != e["size"] or count * dt.itemsize != e["size"]:
                raise ValueError("%s: %d bytes on disk for shape %s of %s" % (name, len(raw), e["shape"], dt))
            if verify and e["crc32c"] is not None and crc32c(raw) != unmask_crc(e["crc32c"]):
                raise ValueError("%s: tensor checksum mismatch" % name)
            out[name] = np.frombuffer(raw, dtype=dt.newbyteorder("<")).astype(dt).reshape(e["shape"])
    finally:
        for f in shards.values():
            f.close()
    return out


def write_bundle(prefix, arrays):
    """{name: array} -> ``<prefix>.index`` + ``<prefix>.data-00000-of-00001`` (one shard, as a single-device Saver
    writes).  Returns the sorted names."""
    os.makedirs(os.path.dirname(os.path.abspath(prefix)), exist_ok=True)
    names = sorted(arrays, key=lambda s: s.encode())
    # BundleHeaderProto: num_shards = 1, endianness LITTLE (default, omitted), version {producer: 1}
    items = [(b"", _msg(_f_varint(1, 1), _f_bytes(3, _f_varint(1, 1))))]
    offset = 0
    with open(_shard_path(prefix, 0, 1), "wb") as f:
        for name in names:
            a = np.asarray(arrays[name])                 # (a 0-d array stays a scalar tensor: shape ())
            if a.dtype not in DT_OF:
                raise TypeError("%s: dtype %s has no TensorFlow counterpart here" % (name, a.dtype))
            raw = a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes(order="C")
            f.write(raw)
            items.append((name.encode(), _entry_bytes(DT_OF[a.dtype], a.shape, offset, len(raw),
                                                      mask_crc(crc32c(raw)))))
            offset += len(raw)
    write_table(prefix + ".index", items)
    return names


def latest_checkpoint(directory):
    """The prefix named by ``model_checkpoint_path`` in ``<directory>/checkpoint`` (tf.train.latest_checkpoint), or
    the newest ``*.index`` file when there is no state file; None when the directory holds no bundle."""
    state = os.path.join(directory, "checkpoint")
    if os.path.exists(state):
        with open(state) as f:
            m = re.search(r'^model_checkpoint_path:\s*"(.*)"\s*$', f.read(), re.M)
        if m:
            p = m.group(1)
            p = p if os.path.isabs(p) else os.path.join(directory, p)
            if os.path.exists(p + ".index"):
                return p
    found = [os.path.join(directory, n[:-6]) for n in os.listdir(directory) if n.endswith(".index")] \
        if os.path.isdir(directory) else []
    return max(found, key=lambda p: os.path.getmtime(p + ".index")) if found else None


def write_checkpoint_state(directory, prefix):
    """The ``checkpoint`` text file a Saver keeps beside its bundles."""
    rel = os.path.basename(prefix)
    with open(os.path.join(directory, "checkpoint"), "w") as f:
        f.write('model_checkpoint_path: "%s"\nall_model_checkpoint_paths: "%s"\n' % (rel, rel))
