"""util/tf_bundle.py: TensorFlow checkpoint bundles read and written without TensorFlow.

PARITY UNPINNED against a TF-written file (none exists here; TF 1.14 cannot run).  Pinned by: RFC 3720's CRC-32C
vectors; a table assembled by hand in this file from the LevelDB table / TensorBundle format descriptions (not by
the module's writer); corruption detection; the module's own round trip across block and restart boundaries."""
import os
import struct

import numpy as np
import pytest

from mtamrecommender_amd.util import tf_bundle as tb


def test_crc32c_rfc3720_vectors():
    for data, want in ((b"123456789", 0xE3069283), (bytes(32), 0x8A9136AA), (b"\xff" * 32, 0x62A8AB43),
                       (bytes(range(32)), 0x46DD794E), (bytes(range(31, -1, -1)), 0x113FDB5C)):
        assert tb._crc32c_python(data) == want
        assert tb.crc32c(data) == want                       # libmtam_host.so's slicing-by-8 when it is built
    rng = np.random.default_rng(0)
    blob = rng.integers(0, 256, 100003, dtype=np.uint8).tobytes()
    assert tb.crc32c(blob) == tb._crc32c_python(blob)
    assert tb.crc32c(blob[50:], tb.crc32c(blob[:50])) == tb.crc32c(blob)      # continuation
    # crc32c::Mask / Unmask (tensorflow/core/lib/hash/crc32c.h): rotate right by 15, add 0xa282ead8
    assert tb.mask_crc(0) == 0xA282EAD8 and tb.unmask_crc(tb.mask_crc(0xDEADBEEF)) == 0xDEADBEEF


def _varint(n):
    out = bytearray()
    while True:
        out.append((n & 0x7F) | (0x80 if n >> 7 else 0))
        n >>= 7
        if not n:
            return bytes(out)


def _block(entries):
    """One uncompressed table block, every entry its own restart point (shared = 0), then its trailer."""
    body, restarts = bytearray(), []
    for k, v in entries:
        restarts.append(len(body))
        body += _varint(0) + _varint(len(k)) + _varint(len(v)) + k + v
    body += b"".join(struct.pack("<I", r) for r in restarts) + struct.pack("<I", len(restarts))
    crc = tb._crc32c_python(bytes(body) + b"\x00")
    masked = (((crc >> 15) | (crc << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    return bytes(body), bytes(body) + b"\x00" + struct.pack("<I", masked)


def test_reads_a_bundle_assembled_by_hand(tmp_path):
    """Index file and data shard written here byte by byte from the format description: header entry under the empty
    key, one float32 [2, 3] tensor and one int64 scalar, one data block, index block, empty metaindex, footer."""
    a = np.arange(6, dtype="<f4").reshape(2, 3) * 0.5
    b = np.array(1234567890123, dtype="<i8")
    raw_a, raw_b = a.tobytes(), b.tobytes()
    with open(tmp_path / "m.ckpt-7.data-00000-of-00001", "wb") as f:
        f.write(raw_a + raw_b)

    def masked(data):
        c = tb._crc32c_python(data)
        return (((c >> 15) | (c << 17)) + 0xA282EAD8) & 0xFFFFFFFF
    # BundleEntryProto: 1 dtype (varint), 2 shape {2: dim {1: size}}, 4 offset, 5 size, 6 crc32c (fixed32)
    dim = lambda n: b"\x12" + _varint(len(b"\x08" + _varint(n))) + b"\x08" + _varint(n)
    shape_a = dim(2) + dim(3)
    entry_a = b"\x08\x01" + b"\x12" + _varint(len(shape_a)) + shape_a + b"\x28" + _varint(len(raw_a)) + \
        b"\x35" + struct.pack("<I", masked(raw_a))
    entry_b = b"\x08\x09" + b"\x12\x00" + b"\x20" + _varint(len(raw_a)) + b"\x28" + _varint(8) + \
        b"\x35" + struct.pack("<I", masked(raw_b))
    header = b"\x08\x01" + b"\x1a\x02\x08\x01"                 # num_shards = 1, version {producer: 1}
    data_contents, data_block = _block([(b"", header), (b"global_step", entry_b), (b"w/kernel", entry_a)])
    meta_contents, meta_block = _block([])
    handle = lambda off, size: _varint(off) + _varint(size)
    index_contents, index_block = _block([(b"w/kernel", handle(0, len(data_contents)))])
    off_meta = len(data_block)
    off_index = off_meta + len(meta_block)
    footer = handle(off_meta, len(meta_contents)) + handle(off_index, len(index_contents))
    footer += b"\x00" * (40 - len(footer)) + struct.pack("<Q", 0xDB4775248B80FB57)
    with open(tmp_path / "m.ckpt-7.index", "wb") as f:
        f.write(data_block + meta_block + index_block + footer)
    with open(tmp_path / "checkpoint", "w") as f:
        f.write('model_checkpoint_path: "m.ckpt-7"\nall_model_checkpoint_paths: "m.ckpt-7"\n')

    prefix = tb.latest_checkpoint(str(tmp_path))
    assert prefix == str(tmp_path / "m.ckpt-7")
    assert tb.list_bundle(prefix) == {"global_step": (np.int64, ()), "w/kernel": (np.float32, (2, 3))}
    got = tb.read_bundle(prefix)
    assert got["w/kernel"].dtype == np.float32 and np.array_equal(got["w/kernel"], a)
    assert got["global_step"].shape == () and int(got["global_step"]) == 1234567890123
    assert list(tb.read_bundle(prefix, names={"w/kernel"})) == ["w/kernel"]


def test_round_trip_across_blocks_and_restart_points(tmp_path):
    rng = np.random.default_rng(3)
    arrays = {"embedding_layer/item": rng.standard_normal((301, 128)).astype(np.float32),
              "beta1_power": np.float32(0.81), "global_step": np.array(17, np.int64),
              "flags": np.array([True, False, True]), "empty": np.zeros((0, 4), np.float32)}
    for i in range(40):                                   # > 16 entries with long common prefixes
        arrays["hidden/time_aware_gru_cell_decay_new/gates/kernel_%02d/Adam_1" % i] = \
            rng.standard_normal((i % 5 + 1, 3)).astype(np.float32 if i % 2 else np.float64)
    prefix = str(tmp_path / "sub" / "model.ckpt-17")
    names = tb.write_bundle(prefix, arrays)
    assert names == sorted(arrays, key=lambda s: s.encode())
    # a small block size forces several data blocks behind the index block
    items = tb.read_table(prefix + ".index")
    tb.write_table(prefix + ".index", items, block_size=300)
    assert len(items) == 46 and tb.read_table(prefix + ".index") == items
    got = tb.read_bundle(prefix)
    assert sorted(got) == sorted(arrays)
    for k, v in arrays.items():
        assert got[k].dtype == np.asarray(v).dtype and got[k].shape == np.asarray(v).shape
        assert np.array_equal(got[k], v)
    tb.write_checkpoint_state(os.path.dirname(prefix), prefix)
    assert tb.latest_checkpoint(os.path.dirname(prefix)) == prefix


def test_corruption_is_detected(tmp_path):
    prefix = str(tmp_path / "c.ckpt")
    tb.write_bundle(prefix, {"a": np.arange(100, dtype=np.float32), "b": np.ones(3, np.int32)})
    data = prefix + ".data-00000-of-00001"
    blob = bytearray(open(data, "rb").read())
    blob[40] ^= 0x10
    open(data, "wb").write(bytes(blob))
    with pytest.raises(ValueError, match="tensor checksum"):
        tb.read_bundle(prefix)
    assert np.array_equal(tb.read_bundle(prefix, names={"b"})["b"], np.ones(3, np.int32))   # the intact tensor reads
    idx = bytearray(open(prefix + ".index", "rb").read())
    idx[5] ^= 0x01
    open(prefix + ".index", "wb").write(bytes(idx))
    with pytest.raises(ValueError, match="block checksum"):
        tb.read_bundle(prefix)
    with pytest.raises(ValueError, match="magic"):
        open(prefix + ".index", "wb").write(bytes(idx[:-1]) + b"\x00")
        tb.read_bundle(prefix)
