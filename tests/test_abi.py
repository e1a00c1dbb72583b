"""The C-ABI shared library loads and exports every symbol include/mtam_hip.h declares (no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "mtam_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(mtam_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_table_agree():
    from mtamrecommender_amd import _lib
    assert declared_symbols() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol(hip_lib):
    from mtamrecommender_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    assert hip_lib.mtam_arch() == b"gfx950"
    assert hip_lib.mtam_version() >= 1


def test_size_queries_need_no_gpu(hip_lib):
    assert hip_lib.mtam_emb_gather_partials(128, 50) % 4 == 0
    assert hip_lib.mtam_emb_scatter_partials(128, 50) % 4 == 0
    assert hip_lib.mtam_ta_attn_decode_save_floats(50, 1) == 3 * 128 + 3 * 50 + 2 * 50 + 1
    assert hip_lib.mtam_softmax_ce_partials(128, 3709) == 128 * 1 * 2
    assert hip_lib.mtam_sqnorm_blocks(4097) == 2


def test_argument_errors_are_reported_not_launched(hip_lib):
    """Bad arguments return MTAM_E_ARG with a message before anything touches a device."""
    rc = hip_lib.mtam_gemm_f32(0, 0, 0, 8, 8, None, 8, None, 8, None, 8, 0, None, None, None, 0, 1, None)
    assert rc == -1 and b"positive" in hip_lib.mtam_last_error()
    rc = hip_lib.mtam_topk(None, 8, 1, 8, 100, None, None, None)
    assert rc == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from mtamrecommender_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.MtamHipError):
        _lib.load()
