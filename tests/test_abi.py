"""CPU: libsglk.so loads and exports every entry point that include/sglk.h declares; the ctypes table of the
host layer covers the same set; host-only calls (no GPU needed) behave."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "sglk.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"SGLK_API\s+[\w\s\*]+?\b(sglk_\w+)\s*\(", text)))


def test_header_declares_entry_points():
    syms = declared_symbols()
    assert len(syms) > 50
    for must in ("sglk_decode_attention", "sglk_awq_gemm", "sglk_radix_match_prefix", "sglk_extend_attention",
                 "sglk_fp8_scaled_mm", "sglk_ar_all_reduce"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    from sglang_security_amd import _lib

    assert os.path.exists(_lib.LIB_PATH), "libsglk.so has not been built (python -m sglang_security_amd.build)"
    L = ctypes.CDLL(_lib.LIB_PATH)
    missing = [s for s in declared_symbols() if not hasattr(L, s)]
    assert not missing, f"declared in sglk.h but not exported: {missing}"


def test_ctypes_table_matches_header():
    from sglang_security_amd import _lib

    assert set(_lib._SIGS) == set(declared_symbols())
    assert not _lib._PENDING


def test_host_only_calls():
    from sglang_security_amd import _lib

    L = _lib.lib()
    assert L.sglk_version() > 0
    assert L.sglk_ar_meta_size() > 0 and L.sglk_ipc_handle_bytes() == 64
    assert L.sglk_awq_gemm_workspace_bytes(64, 4096, 4096) > 0
    # counters + error word + tail tickets (rounded to 256), partials + log-sum-exps of 8 prefix splits and 2 x 4
    # remainder splits, one split count per request (rounded to 256)
    assert L.sglk_decode_shared_workspace_bytes(64, 32, 128, 8) == ((2 * 64 * 32 + 2) * 4 + 255) // 256 * 256 + \
        64 * 32 * (8 + 8) * 129 * 4 + 256
    # argument validation happens before any GPU call
    rc = L.sglk_rmsnorm(None, None, None, 1, 128, 128, 128, 1e-6, 0, 2, None)
    assert rc == -1 and b"null" in L.sglk_last_error().lower()
    with pytest.raises(_lib.SglkError):
        _lib.check(rc, "rmsnorm")


def test_missing_library_fails_loudly(monkeypatch):
    from sglang_security_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsglk.so")
    with pytest.raises(ImportError, match="no CPU fallback"):
        _lib.lib()
