"""The C-ABI library loads and exports what include/fx.h declares (no GPU, no compute)."""
import ctypes
import re
from pathlib import Path

from conftest import ROOT, pkg


def declared_symbols():
    text = (ROOT / "include" / "fx.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(fx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    lib = pkg("_lib")
    handle = lib.load()
    names = declared_symbols()
    assert len(names) >= 11
    for n in names:
        assert hasattr(handle, n), f"libfx.so lacks {n}"
    assert sorted(lib.SIGNATURES) == names, "the ctypes table and the header disagree"
    assert handle.fx_abi_version() == 2


def test_argument_checks_run_without_a_device():
    """Validation happens before any HIP call, so it can be exercised here."""
    lib = pkg("_lib")
    h = lib.load()
    assert h.fx_hu_batch_u8(None, 1, 16, 16, 16, 256, -1, 255, None, None, None, 0, None, None, None) == -22
    assert b"d_img is NULL" in h.fx_last_error()
    assert h.fx_hu_batch_u8(ctypes.c_void_p(256), 1, 0, 16, 16, 256, -1, 255, None, None, ctypes.c_void_p(256), 0, None, None, None) == -22
    assert h.fx_pca_batch_f32(ctypes.c_void_p(256), None, 1, 10, 4, 8, 0, ctypes.c_void_p(256), None, None) == -22
    assert b"dims" in h.fx_last_error()
    assert h.fx_pca_batch_f32(ctypes.c_void_p(256), None, 1, 10, 3, 99, 0, ctypes.c_void_p(256), None, None) == -22
    assert h.fx_hu_batch_u8(None, 0, 16, 16, 16, 256, -1, 255, None, None, None, 0, None, None, None) == 0  # empty batch


def test_product_package_does_not_touch_the_oracle():
    """The shipped path must not import, link or call the CPU checker."""
    for p in (ROOT / "image-processing_amd").rglob("*"):
        if p.suffix in (".py", ".hip", ".h", ".cpp") and p.is_file():
            text = p.read_text()
            assert "oracle" not in text.lower().replace("no cpu fallback", ""), f"{p} mentions the oracle"


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    lib = pkg("_lib")
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", Path(tmp_path / "nope.so"))
    import pytest
    with pytest.raises(ImportError, match="no CPU fallback"):
        lib.load()
