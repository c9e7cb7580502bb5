"""CPU: the C-ABI library builds/loads and exports every symbol include/*.h declares (no compute)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions(path):
    txt = open(path).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    txt = re.sub(r"//[^\n]*", "", txt)
    return sorted(set(re.findall(r"\b(pfhip_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib(pkg):
    if not os.path.exists(pkg.LIB_PATH):
        import __graft_entry__ as ge
        ge.build()
    return pkg.load_lib()


def test_exports_every_declared_symbol(lib):
    names = []
    for h in ("pfhip.h", "pfhip_ops.h"):
        names += header_functions(os.path.join(ROOT, "include", h))
    assert len(names) >= 20
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_binding_list_matches_header(pkg):
    assert sorted(pkg.ABI_SYMBOLS) == header_functions(os.path.join(ROOT, "include", "pfhip.h"))


def test_last_error_and_null_handling(lib):
    assert lib.pfhip_last_error() is not None
    assert lib.pfhip_sample_rate(None) == 0 and lib.pfhip_vocab_size(None) == 0
    lib.pfhip_destroy(None)        # no-op like delete nullptr


def test_missing_library_fails_loudly(pkg, monkeypatch):
    monkeypatch.setattr(pkg, "_lib", None)
    monkeypatch.setattr(pkg, "LIB_PATH", "/nonexistent/libpfhip.so")
    with pytest.raises(pkg.PfhipError):
        pkg.load_lib()
