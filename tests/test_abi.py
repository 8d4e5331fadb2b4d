"""The C-ABI library loads and exports every symbol include/rex.h declares (no compute calls)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    hdr = open(os.path.join(ROOT, "include", "rex.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(rex_[a-z_0-9]+)\s*\(", hdr)))


def test_header_and_binding_agree():
    from random_envs_amd import _native
    assert sorted(_native.SYMBOLS) == _declared()


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    g.build()
    from random_envs_amd import _native
    assert os.path.exists(_native.LIB_PATH)
    assert sorted(_native.exported_symbols()) == _declared()


def test_no_cpu_fallback_in_product():
    """the product package never imports the oracle or the host harness"""
    pkg = os.path.join(ROOT, "random-envs_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert "oracle" not in src.replace("oracle/", "").lower() or f == "__init__.py" and False, f
            assert "host_harness" not in src and "libmjo" not in src, f


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from random_envs_amd import _native
    monkeypatch.setattr(_native, "_lib", None)
    monkeypatch.setattr(_native, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_native.RexError, match="no CPU fallback"):
        _native.lib()
