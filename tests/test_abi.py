"""CPU: the C-ABI library loads and exports every symbol include/diffcodec_hip.h declares (no compute calls)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "diffcodec_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dc_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_what_the_binding_binds():
    from diffcodec_amd import lib
    hdr = _header_symbols()
    assert len(hdr) >= 25
    assert sorted(lib.SIGNATURES) == hdr, (set(hdr) ^ set(lib.SIGNATURES))


def test_library_exports_every_declared_symbol():
    import __graft_entry__ as g
    from diffcodec_amd import lib
    if not os.path.exists(lib.LIB_PATH):
        g.build()
    l = lib.load()
    for name in _header_symbols():
        assert hasattr(l, name), name


def test_conv_desc_layout_matches_header():
    """field order of the ctypes mirror == field order of `dc_conv_desc`"""
    from diffcodec_amd.lib import ConvDesc
    src = open(os.path.join(ROOT, "include", "diffcodec_hip.h")).read()
    body = src[src.index("typedef struct dc_conv_desc {"):src.index("} dc_conv_desc;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = []
    for stmt in body.split("{", 1)[1].split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        decl = stmt.split(",")
        names.append(re.findall(r"[A-Za-z_0-9]+", decl[0])[-1])
        names += [re.findall(r"[A-Za-z_0-9]+", d)[-1] for d in decl[1:]]
    assert [f[0] for f in ConvDesc._fields_] == names


def test_product_path_fails_loudly_without_the_library(monkeypatch):
    from diffcodec_amd import lib
    import pytest
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", "/nonexistent/libdiffcodec_hip.so")
    with pytest.raises(lib.HipLibraryMissing):
        lib.load()
