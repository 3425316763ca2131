"""CPU-side boundary checks: libvltf_hip.so builds for gfx950, loads, and exports exactly the
symbols include/vltf.h declares; the ctypes table mirrors the header.  No compute calls (no GPU)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as g
    from vltf_amd import _ffi
    if not os.path.exists(_ffi.LIB_PATH):
        g.build()
    return _ffi


def header_symbols():
    src = open(os.path.join(ROOT, "include", "vltf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vl_[a-z0-9_]+)\s*\(", src)))


def test_header_matches_ctypes_table(built):
    assert header_symbols() == sorted(built.SIGNATURES)


def test_library_exports_every_symbol(built):
    lib = ctypes.CDLL(built.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name
    l = built.lib()
    assert l.vl_version() >= 1
    assert isinstance(l.vl_last_error(), bytes)


def test_argument_counts_match_header(built):
    src = open(os.path.join(ROOT, "include", "vltf.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    for name, (_, args) in built.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, src, flags=re.S)
        assert m, name
        body = m.group(1).strip()
        n = 0 if body in ("", "void") else body.count(",") + 1
        assert n == len(args), "%s: header has %d args, ctypes table %d" % (name, n, len(args))


def test_missing_library_fails_loudly(monkeypatch, built):
    monkeypatch.setattr(built, "_lib", None)
    monkeypatch.setattr(built, "LIB_PATH", "/nonexistent/libvltf_hip.so")
    with pytest.raises(built.VltfError):
        built.lib()


def test_host_library_exports_header_symbols():
    from vltf_amd import _hostio
    if not os.path.exists(_hostio.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    src = open(os.path.join(ROOT, "include", "vltf_host.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    names = sorted(set(re.findall(r"\b(vlh_[a-z0-9_]+)\s*\(", src)))
    assert names == sorted(_hostio.SIGNATURES)
    lib = ctypes.CDLL(_hostio.LIB_PATH)
    for n in names:
        assert hasattr(lib, n), n
    for name, (_, args) in _hostio.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, src, flags=re.S)
        body = m.group(1).strip()
        assert (0 if body in ("", "void") else body.count(",") + 1) == len(args), name
