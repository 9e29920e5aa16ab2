"""CPU-side checks of the drop-in boundary: libdskgpu.so loads and exports every
symbol include/dskgpu.h declares.  No compute calls (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "dskgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dskgpu_[a-z_0-9]+)\s*\(", src)))


def test_header_lists_expected_symbols():
    from dsk_amd import engine
    assert header_symbols() == sorted(engine.EXPORTS)


def test_library_exports_every_header_symbol():
    from dsk_amd import engine
    if not os.path.exists(engine.library_path()):
        import __graft_entry__ as g
        g.build()
    lib = engine.load_library()
    for name in header_symbols():
        assert hasattr(lib, name), name
    assert b"gfx950" in lib.dskgpu_version()


def test_create_fails_loudly_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from dsk_amd import engine
    if not os.path.exists(engine.library_path()):
        pytest.skip("library not built")
    with pytest.raises(engine.DskGpuError):
        engine.KmerCounter(kmer_size=31)


def test_missing_library_raises(monkeypatch):
    from dsk_amd import engine
    monkeypatch.setattr(engine, "_lib", None)
    monkeypatch.setenv("DSKGPU_LIB", "/nonexistent/libdskgpu.so")
    with pytest.raises(ImportError):
        engine.load_library()


def test_bad_config_rejected():
    from dsk_amd import engine
    if not os.path.exists(engine.library_path()):
        pytest.skip("library not built")
    for kw in (dict(kmer_size=0), dict(kmer_size=129), dict(world_size=3), dict(world_size=2, rank=2)):      # (k = 1..128 is valid: four-word keys)
        with pytest.raises(engine.DskGpuError):
            engine.KmerCounter(**kw)


def test_kmer_to_string():
    from dsk_amd.engine import kmer_to_string
    # test/short.parse_results:1
    s = "ACTGTACGTATAAGA"
    v = 0
    for c in s:
        v = (v << 2) | "ACTG".index(c)
    assert kmer_to_string(v, 15) == s
