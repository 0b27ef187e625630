"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/anatomask_hip.h declares
(no compute calls without a GPU)."""
import ctypes
import os

from anatomask_amd import build, hip


def test_library_builds_and_exports_all_declared_symbols():
    path = build.build(verbose=False)
    assert os.path.exists(path)
    decl = hip.declared_functions()
    assert len(decl) >= 25, sorted(decl)
    lib = ctypes.CDLL(path)
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    assert lib.am_version() >= 1


def test_binding_fails_loudly_without_library(tmp_path):
    import pytest
    with pytest.raises(RuntimeError):
        hip.HipLib(str(tmp_path / "missing.so"))


def test_loader_refuses_a_library_not_built_from_the_sources_next_to_it():
    """build.py stamps the library with the sha256 of its sources + flags; hip.HipLib recomputes it: a prebuilt binary that travelled to
    another box proves itself against the source that travelled with it, a stale one is refused."""
    import pytest
    path = build.build(verbose=False)
    sp = build.stamp_path(path)
    good = open(sp).read()
    assert good.strip() == build.source_digest()
    hip.HipLib(path)
    try:
        with open(sp, "w") as fh:
            fh.write("0" * 64 + "\n")
        with pytest.raises(RuntimeError, match="was not built from"):
            hip.HipLib(path)
    finally:
        with open(sp, "w") as fh:
            fh.write(good)
    hip.HipLib(path)
