"""CPU-side checks of the C-ABI boundary: the library builds, loads and exports every symbol the header declares.
No compute calls (no GPU here)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    from dropoutdecoding_amd import build, _lib
    build.build()
    return _lib.load()


def _header_symbols(name="dropdec.h"):
    src = open(os.path.join(ROOT, "include", name)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dd_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_list_agree():
    from dropoutdecoding_amd import _lib
    assert _header_symbols() == sorted(_lib.SYMBOLS)


def test_library_exports_every_declared_symbol(lib):
    for s in _header_symbols():
        assert hasattr(lib, s), f"libdropdec.so does not export {s}"


def test_measurement_hooks_live_in_the_tools_library_only(lib):
    """include/dropdec_tools.h (timing hooks, experiment knobs) = libdropdec_tools.so; the product library exports none of them."""
    from dropoutdecoding_amd import _lib
    assert _header_symbols("dropdec_tools.h") == sorted(_lib.TOOLS_SYMBOLS)
    assert not set(_lib.TOOLS_SYMBOLS) & set(_lib.SYMBOLS)
    import ctypes
    tools = ctypes.CDLL(_lib.TOOLS_LIB_PATH)                 # plain dlopen: no compute calls without a GPU
    for s in _lib.TOOLS_SYMBOLS + _lib.SYMBOLS:
        assert hasattr(tools, s), f"libdropdec_tools.so does not export {s}"
    for s in _lib.TOOLS_SYMBOLS:
        assert not hasattr(lib, s), f"libdropdec.so exports the measurement hook {s}"


def test_version_and_arch(lib):
    assert lib.dd_version() >= 100
    assert lib.dd_arch() == b"gfx950"


def test_header_cites_reference_lines():
    src = open(os.path.join(ROOT, "include", "dropdec.h")).read()
    for anchor in ("models/llava.py:710-756", "models/llava.py:443-482", "models/llava.py:22-36", "models/llava.py:589-6",
                   "models/instructblip.py:447-460", "models/llavanext.py:779-8", "models/config.py"):
        assert anchor in src, anchor


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from dropoutdecoding_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.DDError, match="no CPU fallback"):
        _lib.load()


def test_product_package_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "dropoutdecoding_amd")
    for dp, _, fs in os.walk(pkg):
        for f in fs:
            if f.endswith(".py"):
                txt = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), os.path.join(dp, f)
    for f in os.listdir(os.path.join(ROOT, "models")) if os.path.isdir(os.path.join(ROOT, "models")) else []:
        if f.endswith(".py"):
            assert "oracle" not in open(os.path.join(ROOT, "models", f)).read()
