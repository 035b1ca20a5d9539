"""CPU-only checks of the C-ABI boundary: the shared library builds for gfx950, loads, and exports
every symbol declared in include/shg_vqa.h (no compute calls - there is no GPU here)."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    text = open(os.path.join(ROOT, "include", "shg_vqa.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(shg_[a-z0-9_]+)\s*\(", text)))


def test_library_builds_and_exports_every_declared_symbol():
    from shg_vqa_amd import build, _lib
    path = build.build()
    assert os.path.exists(path)
    handle = ctypes.CDLL(path)
    names = _declared()
    assert len(names) >= 15
    for n in names:
        assert hasattr(handle, n), "missing export: " + n
    assert set(names) == set(_lib.exported_names()), set(names) ^ set(_lib.exported_names())
    assert _lib.lib().shg_version() >= 100


def test_header_cites_reference_lines():
    text = open(os.path.join(ROOT, "include", "shg_vqa.h")).read()
    assert text.count(".py:") >= 12      # every entry point names the reference arithmetic it replaces


def test_product_does_not_import_the_oracle():
    pkg = os.path.join(ROOT, "shg_vqa_amd")
    for base, _, files in os.walk(pkg):
        for f in files:
            if f.endswith(".py"):
                src = open(os.path.join(base, f)).read()
                assert "import oracle" not in src and "from oracle" not in src, f


def test_fastcall_trampoline_reaches_the_library_and_checks_arity():
    """csrc_host/_fastcall.c: the low-overhead call path (no GPU needed: argument validation fails before any launch)."""
    import pytest
    from shg_vqa_amd import _lib
    if _lib._fast is None:
        pytest.skip("_fastcall extension not built")
    with pytest.raises(_lib.ShgError, match="null pointer"):
        _lib.call("shg_gemm", 0, 0, 0, 0, 1, 1, 8, 8, 8, 8, 8, 8, 1, 1, 0, 0)
    with pytest.raises(_lib.ShgError, match="p_drop"):          # a float argument arrives in its register
        _lib.call("shg_gemm_act", 16, 16, 16, 0, 1, 1, 8, 8, 8, 8, 8, 8, 1, 1, 0, 0, 1.5, 0, 0, 0)
    ent = _lib._FAST["shg_gemm"]
    with pytest.raises(TypeError):
        _lib._fast.call(ent[0], ent[1], 1, 2, 3)


def test_library_reads_no_environment_and_exposes_its_tuning_table():
    """Every switch sits behind shg_set_tuning (host-only calls: no GPU needed); the kernels' sources hold no getenv."""
    import pytest
    from shg_vqa_amd import _lib
    csrc = os.path.join(ROOT, "shg_vqa_amd", "csrc")
    for f in os.listdir(csrc):
        if f.endswith((".hip", ".h")):
            assert "getenv" not in open(os.path.join(csrc, f)).read(), f
    names = _lib.tuning_names()
    assert len(names) >= 16 and "gemm8_min_tiles" in names and "attn_nb_dq" in names
    assert _lib.get_tuning("gemm8_min_tiles") == 120
    old = _lib.set_tuning("gemm8_min_tiles", 130)
    assert old == 120 and _lib.get_tuning("gemm8_min_tiles") == 130
    _lib.set_tuning("gemm8_min_tiles", old)
    with pytest.raises(_lib.ShgError, match="unknown tuning switch"):
        _lib.set_tuning("no_such_switch", 1)
