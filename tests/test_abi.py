"""The C-ABI library loads and exports every symbol include/lemsm.h declares (no GPU needed;
no compute entry is called here)."""
import ctypes
import os
import re

from halo2_liam_eagen_msm_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "lemsm.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lemsm_[a-z0-9_]+)\s*\(", text)))


def test_header_matches_binding_table():
    assert header_symbols() == sorted(_lib.SYMBOLS)


def test_library_builds_loads_and_exports_all():
    _lib.build()
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in header_symbols():
        assert hasattr(lib, name), name


def test_no_oracle_in_product():
    """the product package must not import, link or execute anything under oracle/"""
    pkg = os.path.join(ROOT, "halo2_liam_eagen_msm_amd")
    for dirpath, _, files in os.walk(pkg):
        if "_build" in dirpath:
            continue
        for f in files:
            if f.endswith((".py", ".hip", ".cuh", ".h", ".hpp", "Makefile")):
                text = open(os.path.join(dirpath, f)).read()
                assert not re.search(r'#include\s+["<][^">]*oracle', text), f
                assert "oracle/_build" not in text and "oracle/c" not in text, f
                if f.endswith(".py"):
                    assert not re.search(r"^\s*(import oracle|from oracle)", text, re.M), f
                assert "liblemsm_oracle" not in text, f
