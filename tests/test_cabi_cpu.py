"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/vqa_hip.h declares;
the ctypes table in _lib.py covers exactly the same set (no compute calls without a GPU)."""
import ctypes
import os
import re

import pytest

from _pkg import REPO, sub


def _header_symbols():
    txt = open(os.path.join(REPO, "include", "vqa_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(?:int|long long)\s+(vqa_\w+)\s*\(", txt)))


def test_header_declares_the_bound_symbols():
    L = sub("_lib")
    assert _header_symbols() == sorted(L.SIGNATURES)


def test_library_builds_loads_and_exports_every_symbol():
    import __graft_entry__ as G
    G.build()
    L = sub("_lib")
    lib = ctypes.CDLL(L.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(lib, name), name


def test_header_arity_matches_ctypes_table():
    L = sub("_lib")
    txt = open(os.path.join(REPO, "include", "vqa_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    for m in re.finditer(r"\b(?:int|long long)\s+(vqa_\w+)\s*\((.*?)\)\s*;", txt, flags=re.S):
        name, args = m.group(1), m.group(2)
        n = len([a for a in args.split(",") if a.strip()])
        assert n == len(L.SIGNATURES[name]), (name, n, len(L.SIGNATURES[name]))


def test_missing_library_fails_loudly(monkeypatch):
    L = sub("_lib")
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "LIB_PATH", "/nonexistent/libvqa_hip.so")
    with pytest.raises(RuntimeError):
        L.lib()
