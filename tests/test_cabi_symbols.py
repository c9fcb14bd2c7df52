"""CPU: the C-ABI library loads and exports every symbol include/gd_hip.h declares."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    names = set()
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if fn.endswith(".h"):
            txt = open(os.path.join(inc, fn)).read()
            names |= set(re.findall(r"\b(gd_[a-z0-9_]+)\s*\(", txt))
    return names


def test_library_exports_every_declared_symbol():
    import gd_amd
    from gd_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build the library first (__graft_entry__.build())"
    L = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared()
    assert declared, "include/gd_hip.h declares nothing?"
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/ but not exported"
    assert declared == set(_lib.SIGNATURES), (declared ^ set(_lib.SIGNATURES))
    assert _lib.lib().gd_abi_version() >= 1
