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
    assert _lib.lib().gd_abi_version() == _lib.ABI_VERSION == 3


def _header_prototypes():
    """name -> (return kind, [argument kinds]) parsed from include/*.h; kinds: int, long, float, size_t, ptr."""
    protos = {}
    inc = os.path.join(ROOT, "include")
    for fn in os.listdir(inc):
        if not fn.endswith(".h"):
            continue
        txt = re.sub(r"/\*.*?\*/", " ", open(os.path.join(inc, fn)).read(), flags=re.S)
        txt = "\n".join(l for l in txt.splitlines() if not l.lstrip().startswith("#"))

        def kind(decl):
            decl = decl.strip()
            if "*" in decl:
                return "ptr"
            toks = [t for t in re.findall(r"[A-Za-z_][A-Za-z0-9_]*", decl) if t not in ("const", "unsigned")]
            return toks[0]          # the type word; a parameter name (if any) follows it
        for ret, name, args in re.findall(r"([A-Za-z_][A-Za-z0-9_ \*]*?)\b(gd_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", txt):
            a = [] if args.strip() in ("", "void") else [kind(x) for x in args.split(",")]
            protos[name] = (kind(ret), a)
    return protos


def _ctype_kind(t):
    if t in (ctypes.c_void_p, ctypes.c_char_p) or (isinstance(t, type) and issubclass(t, ctypes._Pointer)):
        return "ptr"
    return {ctypes.c_int: "int", ctypes.c_long: "long", ctypes.c_float: "float", ctypes.c_size_t: "size_t"}[t]


def test_ctypes_signatures_match_the_header():
    """Arity and argument kinds of every entry point agree between include/gd_hip.h and _lib.SIGNATURES (the .hip
    definitions are checked against the same header by the compiler: gd_common.h includes it)."""
    import gd_amd
    from gd_amd import _lib
    protos = _header_prototypes()
    assert set(protos) == set(_lib.SIGNATURES)
    for name, (res, args) in _lib.SIGNATURES.items():
        hret, hargs = protos[name]
        assert [_ctype_kind(a) for a in args] == hargs, (name, [_ctype_kind(a) for a in args], hargs)
        assert _ctype_kind(res) == hret, (name, res, hret)


def test_argument_checks_fail_loudly_without_touching_the_device():
    """Shape / layout guards run before any HIP call: an entry point handed an unsupported problem returns non-zero and
    leaves a message in gd_last_error (src/.../curope.cpp:49-69 is the reference's convention: TORCH_CHECK before dispatch)."""
    import gd_amd
    from gd_amd import _lib
    L = _lib.lib()
    # one image's qkv rows must span < 2^31 bytes (the attention kernels use 32-bit tile offsets): 200 000 tokens x 16 heads
    rc = L.gd_attention_fwd(None, None, None, 1, 200000, 16, 64, ctypes.c_float(0.125), _lib.BF16, None)
    assert rc != 0 and b"2^31" in L.gd_last_error()
    rc = L.gd_attention_bwd(None, None, None, None, None, None, 1, 200000, 16, 64, ctypes.c_float(0.125), _lib.BF16, 0, None)
    assert rc != 0 and b"2^31" in L.gd_last_error()
    rc = L.gd_attention_fwd(None, None, None, 1, 16, 2, 32, ctypes.c_float(0.125), _lib.BF16, None)      # head_dim != 64
    assert rc != 0 and b"head_dim" in L.gd_last_error()
