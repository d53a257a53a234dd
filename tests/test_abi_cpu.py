"""CPU: the C-ABI library loads and exports every symbol include/av_hip.h declares; argument errors surface as
status codes + av_last_error() (no kernel is launched without a GPU)."""
import ctypes
import os

import pytest
import re

from conftest import ROOT, pkg


def _declared():
    txt = open(os.path.join(ROOT, "include", "av_hip.h"), encoding="utf-8").read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(av_[a-z0-9_]+)\s*\(", txt)))


def test_every_declared_symbol_is_exported_and_bound():
    L = pkg("_lib")
    lib = L.lib()
    names = _declared()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/av_hip.h but not exported by libavhip.so"
        assert n in L.SIGNATURES, f"{n} has no ctypes signature in _lib.py"
    assert sorted(L.SIGNATURES) == names, "ctypes table and header disagree"


def test_half_library_exports_the_same_abi():
    """libavhip_f16.so (the same sources built with IEEE-half operands, precision mode "fp16") carries the identical C-ABI."""
    L = pkg("_lib"); P = pkg("precision")
    old = P.get_precision()
    P.set_precision("fp16")
    try:
        lib = L.lib()
        assert lib is not None and lib._name.endswith("libavhip_f16.so")
        for n in _declared():
            assert hasattr(lib, n), f"{n} not exported by libavhip_f16.so"
        assert lib.av_version() >= 1
    finally:
        P.set_precision(old)
    assert L.lib()._name.endswith("libavhip.so")


def test_errors_are_reported_not_aborted():
    L = pkg("_lib")
    lib = L.lib()
    assert lib.av_version() >= 1
    rc = lib.av_gemm(None, None)
    assert rc != 0 and b"null" in lib.av_last_error()
    a = L.GemmArgs()
    a.A = a.B = a.C = 16
    a.M, a.N, a.K, a.batch = 4, 4, 0, 1
    assert lib.av_gemm(ctypes.byref(a), None) != 0 and b"bad shape" in lib.av_last_error()
    assert lib.av_layernorm_fwd(None, 0, None, None, None, 0, None, None, 1, 8, 1e-5, 0, None) != 0
    assert lib.av_attention_fwd(16, 16, 16, 16, None, 1, 1, 1, 4, 4, 48, 0, 0, 0, 0, 0, 0, 0, 0, None, 1.0, 0.0, 0, 0, None) != 0
    assert b"head_dim" in lib.av_last_error()
    try:
        L.check(1, "demo")
        assert False
    except RuntimeError as e:
        assert "libavhip demo failed" in str(e)


def _header_fields(name):
    """(field, is_pointer, C scalar type) of `typedef struct <name> { ... }` in include/av_hip.h, in declaration order."""
    txt = open(os.path.join(ROOT, "include", "av_hip.h"), encoding="utf-8").read()
    body = txt[txt.index("typedef struct %s {" % name):txt.index("} %s;" % name)]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    out = []
    for stmt in body.split("{", 1)[1].split(";"):
        stmt = stmt.strip()
        if not stmt:
            continue
        m = re.match(r"^(const\s+)?(unsigned\s+long\s+long|unsigned\s+int|long\s+long|void|float|int)\s*(\*?)", stmt)
        base, first_ptr = m.group(2), bool(m.group(3))
        rest = stmt[m.end():].strip()
        for k, f in enumerate(rest.split(",")):
            f = f.strip()
            out.append((f.lstrip("*").strip(), f.startswith("*") or (k == 0 and first_ptr), base))
    return out


@pytest.mark.parametrize("cname,pyname", [("av_gemm_args", "GemmArgs"), ("av_w2v2_layer_args", "W2v2LayerArgs"), ("av_w2v2_layer_bwd_args", "W2v2LayerBwdArgs")])
def test_args_structs_match_the_header_layout(cname, pyname):
    """The ctypes mirrors must have the field order AND the field kinds (pointer / int / float / 64-bit) of the structs in include/av_hip.h."""
    import ctypes as C
    L = pkg("_lib")
    hdr = _header_fields(cname)
    mirror = getattr(L, pyname)._fields_
    assert [f[0] for f in hdr] == [f[0] for f in mirror], ([f[0] for f in hdr], [f[0] for f in mirror])
    kinds = {"int": C.c_int, "float": C.c_float, "long long": C.c_longlong, "unsigned long long": C.c_ulonglong, "unsigned int": C.c_uint}
    for (name, is_ptr, base), (_, ctype) in zip(hdr, mirror):
        want = C.c_void_p if is_ptr else kinds[base]
        assert C.sizeof(ctype) == C.sizeof(want) and (ctype is C.c_void_p) == is_ptr, (cname, name, ctype, want)
