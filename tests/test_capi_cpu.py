"""The C-ABI library loads without a GPU and exports exactly what include/t3_hip.h declares
(no compute calls here)."""
import ctypes as C
import os
import re

import pytest

from chatterbox_vllm_amd import _capi

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "t3_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(t3_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_agree():
    declared = _declared_symbols()
    assert len(declared) >= 20
    assert sorted(_capi.SIGNATURES) == declared


def test_library_exports_every_declared_symbol():
    lib = _capi.load()
    for name in _declared_symbols():
        assert hasattr(lib, name), f"{name} is declared in t3_hip.h but not exported"
    assert lib.t3_abi_version() == _capi.ABI_VERSION


def test_struct_layouts_match_header_sizes():
    # t3_config: 16 int32 + 2 float + int64 + 2 int32 ; t3_sampling: 4 float + 4 int32 + uint64
    assert C.sizeof(_capi.T3Config) == 16 * 4 + 2 * 4 + 8 + 2 * 4
    assert C.sizeof(_capi.T3Sampling) == 4 * 4 + 4 * 4 + 8 + 4 * 4
    assert C.sizeof(_capi.T3StepStats) == 4 * 4 + 8 + 4 * 4 + 2 * 4
    # t3_gemm_desc: 5 pointers + 10 int32 + 4 pointers + int64 + pointer + 2 int32 + 4 pointers + int32 + float
    assert C.sizeof(_capi.T3GemmDesc) == 5 * 8 + 10 * 4 + 4 * 8 + 8 + 8 + 2 * 4 + 4 * 8 + 4 + 4 + 8


def test_integration_sketch_declares_the_structs_of_the_binding():
    """INTEGRATION.md shows a maintainer the ctypes structures to declare.  The two struct declarations of that sketch
    are executed here as written and must have the field names, types and sizes of the package's own binding, so the
    document cannot drift from the ABI."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    code = re.search(r"```python\n(.*?)```", doc, flags=re.S).group(1)
    structs = re.findall(r"^class (T3Config|T3Sampling)\(C\.Structure\):\n((?:    .*\n|\n)+?)(?=^\S)", code, flags=re.M)
    assert [n for n, _ in structs] == ["T3Config", "T3Sampling"]
    ns = {"C": C}
    for name, body in structs:
        exec(f"class {name}(C.Structure):\n{body}", ns)
    for name in ("T3Config", "T3Sampling"):
        mine, theirs = getattr(_capi, name), ns[name]
        assert C.sizeof(theirs) == C.sizeof(mine), name
        assert [(f[0], f[1]) for f in theirs._fields_] == [(f[0], f[1]) for f in mine._fields_], name
    # the sketch's positional T3Sampling(...) call fills every field
    call = re.search(r"sp = T3Sampling\((.*?)\)\n", code, flags=re.S).group(1)
    assert len([a for a in re.split(r",\s*(?![^()]*\))", call) if a.strip()]) == len(_capi.T3Sampling._fields_)


def test_argument_errors_are_reported_not_crashed():
    lib = _capi.load()
    h = C.c_void_p()
    bad = _capi.T3Config(n_layers=1, d_model=512, n_heads=8, head_dim=64, d_ff=4096)
    assert lib.t3_engine_create(C.byref(bad), C.byref(h)) != 0
    assert b"d_model" in lib.t3_last_error()
    with pytest.raises(_capi.T3Error):
        _capi.check(lib, lib.t3_engine_finalize(None))
    assert lib.t3_engine_has_unfinished(None) == 0
    assert lib.t3_engine_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setenv("T3_HIP_LIB", str(tmp_path / "nope.so"))
    _capi.load.cache_clear()
    try:
        with pytest.raises(_capi.T3Error):
            _capi.load()
    finally:
        monkeypatch.delenv("T3_HIP_LIB")
        _capi.load.cache_clear()
