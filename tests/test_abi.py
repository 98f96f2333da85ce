"""The C-ABI library loads and exports exactly what include/ngpde.h declares (no compute calls: CPU)."""
import ctypes as C
import os
import re

import ngpde_amd as ng
from ngpde_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "ngpde.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ngpde_[a-z0-9_]+)\s*\(", src)))


def test_header_and_binding_table_agree():
    assert header_functions() == sorted(_lib.SIGNATURES)


def test_library_loads_and_exports_every_symbol():
    lib = _lib.load()
    for name in header_functions():
        assert hasattr(lib, name), name
    assert lib.ngpde_version() == b"0.1.0"


def test_header_cites_reference_lines():
    src = open(os.path.join(ROOT, "include", "ngpde.h")).read()
    for cite in ("src/layers.jl:200-239", "src/layers.jl:211", "graph_node.md:44-66", "test/runtests.jl:89-102"):
        assert cite in src, cite


def test_error_convention_without_gpu():
    lib = _lib.load()
    out = C.c_void_p()
    # NULL edge arrays with n_edges > 0 -> invalid argument, message retrievable, no exception/abort
    st = lib.ngpde_graph_create(3, 4, None, None, 1, 1, C.byref(out))
    assert st == _lib.ERR_INVALID_ARGUMENT
    assert b"NULL" in lib.ngpde_last_error()
    st = lib.ngpde_graph_info(None, None, None, None)
    assert st == _lib.ERR_INVALID_ARGUMENT
    assert lib.ngpde_gcn_workspace_bytes(None, 64, 64, 1) == 0
    assert lib.ngpde_graph_destroy(None) == 0 and lib.ngpde_node_destroy(None) == 0


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "libngpde_hip.so"))
    try:
        _lib.load()
    except ImportError as e:
        assert "no CPU fallback" in str(e)
    else:
        raise AssertionError("loading a missing library must raise")


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "neuralgraphpde.jl_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "from oracle" not in text and "import oracle" not in text and "oracle/" not in text.replace(
                    "never imports anything from oracle/", "").replace("Nothing here imports oracle/", ""), f


def test_collective_entries_reject_null_arguments_without_a_gpu():
    lib = _lib.load()
    out = C.c_void_p()
    assert lib.ngpde_comm_create(None, 0, 1, C.byref(out)) == _lib.ERR_INVALID_ARGUMENT and not out.value
    assert lib.ngpde_comm_unique_id(None, 0) == _lib.ERR_INVALID_ARGUMENT
    assert lib.ngpde_grad_allreduce(None, None, 0, None) == _lib.ERR_INVALID_ARGUMENT
    assert lib.ngpde_comm_destroy(None) == 0
