"""The boundary called from plain C: tests/c_abi/gcn_roundtrip.c (GCNConv forward + pullback) and tests/c_abi/mp_roundtrip.c
(the solver plan -- also on a graph with a hub, where it takes the hub geometry --, the edge-function message path fused and on the primitives, both GNOConv message forms, the one-launch GAT
layer against its composition, ngpde_rk_stage_combine) are compiled with gcc against include/ngpde.h, linked with
libngpde_hip.so, the HIP runtime and the C oracle (the checker), and run on the GPU -- no Python, torch or C++ on the
calling side.  This is the shape of the ccall binding INTEGRATION.md sketches for the Julia package."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build(tmp_path_factory, name):
    out = str(tmp_path_factory.mktemp("c_abi") / name)
    lib_dir = os.path.join(ROOT, "neuralgraphpde.jl_amd")
    odir = os.path.join(ROOT, "oracle")
    if not os.path.exists(os.path.join(odir, "libngpde_oracle.so")):
        subprocess.check_call(["make", "-C", odir])
    cmd = ["gcc", "-O1", "-std=c11", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I", os.path.join(ROOT, "include"), "-I", "/opt/rocm/include",
           os.path.join(ROOT, "tests", "c_abi", name + ".c"), "-o", out,
           os.path.join(lib_dir, "libngpde_hip.so"), os.path.join(odir, "libngpde_oracle.so"),
           "-L/opt/rocm/lib", "-lamdhip64", "-lm",
           f"-Wl,-rpath,{lib_dir}", f"-Wl,-rpath,{odir}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.check_call(cmd)
    return out


@pytest.fixture(scope="module")
def exe(tmp_path_factory):
    return build(tmp_path_factory, "gcn_roundtrip")


@pytest.fixture(scope="module")
def exe_mp(tmp_path_factory):
    return build(tmp_path_factory, "mp_roundtrip")


@pytest.fixture(scope="module")
def exe_layers(tmp_path_factory):
    return build(tmp_path_factory, "layers_roundtrip")


@pytest.fixture(scope="module")
def exe_ode(tmp_path_factory):
    return build(tmp_path_factory, "ode_roundtrip")


def test_solver_level_create_call_from_plain_c(exe_ode):
    # ngpde_ode_create / _forward / _backward (csrc/api_ode.hip): NeuralODE(VMHConv) with saveat and NeuralODE(GCNConv, GCNConv) bit for bit
    # the plans' own entries, one Euler step against a double-precision loop, the refusals (DimensionMismatch / ERR_UNSUPPORTED)
    env = {k: v for k, v in os.environ.items() if not k.startswith("NGPDE_")}
    r = subprocess.run([exe_ode], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAIL" not in r.stdout and "all comparisons within tolerance" in r.stdout
    assert r.stdout.count("ok: ") >= 14, r.stdout


def test_layer_level_entries_from_plain_c(exe_layers):
    # MPPDEConv, VMHConv, ExplicitEdgeConv, GNOConv: forward and pullback with ONE call each (ngpde_edge_layer_*, ngpde_gno_layer_*), values against double-precision
    # loops on the concatenated message inputs, gradients against central differences of those loops
    env = {k: v for k, v in os.environ.items() if not k.startswith("NGPDE_")}
    r = subprocess.run([exe_layers], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAIL" not in r.stdout and "all comparisons within tolerance" in r.stdout
    for name in ("MPPDEConv", "VMHConv", "ExplicitEdgeConv", "GNOConv"):
        assert f"{name} forward" in r.stdout and f"{name} pullback, all gradients" in r.stdout, r.stdout


@pytest.mark.parametrize("d", [64, 24])
def test_gcn_roundtrip_from_plain_c(exe, d):
    r = subprocess.run([exe, str(d)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0.1.0" in r.stdout


def test_solver_plan_message_path_gno_gat_from_plain_c(exe_mp):
    # the program asks for the one-launch GAT layer and the one-launch pair pullback: it lifts the suite-wide switches for them
    # (the C program asserts the forms it exercises -- the persistent plans, the one-launch GAT layer: it runs without the suite's switches)
    env = {k: v for k, v in os.environ.items() if not k.startswith("NGPDE_")}
    r = subprocess.run([exe_mp], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "FAIL" not in r.stdout and "node_gcn2 u(T)" in r.stdout and "gat_layer_forward" in r.stdout
    assert "(hub geometry), fault 0" in r.stdout and "node_gcn2 (hub geometry) dW2" in r.stdout, r.stdout   # (the program runs without switches)
    assert "dense_pair_backward dWq" in r.stdout and "dense_chain2_forward" in r.stdout
    assert "node_vmh_forward_saveat" in r.stdout and r.stdout.count("node_vmh_backward_saveat du0") == 3, r.stdout   # (ran, not skipped)
