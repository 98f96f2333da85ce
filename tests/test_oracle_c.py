"""Cross-check of the two independent CPU restatements: the plain-C reference-faithful port
(oracle/ngpde_oracle.c, float32) against the numpy oracle (float64, finite-difference checked)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from oracle import ngpde_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
ODIR = os.path.join(os.path.dirname(HERE), "oracle")


def load(omp=False):
    name = "libngpde_oracle_omp.so" if omp else "libngpde_oracle.so"
    path = os.path.join(ODIR, name)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", ODIR, name])
    lib = C.CDLL(path)
    vp, i32, i64, f32 = C.c_void_p, C.c_int, C.c_int64, C.c_float
    lib.ngo_gcn_forward.argtypes = [i64, i64, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp]
    lib.ngo_gcn_backward.argtypes = [i64, i64, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp]
    lib.ngo_node_gcn2.argtypes = [i64, i64, vp, vp, i32, i32, i32, i32, f32, i32] + [vp] * 11
    lib.ngo_node_gcn2.restype = i32
    return lib


def P(a):
    return None if a is None else a.ctypes.data


ACT = {"identity": 0, "relu": 1, "tanh": 2, "sigmoid": 3, "swish": 4}


@pytest.mark.parametrize("omp", [False, True])
@pytest.mark.parametrize("act", ["relu", "tanh"])
def test_c_gcn_layer_matches_numpy(omp, act):
    lib = load(omp)
    rng = np.random.default_rng(3)
    N, E, din, dout = 57, 400, 12, 20
    s = rng.integers(0, N, E).astype(np.int64)
    t = rng.integers(0, N, E).astype(np.int64)
    og = O.Graph(s, t, num_nodes=N, index_base=0)
    X = rng.normal(size=(din, N))
    W = rng.normal(size=(dout, din))
    b = rng.normal(size=(dout, 1))
    R = rng.normal(size=(dout, N))
    yo, cache = O.gcn_conv(X, W, b, og, act)
    go = O.gcn_conv_backward(cache, R)
    # kernel layout: x [N][din] == (din x N) column-major
    x = np.ascontiguousarray(X.T, np.float32)
    wt = np.ascontiguousarray(W.T, np.float32)          # [in][out]
    bb = np.ascontiguousarray(b.reshape(-1), np.float32)
    y = np.zeros((N, dout), np.float32)
    x3 = np.zeros((N, din), np.float32)
    z = np.zeros((N, dout), np.float32)
    lib.ngo_gcn_forward(N, E, P(s), P(t), 1, din, dout, ACT[act], P(x), P(wt), P(bb), P(y), P(x3), P(z))
    np.testing.assert_allclose(y.T, yo, rtol=2e-5, atol=2e-5)
    dy = np.ascontiguousarray(R.T, np.float32)
    dx = np.zeros((N, din), np.float32)
    dwt = np.zeros((din, dout), np.float32)
    db = np.zeros(dout, np.float32)
    lib.ngo_gcn_backward(N, E, P(s), P(t), 1, din, dout, ACT[act], P(wt), P(z), P(x3), P(dy), P(dx), P(dwt), P(db))
    np.testing.assert_allclose(dx.T, go["x"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(dwt.T, go["weight"], rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(db, go["bias"].reshape(-1), rtol=1e-4, atol=1e-4)


@pytest.mark.parametrize("tab", ["euler", "tsit5"])
def test_c_node_matches_numpy(tab):
    lib = load(False)
    rng = np.random.default_rng(5)
    N, E, d, nsteps, dt = 40, 220, 8, 3, 0.1
    s = rng.integers(0, N, E).astype(np.int64)
    t = rng.integers(0, N, E).astype(np.int64)
    og = O.Graph(s, t, num_nodes=N, index_base=0)
    params = [dict(weight=rng.normal(size=(d, d)) * 0.4, bias=rng.normal(size=(d, 1)) * 0.1) for _ in range(2)]
    u0 = rng.normal(size=(d, N))
    uTo, du0o, acc = O.gcn2_node_loss_and_grads(params, og, u0, O.TABLEAUS[tab], dt, nsteps, "relu")
    f = lambda a: np.ascontiguousarray(a, np.float32)
    u = f(u0.T)
    w1, w2 = f(params[0]["weight"].T), f(params[1]["weight"].T)
    b1, b2 = f(params[0]["bias"].reshape(-1)), f(params[1]["bias"].reshape(-1))
    uT, du0 = np.zeros_like(u), np.zeros_like(u)
    dw1, dw2 = np.zeros_like(w1), np.zeros_like(w2)
    db1, db2 = np.zeros_like(b1), np.zeros_like(b2)
    rc = lib.ngo_node_gcn2(N, E, P(s), P(t), d, 1, 0 if tab == "euler" else 1, nsteps, dt, 1, P(u), P(w1), P(b1),
                           P(w2), P(b2), P(uT), P(du0), P(dw1), P(db1), P(dw2), P(db2))
    assert rc == 0
    np.testing.assert_allclose(uT.T, uTo, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(du0.T, du0o, rtol=5e-4, atol=5e-4)
    np.testing.assert_allclose(dw1.T, acc[0]["weight"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(dw2.T, acc[1]["weight"], rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(db1, acc[0]["bias"].reshape(-1), rtol=1e-3, atol=1e-3)
    np.testing.assert_allclose(db2, acc[1]["bias"].reshape(-1), rtol=1e-3, atol=1e-3)
