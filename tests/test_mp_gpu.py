"""GPU parity tests of the edge-function layers (ExplicitEdgeConv, VMHConv, MPPDEConv, GNOConv, SpectralConv,
GATConv) and of Dense/Chain: HIP primitives through the C ABI against the float64 numpy oracle, on the
reference's own test cases (/root/reference/test/runtests.jl:27-162) and on larger random graphs.

Tolerances: forward 1e-4 * max|ref| + 1e-5, gradients 5e-4 relative (chains of 2-4 fp32 dense layers).
"""
import ctypes as C

import numpy as np
import pytest
import torch

import ngpde_amd as ng
from ngpde_amd import synth as S
from oracle import ngpde_oracle as O
from ngpde_amd import _lib

pytestmark = pytest.mark.gpu
DEV = "cuda"


def close(a, ref, rtol=1e-4, atol=1e-5, what=""):
    a = a.detach().cpu().double().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)
    ref = np.asarray(ref, dtype=np.float64)
    assert a.shape == ref.shape, (what, a.shape, ref.shape)
    err = np.abs(a - ref).max() if ref.size else 0.0
    bound = rtol * (np.abs(ref).max() if ref.size else 0.0) + atol
    assert err <= bound, f"{what}: max err {err:.3e} > {bound:.3e}"


def omlp(layer, ps):
    """oracle form of a Dense / Chain-of-Dense with the layer's current parameters"""
    if isinstance(layer, ng.Dense):
        pairs = [(layer, ps)]
    else:
        pairs = [(l, ps[n]) for n, l in zip(layer.names(), layer.chain)]
    return [dict(weight=p["weight"].detach().cpu().double().numpy(),
                 bias=p["bias"].detach().cpu().double().numpy() if "bias" in p else None, act=l.activation)
            for l, p in pairs]


def leaves(ps, prefix=""):
    for k, v in ps.items():
        if isinstance(v, dict):
            yield from leaves(v, prefix + k + ".")
        else:
            yield prefix + k, v


def grad_leaves(ps):
    out = []
    for v in ps.values():
        out += grad_leaves(v) if isinstance(v, dict) else [v]
    return out


def prep(ps, seed):
    """device copy with random biases and requires_grad"""
    rng = np.random.default_rng(seed)
    ps = ng.to_device(ps, DEV)

    def walk(d):
        for k, v in d.items():
            if isinstance(v, dict):
                walk(v)
            else:
                if k == "bias":
                    d[k] = torch.as_tensor(rng.normal(size=tuple(v.shape)).astype(np.float32) * 0.3, device=DEV)
                d[k].requires_grad_(True)
    walk(ps)
    return ps


def rgraph(N, E, seed, **kw):
    rng = np.random.default_rng(seed)
    s, t = rng.integers(0, N, E), rng.integers(0, N, E)
    return ng.GNNGraph(s, t, num_nodes=N, index_base=0, **kw), O.Graph(s, t, num_nodes=N, index_base=0, **kw)


def check_grads(ps, ogr_list, x, dx):
    close(x.grad, dx, rtol=5e-4, atol=1e-4, what="dx")
    for (name, p), og in zip(ogr_list[0], ogr_list[1]):
        close(p.grad, og, rtol=5e-4, atol=2e-4, what=f"d{name}")


def mlp_grad_pairs(ps_sub, ograds, layer):
    """[(name, tensor)], [oracle grads] for a Dense / Chain sub-tree"""
    names, og = [], []
    if isinstance(layer, ng.Dense):
        pairs = [("", ps_sub)]
    else:
        pairs = [(n + ".", ps_sub[n]) for n in layer.names()]
    for (pref, p), g in zip(pairs, ograds):
        names.append((pref + "weight", p["weight"])); og.append(g["weight"])
        if "bias" in p:
            names.append((pref + "bias", p["bias"])); og.append(g["bias"])
    return names, og


# ---- SpectralConv: the reference's own known-answer test, on the GPU ------------------------------------------

def test_spectralconv_reference_known_answer():
    # /root/reference/test/runtests.jl:153-162
    s = ng.SpectralConv(100)
    ps, st = ng.setup(0, s)
    assert ps == {} and st["graph"].num_edges == 100 * 99
    x = torch.linspace(0, 2 * np.pi, 101, dtype=torch.float32, device=DEV)[1:]
    e1 = s(torch.sin(x), ps, st)[0] - torch.cos(x)
    e2 = s(torch.cos(x), ps, st)[0] + torch.sin(x)
    assert float((e1 ** 2).sum()) < 1e-3 and float((e2 ** 2).sum()) < 1e-3
    # matrix input and gradient (the operator is antisymmetric: pullback = - forward)
    X = torch.randn(3, 100, device=DEV, requires_grad=True)
    Y, _ = s(X, ps, st)
    og = O.spectral_graph(100)
    close(Y, O.spectral_conv(X.detach().cpu().double().numpy(), og, 100), rtol=2e-4, atol=1e-4)
    R = torch.randn_like(Y)
    (Y * R).sum().backward()
    close(X.grad, -O.spectral_conv(R.cpu().double().numpy(), og, 100), rtol=2e-4, atol=1e-3)


# ---- Dense / Chain ----------------------------------------------------------------------------------------------------

def test_dense_chain_parity():
    model = ng.Chain(ng.Dense(7, 20, "tanh"), ng.Dense(20, 33, "swish"), ng.Dense(33, 5, bias=False))
    ps, st = ng.setup(1, model)
    assert list(ps["layer_3"]) == ["weight"]
    ps = prep(ps, 1)
    x = torch.randn(7, 1000, device=DEV, requires_grad=True)
    y, _ = model(x, ps, st)
    yo, cache = O.mlp_forward(omlp(model, ps), x.detach().cpu().double().numpy())
    close(y, yo)
    R = np.random.default_rng(0).normal(size=yo.shape)
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    dx, gr = O.mlp_backward(omlp(model, ps), cache, R)
    names, og = mlp_grad_pairs(ps, gr, model)
    check_grads(ps, (names, og), x, dx)


@pytest.mark.parametrize("widths,dout,act", [((64, 1, 3, 2), 64, "swish"), ((64, 64), 40, "identity"), ((30, 7), 37, "tanh"),
                                            ((128,), 200, "relu")])
def test_dense_wide_tiles_segmented(widths, dout, act):
    # enough rows that the 128-row tile kernels serve forward and input pullback (>= 512 tiles), ragged last tile,
    # blocks that do / do not allow 16-byte loads, a per-graph block (row_div) as MPPDEConv's theta  (src/layers.jl:397)
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    n, per_graph = 70001, 10000
    rng = np.random.default_rng(21)
    blocks, divs = [], []
    for i, w in enumerate(widths):
        rd = per_graph if (len(widths) == 4 and i == 3) else 1
        rows = (n + rd - 1) // rd
        blocks.append(torch.as_tensor(rng.normal(size=(rows, w)), dtype=torch.float32, device=DEV).requires_grad_(rd == 1))
        divs.append(rd)
    din = sum(widths)
    wt = torch.as_tensor(rng.normal(size=(din, dout)) / np.sqrt(din), dtype=torch.float32, device=DEV).requires_grad_(True)
    b = torch.as_tensor(rng.normal(size=dout), dtype=torch.float32, device=DEV).requires_grad_(True)
    y = F.dense(blocks, wt, b, ng.layers._act_code(act)[1], row_divs=divs, n=n)
    X = np.concatenate([np.repeat(bl.detach().cpu().double().numpy(), rd, axis=0)[:n] for bl, rd in zip(blocks, divs)], axis=1)
    layer = [dict(weight=wt.detach().cpu().double().numpy().T, bias=b.detach().cpu().double().numpy(), act=act)]
    yo, cache = O.mlp_forward(layer, X.T)
    close(y, yo.T)
    R = rng.normal(size=yo.shape)
    (y * torch.as_tensor(R.T, dtype=torch.float32, device=DEV)).sum().backward()
    dx, gr = O.mlp_backward(layer, cache, R)
    close(wt.grad, gr[0]["weight"].T, rtol=3e-4)
    close(b.grad, gr[0]["bias"].reshape(-1), rtol=3e-4)
    o = 0
    for bl, rd, w in zip(blocks, divs, widths):
        if rd == 1:
            close(bl.grad, dx[o:o + w].T)
        o += w


@pytest.mark.parametrize("n,widths,divs,dout,act", [
    (18000, (60,), (1,), 60, "tanh"), (3000, (1, 40), (1, 1), 60, "tanh"), (3000, (60,), (1,), 1, "identity"), (3001, (2,), (1,), 60, "swish"),
    (65, (7, 3), (1, 1), 5, "sigmoid"), (63, (64,), (1,), 64, "relu"), (1000, (33, 2, 29), (1, 100, 1), 37, "gelu"), (129, (1,), (1,), 1, "tanh"),
    (40000, (17,), (1,), 64, "leakyrelu"), (512, (30, 4), (1, 512), 64, "elu")])
def test_dense_small_pullback_one_launch(n, widths, divs, dout, act):
    # dense_small_bwd.hip: the whole pullback of a Dense of at most 64 x 64 at latency-bound row counts in one launch -- the
    # tutorial MLPs' shapes (VMH.md:75-79), ragged last tiles, fewer rows than a tile, widths that are no multiple of 4 or 16, blocks of
    # a virtual vcat with and without a gradient (per-graph blocks, src/layers.jl:397), one input / one output -- against float64
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    rng = np.random.default_rng(n + dout)
    blocks = []
    for w, rd in zip(widths, divs):
        rows = (n + rd - 1) // rd
        blocks.append(torch.as_tensor(rng.normal(size=(rows, w)), dtype=torch.float32, device=DEV).requires_grad_(rd == 1))
    din = sum(widths)
    wt = torch.as_tensor(rng.normal(size=(din, dout)) / np.sqrt(din), dtype=torch.float32, device=DEV).requires_grad_(True)
    b = torch.as_tensor(rng.normal(size=dout), dtype=torch.float32, device=DEV).requires_grad_(True)
    y = F.dense(blocks, wt, b, ng.layers._act_code(act)[1], row_divs=list(divs), n=n)
    X = np.concatenate([np.repeat(bl.detach().cpu().double().numpy(), rd, axis=0)[:n] for bl, rd in zip(blocks, divs)], axis=1)
    layer = [dict(weight=wt.detach().cpu().double().numpy().T, bias=b.detach().cpu().double().numpy(), act=act)]
    yo, cache = O.mlp_forward(layer, X.T)
    close(y, yo.T)
    R = rng.normal(size=yo.shape)
    (y * torch.as_tensor(R.T, dtype=torch.float32, device=DEV)).sum().backward()
    dx, gr = O.mlp_backward(layer, cache, R)
    close(wt.grad, gr[0]["weight"].T, rtol=3e-4, atol=1e-4)
    close(b.grad, gr[0]["bias"].reshape(-1), rtol=3e-4, atol=1e-4)
    o = 0
    for bl, rd, w in zip(blocks, divs, widths):
        if rd == 1:
            close(bl.grad, dx[o:o + w].T)
        o += w
    y2 = F.dense(blocks, wt, b, ng.layers._act_code(act)[1], row_divs=list(divs), n=n)      # reproducible: fixed-order slab sums
    g1 = wt.grad.clone(); wt.grad = None
    (y2 * torch.as_tensor(R.T, dtype=torch.float32, device=DEV)).sum().backward()
    assert torch.equal(wt.grad, g1)


@pytest.mark.parametrize("n,din,dout,bias", [(4096, 128, 8192, False), (1040, 132, 2052, False), (4096, 128, 8192, True),
                                             (2048, 256, 4096, False)])
def test_dense_few_rows_wide_output_pullbacks(n, din, dout, bias, monkeypatch):
    # GNOConv's node-level T = W2 (x) h (src/layers.jl:523-530 reassociated): 4096 x 128 => 8192.  Forward on 128 x 128 tiles;
    # both pullbacks on the same tiles with the contraction split over workgroups (dense_gemm128_split_kernel: input pullback
    # over the 8192 outputs, weight pullback over the rows) when the layer has no bias gradient -- against float64, and against
    # the older kernels (NGPDE_DENSE_NO_GEMM128 is read once per process, so that comparison is with the float64 values only)
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    rng = np.random.default_rng(n + din)
    x = torch.as_tensor(rng.normal(size=(n, din)), dtype=torch.float32, device=DEV).requires_grad_(True)
    wt = torch.as_tensor(rng.normal(size=(din, dout)) / np.sqrt(din), dtype=torch.float32, device=DEV).requires_grad_(True)
    b = torch.as_tensor(rng.normal(size=dout), dtype=torch.float32, device=DEV).requires_grad_(True) if bias else None
    y = F.dense([x], wt, b, 0)
    X, W = x.detach().cpu().double().numpy(), wt.detach().cpu().double().numpy()
    yo = X @ W + (b.detach().cpu().double().numpy() if bias else 0.0)
    close(y, yo)
    R = rng.normal(size=yo.shape) / np.sqrt(dout)
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    close(x.grad, R @ W.T, rtol=3e-4)
    close(wt.grad, X.T @ R, rtol=3e-4)
    if bias:
        close(b.grad, R.sum(axis=0), rtol=3e-4)


@pytest.mark.parametrize("widths,grads,act", [((64,), (True,), "identity"), ((64, 2), (True, False), "swish"),
                                              ((64, 1, 1, 2), (True, False, False, False), "swish"),
                                              ((64, 64, 2), (True, True, False), "swish"), ((2, 64, 64), (False, False, True), "relu"),
                                              ((64,), (False,), "tanh"), ((64, 64), (True, False), "tanh")])
def test_dense_streaming_pullback(widths, grads, act, monkeypatch):
    _dense_streaming_pullback_case(widths, grads, act, 70001, monkeypatch)


@pytest.mark.parametrize("widths,grads,n", [((64, 2), (True, False), 32768), ((64, 64, 2), (True, True, False), 32768),
                                            ((64, 64), (False, True), 32768 + 64 * 511 + 1)])
def test_dense_streaming_pullback_tile_counts(widths, grads, n, monkeypatch):
    # one tile per workgroup (no next tile to prefetch: 32 768 rows is the launch's lower limit), and a workgroup count that leaves
    # all but one workgroup with a single tile and the last tile ragged
    _dense_streaming_pullback_case(widths, grads, "swish", n, monkeypatch)


def _dense_streaming_pullback_case(widths, grads, act, n, monkeypatch):
    # dense_stream_bwd.hip: dz, input pullbacks, weight pullback and bias gradient of a 64-output Dense in one launch (one or two
    # 64-wide blocks + narrow blocks without gradient, the last block per graph as MPPDEConv's theta, src/layers.jl:397, :418);
    # ragged last tile; against the oracle and against the composed path (NGPDE_DENSE_NO_STREAM_BWD=1)
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    monkeypatch.delenv("NGPDE_DENSE_NO_STREAM_BWD", raising=False)
    per_graph, dout = 10000, 64
    rng = np.random.default_rng(33)
    res = []
    for mode in (0, 1):
        rng = np.random.default_rng(33)
        blocks, divs = [], []
        for i, (w, gq) in enumerate(zip(widths, grads)):
            rd = per_graph if (w == 2 and len(widths) >= 3 and i in (0, len(widths) - 1) and not gq and widths.count(64) >= 1 and i == len(widths) - 1) else 1
            rows = (n + rd - 1) // rd
            blocks.append(torch.as_tensor(rng.normal(size=(rows, w)), dtype=torch.float32, device=DEV).requires_grad_(gq))
            divs.append(rd)
        din = sum(widths)
        wt = torch.as_tensor(rng.normal(size=(din, dout)) / np.sqrt(din), dtype=torch.float32, device=DEV).requires_grad_(True)
        b = torch.as_tensor(rng.normal(size=dout), dtype=torch.float32, device=DEV).requires_grad_(True)
        R = rng.normal(size=(dout, n))
        if mode == 1:
            monkeypatch.setenv("NGPDE_DENSE_NO_STREAM_BWD", "1")
        y = F.dense(blocks, wt, b, ng.layers._act_code(act)[1], row_divs=divs, n=n)
        (y * torch.as_tensor(R.T, dtype=torch.float32, device=DEV)).sum().backward()
        if mode == 1:
            monkeypatch.delenv("NGPDE_DENSE_NO_STREAM_BWD")
        res.append((blocks, wt, b))
    (blocks, wt, b), (blocks2, wt2, b2) = res
    X = np.concatenate([np.repeat(bl.detach().cpu().double().numpy(), rd, axis=0)[:n] for bl, rd in zip(blocks, divs)], axis=1)
    layer = [dict(weight=wt.detach().cpu().double().numpy().T, bias=b.detach().cpu().double().numpy(), act=act)]
    yo, cache = O.mlp_forward(layer, X.T)
    dx, gr = O.mlp_backward(layer, cache, R)
    close(wt.grad, gr[0]["weight"].T, rtol=3e-4)
    close(b.grad, gr[0]["bias"].reshape(-1), rtol=3e-4)
    close(wt.grad, wt2.grad.cpu().double().numpy(), rtol=3e-4)
    close(b.grad, b2.grad.cpu().double().numpy(), rtol=3e-4)
    o = 0
    for bl, bl2, gq, w in zip(blocks, blocks2, grads, widths):
        if gq:
            close(bl.grad, dx[o:o + w].T)
            close(bl.grad, bl2.grad.cpu().double().numpy(), rtol=2e-5, atol=2e-6)
        else:
            assert bl.grad is None
        o += w


@pytest.mark.parametrize("nwa,nwb", [((2, 2), (2,)), ((), ()), ((1, 1, 2), (3,))])
def test_dense_pair_streaming_backward(nwa, nwb, monkeypatch):
    # ngpde_dense_pair_backward: both pullbacks of an activation-free pair in one launch (dx already summed, narrow features and
    # bias on the matrix pipe as a 16-wide block), against the oracle and the composed path
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    monkeypatch.delenv("NGPDE_DENSE_NO_STREAM_BWD", raising=False)
    n, per_graph = 70001, 10000
    res = []
    for mode in (0, 1):
        rng = np.random.default_rng(79)
        x = torch.as_tensor(rng.normal(size=(n, 64)), dtype=torch.float32, device=DEV).requires_grad_(True)

        def side(nw, last_per_graph):
            blocks, divs = [x], [1]
            for i, w in enumerate(nw):
                rd = per_graph if (last_per_graph and i == len(nw) - 1) else 1
                blocks.append(torch.as_tensor(rng.normal(size=((n + rd - 1) // rd, w)), dtype=torch.float32, device=DEV))
                divs.append(rd)
            din = 64 + sum(nw)
            wt = torch.as_tensor(rng.normal(size=(din, 64)) / np.sqrt(din), dtype=torch.float32, device=DEV).requires_grad_(True)
            b = torch.as_tensor(rng.normal(size=64), dtype=torch.float32, device=DEV).requires_grad_(True)
            return blocks, divs, wt, b

        A, B = side(nwa, True), side(nwb, False)
        Ra, Rb = rng.normal(size=(64, n)), rng.normal(size=(64, n))
        if mode == 1:
            monkeypatch.setenv("NGPDE_DENSE_NO_STREAM_BWD", "1")
        # the shared block also comes back routed through the pair (passthrough): a further consumer's gradient w.r.t. it is
        # added inside the pair's pullback launch (dx_addend)
        Rx = rng.normal(size=(n, 64))
        ya, yb, xp = F.dense_pair(A[0], A[2], A[3], 0, B[0], B[2], None, 0, row_divs_a=A[1], row_divs_b=B[1], n=n, passthrough=True)
        ((ya * torch.as_tensor(Ra.T, dtype=torch.float32, device=DEV)).sum() + (yb * torch.as_tensor(Rb.T, dtype=torch.float32, device=DEV)).sum()
         + (xp * torch.as_tensor(Rx, dtype=torch.float32, device=DEV)).sum()).backward()
        if mode == 1:
            monkeypatch.delenv("NGPDE_DENSE_NO_STREAM_BWD")
        res.append((x, A, B, Ra, Rb, Rx))
    (x, A, B, Ra, Rb, Rx), (x2, A2, B2, _, _, _) = res
    close(x.grad, x2.grad.cpu().double().numpy(), rtol=2e-5, atol=2e-6)
    close(A[2].grad, A2[2].grad.cpu().double().numpy(), rtol=3e-4)
    close(B[2].grad, B2[2].grad.cpu().double().numpy(), rtol=3e-4)
    close(A[3].grad, A2[3].grad.cpu().double().numpy(), rtol=3e-4)
    dxs = []
    for (blocks, divs, wt, b), R, bias in ((A, Ra, True), (B, Rb, False)):
        X = np.concatenate([np.repeat(bl.detach().cpu().double().numpy(), rd, axis=0)[:n] for bl, rd in zip(blocks, divs)], axis=1)
        layer = [dict(weight=wt.detach().cpu().double().numpy().T, bias=b.detach().cpu().double().numpy() if bias else None, act="identity")]
        yo, cache = O.mlp_forward(layer, X.T)
        dx, gr = O.mlp_backward(layer, cache, R)
        dxs.append(dx[:64])
        close(wt.grad, gr[0]["weight"].T, rtol=3e-4)
        if bias:
            close(b.grad, gr[0]["bias"].reshape(-1), rtol=3e-4)
    close(x.grad, (dxs[0] + dxs[1]).T + Rx)


@pytest.mark.parametrize("nwa,nwb,douts", [((2, 2), (2,), (64, 64)), ((), (), (64, 48)), ((1, 3, 2), (3,), (40, 64))])
def test_dense_pair_streaming_forward(nwa, nwb, douts, monkeypatch):
    # ngpde_dense_pair_forward: two Dense layers from one pass over their shared 64-wide block (dense_pair_fwd_kernel, weights in
    # registers), against the oracle and the two-launch path (NGPDE_DENSE_NO_STREAM2=1); ragged last tile, a per-graph block
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    monkeypatch.delenv("NGPDE_DENSE_NO_STREAM2", raising=False)
    n, per_graph = 70001, 10000
    rng = np.random.default_rng(71)
    x = torch.as_tensor(rng.normal(size=(n, 64)), dtype=torch.float32, device=DEV).requires_grad_(True)

    def side(nw, dout, last_per_graph):
        blocks, divs = [x], [1]
        for i, w in enumerate(nw):
            rd = per_graph if (last_per_graph and i == len(nw) - 1) else 1
            blocks.append(torch.as_tensor(rng.normal(size=((n + rd - 1) // rd, w)), dtype=torch.float32, device=DEV))
            divs.append(rd)
        din = 64 + sum(nw)
        wt = torch.as_tensor(rng.normal(size=(din, dout)) / np.sqrt(din), dtype=torch.float32, device=DEV).requires_grad_(True)
        b = torch.as_tensor(rng.normal(size=dout), dtype=torch.float32, device=DEV).requires_grad_(True)
        return blocks, divs, wt, b

    A, B = side(nwa, douts[0], True), side(nwb, douts[1], False)
    ya, yb = F.dense_pair(A[0], A[2], A[3], 4, B[0], B[2], None, 0, row_divs_a=A[1], row_divs_b=B[1], n=n)
    monkeypatch.setenv("NGPDE_DENSE_NO_STREAM2", "1")
    with torch.no_grad():
        ya2, yb2 = F.dense_pair(A[0], A[2], A[3], 4, B[0], B[2], None, 0, row_divs_a=A[1], row_divs_b=B[1], n=n)
    monkeypatch.delenv("NGPDE_DENSE_NO_STREAM2")
    close(ya, ya2.cpu().double().numpy(), rtol=2e-5, atol=2e-6)
    close(yb, yb2.cpu().double().numpy(), rtol=2e-5, atol=2e-6)
    outs, caches, layers = [], [], []
    for (blocks, divs, wt, b), act, bias in ((A, "swish", True), (B, "identity", False)):
        X = np.concatenate([np.repeat(bl.detach().cpu().double().numpy(), rd, axis=0)[:n] for bl, rd in zip(blocks, divs)], axis=1)
        layer = [dict(weight=wt.detach().cpu().double().numpy().T, bias=b.detach().cpu().double().numpy() if bias else None, act=act)]
        yo, cache = O.mlp_forward(layer, X.T)
        outs.append(yo); caches.append(cache); layers.append(layer)
    close(ya, outs[0].T)
    close(yb, outs[1].T)
    Ra, Rb = rng.normal(size=outs[0].shape), rng.normal(size=outs[1].shape)
    ((ya * torch.as_tensor(Ra.T, dtype=torch.float32, device=DEV)).sum() + (yb * torch.as_tensor(Rb.T, dtype=torch.float32, device=DEV)).sum()).backward()
    dxa, gra = O.mlp_backward(layers[0], caches[0], Ra)
    dxb, grb = O.mlp_backward(layers[1], caches[1], Rb)
    close(x.grad, (dxa[:64] + dxb[:64]).T)
    close(A[2].grad, gra[0]["weight"].T, rtol=3e-4)
    close(B[2].grad, grb[0]["weight"].T, rtol=3e-4)
    close(A[3].grad, gra[0]["bias"].reshape(-1), rtol=3e-4)


@pytest.mark.parametrize("widths,dout,acts", [((64, 64, 2), 64, ("swish", "identity")), ((64,), 64, ("tanh", "relu")),
                                              ((64, 3), 40, ("swish", "swish")), ((60, 4), 64, ("swish", "identity"))])
def test_dense_chain2_streaming_forward(widths, dout, acts, monkeypatch):
    # ngpde_dense_chain2_forward: Chain(Dense(. => 64), Dense(64 => dout)) with the intermediate on chip (dense_chain_fwd_kernel),
    # inference (nothing saved) and training (z1 / a1 kept, both pullbacks), against the oracle and the two-launch path; the last
    # case (no 64-wide leading block) takes the two-launch path by itself
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    monkeypatch.delenv("NGPDE_DENSE_NO_STREAM2", raising=False)
    n, per_graph = 70001, 10000
    rng = np.random.default_rng(73)
    blocks, divs = [], []
    for i, w in enumerate(widths):
        rd = per_graph if (len(widths) == 3 and i == 2) else 1
        blocks.append(torch.as_tensor(rng.normal(size=((n + rd - 1) // rd, w)), dtype=torch.float32, device=DEV).requires_grad_(rd == 1 and w >= 60))
        divs.append(rd)
    din = sum(widths)
    wt1 = torch.as_tensor(rng.normal(size=(din, 64)) / np.sqrt(din), dtype=torch.float32, device=DEV).requires_grad_(True)
    b1 = torch.as_tensor(rng.normal(size=64), dtype=torch.float32, device=DEV).requires_grad_(True)
    wt2 = torch.as_tensor(rng.normal(size=(64, dout)) / 8.0, dtype=torch.float32, device=DEV).requires_grad_(True)
    b2 = torch.as_tensor(rng.normal(size=dout), dtype=torch.float32, device=DEV).requires_grad_(True)
    a1c, a2c = ng.layers._act_code(acts[0])[1], ng.layers._act_code(acts[1])[1]
    with torch.no_grad():
        y_inf = F.dense_chain2(blocks, wt1, b1, a1c, wt2, b2, a2c, row_divs=divs, n=n)
        monkeypatch.setenv("NGPDE_DENSE_NO_STREAM2", "1")
        y_two = F.dense_chain2(blocks, wt1, b1, a1c, wt2, b2, a2c, row_divs=divs, n=n)
        monkeypatch.delenv("NGPDE_DENSE_NO_STREAM2")
    y = F.dense_chain2(blocks, wt1, b1, a1c, wt2, b2, a2c, row_divs=divs, n=n)
    assert torch.equal(y.detach(), y_inf)
    close(y, y_two.cpu().double().numpy(), rtol=2e-5, atol=2e-6)
    X = np.concatenate([np.repeat(bl.detach().cpu().double().numpy(), rd, axis=0)[:n] for bl, rd in zip(blocks, divs)], axis=1)
    layers = [dict(weight=wt1.detach().cpu().double().numpy().T, bias=b1.detach().cpu().double().numpy(), act=acts[0]),
              dict(weight=wt2.detach().cpu().double().numpy().T, bias=b2.detach().cpu().double().numpy(), act=acts[1])]
    yo, cache = O.mlp_forward(layers, X.T)
    close(y, yo.T)
    R = rng.normal(size=yo.shape)
    (y * torch.as_tensor(R.T, dtype=torch.float32, device=DEV)).sum().backward()
    dx, gr = O.mlp_backward(layers, cache, R)
    close(wt1.grad, gr[0]["weight"].T, rtol=3e-4)
    close(b1.grad, gr[0]["bias"].reshape(-1), rtol=3e-4)
    close(wt2.grad, gr[1]["weight"].T, rtol=3e-4)
    close(b2.grad, gr[1]["bias"].reshape(-1), rtol=3e-4)
    o = 0
    for bl, w in zip(blocks, widths):
        if bl.requires_grad:
            close(bl.grad, dx[o:o + w].T)
        o += w


# ---- ExplicitEdgeConv ---------------------------------------------------------------------------------------------------

def test_edgeconv_reference_fixture():
    # test/runtests.jl:27-37
    g = ng.GNNGraph([1, 1, 2, 3], [2, 3, 1, 1])
    pos = np.random.default_rng(0).random((3, 3)).astype(np.float32)
    gh = ng.GNNGraph(g, ndata={"x": pos})
    l = ng.ExplicitEdgeConv(ng.Dense(4 + 4 + 3, 5), initialgraph=gh)
    ps, st = ng.setup(0, l)
    assert st == {"ϕ": {}, "graph": gh} and list(ps) == ["weight", "bias"]     # single sub-layer: params un-nested
    u = torch.randn(4, 3, device=DEV)
    y, _ = l(u, ng.to_device(ps, DEV), st)
    assert tuple(y.shape) == (5, 3)
    og = O.Graph([1, 1, 2, 3], [2, 3, 1, 1], ndata={"x": pos.astype(np.float64)})
    yo, _ = O.explicit_edge_conv(u.cpu().double().numpy(), omlp(l.ϕ, ps), og)
    close(y, yo)


@pytest.mark.parametrize("aggr", ["mean", "+", "max"])
def test_edgeconv_parity_and_grads(aggr):
    N, E, h = 300, 2500, 6
    rng = np.random.default_rng(3)
    nd = {"x": rng.random((2, N))}
    g, og = rgraph(N, E, 3, ndata=nd)
    phi = ng.Chain(ng.Dense(2 * h + 2, 16, "tanh"), ng.Dense(16, 9, "tanh"))
    l = ng.ExplicitEdgeConv(phi, initialgraph=g, aggr=aggr)
    ps, st = ng.setup(3, l)
    ps = prep(ps, 3)
    x = torch.randn(h, N, device=DEV, requires_grad=True)
    y, _ = l(x, ps, st)
    yo, c = O.explicit_edge_conv(x.detach().cpu().double().numpy(), omlp(phi, ps), og, aggr)
    if aggr == "max":
        yo = np.where(np.isfinite(yo), yo, 0.0)
        yv = torch.where(torch.isfinite(y), y, torch.zeros_like(y))
        close(yv, yo)
        return
    close(y, yo)
    R = rng.normal(size=yo.shape)
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    gr = O.explicit_edge_conv_backward(c, R)
    names, ogr = mlp_grad_pairs(ps, gr["phi"], phi)
    check_grads(ps, (names, ogr), x, gr["x"])


def test_edgeconv_product_aggregation():
    # aggr = * is one of the five the reference documents (src/layers.jl:49,257,348,441): a node without incoming edges gets
    # the neutral element 1, the pullback gives every message the product of the row's other messages
    N, E, h = 300, 700, 5
    rng = np.random.default_rng(11)
    g, og = rgraph(N, E, 11, ndata={"x": rng.random((2, N))})
    assert (np.bincount(og.t, minlength=N) == 0).any()
    phi = ng.Chain(ng.Dense(2 * h + 2, 12, "tanh"), ng.Dense(12, 7, "tanh"))
    l = ng.ExplicitEdgeConv(phi, initialgraph=g, aggr="*")
    ps, st = ng.setup(5, l)
    ps = prep(ps, 5)
    x = torch.randn(h, N, device=DEV, requires_grad=True)
    y, _ = l(x, ps, st)
    yo, c = O.explicit_edge_conv(x.detach().cpu().double().numpy(), omlp(phi, ps), og, "*")
    close(y, yo)
    R = rng.normal(size=yo.shape)
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    gr = O.explicit_edge_conv_backward(c, R)
    names, ogr = mlp_grad_pairs(ps, gr["phi"], phi)
    check_grads(ps, (names, ogr), x, gr["x"])


@pytest.mark.parametrize("widths,last_act", [((8,), "tanh"), ((16, 8), "tanh"), ((16, 12), "relu"), ((64, 64), "relu")])
def test_product_aggregation_in_the_one_launch_pullback(widths, last_act):
    # aggr = * with gradients on the fused message path (round 5): the one-launch pullback makes a first pass over a tile's edges for the
    # per-target product of the nonzero messages and the number of zeros, then gives every message the product of the others -- g P / m_e,
    # the row's one zero message the product of the rest, nothing where two are zero (a relu last layer produces exact zeros).  Values and
    # all gradients against the float64 oracle; a single Dense phi and a two-layer one (the 64-wide pair takes this kernel too: the
    # specialised ones are for + / mean).
    from ngpde_amd import _lib
    N, h = 600, 6
    rng = np.random.default_rng(17)
    _, s, t = S.closest_pairs_graph(N, 2 * N, seed=9)
    nd = {"x": rng.random((2, N))}
    g, og = ng.GNNGraph(s, t, num_nodes=N, index_base=0, ndata=nd), O.Graph(s, t, num_nodes=N, index_base=0, ndata=nd)
    dims = (2 * h + 2,) + widths
    phi = ng.Chain(*[ng.Dense(dims[l], dims[l + 1], "tanh" if l + 1 < len(widths) else last_act) for l in range(len(widths))]) if len(widths) > 1 else \
        ng.Dense(dims[0], dims[1], last_act)
    l = ng.ExplicitEdgeConv(phi, initialgraph=g, aggr="*")
    assert _lib.load().ngpde_edge_mlp_backward_supported(g.handle((False, None, False)).ptr, widths[0], len(widths) - 1,
                                                         (C.c_int32 * 1)(widths[-1]) if len(widths) > 1 else None, _lib.AGGR["*"]) == 1
    ps, st = ng.setup(5, l)
    ps = prep(ps, 5)
    x = torch.randn(h, N, device=DEV, requires_grad=True)
    y, _ = l(x, ps, st)
    yo, c = O.explicit_edge_conv(x.detach().cpu().double().numpy(), omlp(phi, ps), og, "*")
    if last_act == "relu":
        assert (yo == 0).mean() > 0.2          # rows whose product holds a zero message
    close(y, yo, rtol=2e-4)
    R = rng.normal(size=yo.shape)
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    gr = O.explicit_edge_conv_backward(c, R)
    names, ogr = mlp_grad_pairs(ps, gr["phi"], phi)
    check_grads(ps, (names, ogr), x, gr["x"])


@pytest.mark.parametrize("aggr", ["max", "min"])
@pytest.mark.parametrize("widths", [(8,), (16, 12), (64, 64)])
def test_extremum_aggregation_in_the_one_launch_pullback(aggr, widths):
    # aggr = max / min with gradients on the fused message path (round 5): the pullback's first pass over a tile's edges leaves every
    # target's extremum, the second gives the gradient to the messages equal to it (NNlib's pullback of scatter(max): ties all receive it).
    # Values and all gradients against the float64 oracle.
    from ngpde_amd import _lib
    N, h = 600, 6
    rng = np.random.default_rng(23)
    _, s, t = S.closest_pairs_graph(N, 2 * N, seed=9)
    nd = {"x": rng.random((2, N))}
    g, og = ng.GNNGraph(s, t, num_nodes=N, index_base=0, ndata=nd), O.Graph(s, t, num_nodes=N, index_base=0, ndata=nd)
    dims = (2 * h + 2,) + widths
    phi = ng.Chain(*[ng.Dense(dims[l], dims[l + 1], "tanh") for l in range(len(widths))]) if len(widths) > 1 else ng.Dense(dims[0], dims[1], "tanh")
    l = ng.ExplicitEdgeConv(phi, initialgraph=g, aggr=aggr)
    assert _lib.load().ngpde_edge_mlp_backward_supported(g.handle((False, None, False)).ptr, widths[0], len(widths) - 1,
                                                         (C.c_int32 * 1)(widths[-1]) if len(widths) > 1 else None, _lib.AGGR[aggr]) == 1
    ps, st = ng.setup(5, l)
    ps = prep(ps, 5)
    x = torch.randn(h, N, device=DEV, requires_grad=True)
    y, _ = l(x, ps, st)
    yo, c = O.explicit_edge_conv(x.detach().cpu().double().numpy(), omlp(phi, ps), og, aggr)
    fin = np.isfinite(yo)                       # (a node without incoming edges keeps -inf / +inf, as NNlib's scatter leaves it)
    assert torch.equal(torch.isfinite(y).cpu(), torch.as_tensor(fin))
    close(torch.where(torch.isfinite(y), y, torch.zeros_like(y)), np.where(fin, yo, 0.0), rtol=2e-4)
    R = rng.normal(size=yo.shape) * fin
    (torch.where(torch.isfinite(y), y, torch.zeros_like(y)) * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    gr = O.explicit_edge_conv_backward(c, R)
    names, ogr = mlp_grad_pairs(ps, gr["phi"], phi)
    check_grads(ps, (names, ogr), x, gr["x"])


# ---- VMHConv ---------------------------------------------------------------------------------------------------------------

def test_vmh_reference_fixture_and_parity():
    # test/runtests.jl:39-54
    g = ng.GNNGraph([1, 1, 2, 3], [2, 3, 1, 1])
    pos = np.random.default_rng(0).random((3, 3)).astype(np.float32)
    gh = ng.GNNGraph(g, ndata={"x": pos})
    l = ng.VMHConv(ng.Dense(4 + 4 + 3, 5), ng.Dense(5 + 4, 7), initialgraph=gh)
    ps, st = ng.setup(0, l)
    assert st == {"ϕ": {}, "γ": {}, "graph": gh} and list(ps) == ["ϕ", "γ"]
    u = torch.randn(4, 3, device=DEV)
    y, _ = l(u, ng.to_device(ps, DEV), st)
    assert tuple(y.shape) == (7, 3)
    og = O.Graph([1, 1, 2, 3], [2, 3, 1, 1], ndata={"x": pos.astype(np.float64)})
    yo, _ = O.vmh_conv(u.cpu().double().numpy(), omlp(l.ϕ, ps["ϕ"]), omlp(l.γ, ps["γ"]), og)
    close(y, yo)


def test_vmh_tutorial_shape_parity_and_grads():
    # docs/src/tutorials/VMH.md:75-83: h = 1, pos = 2, 4-layer tanh MLPs 60 wide, message width 40
    N, E = 400, 3000
    rng = np.random.default_rng(5)
    g, og = rgraph(N, E, 5, ndata={"x": rng.random((2, N))})
    phi = ng.Chain(ng.Dense(4, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 40))
    gam = ng.Chain(ng.Dense(41, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 1))
    l = ng.VMHConv(phi, gam, initialgraph=g)
    ps, st = ng.setup(5, l)
    ps = prep(ps, 5)
    x = torch.randn(1, N, device=DEV, requires_grad=True)
    y, _ = l(x, ps, st)
    yo, c = O.vmh_conv(x.detach().cpu().double().numpy(), omlp(phi, ps["ϕ"]), omlp(gam, ps["γ"]), og)
    close(y, yo, rtol=2e-4)
    R = rng.normal(size=yo.shape)
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    gr = O.vmh_conv_backward(c, R)
    n1, o1 = mlp_grad_pairs(ps["ϕ"], gr["phi"], phi)
    n2, o2 = mlp_grad_pairs(ps["γ"], gr["gamma"], gam)
    check_grads(ps, (n1 + n2, o1 + o2), x, gr["x"])


@pytest.mark.parametrize("depth,aggr,N", [(3, "mean", 700), (4, "mean", 1000), (4, "+", 333)])
def test_vmh_deep_message_mlp_fused_pullback_on_a_spatial_graph(depth, aggr, N, monkeypatch):
    # message MLPs of three / four Dense layers (docs/src/tutorials/VMH.md:75-83: 4 => 60 => 60 => 60 => 40) on a graph whose
    # tiles fit the LDS halo: forward in one launch without per-edge saves, pullback in one launch that recomputes the chain
    # (edge_mlp_deep_bwd.hip) -- against the float64 oracle, and against the primitives' pullback (NGPDE_NO_FUSED_EDGE_BWD=1)
    from ngpde_amd import _lib, synth as S
    pts, s, t = S.closest_pairs_graph(N, 3 * N, seed=17 + depth)
    nd = {"x": pts.T.copy()}
    g, og = ng.GNNGraph(s, t, num_nodes=N, index_base=0, ndata=nd), O.Graph(s, t, num_nodes=N, index_base=0, ndata=nd)
    hidden = [ng.Dense(60, 60, "tanh") for _ in range(depth - 2)]
    phi = ng.Chain(ng.Dense(4, 60, "tanh"), *hidden, ng.Dense(60, 40))
    gam = ng.Chain(ng.Dense(41, 60, "tanh"), ng.Dense(60, 1))
    l = ng.VMHConv(phi, gam, aggr=aggr, initialgraph=g)
    ps0, st = ng.setup(7, l)
    lib = _lib.load()
    import ctypes as C
    douts = (C.c_int32 * (depth - 1))(*([60] * (depth - 2) + [40]))
    assert lib.ngpde_edge_mlp_backward_supported(g.handle().ptr, 60, depth - 1, douts, _lib.AGGR[aggr]) == 1
    rng = np.random.default_rng(5)
    x0 = rng.normal(size=(1, N)).astype(np.float32)
    yo, c = O.vmh_conv(x0.astype(np.float64), omlp(phi, prep(ps0, 5)["ϕ"]), omlp(gam, prep(ps0, 5)["γ"]), og, aggr=aggr)
    R = rng.normal(size=yo.shape)
    gr = O.vmh_conv_backward(c, R)
    got = {}
    monkeypatch.setenv("NGPDE_DEEP_EDGE_BWD", "1")       # (below 32 768 nodes the layer takes the primitives' pullback by default)
    for mode in ("fused", "primitives"):
        if mode == "primitives":
            monkeypatch.setenv("NGPDE_NO_FUSED_EDGE_BWD", "1")
        ps = prep(ps0, 5)
        x = torch.as_tensor(x0, device=DEV).requires_grad_(True)
        y, _ = l(x, ps, st)
        close(y, yo, rtol=2e-4)
        (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
        n1, o1 = mlp_grad_pairs(ps["ϕ"], gr["phi"], phi)
        n2, o2 = mlp_grad_pairs(ps["γ"], gr["gamma"], gam)
        check_grads(ps, (n1 + n2, o1 + o2), x, gr["x"])
        got[mode] = [x.grad.clone()] + [p.grad.clone() for _, p in n1]
    for a, b in zip(got["fused"], got["primitives"]):
        assert torch.allclose(a, b, rtol=2e-4, atol=2e-5)


# ---- MPPDEConv: the four variants of the reference's tests ----------------------------------------------------------------

def mppde_case(gh, ogh, dphi_in, dpsi_in, N, seed=0, h=5):
    l = ng.MPPDEConv(ng.Dense(dphi_in, 5), ng.Dense(dpsi_in, 7), initialgraph=gh)
    ps, st = ng.setup(seed, l)
    assert st["graph"] == gh
    hh = torch.randn(h, N, device=DEV)
    y, st2 = l(hh, ng.to_device(ps, DEV), st)
    assert tuple(y.shape) == (7, N) and st2["graph"] == gh
    yo, _ = O.mppde_conv(hh.cpu().double().numpy(), omlp(l.ϕ, ps["ϕ"]), omlp(l.ψ, ps["ψ"]), ogh)
    close(y, yo)


def test_mppde_reference_variants():
    rng = np.random.default_rng(0)
    s, t = [1, 1, 2, 3], [2, 3, 1, 1]
    u, x, th = rng.random((2, 3)), rng.random((3, 3)), rng.random(4)       # Float64 graph data, as in the reference tests
    # with theta (:57-73)
    mppde_case(ng.GNNGraph(s, t, ndata={"u": u, "x": x}, gdata={"θ": th}),
               O.Graph(s, t, ndata={"u": u, "x": x}, gdata={"θ": th}), 5 + 5 + 2 + 3 + 4, 5 + 5 + 4, 3)
    # features in edata (:75-87)
    eu, ex = rng.random((2, 4)), rng.random((3, 4))
    mppde_case(ng.GNNGraph(s, t, edata={"u": eu, "x": ex}, gdata={"θ": th}),
               O.Graph(s, t, edata={"u": eu, "x": ex}, gdata={"θ": th}), 5 + 5 + 2 + 3 + 4, 5 + 5 + 4, 3)
    # batched graph (:89-102)
    g1 = ng.GNNGraph(s, t, ndata={"u": u, "x": x}, gdata={"θ": th})
    o1 = O.Graph(s, t, ndata={"u": u, "x": x}, gdata={"θ": th})
    mppde_case(ng.batch([g1, g1.copy()]), O.batch([o1, o1.copy()]), 19, 14, 6)
    # without theta (:104-120)
    mppde_case(ng.GNNGraph(s, t, ndata={"u": u, "x": x}), O.Graph(s, t, ndata={"u": u, "x": x}), 15, 10, 3)


@pytest.mark.parametrize("aggr", ["mean", "+"])
def test_mppde_batched_parity_and_grads(aggr):
    # C4-like structure at test size: periodic 1-D mesh, 3 neighbours each side, several trajectories, h = 16
    n, G, h = 64, 5, 16
    idx = np.arange(n)
    s = np.concatenate([idx for k in (-3, -2, -1, 1, 2, 3)])
    t = np.concatenate([(idx + k) % n for k in (-3, -2, -1, 1, 2, 3)])
    rng = np.random.default_rng(7)
    gs, ogs = [], []
    for _ in range(G):
        nd = {"u": rng.random((1, n)), "x": (idx / n).reshape(1, n)}
        gd = {"θ": rng.random(2)}
        gs.append(ng.GNNGraph(s, t, num_nodes=n, index_base=0, ndata=nd, gdata=gd))
        ogs.append(O.Graph(s, t, num_nodes=n, index_base=0, ndata=nd, gdata=gd))
    g, og = ng.batch(gs), O.batch(ogs)
    phi = ng.Chain(ng.Dense(2 * h + 2 + 2, 32, "swish"), ng.Dense(32, 24, "swish"))
    psi = ng.Chain(ng.Dense(h + 24 + 2, 32, "swish"), ng.Dense(32, h))
    l = ng.MPPDEConv(phi, psi, initialgraph=g, aggr=aggr)
    ps, st = ng.setup(7, l)
    ps = prep(ps, 7)
    N = n * G
    x = torch.randn(h, N, device=DEV, requires_grad=True)
    y, _ = l(x, ps, st)
    yo, c = O.mppde_conv(x.detach().cpu().double().numpy(), omlp(phi, ps["ϕ"]), omlp(psi, ps["ψ"]), og, aggr)
    close(y, yo)
    R = rng.normal(size=yo.shape)
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    gr = O.mppde_conv_backward(c, R)
    n1, o1 = mlp_grad_pairs(ps["ϕ"], gr["phi"], phi)
    n2, o2 = mlp_grad_pairs(ps["ψ"], gr["psi"], psi)
    check_grads(ps, (n1 + n2, o1 + o2), x, gr["x"])
    # block-diagonal batching == per-graph evaluation
    l1 = ng.MPPDEConv(phi, psi, initialgraph=gs[2], aggr=aggr)
    y1, _ = l1(x.detach()[:, 2 * n:3 * n], ps, ng.setup(0, l1)[1])
    assert torch.allclose(y1, y.detach()[:, 2 * n:3 * n], rtol=1e-5, atol=1e-6)


def test_mppde_structure_mismatch_is_rejected():
    g1 = ng.GNNGraph([1, 2], [2, 1], num_nodes=2, gdata={"θ": np.ones(1)})
    g2 = ng.GNNGraph([1, 2, 3], [2, 3, 1], num_nodes=4, gdata={"θ": np.ones(1)})
    gb = ng.batch([g1, g2])        # 6 nodes, 5 edges, 2 graphs: edges do not split evenly
    l = ng.MPPDEConv(ng.Dense(4 + 4 + 1, 3), ng.Dense(4 + 3 + 1, 2), initialgraph=gb)
    ps, st = ng.setup(0, l)
    with pytest.raises(ng.DimensionMismatch):
        l(torch.randn(4, 6, device=DEV), ng.to_device(ps, DEV), st)


# ---- GNOConv --------------------------------------------------------------------------------------------------------------------

def test_gno_reference_case_and_updategraph():
    # test/runtests.jl:123-151
    gh = ng.rand_graph(10, 6, seed=0)
    rng = np.random.default_rng(0)
    a, xx = rng.random((2, 10)), rng.random((3, 10))
    gh = ng.GNNGraph(gh, ndata={"a": a, "x": xx})
    s, t = gh.edge_index(0)
    cin, cout = 5, 7
    phi = ng.Dense(2 + 2 + 3 + 3, cin * cout)
    l = ng.GNOConv((cin, cout), phi, initialgraph=gh)
    ps, st = ng.setup(0, l)
    assert list(ps) == ["linear", "ϕ"] and list(st) == ["linear", "ϕ", "graph"]
    h = torch.randn(cin, 10, device=DEV)
    psd = ng.to_device(ps, DEV)
    y, st = l(h, psd, st)
    assert tuple(y.shape) == (cout, 10)
    og = O.Graph(s, t, num_nodes=10, index_base=0, ndata={"a": a, "x": xx})
    W, b = ps["linear"]["weight"].double().numpy(), ps["linear"]["bias"].double().numpy()
    yo, _ = O.gno_conv(h.cpu().double().numpy(), omlp(phi, ps["ϕ"]), W, b, og, cin, cout)
    close(y, yo)
    # positional constructor form GNOConv(in, out, ϕ)  (:494)
    assert ng.GNOConv(cin, cout, phi).out_chs == cout
    # swap in a graph that carries the kernel inputs as edge features (:145-150)
    e = rng.random((10, 6))
    ge = ng.GNNGraph(gh, ndata={}, edata=e)
    st = ng.updategraph(st, ge)
    y2, _ = l(h, psd, st)
    assert tuple(y2.shape) == (cout, 10)
    oge = O.Graph(s, t, num_nodes=10, index_base=0, edata=e)
    yo2, _ = O.gno_conv(h.cpu().double().numpy(), omlp(phi, ps["ϕ"]), W, b, oge, cin, cout)
    close(y2, yo2)


@pytest.mark.parametrize("variant", ["reassociated", "materialized", "three-layer", "no-bias"])
def test_gno_parity_and_grads(variant, monkeypatch):
    # "reassociated": K_e h_j = T_j z_e + B2 h_j (ngpde_gno_apply_*), the default whenever the last layer of phi is an
    # identity Dense; "materialized": the literal reshape/batched_mul of src/layers.jl:527-530 (ngpde_gno_contract_*)
    N, E, cin, cout = 120, 900, 6, 5
    rng = np.random.default_rng(11)
    nd = {"a": rng.random((1, N)), "x": rng.random((2, N))}
    g, og = rgraph(N, E, 11, ndata=nd)
    if variant == "materialized":
        monkeypatch.setenv("NGPDE_GNO_MATERIALIZE", "1")
    if variant == "three-layer":
        phi = ng.Chain(ng.Dense(6, 16, "relu"), ng.Dense(16, 12, "tanh"), ng.Dense(12, cin * cout))
    elif variant == "no-bias":
        phi = ng.Chain(ng.Dense(6, 16, "relu"), ng.Dense(16, cin * cout, bias=False))
    else:
        phi = ng.Chain(ng.Dense(6, 16, "relu"), ng.Dense(16, cin * cout))
    l = ng.GNOConv((cin, cout), phi, "tanh", initialgraph=g)
    ps, st = ng.setup(11, l)
    ps = prep(ps, 11)
    x = torch.randn(cin, N, device=DEV, requires_grad=True)
    y, _ = l(x, ps, st)
    W, b = ps["linear"]["weight"].detach().cpu().double().numpy(), ps["linear"]["bias"].detach().cpu().double().numpy()
    yo, c = O.gno_conv(x.detach().cpu().double().numpy(), omlp(phi, ps["ϕ"]), W, b, og, cin, cout, "tanh")
    close(y, yo)
    R = rng.normal(size=yo.shape)
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    gr = O.gno_conv_backward(c, R)
    n1, o1 = mlp_grad_pairs(ps["ϕ"], gr["phi"], phi)
    names = n1 + [("linear.weight", ps["linear"]["weight"]), ("linear.bias", ps["linear"]["bias"])]
    check_grads(ps, (names, o1 + [gr["weight"], gr["bias"]]), x, gr["x"])


@pytest.mark.parametrize("aggr", ["+", "mean", "max"])
def test_gno_message_aggregate_pullback_from_node_gradient(aggr):
    # ngpde_gno_message_backward_from_nodes: the pullback of message + sum / mean aggregation (src/layers.jl:527-534) formed from the
    # node-level gradient inside the per-source launch -- the same arithmetic as ngpde_segment_reduce_backward followed by
    # ngpde_gno_apply_backward, so every output must be bit-identical to the composed path; isolated nodes included; max takes
    # the composed path through the same entry point
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    N, E, cout, kdim = 300, 2600, 32, 16
    rng = np.random.default_rng(5)
    s, t = rng.integers(0, N - 7, size=E), rng.integers(0, N - 7, size=E)          # the last 7 nodes are isolated
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
    h = g.handle((False, None, False))
    mk = lambda *shape: torch.as_tensor(rng.normal(size=shape), dtype=torch.float32, device=DEV)
    leaves = [mk(N, kdim), mk(N, kdim), mk(E, kdim), mk(N, cout * kdim) * 0.2, mk(N, cout)]
    R = mk(N, cout)

    def run(fused):
        xs = [v.clone().requires_grad_(True) for v in leaves]
        P, Q, Et, T, Bh = xs
        if fused:
            agg = F.gno_message_aggregate(P, Q, Et, T, Bh, h, 1, cout, kdim, E, aggr, N)
        else:
            agg = F.segment_reduce(F.gno_message(P, Q, Et, T, Bh, h, 1, cout, kdim, E), h, aggr, N)
        (agg * R).sum().backward()
        return [agg.detach()] + [v.grad for v in xs]

    if not F.gno_message_supported(cout, kdim, 1):
        pytest.skip("the matrix-pipe GNO kernels are switched off (NGPDE_NO_GNO_MFMA): this entry point has no other form")
    a, b = run(True), run(False)
    for u, v, name in zip(a, b, ["agg", "dP", "dQ", "dE", "dT", "dBh"]):
        if name == "dQ" and aggr != "max":     # summed over the source's edges inside the launch: another order of the same terms
            close(u, v.cpu().double().numpy(), rtol=1e-5, atol=1e-5)
        else:
            assert torch.equal(u, v), name
    assert float(a[4].abs().sum()) > 0 and bool(torch.isfinite(a[1]).all())


# ---- GAT-style layer -----------------------------------------------------------------------------------------------------------

@pytest.mark.parametrize("concat", [True, False])
def test_gat_parity_and_grads(concat):
    N, E, H, C, Din = 500, 4000, 4, 16, 64          # config 3 shape at test size: 64 => 4 heads x 16
    rng = np.random.default_rng(13)
    g, og = rgraph(N, E, 13)
    l = ng.GATConv((Din, C), "relu", heads=H, concat=concat, initialgraph=g)
    ps, st = ng.setup(13, l)
    assert tuple(ps["weight"].shape) == (C * H, Din) and tuple(ps["a"].shape) == (2 * C, H)
    ps = prep(ps, 13)
    x = torch.randn(Din, N, device=DEV, requires_grad=True)
    y, _ = l(x, ps, st)
    assert tuple(y.shape) == ((C * H) if concat else C, N)
    p = lambda k: ps[k].detach().cpu().double().numpy()
    yo, c = O.gat_conv(x.detach().cpu().double().numpy(), p("weight"), p("a"), p("bias"), og, H, C, "relu", concat=concat)
    close(y, yo)
    R = rng.normal(size=yo.shape)
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    gr = O.gat_conv_backward(c, R)
    close(x.grad, gr["x"], rtol=5e-4, atol=1e-4, what="dx")
    close(ps["weight"].grad, gr["weight"], rtol=5e-4, atol=5e-4, what="dW")
    close(ps["a"].grad, gr["a"], rtol=5e-4, atol=5e-4, what="da")
    close(ps["bias"].grad, gr["bias"].reshape(-1, 1), rtol=5e-4, atol=5e-4, what="db")


def test_gat_c3_full_size_forward():
    # BASELINE config 3: 4-head x 16 on the C2 graph (16384 nodes, 131072 edges + self loops)
    from ngpde_amd import synth as S
    _, s, t = S.closest_pairs_graph(16384, 65536, seed=2)
    g = ng.GNNGraph(s, t, num_nodes=16384, index_base=0)
    og = O.Graph(s, t, num_nodes=16384, index_base=0)
    l = ng.GATConv((64, 16), "identity", heads=4, initialgraph=g)
    ps, st = ng.setup(3, l)
    psd = ng.to_device(ps, DEV)
    x = torch.as_tensor(S.normal(33, 64 * 16384).reshape(64, 16384).astype(np.float32), device=DEV)
    y, _ = l(x, psd, st)
    yo, c = O.gat_conv(x.cpu().double().numpy(), ps["weight"].double().numpy(), ps["a"].double().numpy(),
                       ps["bias"].double().numpy(), og, 4, 16)
    close(y, yo)
    # attention rows sum to one over each node's incoming edges (incl. the self loop): size-independent property
    assert np.allclose(O.scatter("+", c["alpha"], c["g"].t, 16384), 1.0)


# ---- aggregation edge cases ----------------------------------------------------------------------------------------------------

def test_segment_reduce_edge_cases():
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    g = ng.GNNGraph([1, 2, 2], [2, 1, 1], num_nodes=4)            # nodes 3, 4 isolated; node 1 has two incoming edges
    h = g.handle()
    M = torch.tensor([[1.0, -2.0], [3.0, 5.0], [7.0, 5.0]], device=DEV)   # COO order
    Mp = F.edge_permute(M, h)
    mean = F.segment_reduce(Mp, h, "mean", 4)
    assert torch.equal(mean[2:], torch.zeros(2, 2, device=DEV))   # mean of an empty neighbourhood is 0
    assert torch.allclose(mean[0], torch.tensor([5.0, 5.0], device=DEV)) and torch.allclose(mean[1], M[0])
    mx = F.segment_reduce(Mp, h, "max", 4)
    assert torch.allclose(mx[0], torch.tensor([7.0, 5.0], device=DEV)) and torch.isinf(mx[2]).all()
    # max pullback: every extremal entry receives the gradient (NNlib)
    Mp2 = Mp.clone().requires_grad_(True)
    F.segment_reduce(Mp2, h, "max", 4)[:2].sum().backward()
    back = F.edge_permute(Mp2.grad, h, inverse=True)
    assert back.tolist() == [[1.0, 1.0], [0.0, 1.0], [1.0, 1.0]]
    assert torch.equal(F.edge_permute(Mp, h, inverse=True), M)


@pytest.mark.parametrize("heads", [1, 2, 4])
def test_gat_fused_halo_forward_and_grads(heads, monkeypatch):
    # a local graph whose tiles fit the LDS halo: the softmax aggregation runs on the tile / halo kernel (heads * c == 64);
    # forward, gradients (the pullback consumes the alpha the fused kernel wrote) and agreement with the row-per-wave kernel
    n, C = 203, 64 // heads
    rng = np.random.default_rng(51)
    ss, tt = [], []
    for i in range(n):
        for off in rng.choice(np.arange(-6, 7), size=rng.integers(0, 11), replace=False):
            if off != 0:
                ss.append((i + off) % n); tt.append(i)
    s, t = np.array(ss), np.array(tt)
    g = ng.GNNGraph(s, t, num_nodes=n, index_base=0)
    og = O.Graph(s, t, num_nodes=n, index_base=0)
    l = ng.GATConv((24, C), "tanh", heads=heads, concat=True, initialgraph=g)
    ps, st = ng.setup(51, l)
    ps = prep(ps, 51)
    x = torch.randn(24, n, device=DEV, requires_grad=True)
    y, _ = l(x, ps, st)
    pw = lambda k: ps[k].detach().cpu().double().numpy()
    yo, c = O.gat_conv(x.detach().cpu().double().numpy(), pw("weight"), pw("a"), pw("bias"), og, heads, C, "tanh", concat=True)
    close(y, yo)
    R = rng.normal(size=yo.shape)
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    gr = O.gat_conv_backward(c, R)
    close(x.grad, gr["x"], rtol=5e-4, atol=1e-4)
    for k in ("weight", "a", "bias"):
        close(ps[k].grad, np.asarray(gr[k]).reshape(tuple(ps[k].shape)), rtol=5e-4, atol=2e-4, what=k)
    monkeypatch.setenv("NGPDE_NO_FUSED_GAT", "1")       # the row-per-wave kernel
    with torch.no_grad():
        y2, _ = l(x, ps, st)
    close(y2, y.detach().cpu().double().numpy(), rtol=2e-5, atol=2e-6)


def _local_graph(n, seed, max_deg=10, reach=6):
    rng = np.random.default_rng(seed)
    ss, tt = [], []
    for i in range(n):
        for off in rng.choice(np.arange(-reach, reach + 1), size=rng.integers(0, max_deg + 1), replace=False):
            if off != 0:
                ss.append((i + off) % n); tt.append(i)
    return np.array(ss), np.array(tt)


@pytest.mark.parametrize("heads,act,bias,loops", [(4, "relu", True, True), (4, "tanh", True, True), (4, "identity", False, True),
                                                  (2, "relu", True, False), (1, "swish", True, True), (2, "identity", True, True)])
def test_gat_layer_one_launch_forward_and_pullback(heads, act, bias, loops, monkeypatch):
    # 64 => heads x c = 64 on a graph whose tiles fit the LDS halo: the WHOLE layer is ngpde_gat_layer_forward (logits from
    # the staged input rows, per-head aggregation before the weight) and its two-launch pullback; values and all gradients
    # against the oracle, and agreement with the composed path (Dense + softmax aggregation + bias/activation kernels).
    # loops=False leaves nodes without incoming edges (empty softmax: the row is act(b)).
    monkeypatch.delenv("NGPDE_NO_FUSED_GAT_LAYER", raising=False)
    n, C = 333, 64 // heads
    rng = np.random.default_rng(60 + heads)
    s, t = _local_graph(n, 60 + heads)
    g = ng.GNNGraph(s, t, num_nodes=n, index_base=0)
    og = O.Graph(s, t, num_nodes=n, index_base=0)
    l = ng.GATConv((64, C), act, heads=heads, concat=True, add_self_loops=loops, bias=bias, initialgraph=g)
    ps, st = ng.setup(61, l)
    ps = prep(ps, 61)
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    assert F.gat_layer_supported(l._graph(g).handle(), 64, heads, C)
    x = torch.randn(64, n, device=DEV, requires_grad=True)
    y, _ = l(x, ps, st)
    pw = lambda k: ps[k].detach().cpu().double().numpy() if k in ps else None
    yo, c = O.gat_conv(x.detach().cpu().double().numpy(), pw("weight"), pw("a"), pw("bias"), og, heads, C, act, concat=True,
                       add_self_loops_=loops)
    close(y, yo)
    R = rng.normal(size=yo.shape)
    Rt = torch.as_tensor(R, dtype=torch.float32, device=DEV)
    (y * Rt).sum().backward()
    gr = O.gat_conv_backward(c, R)
    close(x.grad, gr["x"], rtol=5e-4, atol=1e-4, what="dx")
    keys = ("weight", "a", "bias") if bias else ("weight", "a")
    for k in keys:
        close(ps[k].grad, np.asarray(gr[k]).reshape(tuple(ps[k].shape)), rtol=5e-4, atol=2e-4, what=k)
    fused = {k: ps[k].grad.clone() for k in keys}
    fused_dx, fused_y = x.grad.clone(), y.detach().clone()
    # bitwise reproducible (no atomics): a second evaluation gives identical bits
    x.grad = None
    for k in keys:
        ps[k].grad = None
    y1, _ = l(x, ps, st)
    (y1 * Rt).sum().backward()
    assert torch.equal(y1.detach(), fused_y) and torch.equal(x.grad, fused_dx)
    assert all(torch.equal(ps[k].grad, fused[k]) for k in keys)
    # the composed path agrees to rounding
    monkeypatch.setenv("NGPDE_NO_FUSED_GAT_LAYER", "1")
    x.grad = None
    for k in keys:
        ps[k].grad = None
    y2, _ = l(x, ps, st)
    (y2 * Rt).sum().backward()
    close(y2, fused_y.cpu().double().numpy(), rtol=2e-5, atol=2e-6)
    close(x.grad, fused_dx.cpu().double().numpy(), rtol=1e-4, atol=2e-5)
    for k in keys:
        close(ps[k].grad, fused[k].cpu().double().numpy(), rtol=1e-4, atol=5e-5, what=k)


@pytest.mark.parametrize("heads", [4, 2, 1])
def test_gat_layer_and_solver_with_long_rows(heads, monkeypatch):
    # rows of up to 24 entries + the self loop (a slot list holds 32): the second half of every lane's entry pair (entries
    # 16 .. 31) and halos near their capacity -- the layer against the oracle, the device-resident solver against the generic one
    monkeypatch.delenv("NGPDE_NO_FUSED_GAT_LAYER", raising=False)
    monkeypatch.delenv("NGPDE_NO_PERSISTENT", raising=False)
    n, C = 420, 64 // heads
    s, t = _local_graph(n, 130 + heads, max_deg=24, reach=14)
    deg = np.bincount(t, minlength=n)
    assert deg.max() >= 20
    g = ng.GNNGraph(s, t, num_nodes=n, index_base=0)
    og = O.Graph(s, t, num_nodes=n, index_base=0)
    l = ng.GATConv((64, C), "tanh", heads=heads, concat=True, initialgraph=g)
    ps, st = ng.setup(131, l)
    ps = prep(ps, 131)
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    assert F.gat_layer_supported(l._graph(g).handle(), 64, heads, C)
    x = torch.randn(64, n, device=DEV, requires_grad=True)
    y, _ = l(x, ps, st)
    pw = lambda k: ps[k].detach().cpu().double().numpy()
    yo, c = O.gat_conv(x.detach().cpu().double().numpy(), pw("weight"), pw("a"), pw("bias"), og, heads, C, "tanh", concat=True)
    close(y, yo)
    R = np.random.default_rng(132).normal(size=yo.shape)
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    gr = O.gat_conv_backward(c, R)
    close(x.grad, gr["x"], rtol=5e-4, atol=1e-4, what="dx")
    for k in ("weight", "a", "bias"):
        close(ps[k].grad, np.asarray(gr[k]).reshape(tuple(ps[k].shape)), rtol=5e-4, atol=3e-4, what=k)

    def run(resident):
        if resident:
            monkeypatch.delenv("NGPDE_NO_PERSISTENT", raising=False)
        else:
            monkeypatch.setenv("NGPDE_NO_PERSISTENT", "1")
        node = ng.NeuralODE(l, solver="tsit5", n_steps=2, dt=0.05)
        _, st2 = ng.setup(131, node)
        p2 = {k: v.detach().clone().requires_grad_(True) for k, v in ps.items()}
        u = x.detach().clone().requires_grad_(True)
        uT, _ = node(u, p2, st2)
        uT.sum().backward()
        return uT.detach(), u.grad, [p for pool in node._plans.values() for p in pool]
    a, b = run(True), run(False)
    assert a[2] and "gat" in a[2][0].flags() and not a[2][0].fault() and not b[2]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_gat_as_ode_right_hand_side_generic_solver_path():
    # BASELINE config 3 "as ODE RHS": du/dt = GATConv(64 => 4 x 16, concat)(u), stepped by NeuralODE's generic path (explicit
    # RK through the layer's kernels, gradients by autograd through every stage) against rk_solve / rk_adjoint of the oracle
    N, E, H, C = 300, 2400, 4, 16
    rng = np.random.default_rng(17)
    g, og = rgraph(N, E, 17)
    l = ng.GATConv((H * C, C), "tanh", heads=H, concat=True, initialgraph=g)
    node = ng.NeuralODE(l, solver="tsit5", n_steps=2, dt=0.05)
    ps, st = ng.setup(17, node)
    ps = prep(ps, 17)
    u0 = torch.randn(H * C, N, device=DEV, requires_grad=True)
    uT, _ = node(u0, ps, st)
    pw = lambda k: ps[k].detach().cpu().double().numpy()
    W, a, b = pw("weight"), pw("a"), pw("bias")

    def rhs(u):
        return O.gat_conv(u, W, a, b, og, H, C, "tanh", concat=True)

    acc = dict(weight=np.zeros_like(W), a=np.zeros_like(a), bias=np.zeros_like(b))

    def vjp(cache, kbar):
        gr = O.gat_conv_backward(cache, kbar)
        return gr["x"], gr

    def accumulate(gr):
        for k in acc:
            acc[k] += np.asarray(gr[k]).reshape(acc[k].shape)

    uTo, tape = O.rk_solve(rhs, u0.detach().cpu().double().numpy(), O.TABLEAUS["tsit5"], 0.05, 2)
    close(uT, uTo, rtol=2e-4)
    du0 = O.rk_adjoint(vjp, tape, np.ones_like(uTo), O.TABLEAUS["tsit5"], 0.05, accumulate)
    uT.sum().backward()
    close(u0.grad, du0, rtol=5e-4, atol=1e-4)
    for k in acc:
        close(ps[k].grad, acc[k], rtol=5e-4, atol=5e-4, what=k)


def _rk_oracle(rhs_fwd, rhs_bwd, names, shapes, u0, tab, dt, steps):
    """u(T), du0 and summed parameter cotangents of loss = sum(u(T)) through the oracle's rk_solve / rk_adjoint"""
    acc = {k: np.zeros(shapes[k]) for k in names}

    def vjp(cache, kbar):
        gr = rhs_bwd(cache, kbar)
        return gr["x"], gr

    def accumulate(gr):
        for k in names:
            acc[k] += np.asarray(gr[k]).reshape(acc[k].shape)
    uT, tape = O.rk_solve(rhs_fwd, u0, O.TABLEAUS[tab], dt, steps)
    du0 = O.rk_adjoint(vjp, tape, np.ones_like(uT), O.TABLEAUS[tab], dt, accumulate)
    return uT, du0, acc


@pytest.mark.parametrize("solver,steps", [("tsit5", 2), ("euler", 3)])
def test_gat_one_launch_layer_as_ode_right_hand_side(solver, steps, monkeypatch):
    monkeypatch.delenv("NGPDE_NO_FUSED_GAT_LAYER", raising=False)       # (the suite may run under that switch)
    # BASELINE config 3 "as ODE RHS" on a graph whose tiles fit the LDS halo: every right-hand-side evaluation is the one-launch
    # GAT layer, every Runge-Kutta combination (and every combination of the discrete adjoint) one ngpde_rk_stage_combine launch
    n, H, C_ = 300, 4, 16
    s, t = _local_graph(n, 77)
    g = ng.GNNGraph(s, t, num_nodes=n, index_base=0)
    og = O.Graph(s, t, num_nodes=n, index_base=0)
    l = ng.GATConv((64, C_), "tanh", heads=H, concat=True, initialgraph=g)
    node = ng.NeuralODE(l, solver=solver, n_steps=steps, dt=0.05)
    ps, st = ng.setup(77, node)
    ps = prep(ps, 77)
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    assert F.gat_layer_supported(l._graph(g).handle(), 64, H, C_)
    u0 = torch.randn(64, n, device=DEV, requires_grad=True)
    uT, _ = node(u0, ps, st)
    pw = lambda k: ps[k].detach().cpu().double().numpy()
    W, a, b = pw("weight"), pw("a"), pw("bias")
    uTo, du0, acc = _rk_oracle(lambda u: O.gat_conv(u, W, a, b, og, H, C_, "tanh", concat=True), O.gat_conv_backward,
                               ("weight", "a", "bias"), dict(weight=W.shape, a=a.shape, bias=b.shape),
                               u0.detach().cpu().double().numpy(), solver, 0.05, steps)
    close(uT, uTo, rtol=2e-4)
    uT.sum().backward()
    close(u0.grad, du0, rtol=5e-4, atol=1e-4)
    for k in acc:
        close(ps[k].grad, acc[k], rtol=5e-4, atol=5e-4, what=k)
    # inference call (no tape) gives the same bits
    with torch.no_grad():
        uT2, _ = node(u0, ps, st)
    assert torch.equal(uT2, uT.detach())


@pytest.mark.parametrize("heads,act,bias,loops,solver,steps", [(4, "relu", True, True, "tsit5", 3), (4, "tanh", True, True, "tsit5", 2),
                                                               (2, "identity", False, True, "tsit5", 2), (1, "swish", True, True, "euler", 4),
                                                               (2, "relu", True, False, "tsit5", 2), (4, "leakyrelu", True, True, "euler", 1)])
def test_gat_device_resident_solver_equals_the_generic_solver(heads, act, bias, loops, solver, steps, monkeypatch):
    # NeuralODE(GATConv 64 => heads x c = 64) on a graph whose tiles fit the LDS halo: ngpde_node_gat_* -- ONE persistent launch for the
    # solve, ONE for the discrete adjoint, tiles synchronised by per-tile phase flags -- against the generic solver (every stage
    # the one-launch layer, every combination ngpde_rk_stage_combine; NGPDE_NO_PERSISTENT=1).  Same per-tile code, same
    # coefficients, same order: u(T) and du0 bit for bit; parameter gradients to rounding (summed per tile over the whole adjoint).
    monkeypatch.delenv("NGPDE_NO_FUSED_GAT_LAYER", raising=False)
    n, C_ = 700, 64 // heads
    s, t = _local_graph(n, 90 + heads)
    g = ng.GNNGraph(s, t, num_nodes=n, index_base=0)
    l = ng.GATConv((64, C_), act, heads=heads, concat=True, add_self_loops=loops, bias=bias, initialgraph=g)
    ps0, _ = ng.setup(91, l)
    ps0 = prep(ps0, 91)
    u0 = torch.randn(64, n, device=DEV)
    R = torch.randn(64, n, device=DEV)

    def run(resident):
        if resident:
            monkeypatch.delenv("NGPDE_NO_PERSISTENT", raising=False)
        else:
            monkeypatch.setenv("NGPDE_NO_PERSISTENT", "1")
        node = ng.NeuralODE(l, solver=solver, n_steps=steps, dt=0.05)
        _, st = ng.setup(91, node)
        ps = {k: v.detach().clone().requires_grad_(True) for k, v in ps0.items()}
        u = u0.clone().requires_grad_(True)
        uT, _ = node(u, ps, st)
        (uT * R).sum().backward()
        plans = [p for pool in node._plans.values() for p in pool]
        with torch.no_grad():
            uT2, _ = node(u, ps, st)             # forward-only plan (two ping-pong slots instead of a tape)
        assert torch.equal(uT2, uT.detach())
        return uT.detach(), u.grad, {k: v.grad for k, v in ps.items()}, plans

    a = run(True)
    assert a[3] and all("gat" in p.flags() and not p.fault() for p in a[3]), [p.flags() for p in a[3]]
    b = run(False)
    assert not b[3]
    assert torch.equal(a[0], b[0]), "u(T)"
    assert torch.equal(a[1], b[1]), "du0"
    for k in a[2]:
        close(a[2][k], b[2][k].cpu().double().numpy(), rtol=2e-5, atol=1e-5, what=k)
    if heads == 4 and act == "relu":      # Chain(GATConv(...)) is the same right-hand side: the same plan kind, the same bits
        monkeypatch.delenv("NGPDE_NO_PERSISTENT", raising=False)
        nodec = ng.NeuralODE(ng.Chain(l), solver=solver, n_steps=steps, dt=0.05)
        _, stc = ng.setup(91, nodec)
        psc = {"layer_1": {k: v.detach().clone().requires_grad_(True) for k, v in ps0.items()}}
        uc = u0.clone().requires_grad_(True)
        uTc, _ = nodec(uc, psc, stc)
        (uTc * R).sum().backward()
        plansc = [p for pool in nodec._plans.values() for p in pool]
        assert plansc and all("gat" in p.flags() for p in plansc)
        assert torch.equal(uTc.detach(), a[0]) and torch.equal(uc.grad, a[1])


@pytest.mark.parametrize("n", [5, 31, 33, 95])
def test_gat_device_resident_solver_with_partial_tiles(n, monkeypatch):
    # node counts that are not multiples of the 32-row tile: the padding rows' threads address node 0, and a store of theirs that is
    # not masked lands in node 0's rows (found by tools/fuzz_gat_node.py at n = 31: the stage derivatives of node 0 were
    # overwritten by a padding row of the same tile).  Tsit5 x 2 with relu and a bias, against the generic solver, bit for bit.
    monkeypatch.delenv("NGPDE_NO_FUSED_GAT_LAYER", raising=False)
    s, t = _local_graph(n, 140 + n, max_deg=4, reach=min(3, max(1, n // 2 - 1)))
    if s.size == 0:
        s, t = np.array([0]), np.array([1])
    g = ng.GNNGraph(s, t, num_nodes=n, index_base=0)
    l = ng.GATConv((64, 16), "relu", heads=4, initialgraph=g)
    ps0, _ = ng.setup(141, l)
    ps0 = prep(ps0, 141)
    u0 = torch.randn(64, n, device=DEV)
    R = torch.randn(64, n, device=DEV)
    outs = []
    for resident in (True, False):
        if resident:
            monkeypatch.delenv("NGPDE_NO_PERSISTENT", raising=False)
        else:
            monkeypatch.setenv("NGPDE_NO_PERSISTENT", "1")
        node = ng.NeuralODE(l, solver="tsit5", n_steps=2, dt=0.05)
        _, st = ng.setup(141, node)
        ps = {k: v.detach().clone().requires_grad_(True) for k, v in ps0.items()}
        u = u0.clone().requires_grad_(True)
        uT, _ = node(u, ps, st)
        (uT * R).sum().backward()
        outs.append((uT.detach(), u.grad, {k: v.grad for k, v in ps.items()}, [p for pool in node._plans.values() for p in pool]))
    a, b = outs
    assert a[3] and "gat" in a[3][0].flags() and not a[3][0].fault() and not b[3]
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        close(a[2][k], b[2][k].cpu().double().numpy(), rtol=2e-5, atol=1e-5, what=k)


@pytest.mark.parametrize("members", [2, 3])
def test_gat_device_resident_solver_on_a_batch_of_identical_structures(members, monkeypatch):
    # batch([g] * K) as ODE state (test/runtests.jl:89-102): ngpde_node_gat_create_batch on ONE member, two members at a time per
    # workgroup (an odd last member alone) -- every member's u(T) and du0 bit for bit those of the generic solver on the
    # block-diagonal graph, parameter gradients (summed over the members) to rounding
    monkeypatch.delenv("NGPDE_NO_FUSED_GAT_LAYER", raising=False)
    n, H, C_ = 500, 4, 16
    s, t = _local_graph(n, 97)
    g1 = ng.GNNGraph(s, t, num_nodes=n, index_base=0)
    g = ng.batch([g1] * members)
    l = ng.GATConv((64, C_), "tanh", heads=H, initialgraph=g)
    ps0, _ = ng.setup(97, l)
    ps0 = prep(ps0, 97)
    u0 = torch.randn(64, n * members, device=DEV)
    R = torch.randn(64, n * members, device=DEV)

    def run(resident):
        if resident:
            monkeypatch.delenv("NGPDE_NO_PERSISTENT", raising=False)
        else:
            monkeypatch.setenv("NGPDE_NO_PERSISTENT", "1")
        node = ng.NeuralODE(l, solver="tsit5", n_steps=2, dt=0.05)
        _, st = ng.setup(97, node)
        ps = {k: v.detach().clone().requires_grad_(True) for k, v in ps0.items()}
        u = u0.clone().requires_grad_(True)
        uT, _ = node(u, ps, st)
        (uT * R).sum().backward()
        return uT.detach(), u.grad, {k: v.grad for k, v in ps.items()}, [p for pool in node._plans.values() for p in pool]

    a = run(True)
    assert a[3] and all("gat" in p.flags() and p.members == members and not p.fault() for p in a[3])
    b = run(False)
    assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
    for k in a[2]:
        close(a[2][k], b[2][k].cpu().double().numpy(), rtol=2e-5, atol=1e-5, what=k)


def test_gat_device_resident_solver_replays_are_bit_identical_at_c3_size(monkeypatch):
    # the race screen: 12 solves + adjoints of BASELINE config 3 as ODE right-hand side (512 tiles, 50 Tsit5 steps = 300 hand-offs
    # per direction), every output of every replay equal to the first, bit for bit; no launch gave up waiting
    monkeypatch.delenv("NGPDE_NO_PERSISTENT", raising=False)
    monkeypatch.delenv("NGPDE_NO_FUSED_GAT_LAYER", raising=False)
    _, s, t = S.closest_pairs_graph(16384, 65536, seed=2)
    g = ng.GNNGraph(s, t, num_nodes=16384, index_base=0)
    l = ng.GATConv((64, 16), "relu", heads=4, initialgraph=g)
    node = ng.NeuralODE(l, solver="tsit5", n_steps=50, dt=0.02)
    ps, st = ng.setup(5, node)
    ps = prep(ps, 5)
    u = torch.randn(64, 16384, device=DEV, requires_grad=True)
    first = None
    for rep in range(12):
        for v in list(ps.values()) + [u]:
            v.grad = None
        uT, _ = node(u, ps, st)
        uT.sum().backward()
        got = [uT.detach().clone(), u.grad.clone()] + [ps[k].grad.clone() for k in sorted(ps)]
        if first is None:
            first = got
            assert all(bool(torch.isfinite(x).all()) for x in got)
        else:
            assert all(torch.equal(x, y) for x, y in zip(got, first)), f"replay {rep} differs"
    plans = [p for pool in node._plans.values() for p in pool]
    assert plans and all("gat" in p.flags() and not p.fault() for p in plans)


def test_gat_device_resident_solver_abort_poisons_outputs_and_the_plan_refuses_further_work(monkeypatch):
    monkeypatch.delenv("NGPDE_NO_PERSISTENT", raising=False)
    monkeypatch.delenv("NGPDE_NO_FUSED_GAT_LAYER", raising=False)
    n = 700
    s, t = _local_graph(n, 95)
    g = ng.GNNGraph(s, t, num_nodes=n, index_base=0)
    l = ng.GATConv((64, 16), "relu", heads=4, initialgraph=g)
    node = ng.NeuralODE(l, solver="tsit5", n_steps=2, dt=0.05)
    ps, st = ng.setup(95, node)
    ps = prep(ps, 95)
    u = torch.randn(64, n, device=DEV)
    monkeypatch.setenv("NGPDE_DEBUG_FORCE_ABORT", "1")
    with torch.no_grad():
        uT, _ = node(u, ps, st)
    torch.cuda.synchronize()
    monkeypatch.delenv("NGPDE_DEBUG_FORCE_ABORT")
    plans = [p for pool in node._plans.values() for p in pool]
    assert plans and plans[0].fault()
    assert bool(torch.isnan(uT).all())
    with pytest.raises(ng._lib.NgpdeError) as e:
        with torch.no_grad():
            node(u, ps, st)
    assert e.value.code == ng._lib.ERR_STATE and "gave up waiting" in str(e.value)


def test_captured_generic_solve_replays_and_follows_parameter_updates(monkeypatch):
    monkeypatch.delenv("NGPDE_NO_FUSED_GAT_LAYER", raising=False)
    monkeypatch.setenv("NGPDE_NO_PERSISTENT", "1")     # the GENERIC solver is the subject (a lone GATConv would take the device-resident one)
    # NeuralODE(..., capture=True): the whole stepping loop and the whole discrete adjoint are HIP graphs captured at the first
    # call; replays must reproduce the eager path bit for bit, for new inputs and after an in-place parameter update
    n, H, C_ = 300, 4, 16
    s, t = _local_graph(n, 78)
    g = ng.GNNGraph(s, t, num_nodes=n, index_base=0)
    l = ng.GATConv((64, C_), "tanh", heads=H, concat=True, initialgraph=g)
    eager = ng.NeuralODE(l, solver="tsit5", n_steps=2, dt=0.05)
    captured = ng.NeuralODE(l, solver="tsit5", n_steps=2, dt=0.05, capture=True)
    ps, st = ng.setup(78, eager)
    ps = prep(ps, 78)

    def run(node, u):
        for v in ps.values():
            v.grad = None
        u = u.clone().requires_grad_(True)
        uT, _ = node(u, ps, st)
        (uT * uT).sum().backward()
        return uT.detach().clone(), u.grad.clone(), {k: v.grad.clone() for k, v in ps.items()}

    for trial in range(3):
        u = torch.randn(64, n, device=DEV)
        if trial == 2:
            with torch.no_grad():
                ps["bias"].mul_(0.5)
                ps["weight"].add_(0.01)
        ref = run(eager, u)
        got = run(captured, u)
        assert torch.equal(got[0], ref[0]) and torch.equal(got[1], ref[1]), f"trial {trial}"
        for k in ps:
            assert torch.equal(got[2][k], ref[2][k]), (trial, k)
    assert len(captured._captured) == 1
    with torch.no_grad():                       # inference capture (no tape) next to the training capture
        u = torch.randn(64, n, device=DEV)
        a1, _ = captured(u, ps, st)
        a2, _ = eager(u, ps, st)
    assert torch.equal(a1, a2) and len(captured._captured) == 2


def test_captured_solve_follows_updategraph_in_a_container_and_guards_its_single_tape(monkeypatch):
    monkeypatch.setenv("NGPDE_NO_PERSISTENT", "1")     # the captured GENERIC solver is the subject (Chain(GATConv) alone takes the device-resident one)
    # (i) a container right-hand side keeps its graphs in st["layer_k"]["graph"]: updategraph(st, g2) must capture anew instead of
    # replaying launches with the old graph's arrays baked in; (ii) a captured solve holds ONE tape: the backward of a solve
    # whose tape a later forward replaced raises instead of returning the other solve's gradients
    n, H, C_ = 300, 4, 16
    s1, t1 = _local_graph(n, 81)
    s2, t2 = _local_graph(n, 82)
    g1 = ng.GNNGraph(s1, t1, num_nodes=n, index_base=0)
    g2 = ng.GNNGraph(s2, t2, num_nodes=n, index_base=0)
    mk = lambda g: ng.Chain(ng.GATConv((64, C_), "tanh", heads=H, concat=True, initialgraph=g))
    eager = ng.NeuralODE(mk(g1), solver="euler", n_steps=2, dt=0.05)
    captured = ng.NeuralODE(mk(g1), solver="euler", n_steps=2, dt=0.05, capture=True)
    ps, st = ng.setup(81, eager)
    ps = prep(ps, 81)
    u = torch.randn(64, n, device=DEV)
    with torch.no_grad():
        a1, _ = captured(u, ps, st)
        r1, _ = eager(u, ps, st)
        st2 = ng.updategraph(st, g2)
        a2, _ = captured(u, ps, st2)
        r2, _ = eager(u, ps, st2)
    assert torch.equal(a1, r1) and torch.equal(a2, r2) and not torch.equal(r1, r2)
    assert len(captured._captured) == 2
    ua, ub = torch.randn(64, n, device=DEV, requires_grad=True), torch.randn(64, n, device=DEV, requires_grad=True)
    ya, _ = captured(ua, ps, st)
    yb, _ = captured(ub, ps, st)
    with pytest.raises(_lib.NgpdeError, match="ONE tape"):
        (ya.sum() + yb.sum()).backward()
    yc, _ = captured(ua, ps, st)          # forward then backward, in order: fine
    yc.sum().backward()


def test_vmh_as_ode_right_hand_side():
    # docs/src/tutorials/VMH.md:85-89: NeuralODE(VMHConv(phi, gamma)) -- the layer maps h features to h features and is
    # integrated as du/dt; values and all gradients of two Tsit5 steps against the oracle's rk_solve / rk_adjoint
    N, E, h = 250, 1800, 3
    rng = np.random.default_rng(23)
    g, og = rgraph(N, E, 23, ndata={"x": rng.random((2, N))})
    phi = ng.Chain(ng.Dense(2 * h + 2, 24, "tanh"), ng.Dense(24, 16, "tanh"))
    gam = ng.Chain(ng.Dense(h + 16, 20, "tanh"), ng.Dense(20, h))
    l = ng.VMHConv(phi, gam, initialgraph=g)
    node = ng.NeuralODE(l, solver="tsit5", n_steps=2, dt=0.1)
    ps, st = ng.setup(23, node)
    ps = prep(ps, 23)
    u0 = torch.randn(h, N, device=DEV, requires_grad=True)
    uT, _ = node(u0, ps, st)
    ophi, ogam = omlp(phi, ps["ϕ"]), omlp(gam, ps["γ"])
    tab, dt = O.TABLEAUS["tsit5"], 0.1
    gphi = [dict(weight=np.zeros_like(L["weight"]), bias=np.zeros_like(L["bias"])) for L in ophi]
    ggam = [dict(weight=np.zeros_like(L["weight"]), bias=np.zeros_like(L["bias"])) for L in ogam]

    def vjp(cache, kbar):
        gr = O.vmh_conv_backward(cache, kbar)
        return gr["x"], gr

    def accumulate(gr):
        for dst, src in ((gphi, gr["phi"]), (ggam, gr["gamma"])):
            for d_, s_ in zip(dst, src):
                d_["weight"] += s_["weight"]
                d_["bias"] += np.asarray(s_["bias"]).reshape(d_["bias"].shape)
    uTo, tape = O.rk_solve(lambda u: O.vmh_conv(u, ophi, ogam, og), u0.detach().cpu().double().numpy(), tab, dt, 2)
    close(uT, uTo, rtol=2e-4)
    du0 = O.rk_adjoint(vjp, tape, np.ones_like(uTo), tab, dt, accumulate)
    uT.sum().backward()
    n1, o1 = mlp_grad_pairs(ps["ϕ"], gphi, phi)
    n2, o2 = mlp_grad_pairs(ps["γ"], ggam, gam)
    check_grads(ps, (n1 + n2, o1 + o2), u0, du0)


# ---- fused message path (one launch) against the primitives and the oracle -------------------------------------------------

@pytest.mark.parametrize("aggr", ["mean", "+", "max", "min", "*"])
def test_fused_message_path_matches_primitives_and_oracle(aggr, monkeypatch):
    monkeypatch.delenv("NGPDE_NO_FUSED_EDGE", raising=False)      # (the suite may run under that switch)
    # MPPDE shape of BASELINE config 4 at test size: h = 64, phi 132 => 64 => 64 swish, periodic mesh, 3 trajectories
    n, G, h = 256, 3, 64
    idx = np.arange(n)
    s = np.concatenate([idx for k in (-3, -2, -1, 1, 2, 3)])
    t = np.concatenate([(idx + k) % n for k in (-3, -2, -1, 1, 2, 3)])
    rng = np.random.default_rng(21)
    gs, ogs = [], []
    for _ in range(G):
        nd = {"u": rng.random((1, n)), "x": (idx / n).reshape(1, n)}
        gd = {"θ": rng.random(2)}
        gs.append(ng.GNNGraph(s, t, num_nodes=n, index_base=0, ndata=nd, gdata=gd))
        ogs.append(O.Graph(s, t, num_nodes=n, index_base=0, ndata=nd, gdata=gd))
    g, og = ng.batch(gs), O.batch(ogs)
    phi = ng.Chain(ng.Dense(2 * h + 2 + 2, 64, "swish"), ng.Dense(64, 64, "swish"))
    psi = ng.Chain(ng.Dense(h + 64 + 2, 64, "swish"), ng.Dense(64, h))
    l = ng.MPPDEConv(phi, psi, initialgraph=g, aggr=aggr)
    ps, st = ng.setup(21, l)
    ps = prep(ps, 21)
    x = torch.randn(h, n * G, device=DEV)
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    fh = g.handle((False, None, False))
    assert F.edge_mlp_supported(fh, 64, [64])
    with torch.no_grad():
        y_fused, _ = l(x, ps, st)
        monkeypatch.setenv("NGPDE_NO_FUSED_EDGE", "1")
        y_prim, _ = l(x, ps, st)
        monkeypatch.delenv("NGPDE_NO_FUSED_EDGE")
    yo, c = O.mppde_conv(x.cpu().double().numpy(), omlp(phi, ps["ϕ"]), omlp(psi, ps["ψ"]), og, aggr)
    close(y_fused, yo)
    close(y_prim, yo)
    close(y_fused, y_prim.cpu().double().numpy(), rtol=2e-5, atol=2e-6)
    y2, _ = l(x, ps, st)
    assert torch.equal(y2.detach(), y_fused) or aggr in ("max", "min")       # fused path is bitwise reproducible
    if aggr in ("mean", "+"):                                                  # training: fused forward, primitive pullback
        xg = x.clone().requires_grad_(True)
        yg, _ = l(xg, ps, st)
        R = rng.normal(size=yo.shape)
        (yg * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
        gr = O.mppde_conv_backward(c, R)
        n1, o1 = mlp_grad_pairs(ps["ϕ"], gr["phi"], phi)
        n2, o2 = mlp_grad_pairs(ps["ψ"], gr["psi"], psi)
        check_grads(ps, (n1 + n2, o1 + o2), xg, gr["x"])


@pytest.mark.parametrize("widths,acts", [((32,), ("tanh",)), ((16, 24, 8), ("swish", "relu", "identity")),
                                         ((60, 60, 60, 40), ("tanh", "tanh", "tanh", "identity"))])
def test_fused_message_path_layer_counts_and_ragged_rows(widths, acts):
    # 0, 2 and 3 Dense layers after the first (the last shape is the VMH tutorial's message MLP, VMH.md:75-83), rows of
    # varying degree (0..9), a ragged last tile, per-edge features in the first layer -- fused kernel vs oracle, fwd + grads
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    n, h = 203, 6
    rng = np.random.default_rng(31)
    ss, tt = [], []
    for i in range(n):
        for off in rng.choice(np.arange(-5, 6), size=rng.integers(0, 10), replace=False):
            if off != 0:
                ss.append((i + off) % n); tt.append(i)
    s, t = np.array(ss), np.array(tt)
    nd = {"x": rng.random((2, n))}
    g = ng.GNNGraph(s, t, num_nodes=n, index_base=0, ndata=nd)
    og = O.Graph(s, t, num_nodes=n, index_base=0, ndata=nd)
    dims = [2 * h + 2] + list(widths)
    phi = ng.Chain(*[ng.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(widths))]) if len(widths) > 1 \
        else ng.Dense(dims[0], dims[1], acts[0])
    assert F.edge_mlp_supported(g.handle((False, None, False)), widths[0], list(widths[1:]))
    l = ng.ExplicitEdgeConv(phi, initialgraph=g, aggr="mean")
    ps, st = ng.setup(31, l)
    ps = prep(ps, 31)
    x = torch.randn(h, n, device=DEV, requires_grad=True)
    y, _ = l(x, ps, st)
    yo, c = O.explicit_edge_conv(x.detach().cpu().double().numpy(), omlp(phi, ps), og, "mean")
    close(y, yo)
    R = rng.normal(size=yo.shape)
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    gr = O.explicit_edge_conv_backward(c, R)
    names, ogr = mlp_grad_pairs(ps, gr["phi"], phi)
    check_grads(ps, (names, ogr), x, gr["x"])


@pytest.mark.parametrize("acts", [("swish", "swish"), ("swish", "identity"), ("relu", "relu"), ("tanh", "tanh"), ("tanh", "identity")])
@pytest.mark.parametrize("aggr", ["mean", "+"])
def test_pipelined_message_kernel_64_equals_general_kernel(acts, aggr, monkeypatch):
    # edge_mlp64.hip (software-pipelined specialisation for the 64-wide two-layer message MLP, BASELINE config 4's shape) against
    # the general fused kernel (NGPDE_NO_EDGE64=1) -- same operations in the same order, so bit for bit -- and the oracle, on rows
    # of varying degree (0..9: tiles of 0 / 1 / several 64-edge chunks with ragged tails), a ragged last tile, isolated nodes
    monkeypatch.delenv("NGPDE_NO_FUSED_EDGE", raising=False)
    monkeypatch.delenv("NGPDE_NO_EDGE64", raising=False)
    n, h = 1003, 64
    rng = np.random.default_rng(47)
    ss, tt = [], []
    for i in range(n):
        if 300 <= i < 340:                      # a run of isolated nodes: a whole tile without edges
            continue
        for off in rng.choice(np.arange(-5, 6), size=rng.integers(0, 10), replace=False):
            if off != 0:
                ss.append((i + off) % n); tt.append(i)
    s, t = np.array(ss), np.array(tt)
    nd = {"x": rng.random((2, n))}
    g = ng.GNNGraph(s, t, num_nodes=n, index_base=0, ndata=nd)
    og = O.Graph(s, t, num_nodes=n, index_base=0, ndata=nd)
    phi = ng.Chain(ng.Dense(2 * h + 2, 64, acts[0]), ng.Dense(64, 64, acts[1]))
    l = ng.ExplicitEdgeConv(phi, initialgraph=g, aggr=aggr)
    ps, st = ng.setup(47, l)
    ps = prep(ps, 47)
    x = torch.randn(h, n, device=DEV)
    with torch.no_grad():
        y64, _ = l(x, ps, st)
        y64b, _ = l(x, ps, st)
        monkeypatch.setenv("NGPDE_NO_EDGE64", "1")
        ygen, _ = l(x, ps, st)
        monkeypatch.delenv("NGPDE_NO_EDGE64")
    assert torch.equal(y64, y64b)
    assert torch.equal(y64, ygen)
    yo, c = O.explicit_edge_conv(x.cpu().double().numpy(), omlp(phi, ps), og, aggr)
    close(y64, yo)
    # the pullback (edge_mlp64_bwd_kernel: 64-edge chunks, two workgroups per CU) against the oracle and the general kernel
    R = rng.normal(size=yo.shape)
    Rt = torch.as_tensor(R, dtype=torch.float32, device=DEV)
    grads = []
    for general in (False, True):
        if general:
            monkeypatch.setenv("NGPDE_NO_EDGE64", "1")
        leaves = [v for v in grad_leaves(ps)]
        for v in leaves:
            v.grad = None
        xg = x.clone().requires_grad_(True)
        yg, _ = l(xg, ps, st)
        (yg * Rt).sum().backward()
        grads.append((xg.grad.clone(), [v.grad.clone() for v in leaves]))
        if general:
            monkeypatch.delenv("NGPDE_NO_EDGE64")
    gr = O.explicit_edge_conv_backward(c, R)
    names, ogr = mlp_grad_pairs(ps, gr["phi"], phi)
    xg = x.clone().requires_grad_(True)
    for v in grad_leaves(ps):
        v.grad = None
    yg, _ = l(xg, ps, st)
    (yg * Rt).sum().backward()
    check_grads(ps, (names, ogr), xg, gr["x"])
    close(grads[0][0], grads[1][0].cpu().double().numpy(), rtol=2e-5, atol=2e-6)
    for a, b in zip(grads[0][1], grads[1][1]):
        close(a, b.cpu().double().numpy(), rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("phi_widths", [(8,), (12, 8)])
def test_fused_paths_with_edge_features_and_fused_pullback(phi_widths, monkeypatch):
    # MPPDEConv with per-edge features (the E term of the split first layer) and a per-graph theta on a local graph whose
    # tiles fit the LDS halo: fused forward AND fused pullback (0 / 1 layer after the first) against the oracle, and the
    # fused pullback against the primitives' pullback
    n, G, h = 96, 2, 8
    rng = np.random.default_rng(41)
    idx = np.arange(n)
    offs = (-4, -2, -1, 1, 3)
    s1 = np.concatenate([idx for k in offs]); t1 = np.concatenate([(idx + k) % n for k in offs])
    gs, ogs = [], []
    for _ in range(G):
        nd = {"u": rng.random((1, n)), "x": rng.random((1, n))}
        ed = {"e": rng.random((2, s1.size))}
        gd = {"θ": rng.random(2)}
        gs.append(ng.GNNGraph(s1, t1, num_nodes=n, index_base=0, ndata=nd, edata=ed, gdata=gd))
        ogs.append(O.Graph(s1, t1, num_nodes=n, index_base=0, ndata=nd, edata=ed, gdata=gd))
    g, og = ng.batch(gs), O.batch(ogs)
    dims = [2 * h + 2 + 2 + 2] + list(phi_widths)
    acts = ["swish"] * len(phi_widths)
    phi = ng.Chain(*[ng.Dense(dims[i], dims[i + 1], acts[i]) for i in range(len(phi_widths))]) if len(phi_widths) > 1 \
        else ng.Dense(dims[0], dims[1], acts[0])
    psi = ng.Dense(h + phi_widths[-1] + 2, h, "tanh")
    l = ng.MPPDEConv(phi, psi, initialgraph=g, aggr="mean")
    ps, st = ng.setup(41, l)
    ps = prep(ps, 41)
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    assert F.edge_mlp_supported(g.handle(), phi_widths[0], list(phi_widths[1:]))
    R = rng.normal(size=(h, n * G))
    Rt = torch.as_tensor(R, dtype=torch.float32, device=DEV)

    def run():
        for v in [t for _, t in leaves(ps)]:
            v.grad = None
        x = xin.clone().requires_grad_(True)
        y, _ = l(x, ps, st)
        (y * Rt).sum().backward()
        return y.detach(), x.grad.clone(), {k: t.grad.clone() for k, t in leaves(ps)}

    xin = torch.randn(h, n * G, device=DEV)
    y, dx, gp = run()
    yo, c = O.mppde_conv(xin.cpu().double().numpy(), omlp(phi, ps["ϕ"]), omlp(psi, ps["ψ"]), og, "mean")
    close(y, yo)
    gr = O.mppde_conv_backward(c, R)
    close(dx, gr["x"], rtol=5e-4, atol=1e-4)
    n1, o1 = mlp_grad_pairs(ps["ϕ"], gr["phi"], phi)
    n2, o2 = mlp_grad_pairs(ps["ψ"], gr["psi"], psi)
    for (name, p), ogv in zip(n1 + n2, o1 + o2):
        close(p.grad, ogv, rtol=5e-4, atol=2e-4, what=name)
    monkeypatch.setenv("NGPDE_NO_FUSED_EDGE_BWD", "1")
    y2, dx2, gp2 = run()
    assert torch.equal(y, y2)
    close(dx, dx2.cpu().double().numpy(), rtol=2e-5, atol=2e-6)
    for k in gp:
        close(gp[k], gp2[k].cpu().double().numpy(), rtol=5e-5, atol=5e-6, what=k)


def test_fused_message_path_falls_back_when_unsupported():
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    g = ng.rand_graph(50, 200, seed=1)
    fh = g.handle((False, None, False))
    assert not F.edge_mlp_supported(fh, 66, [64])       # wider than 64
    assert not F.edge_mlp_supported(fh, 64, [9])        # not a multiple of 4
    assert not F.edge_mlp_supported(fh, 64, [64, 64, 64, 64])   # more than 3 further layers
    assert F.edge_mlp_supported(fh, 60, [60, 60, 40])   # the VMH tutorial's message MLP


# ---- small primitives added in round 2 ---------------------------------------------------------------------------------------

@pytest.mark.parametrize("count", [4096, 4099])
def test_rk_stage_combine_all_term_counts_and_aliasing(count):
    # out = c_self * base + sum_k coefs[k] * terms[k] for 0 .. 8 terms, 16-byte and scalar paths, base = NULL, out aliasing base
    from ngpde_amd.node import _combine
    rng = np.random.default_rng(count)
    base = torch.as_tensor(rng.normal(size=count).astype(np.float32), device=DEV)
    terms = [torch.as_tensor(rng.normal(size=count).astype(np.float32), device=DEV) for _ in range(8)]
    coefs = [float(c) for c in rng.normal(size=8)]
    for n in range(9):
        got = _combine(base, 0.75, terms[:n], coefs[:n])
        ref = 0.75 * base.double()
        for t, c in zip(terms[:n], coefs[:n]):
            ref = ref + np.float32(c).astype(np.float64) * t.double()
        close(got, ref.cpu().numpy(), rtol=2e-6, atol=1e-6, what=f"{n} terms")
    got = _combine(None, 0.0, terms[:3], coefs[:3])
    ref = sum(np.float32(c).astype(np.float64) * t.double() for t, c in zip(terms[:3], coefs[:3]))
    close(got, ref.cpu().numpy(), rtol=2e-6, atol=1e-6, what="no base")
    acc = base.clone()
    _combine(acc, 1.0, [terms[0]], [1.0], out=acc)                      # in place: the parameter-gradient accumulation
    assert torch.equal(acc, base + terms[0])
    from ngpde_amd import _lib
    with pytest.raises(_lib.NgpdeError):
        _combine(base, 1.0, terms + [terms[0]], coefs + [1.0])          # more than 8 terms: status, not a crash


@pytest.mark.parametrize("act", ["identity", "relu", "swish"])
@pytest.mark.parametrize("d", [128, 7])
def test_bias_act_tail_forward_and_pullback(act, d):
    # y = act(a + addend + b): GNOConv's sigma(W x + m + b) tail (src/layers.jl:536-547); 16-byte and scalar paths
    import composed as F          # the primitives' autograd wrappers (tests/composed.py)
    n = 1000
    rng = np.random.default_rng(d)
    mk = lambda *s: torch.as_tensor(rng.normal(size=s).astype(np.float32), device=DEV).requires_grad_(True)
    a, add, b = mk(n, d), mk(n, d), mk(d)
    code = ng.layers._act_code(act)[1]
    y = F.bias_act(a, add, b, code)
    z = (a + add + b).detach().cpu().double().numpy()
    close(y, O.act(act, z))
    R = rng.normal(size=(n, d))
    (y * torch.as_tensor(R, dtype=torch.float32, device=DEV)).sum().backward()
    dz = R * O.dact(act, z)
    close(a.grad, dz, rtol=2e-4)
    close(add.grad, dz, rtol=2e-4)
    close(b.grad, dz.sum(axis=0), rtol=3e-4, atol=1e-3)
    y2 = F.bias_act(a.detach(), None, None, code)                       # no addend, no bias
    close(y2, O.act(act, a.detach().cpu().double().numpy()))
