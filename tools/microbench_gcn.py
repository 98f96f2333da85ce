"""Diagnostic: device time of the fused GCN layer kernels on the C2 graph under different node orders."""
import ctypes as C
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S

N, PAIRS, D = 16384, 65536, 64
pts, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)

def morton(pts, bits=10):
    q = np.minimum((pts * (1 << bits)).astype(np.int64), (1 << bits) - 1)
    code = np.zeros(len(pts), np.int64)
    for b in range(bits):
        code |= ((q[:, 0] >> b) & 1) << (2 * b + 1)
        code |= ((q[:, 1] >> b) & 1) << (2 * b)
    return np.argsort(code, kind="stable")

def timed(fn, reps=200):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps

lib = _lib.load()
for name, order in (("random", np.arange(N)), ("morton", morton(pts))):
    inv = np.empty(N, np.int64); inv[order] = np.arange(N)
    g = ng.GNNGraph(inv[s], inv[t], num_nodes=N, index_base=0)
    h = g.handle((True, None, False))
    x = torch.randn(N, D, device="cuda"); w = torch.randn(D, D, device="cuda") * 0.1; b = torch.zeros(D, device="cuda")
    y = torch.empty_like(x); agg = torch.empty_like(x); z = torch.empty_like(x)
    ws = torch.empty(lib.ngpde_gcn_workspace_bytes(h.ptr, D, D, 1), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    p = _lib.ptr
    fwd = lambda: _lib.check(lib.ngpde_gcn_forward(h.ptr, D, D, 1, p(x), p(w), p(b), p(y), p(agg), None, p(ws), ws.numel(), st))
    fwd_nosave = lambda: _lib.check(lib.ngpde_gcn_forward(h.ptr, D, D, 1, p(x), p(w), p(b), p(y), None, None, p(ws), ws.numel(), st))
    dy = torch.randn_like(x); dx = torch.empty_like(x); dw = torch.empty_like(w); db = torch.empty_like(b)
    bwd = lambda: _lib.check(lib.ngpde_gcn_backward(h.ptr, D, D, 1, p(x), p(w), p(y), p(agg), p(dy), p(dx), p(dw), p(db), p(ws), ws.numel(), st))
    print(f"{name:8s} fwd(save_agg) {timed(fwd):7.2f} us   fwd(nosave) {timed(fwd_nosave):7.2f} us   layer-bwd(4 launches+memset) {timed(bwd, 100):7.2f} us")
