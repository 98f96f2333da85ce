"""Where does a phase of the persistent solver go?  Needs the diagnostic library (make -C neuralgraphpde.jl_amd/csrc diag).
Shader-clock stamps of thread 0 of every workgroup at 8 points of the first PH phases of the forward and of the adjoint
launch; prints the mean (over workgroups and phases) cycles between consecutive points, per phase kind (odd / even phase)."""
import ctypes as C
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S

_lib.LIB_PATH = os.path.join(ROOT, "neuralgraphpde.jl_amd", "libngpde_diag.so")
from ngpde_amd.node import _Plan

# env GRAPH=cora: BASELINE config 1's shape (2 708 nodes with hubs: the hub geometry) instead of the C2 graph
CORA = os.environ.get("GRAPH", "c2") == "cora"
N, PAIRS, D, STEPS, PH = (2708, 5278, 64, 10, 100) if CORA else (16384, 65536, 64, 50, 240)
dev = "cuda:0"
if CORA:
    s, t = S.preferential_pairs_graph(N, PAIRS, seed=1)
else:
    _, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
lib, p = _lib.load(), _lib.ptr
lib.ngpde_debug_set_persistent_stamps.argtypes = [C.c_void_p, C.c_int32]
dv = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev)
u0 = dv(S.normal(1000, D * N).reshape(N, D))
w1, w2 = dv(S.glorot_uniform(11, D, D).T), dv(S.glorot_uniform(12, D, D).T)
b1, b2 = dv(np.zeros(D)), dv(np.zeros(D))
seed = torch.ones_like(u0)
stream = torch.cuda.current_stream().cuda_stream
plan = _Plan(g.handle((True, None, False)), D, _lib.ACT["relu"], "tsit5", STEPS, 1.0 / 50, True)
outs = [torch.empty_like(u0), torch.empty_like(u0), torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)]
NT = (N + 31) // 32
names_f = ["wait for the neighbours' flags", "gather foreign rows (LDS-DMA) + barrier", "LDS aggregation, tile write, barrier", "MFMA + barrier",
           "epilogue + row stores issued", "drain + barrier + flag"]
names_b = ["prefetch + wait for the flags", "gather foreign rows + barrier", "LDS aggregation, stage terms, tile writes, barrier", "MFMA (G) + barrier",
           "row stores issued", "drain + barrier + flag", "dW / db products"]


def run(which):
    buf = torch.zeros(NT * PH * 8, dtype=torch.int64, device=dev)
    for rep in range(3):
        lib.ngpde_debug_set_persistent_stamps(p(buf) if (rep == 2 and which == "fwd") else None, PH)
        _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(outs[0]), stream))
        lib.ngpde_debug_set_persistent_stamps(p(buf) if (rep == 2 and which == "bwd") else None, PH)
        _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]), p(outs[5]), stream))
        torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(NT, PH, 8).astype(np.float64)
    names = names_f if which == "fwd" else names_b
    for parity, label in ((0, "odd phases (layer 1)"), (1, "even phases (layer 2 + stage terms)")):
        sel = st[:, 8 + parity:PH:2, :]
        d = np.diff(sel[:, :, :len(names) + 1], axis=2)
        whole = sel[:, 1:, 0] - sel[:, :-1, 0]          # start to start of the same kind = 2 phases
        print(f"{which} {label}: " + "; ".join(f"{nm}: {d[:, :, k].mean():.0f}" for k, nm in enumerate(names)) +
              f" | two consecutive phases start-to-start: {whole.mean():.0f} cycles")
    per_tile = (st[:, PH - 1, 0] - st[:, 8, 0]) / (PH - 9)          # mean cycles per phase, tile by tile
    waits = np.diff(st[:, 8:, :2], axis=2)[:, :, 0].mean(axis=1)    # mean wait per tile
    order = np.argsort(waits)
    print(f"{which}: cycles per phase {per_tile.mean():.0f} (s_memtime ticks); tiles waiting least (the ones waited for): " +
          ", ".join(f"tile {k}: wait {waits[k]:.0f}" for k in order[:6]))
    if os.environ.get("DUMP"):
        own = np.diff(st[:, 8:, :len(names) + 1], axis=2).mean(axis=1)       # [tile(workgroup)][segment]
        np.save(os.path.join(ROOT, "gpurun_out", f"stamps_{which}.npy"), own)
    k0 = int(order[0])
    d0 = np.diff(st[k0, 8:, :len(names) + 1], axis=1).mean(axis=0)
    print(f"{which}: workgroup {k0} (waits least), mean cycles per step of a phase: " + "; ".join(f"{nm}: {d0[k]:.0f}" for k, nm in enumerate(names)))
    print("flags", sorted(plan.flags()), "fault", plan.fault())


run("fwd")
run("bwd")
