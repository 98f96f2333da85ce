"""Where does a slot-phase of the interleaved (two members per workgroup) persistent kernels go?  Needs the diagnostic library
(make -C neuralgraphpde.jl_amd/csrc diag).  Shader-clock stamps of thread 0 of every workgroup; both slots stamp the same phase
index, so what survives is slot 1's slot-phase: 0 top, 2 halo complete (T0: drain + barrier, or the blocking path), 3 operand tile
written + T2 barrier, 4 matrix products + T4 (flags of the next slot-phase + barrier), 5 epilogue, ahead-gather issued, row stores
issued, 6 end.  Start-to-start of consecutive phases of one slot = two slot-phases."""
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

N, PAIRS, D, STEPS, PH, K = 16384, 65536, 64, 50, 240, 2
dev = "cuda:0"
_, s, t = S.closest_pairs_graph(N, PAIRS, seed=2)
g = ng.GNNGraph(s, t, num_nodes=N, index_base=0)
lib, p = _lib.load(), _lib.ptr
lib.ngpde_debug_set_persistent_stamps.argtypes = [C.c_void_p, C.c_int32]
dv = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev)
u0 = dv(S.normal(1000, D * N * K).reshape(N * K, D))
w1, w2 = dv(S.glorot_uniform(11, D, D).T), dv(S.glorot_uniform(12, D, D).T)
b1, b2 = dv(np.zeros(D)), dv(np.zeros(D))
seed = torch.ones_like(u0)
stream = torch.cuda.current_stream().cuda_stream
plan = _Plan(g.handle((True, None, False)), D, _lib.ACT["relu"], "tsit5", STEPS, 1.0 / 50, True, members=K)
outs = [torch.empty_like(u0), torch.empty_like(u0), torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)]
NT = N // 32
pts = [(0, 2, "T0: drain + barrier (or blocking path)"), (2, 3, "aggregate + operand tile + T2"), (3, 4, "products + T4 (next flags, barrier)"),
       (4, 5, "epilogue + ahead-gather + stores issued"), (5, 6, "tail")]


def run(which):
    buf = torch.zeros(NT * PH * 8, dtype=torch.int64, device=dev)
    for rep in range(3):
        lib.ngpde_debug_set_persistent_stamps(p(buf) if (rep == 2 and which == "fwd") else None, PH)
        _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(outs[0]), stream))
        lib.ngpde_debug_set_persistent_stamps(p(buf) if (rep == 2 and which == "bwd") else None, PH)
        _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(outs[1]), p(outs[2]), p(outs[3]), p(outs[4]), p(outs[5]), stream))
        torch.cuda.synchronize()
    st = buf.cpu().numpy().reshape(NT, PH, 8).astype(np.float64)
    for parity, label in ((0, "odd phases (layer 1)"), (1, "even phases (layer 2)")):
        sel = st[:, 8 + parity:PH:2, :]
        whole = sel[:, 1:, 0] - sel[:, :-1, 0]          # slot 1, start to start of the same kind = 2 phases = 4 slot-phases
        own = sel[:, :, 6] - sel[:, :, 0]
        print(f"{which} slot 1, {label}: " + "; ".join(f"{nm}: {(sel[:, :, b] - sel[:, :, a]).mean():.0f}" for a, b, nm in pts) +
              f" | whole slot-phase {own.mean():.0f} | four slot-phases start-to-start {whole.mean():.0f} cycles (100 MHz counter x ... see s_memtime)")
    print("flags", sorted(plan.flags()), "fault", plan.fault())


run("fwd")
run("bwd")
