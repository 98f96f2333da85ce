"""Solve + discrete adjoint of the C2 right-hand side on graphs of 2x / 4x / 8x the bench's size (tile pairs, tile rounds), through the
C ABI: ms per direction by the plan's own dispatch events, and the fraction of the HBM bound of SURVEY 8(d)'s algorithmic bytes.
  python3 tools/bench_tile_rounds.py [factors...]      (NGPDE_NO_TILE_PIPE=1: the plain tile-round kernels)"""
import ctypes as C, json, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ngpde_amd as ng
from ngpde_amd import _lib, synth as S
from ngpde_amd.node import _Plan

D, STEPS = 64, 50
BYTES = 2 * 6 * STEPS * (9.06e6 + 17.47e6)       # one C2 solve + adjoint
lib, p, dev = _lib.load(), _lib.ptr, "cuda:0"
stream = torch.cuda.current_stream().cuda_stream
for f in [int(a) for a in sys.argv[1:]] or [2, 4, 8]:
    n = 16384 * f
    _, s, t = S.closest_pairs_graph(n, 65536 * f, seed=100 + f)
    g = ng.GNNGraph(s, t, num_nodes=n, index_base=0)
    plan = _Plan(g.handle((True, None, False)), D, _lib.ACT["relu"], "tsit5", STEPS, 1.0 / STEPS, True)
    dv = lambda a: torch.as_tensor(np.ascontiguousarray(a, np.float32), device=dev)
    u = dv(S.normal(7, D * n).reshape(n, D))
    w1, w2 = dv(S.glorot_uniform(1, D, D)), dv(S.glorot_uniform(2, D, D))
    b1, b2 = torch.zeros(D, device=dev), torch.zeros(D, device=dev)
    uT, du, seed = torch.empty_like(u), torch.empty_like(u), torch.ones_like(u)
    gw = [torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)]
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    best = [1e9, 1e9]
    for rep in range(5):
        ev[0].record()
        _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u), p(w1), p(b1), p(w2), p(b2), p(uT), stream))
        ev[1].record()
        _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed), p(du), p(gw[0]), p(gw[1]), p(gw[2]), p(gw[3]), stream))
        ev[2].record()
        torch.cuda.synchronize()
        if rep:
            best = [min(best[0], ev[0].elapsed_time(ev[1])), min(best[1], ev[1].elapsed_time(ev[2]))]
    af, ab, tt = C.c_int64(), C.c_int64(), C.c_int64()
    _lib.check(lib.ngpde_node_pipeline_stats(plan.ptr, stream, C.byref(af), C.byref(ab), C.byref(tt)))
    tot = best[0] + best[1]
    print(json.dumps({"nodes": n, "tiles": n // 32, "plan": sorted(plan.flags()), "fault": bool(plan.fault()), "ms_forward": round(best[0], 3),
                      "ms_adjoint": round(best[1], 3), "ms_total": round(tot, 3), "ode_steps_per_s": round(STEPS / (tot * 1e-3), 1),
                      "frac_of_hbm_bound": round(f * BYTES / (tot * 1e-3) / 8e12, 4), "finite": bool(torch.isfinite(du).all()),
                      "turns": int(tt.value), "gathered_ahead_forward": int(af.value), "gathered_ahead_adjoint": int(ab.value)}), flush=True)
    plan = None
