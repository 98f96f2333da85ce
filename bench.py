#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on BASELINE config 2 (C2):

    ODE-steps/sec, forward + backward, of the neural graph ODE  du/dt = Chain(GCNConv(64=>64, relu),
    GCNConv(64=>64, relu))(u)  on a 16384-node / 131072-edge radius-style graph with 64-d features,
    Tsit5 with a fixed step dt = 1/50 for 50 steps (6 right-hand-side evaluations per step), gradient of
    sum(u(T)) w.r.t. u0 and all parameters by the discrete adjoint.

A bench "step" (what --steps counts) is ONE full solve, forward then backward (+ the RCCL all-reduce
of the parameter gradients when N > 1) and the fused Adam update of the flat parameter vector, over one
batch of synthetic input already resident in HBM.
`value` = ODE steps integrated per second by the whole job = n_gpus * 50 * K / T.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --gpus N --steps K --warmup W        # the same: starts its own N ranks (one fresh process per GPU);
                                                         # prints a {"skipped": true, ...} record when the box has < N devices

Multi-GPU is data parallel over independent trajectories (different u0 per rank, same graph and
parameters): per-GPU work is fixed ("weak"), one all-reduce(sum) of the 8320-float gradient per step.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import ngpde_amd as ng  # noqa: E402
from ngpde_amd import _lib, synth as S  # noqa: E402
from ngpde_amd.node import _Plan  # noqa: E402

N_NODES, N_PAIRS, D = 16384, 65536, 64
ODE_STEPS, DT = 50, 1.0 / 50.0
GRAPH_SEED = 2
# SURVEY.md §8(d) algorithmic (compulsory) HBM bytes of ONE fused layer launch at C2
BYTES_FWD_LAYER = 9.06e6
BYTES_BWD_LAYER = 17.47e6
HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8 TB/s spec


def make_inputs(rank):
    _, s, t = S.closest_pairs_graph(N_NODES, N_PAIRS, seed=GRAPH_SEED)
    u0 = S.normal(1000 + rank, D * N_NODES).reshape(N_NODES, D).astype(np.float32)      # [N][D] = (D x N) col-major
    w1 = np.ascontiguousarray(S.glorot_uniform(11, D, D).T, np.float32)                 # [in][out]
    w2 = np.ascontiguousarray(S.glorot_uniform(12, D, D).T, np.float32)
    b1 = np.zeros(D, np.float32)
    b2 = np.zeros(D, np.float32)
    return s, t, u0, w1, b1, w2, b2


def cpu_baseline(s, t, u0, w1, b1, w2, b2, budget_s=12.0):
    """The C restatement of the reference algorithm (oracle/, kind "port") timed on this box's host
    cores on a bounded sample of the same workload: n Tsit5 steps forward + backward on the C2 graph."""
    import subprocess
    odir = os.path.join(ROOT, "oracle")
    path = os.path.join(odir, "libngpde_oracle_omp.so")
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", odir])
    lib = C.CDLL(path)
    vp = C.c_void_p
    lib.ngo_node_gcn2.argtypes = [C.c_int64, C.c_int64, vp, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float,
                                  C.c_int] + [vp] * 11
    lib.ngo_node_gcn2.restype = C.c_int
    cores = os.cpu_count() or 1
    try:
        gomp = C.CDLL("libgomp.so.1")
    except OSError:
        gomp = None
    s64, t64 = np.ascontiguousarray(s, np.int64), np.ascontiguousarray(t, np.int64)
    outs = [np.zeros_like(u0), np.zeros_like(u0), np.zeros_like(w1), np.zeros_like(b1), np.zeros_like(w2),
            np.zeros_like(b2)]
    P = lambda a: a.ctypes.data

    def run(nsteps):
        t0 = time.perf_counter()
        rc = lib.ngo_node_gcn2(N_NODES, s64.size, P(s64), P(t64), D, 1, 1, nsteps, DT, 1, P(u0), P(w1), P(b1), P(w2),
                               P(b2), *[P(o) for o in outs])
        assert rc == 0
        return time.perf_counter() - t0

    # the port's loops are short: more threads than the sparse products can feed only add fork/join and NUMA cost, so
    # probe a few team sizes with one step each and keep the fastest (the first call also pays the page faults)
    run(1)
    best_t, best_n = None, cores
    if gomp is not None:
        for nt in sorted({min(cores, k) for k in (8, 16, 32, 64, cores)}):
            gomp.omp_set_num_threads(nt)
            t1 = run(1)
            if best_t is None or t1 < best_t:
                best_t, best_n = t1, nt
        gomp.omp_set_num_threads(best_n)
    else:
        best_t = run(1)
    n = int(max(1, min(ODE_STEPS, budget_s / max(best_t, 1e-6))))
    tn = run(n) if n > 1 else best_t
    out = {"value": n / tn, "unit": "ODE-steps/s", "cores": best_n, "kind": "port",
           "sample": f"{n} Tsit5 step(s) fwd+bwd of the same C2 workload (C restatement of the reference "
                     f"algorithm, OpenMP, fastest of the probed team sizes = {best_n} of {cores} host threads; "
                     f"not the Julia package)", "variants": {}}
    # variant (i) of SURVEY 8(d): serial gather / scatter / sparse product (NNlib 0.8 on CPU) + threaded dense products
    if gomp is not None and hasattr(lib, "ngo_set_serial_sparse"):
        lib.ngo_set_serial_sparse(1)
        t1 = run(1)
        k = int(max(1, min(10, 4.0 / max(t1, 1e-6))))
        tk = run(k) if k > 1 else t1
        lib.ngo_set_serial_sparse(0)
        out["variants"]["serial_sparse_threaded_gemm"] = {
            "value": k / tk, "unit": "ODE-steps/s", "cores": best_n,
            "sample": f"{k} Tsit5 step(s): sparse products on one thread (NNlib 0.8's serial gather / scatter), dense products on "
                      f"{best_n} threads"}
    try:
        out["variants"]["torch_cpu"] = torch_cpu_point(s, t, u0, w1, b1, w2, b2)
    except Exception as ex:          # an independent third point: never fails the bench
        out["variants"]["torch_cpu"] = {"error": repr(ex)[:200]}
    return out, outs


def torch_cpu_point(s, t, u0, w1, b1, w2, b2, budget_s=4.0):
    """An independent CPU point: the same two-layer GCN right-hand side and Tsit5 steps written with torch CPU ops (index_add_
    aggregation, mm, autograd for the discrete adjoint), float32, torch's own thread pool"""
    a = [[], [0.161], [-0.008480655492356989, 0.335480655492357], [2.8971530571054935, -6.359448489975075, 4.3622954328695815],
         [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525],
         [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383]]
    b = [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774]
    n = u0.shape[0]
    ss = torch.as_tensor(np.concatenate([s, np.arange(n)]), dtype=torch.int64)
    tt = torch.as_tensor(np.concatenate([t, np.arange(n)]), dtype=torch.int64)
    deg = torch.zeros(n).index_add_(0, tt, torch.ones(tt.numel()))
    c = deg.rsqrt().unsqueeze(1)
    W = [torch.tensor(w1, requires_grad=True), torch.tensor(w2, requires_grad=True)]
    B = [torch.tensor(b1, requires_grad=True), torch.tensor(b2, requires_grad=True)]

    def layer(x, k):
        agg = torch.zeros_like(x).index_add_(0, tt, (x * c)[ss]) * c
        return torch.relu(agg @ W[k] + B[k])

    def steps(k):
        u = torch.tensor(u0, requires_grad=True)
        t0 = time.perf_counter()
        x = u
        for _ in range(k):
            ks = []
            for i in range(6):
                U = x
                for j in range(i):
                    U = U + (DT * a[i][j]) * ks[j]
                ks.append(layer(layer(U, 0), 1))
            for i in range(6):
                x = x + (DT * b[i]) * ks[i]
        x.sum().backward()
        return time.perf_counter() - t0
    t1 = steps(1)
    k = int(max(1, min(10, budget_s / max(t1, 1e-6))))
    tk = steps(k) if k > 1 else t1
    return {"value": k / tk, "unit": "ODE-steps/s", "cores": torch.get_num_threads(),
            "sample": f"{k} Tsit5 step(s) fwd+bwd, torch CPU ops (index_add_ + mm + autograd), {torch.get_num_threads()} threads"}


# ---- secondary workloads: one layer of BASELINE configs 3-5 (never in `value`) ---------------------------------------------
FP32_MFMA_PEAK_TFS = 157.3   # MI355X_MICROARCH.md: fp32-input MFMA, dense
TIMING_NOTE = ("ms_forward / ms_forward_backward (the figures of record, ONE method for every layer leg): the layer call (and its autograd "
               "pullback) captured once into a HIP graph and replayed, at least 30 timed replays after 5 warm-ups -- the device-side time of "
               "the library's launches; *_eager_api: the same call issued from Python every time (short layers are bound by the Python "
               "dispatch there; C4's ~ 20 kernels run ~ 1 % faster than their replayed graph); roofline.kernel_sum_*: the sum of the "
               "launches' own durations by rocprofv3 --kernel-trace --stats in a child pass (no gaps between kernels), the conservative "
               "source for roofline.frac where present")
# algorithmic figures of ONE layer forward
C3_FWD_BYTES = 9.50e6        # GATConv on the C2 graph: compulsory traffic
# ... as ODE right-hand side (one evaluation and its pullback): forward 9.50 MB + the saved attention coefficients
# (E' x 4 heads x 4 B = 2.36 MB) = 11.86 MB; pullback: dY, Y (relu mask), X (W x is rebuilt), dX = 4 x 4.194 MB, the coefficients
# 2.36 MB, the lists of both directions 1.31 MB, W / a and their gradients 0.03 MB = 20.48 MB
C3_RHS_BYTES = 11.86e6 + 20.48e6
C4_FWD_FLOP = 49.0e9         # MPPDEConv shard (64 trajectories), first-layer-split form
# (C5: c5_fwd_flop(E) -- SURVEY's 10.6 / 16.8 GFLOP are for E ~ 115 k / 492 k; the generator gives 140 860 / 480 784 edges)


def _time_ms(fn, reps):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def _graph_ms(fn, reps):
    """ms per call of `fn` replayed from a HIP graph (torch.cuda.graph capture of the library's launches on the capture stream):
    the device-side time of the layer, without the Python / autograd dispatch of the eager call"""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(2):
            fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gr = torch.cuda.CUDAGraph()
    import gc
    gc_was = gc.isenabled()
    gc.disable()      # (a cyclic collection inside the capture could finalise an older captured solve: destroying its graphs is not
    try:              # permitted on a capturing stream)
        with torch.cuda.graph(gr):
            fn()
    finally:
        if gc_was:
            gc.enable()
    return _time_ms(gr.replay, reps)


def _layer_times(layer, x, ps, st, reps):
    """ms of one forward (no autograd) and of forward + backward (gradients w.r.t. x and every parameter) through the layer
    API: (eager forward, eager forward+backward, graph-replayed forward, graph-replayed forward+backward); the graph replays are the
    figures of record (TIMING_NOTE)"""
    x = x.detach().requires_grad_(True)
    leaves = [x] + _grad_leaves(ps)
    reps = max(reps, 30)
    with torch.no_grad():
        y0 = layer(x, ps, st)[0]
        for _ in range(3):
            layer(x, ps, st)
        ms_f = _time_ms(lambda: layer(x, ps, st)[0], reps)
        ms_fg = _graph_ms(lambda: layer(x, ps, st)[0], reps)
    R = torch.randn(y0.shape[1], y0.shape[0], device=x.device).T      # cotangent in the output's (column-major) layout

    def fb():
        for v in leaves:
            v.grad = None                                              # gradients are written, not accumulated
        layer(x, ps, st)[0].backward(R)
    for _ in range(3):
        fb()
    ms_fb = _time_ms(fb, reps)
    ms_fbg = _graph_ms(fb, reps)
    _layer_times.graph_replay = (ms_fg, ms_fbg)
    return ms_f, ms_fb, ms_fg, ms_fbg                # (the figures of record are the graph replays: one method for every leg)


def _grad_leaves(ps):
    out = []
    for v in ps.values():
        out += _grad_leaves(v) if isinstance(v, dict) else [v]
    return out


def c4_layer(dev, traj, seed):
    n, h = 8192, 64
    s, t = S.periodic_mesh_batch(n, traj)
    N = n * traj
    g = ng.GNNGraph(s, t, num_nodes=N, index_base=0, num_graphs=traj,
                    ndata={"u": S.uniform01(40 + seed, N).reshape(1, N).astype(np.float32),
                           "x": np.tile(np.arange(n) / n, traj)[None, :].astype(np.float32)},
                    gdata={"θ": S.uniform01(41 + seed, 2 * traj).reshape(2, traj).astype(np.float32)})
    phi = ng.Chain(ng.Dense(132, 64, "swish"), ng.Dense(64, 64, "swish"))
    psi = ng.Chain(ng.Dense(130, 64, "swish"), ng.Dense(64, 64))
    layer = ng.MPPDEConv(phi, psi, initialgraph=g)
    ps, st = ng.setup(4, layer)
    x = torch.as_tensor(S.normal(42 + seed, h * N).reshape(N, h).astype(np.float32), device=dev).T
    return layer, ps, st, x, int(s.size)


def c5_layer(dev, radius, width=128, seed=0):
    """GNOConv width => width on the 64 x 64 grid radius graph of BASELINE config 5: ndata = (a (1), x (2)),
    phi = Dense(6 => 64, relu) -> Dense(64 => width^2).  Shared with tools/bench_layers.py so that profiles and bench line run
    the same edges."""
    pts, s5, t5 = S.grid_radius_graph(64, radius)
    g5 = ng.GNNGraph(s5, t5, num_nodes=4096, index_base=0,
                     ndata={"a": S.uniform01(50, 4096).reshape(1, 4096).astype(np.float32), "x": pts.astype(np.float32)})
    phi = ng.Chain(ng.Dense(6, 64, "relu"), ng.Dense(64, width * width))
    l5 = ng.GNOConv((width, width), phi, "relu", initialgraph=g5)
    ps5, st5 = ng.setup(5, l5)
    x5 = torch.as_tensor(S.normal(51 + seed, width * 4096).reshape(4096, width).astype(np.float32), device=dev).T
    return l5, ps5, st5, x5, int(s5.size)


def c5_fwd_flop(n_edges, width=128, k=64, n=4096):
    """SURVEY.md 8(d)'s official (reassociated) count for the edges actually run: node-level T = W2 (x) h (2 n in out k), per-edge
    T_j z_e (2 out k E), the bias term B2 h (2 n in out)"""
    return 2.0 * n * width * width * k + 2.0 * width * k * n_edges + 2.0 * n * width * width


def secondary(dev, world, rank, dist, rocprof=False):
    """One layer forward / forward + backward of BASELINE configs 3, 4 (per-GPU shard) and 5 through the layer API, with the
    roofline that bounds each.  N > 1: the C4 leg only, as a data-parallel training step (64 trajectories per
    rank, replicated parameters, bucketed all-reduce of the flat gradient overlapped with the pullback, fused Adam)."""
    out = {}
    if world == 1:
        _, s, t = S.closest_pairs_graph(N_NODES, N_PAIRS, seed=GRAPH_SEED)
        g = ng.GNNGraph(s, t, num_nodes=N_NODES, index_base=0)
        layer = ng.GATConv((64, 16), "relu", heads=4, initialgraph=g)
        ps, st = ng.setup(3, layer)
        ps = ng.to_device(ps, dev)
        for v in _grad_leaves(ps):
            v.requires_grad_(True)
        x = torch.as_tensor(S.normal(33, 64 * N_NODES).reshape(N_NODES, 64).astype(np.float32), device=dev).T
        fe, fbe, f, fb = _layer_times(layer, x, ps, st, 50)
        ach = C3_FWD_BYTES / (f * 1e-3) / 1e9
        out["C3_gat_4x16_layer"] = {"ms_forward": round(f, 4), "ms_forward_backward": round(fb, 4),
                                    "ms_forward_eager_api": round(fe, 4), "ms_forward_backward_eager_api": round(fbe, 4),
                                    "ms_forward_graph_replay": round(_layer_times.graph_replay[0], 4),
                                    "ms_forward_backward_graph_replay": round(_layer_times.graph_replay[1], 4),
                                    "timing": TIMING_NOTE,
                                    "roofline": {"bound": "hbm", "achieved": round(ach, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                 "frac": round(ach / HBM_PEAK_GBS, 4), "algorithmic_MB_forward": C3_FWD_BYTES / 1e6}}
        # BASELINE config 3 "as ODE RHS": NeuralODE(GATConv 64 => 4 x 16) on the C2 graph, Tsit5 x 50, forward + discrete adjoint,
        # the generic solver path captured into two HIP graphs (every stage = the one-launch GAT layer, every Runge-Kutta
        # combination one ngpde_rk_stage_combine launch)
        lg = ng.GATConv((64, 16), "relu", heads=4, initialgraph=g)
        nodeg = ng.NeuralODE(lg, solver="tsit5", n_steps=ODE_STEPS, dt=DT, capture=True)
        psg, stg = ng.setup(3, nodeg)
        psg = ng.to_device(psg, dev)
        for v in _grad_leaves(psg):
            v.requires_grad_(True)
        leaves = [x] + _grad_leaves(psg)
        xg = x.detach().requires_grad_(True)

        def solve():
            for v in [xg] + _grad_leaves(psg):
                v.grad = None
            uT, _ = nodeg(xg, psg, stg)
            uT.sum().backward()
        ms_node = _time_ms(solve, 5)
        gplans = [p for pool in nodeg._plans.values() for p in pool]
        resident = bool(gplans) and all("gat" in p.flags() for p in gplans)
        out["C3_gat_node_tsit5x50"] = {"ms_solve_forward_backward": round(ms_node, 3), "value": round(ODE_STEPS / (ms_node * 1e-3), 1),
                                       "unit": "ODE-steps/s",
                                       "path": ("NeuralODE(GATConv): device-resident solver, one persistent launch per direction (ngpde_node_gat_*)"
                                                if resident else "NeuralODE(GATConv, capture=True): HIP-graph replay of the generic solver"),
                                       "fault": any(p.fault() for p in gplans) if resident else False,
                                       "rhs_evals_per_ode_step": 6}
        ach_n = 6 * ODE_STEPS * C3_RHS_BYTES / (ms_node * 1e-3) / 1e9
        out["C3_gat_node_tsit5x50"]["roofline"] = {"bound": "hbm", "achieved": round(ach_n, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                   "frac": round(ach_n / HBM_PEAK_GBS, 4),
                                                   "algorithmic_MB_per_rhs_evaluation_fwd_bwd": C3_RHS_BYTES / 1e6,
                                                   "algorithmic_GB_solve_and_adjoint": round(6 * ODE_STEPS * C3_RHS_BYTES / 1e9, 3)}
        if resident:      # 8 trajectories per GPU on this right-hand side: a block-diagonal batch of identical structures, two members per workgroup
            traj = 8
            lb = ng.GATConv((64, 16), "relu", heads=4, initialgraph=ng.batch([g] * traj))
            nodeb = ng.NeuralODE(lb, solver="tsit5", n_steps=ODE_STEPS, dt=DT)
            _, stb = ng.setup(3, nodeb)
            xb = torch.as_tensor(S.normal(37, 64 * N_NODES * traj).reshape(N_NODES * traj, 64).astype(np.float32), device=dev).T.requires_grad_(True)

            def solveb():
                for v in [xb] + _grad_leaves(psg):
                    v.grad = None
                uT, _ = nodeb(xb, psg, stb)
                uT.sum().backward()
            msb = _time_ms(solveb, 3)
            bplans = [p for pool in nodeb._plans.values() for p in pool]
            out["C3_gat_node_tsit5x50"]["batched"] = {"trajectories_per_gpu": traj, "ms_solve_forward_backward": round(msb, 3),
                                                      "value": round(traj * ODE_STEPS / (msb * 1e-3), 1), "unit": "trajectory ODE-steps/s",
                                                      "device_resident": bool(bplans) and all("gat" in p.flags() for p in bplans),
                                                      "fault": any(p.fault() for p in bplans),
                                                      "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                                                   "achieved": round(traj * 6 * ODE_STEPS * C3_RHS_BYTES / (msb * 1e-3) / 1e9, 1),
                                                                   "frac": round(traj * 6 * ODE_STEPS * C3_RHS_BYTES / (msb * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
            del nodeb, stb, xb, lb
        if resident:      # the generic solver (every stage the one-launch layer, captured into HIP graphs) on the same workload
            os.environ["NGPDE_NO_PERSISTENT"] = "1"
            try:
                nodeg2 = ng.NeuralODE(lg, solver="tsit5", n_steps=ODE_STEPS, dt=DT, capture=True)

                def solve2():
                    for v in [xg] + _grad_leaves(psg):
                        v.grad = None
                    uT, _ = nodeg2(xg, psg, stg)
                    uT.sum().backward()
                ms2 = _time_ms(solve2, 5)
                out["C3_gat_node_tsit5x50"]["generic_captured_ms"] = round(ms2, 3)
                out["C3_gat_node_tsit5x50"]["generic_captured_value"] = round(ODE_STEPS / (ms2 * 1e-3), 1)
            finally:
                os.environ.pop("NGPDE_NO_PERSISTENT", None)
    if world == 1:
        # BASELINE config 1 (the reference's own CPU-runnable case, docs/src/tutorials/graph_node.md:44-83): NeuralODE(2 x GCNConv(32 => 32,
        # relu)) on a Cora-sized graph with Cora's degree skew (2 708 nodes, 5 278 pairs, hubs of degree ~100 beside a median of 3),
        # Euler x 10.  Its hub tiles do not fit the handle's 96-row halo lists: the solver runs the persistent kernels' hub geometry
        # (node_persistent.hip: its own tile partition, 256-row halos, hub rows summed by all lane groups; round 4 replayed the per-layer
        # kernels with the per-row gather: 0.43 ms); `same_size_without_hubs` is the 96-row geometry on a closest-pairs graph of that size
        n1, pairs1, d1, steps1 = 2708, 5278, 32, 10
        c1 = {}
        for name, (s1, t1) in (("cora_like", S.preferential_pairs_graph(n1, pairs1, seed=1)),
                               ("same_size_without_hubs", S.closest_pairs_graph(n1, pairs1, seed=1)[1:])):
            g1 = ng.GNNGraph(s1, t1, num_nodes=n1, index_base=0)
            rhs1 = ng.Chain(ng.GCNConv((d1, d1), "relu", initialgraph=g1), ng.GCNConv((d1, d1), "relu", initialgraph=g1))
            node1 = ng.NeuralODE(rhs1, solver="euler", n_steps=steps1, dt=0.1)
            ps1, st1 = ng.setup(0, node1)
            ps1 = ng.to_device(ps1, dev)
            for v in _grad_leaves(ps1):
                v.requires_grad_(True)
            u1 = torch.as_tensor(S.normal(51, d1 * n1).reshape(n1, d1).astype(np.float32), device=dev).T.requires_grad_(True)

            def solve1():
                for v in [u1] + _grad_leaves(ps1):
                    v.grad = None
                uT, _ = node1(u1, ps1, st1)
                uT.sum().backward()
            ms1e = _time_ms(solve1, 10)
            try:      # the solve is ~0.2 ms of device time: the eager call is bound by Python's dispatch, so the figure of record is the HIP-graph replay
                ms1 = _graph_ms(solve1, 20)
            except Exception:  # noqa: BLE001 -- a plan that cannot be captured keeps the eager figure
                ms1 = ms1e
            c1[name] = {"ms_solve_forward_backward": round(ms1, 3), "ms_solve_forward_backward_eager_api": round(ms1e, 3),
                        "value": round(steps1 / (ms1 * 1e-3), 1), "unit": "ODE-steps/s",
                        "plan": sorted({f for pool in node1._plans.values() for q in pool for f in q.flags()})}
        out["C1_cora_gcn32_eulerx10"] = {"nodes": n1, "edges": 2 * pairs1, **c1["cora_like"],
                                         "same_size_without_hubs": c1["same_size_without_hubs"]}
    if world == 1:
        # NeuralODE(VMHConv(phi, gamma)) of docs/src/tutorials/VMH.md:75-89 at the tutorial's size: 3 000 points in the unit square,
        # the 6 nearest neighbours of every point (the tutorial links Delaunay neighbours), h = 1, positions as node data, phi = 4 =>
        # 60 => 60 => 60 => 40 and gamma = 41 => 60 => 60 => 60 => 1, tanh.  Generic solver (every stage the layer's own kernels --
        # phi is deeper than the fused message path takes --, every combination one launch), captured into HIP graphs.
        nv, kv, steps_v = 3000, 6, 20
        pts = torch.as_tensor(S.uniform01(41, 2 * nv).reshape(2, nv).astype(np.float32), device=dev)
        gk = ng.knn_graph(pts, kv)
        gv = ng.GNNGraph(gk, ndata={"x": pts})
        phi = ng.Chain(ng.Dense(4, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 40))
        gam = ng.Chain(ng.Dense(41, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 60, "tanh"), ng.Dense(60, 1))
        nodev = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=gv), solver="tsit5", n_steps=steps_v, dt=0.2 / steps_v, capture=True)
        psv_, stv_ = ng.setup(4, nodev)
        psv_ = ng.to_device(psv_, dev)
        for v in _grad_leaves(psv_):
            v.requires_grad_(True)
        uv = torch.as_tensor(S.normal(42, nv).reshape(1, nv).astype(np.float32), device=dev).requires_grad_(True)

        def solvev():
            for v in [uv] + _grad_leaves(psv_):
                v.grad = None
            uT, _ = nodev(uv, psv_, stv_)
            uT.sum().backward()
        msv = _time_ms(solvev, 5)
        flags_v = sorted({f for pool in nodev._plans.values() for pl in pool for f in pl.flags()})
        out["VMH_node_tsit5x20"] = {"nodes": nv, "edges": int(gv.num_edges), "ode_steps": steps_v, "ms_solve_forward_backward": round(msv, 3),
                                    "value": round(steps_v / (msv * 1e-3), 1), "unit": "ODE-steps/s",
                                    "path": ("device-resident plan (ngpde_node_vmh_*): one forward and one adjoint launch per solve + a "
                                             "weight-pullback GEMM per layer over the tapes") if "vmh" in flags_v
                                            else "NeuralODE(VMHConv, capture=True): HIP-graph replay of the generic solver",
                                    "plan_flags": flags_v}
        # the arithmetic of one right-hand-side evaluation and its pullback (products of the two Dense chains: forward, input gradient,
        # weight gradient), priced against the fp32-MFMA peak as C4 / C5 are
        ne_v = int(gv.num_edges)
        mac_phi = 4 * 60 + 60 * 60 + 60 * 60 + 60 * 40
        mac_gam = 41 * 60 + 60 * 60 + 60 * 60 + 60 * 1
        gflop_v = 3 * 2 * (ne_v * mac_phi + nv * mac_gam) * 6 * steps_v / 1e9
        ach_v = gflop_v / msv
        out["VMH_node_tsit5x20"]["roofline"] = {"bound": "mfma", "achieved": round(ach_v, 2), "peak": FP32_MFMA_PEAK_TFS, "unit": "TFLOP/s",
                                                "frac": round(ach_v / FP32_MFMA_PEAK_TFS, 4), "algorithmic_GFLOP_solve_and_adjoint": round(gflop_v, 2)}
        if "vmh" in flags_v:   # the same solve on the generic solver (every stage the layer's own kernels), captured into HIP graphs
            os.environ["NGPDE_NO_VMH_NODE"] = "1"
            try:
                nodeg = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=gv), solver="tsit5", n_steps=steps_v, dt=0.2 / steps_v, capture=True)

                def solveg():
                    for v in [uv] + _grad_leaves(psv_):
                        v.grad = None
                    uT, _ = nodeg(uv, psv_, stv_)
                    uT.sum().backward()
                msg_ = _time_ms(solveg, 5)
                out["VMH_node_tsit5x20"]["generic_captured_ms"] = round(msg_, 3)
                out["VMH_node_tsit5x20"]["generic_captured_value"] = round(steps_v / (msg_ * 1e-3), 1)
            finally:
                os.environ.pop("NGPDE_NO_VMH_NODE", None)
        # the tutorial's minibatch form (VMH.md:120-134: a DataLoader batch of point clouds as ONE block-diagonal graph per step), 8 clouds of
        # 3 000 points: 752 tiles on 256 compute units -- the device-resident plan walks them in tile rounds.  (3 000 is not a multiple of
        # the 32-row tile: NeuralODE pads every cloud to whole tiles with isolated nodes, node.py: _padded_batch; plan_flags says which
        # path ran.)
        nb, nvb = 8, 3000
        gcl = []
        for kb in range(nb):
            pk = torch.as_tensor(S.uniform01(200 + kb, 2 * nvb).reshape(2, nvb).astype(np.float32), device=dev)
            gcl.append(ng.GNNGraph(ng.knn_graph(pk, kv), ndata={"x": pk}))
        gb = ng.batch(gcl)
        ub = torch.as_tensor(S.normal(44, nb * nvb).reshape(1, nb * nvb).astype(np.float32), device=dev).requires_grad_(True)
        res_b = {}
        for mode in ("plan", "generic"):
            if mode == "generic":
                os.environ["NGPDE_NO_VMH_NODE"] = "1"
            try:
                nodeb = ng.NeuralODE(ng.VMHConv(phi, gam, initialgraph=gb), solver="tsit5", n_steps=steps_v, dt=0.2 / steps_v, capture=(mode == "generic"))

                def solveb():
                    for v in [ub] + _grad_leaves(psv_):
                        v.grad = None
                    uT, _ = nodeb(ub, psv_, ng.updategraph(stv_, gb))
                    uT.sum().backward()
                res_b[mode] = (_time_ms(solveb, 3), sorted({f for pool in nodeb._plans.values() for pl in pool for f in pl.flags()}))
                del nodeb
            finally:
                os.environ.pop("NGPDE_NO_VMH_NODE", None)
        msb = res_b["plan"][0]
        out["VMH_node_batch8_tsit5x20"] = {"clouds": nb, "nodes": nb * nvb, "edges": int(gb.num_edges), "ms_solve_forward_backward": round(msb, 3),
                                           "value": round(nb * steps_v / (msb * 1e-3), 1), "unit": "trajectory ODE-steps/s",
                                           "plan_flags": res_b["plan"][1],
                                           "roofline": {"bound": "mfma", "achieved": round(nb * gflop_v / msb, 2), "peak": FP32_MFMA_PEAK_TFS, "unit": "TFLOP/s",
                                                        "frac": round(nb * gflop_v / msb / FP32_MFMA_PEAK_TFS, 4)},
                                           "generic_captured_ms": round(res_b["generic"][0], 3),
                                           "generic_captured_value": round(nb * steps_v / (res_b["generic"][0] * 1e-3), 1)}
    # C4: the per-GPU shard of the 512-trajectory config
    layer, ps, st, x, n_edges = c4_layer(dev, 64, rank)
    flat, psv = ng.optim.flatten_parameters(ng.to_device(ps, dev))
    if world == 1:
        fe, fbe, f, fb = _layer_times(layer, x, psv, st, 10)
        ach = C4_FWD_FLOP / (f * 1e-3) / 1e12
        out["C4_mppde_shard_layer"] = {"trajectories": 64, "nodes": 64 * 8192, "edges": n_edges, "ms_forward": round(f, 4),
                                       "ms_forward_backward": round(fb, 4),
                                       "ms_forward_eager_api": round(fe, 4), "ms_forward_backward_eager_api": round(fbe, 4),
                                    "ms_forward_graph_replay": round(_layer_times.graph_replay[0], 4),
                                    "ms_forward_backward_graph_replay": round(_layer_times.graph_replay[1], 4),
                                       "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFS,
                                                    "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFS, 4),
                                                    "algorithmic_GFLOP_forward": C4_FWD_FLOP / 1e9}}
        _kernel_sum_roofline(out["C4_mppde_shard_layer"], "c4", C4_FWD_FLOP, rocprof)
    else:
        st_opt = ng.optim.setup(ng.optim.Adam(1e-4), flat)
        red = ng.dist.OverlappedGradReduce(flat, psv, [("ψ.",), ("ϕ.",)])     # psi's gradient is final first: its collective
        xin = x.detach().requires_grad_(True)                                   # runs under phi's pullback
        R = torch.ones(x.shape[1], x.shape[0], device=dev).T

        def step():
            flat.zero_grad()
            layer(xin, psv, st)[0].backward(R)
            red.finish()
            ng.optim.update(st_opt, flat, reduced=True)
        for _ in range(2):
            step()
        dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 10
        for _ in range(reps):
            step()
        dist.barrier()
        torch.cuda.synchronize()
        tt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        ms = 1e3 * float(tt.item()) / reps
        out["C4_mppde_data_parallel_step"] = {"trajectories_per_rank": 64, "ranks": world, "ms_step": round(ms, 4),
                                              "value": round(world * 64 / (ms * 1e-3), 1), "unit": "trajectory-layers/s (fwd+bwd+all-reduce+Adam)",
                                              "gradient_floats": int(flat.numel()), "scaling": "weak"}
    if world == 1:
        for radius in (0.05, 0.1):
            l5, ps5, st5, x5, n_e5 = c5_layer(dev, radius)
            ps5 = ng.to_device(ps5, dev)
            for v in _grad_leaves(ps5):
                v.requires_grad_(True)
            fe, fbe, f, fb = _layer_times(l5, x5, ps5, st5, 10)
            flop5 = c5_fwd_flop(n_e5)
            ach = flop5 / (f * 1e-3) / 1e12
            out[f"C5_gno_128_r{radius}_layer"] = {"edges": n_e5, "ms_forward": round(f, 4), "ms_forward_backward": round(fb, 4),
                                                  "ms_forward_eager_api": round(fe, 4), "ms_forward_backward_eager_api": round(fbe, 4),
                                    "ms_forward_graph_replay": round(_layer_times.graph_replay[0], 4),
                                    "ms_forward_backward_graph_replay": round(_layer_times.graph_replay[1], 4),
                                                  "roofline": {"bound": "mfma", "achieved": round(ach, 2), "peak": FP32_MFMA_PEAK_TFS,
                                                               "unit": "TFLOP/s", "frac": round(ach / FP32_MFMA_PEAK_TFS, 4),
                                                               "algorithmic_GFLOP_forward": round(flop5 / 1e9, 2)}}
            _kernel_sum_roofline(out[f"C5_gno_128_r{radius}_layer"], f"c5_r{radius}", flop5, rocprof)
    return out


def visible_gpu_count(topology="/sys/class/kfd/kfd/topology/nodes", dri="/dev/dri"):
    """GPUs this process would see, WITHOUT the HIP runtime: the launcher parent must never touch the GPU (a process that has
    initialised HIP may not start ranks that re-use its state, and on this pool may not exec at all).  The kernel driver lists every
    agent under /sys/class/kfd/kfd/topology/nodes/<k>/properties; GPU agents are the ones with simd_count > 0.  The visibility
    variables of the runtime (ROCR_VISIBLE_DEVICES, HIP_VISIBLE_DEVICES, CUDA_VISIBLE_DEVICES: comma lists of indices / UUIDs; a
    negative index ends the list) narrow that set, and so does the container: an agent whose render node (drm_render_minor ->
    /dev/dri/renderD<minor>) this process cannot open is not counted.  Returns (count, source)."""
    import glob
    n = 0
    nodes = sorted(glob.glob(os.path.join(topology, "*", "properties")))
    for path in nodes:
        try:
            props = dict(line.split(None, 1) for line in open(path).read().splitlines() if " " in line)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            # a container or cgroup may expose only some /dev/dri/renderD* nodes while sysfs lists every GPU of the host: count the
            # agents whose render node this process can open (no drm_render_minor in the properties: count the agent)
            minor = props.get("drm_render_minor", "").strip()
            if minor.isdigit() and int(minor) > 0 and os.path.isdir(dri):
                if not os.access(os.path.join(dri, f"renderD{int(minor)}"), os.R_OK | os.W_OK):
                    continue
            n += 1
    if not nodes:
        return None, "no /sys/class/kfd"
    for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):
        v = os.environ.get(var)
        if v is not None:
            ids = []
            for t in v.split(","):
                t = t.strip()
                if t == "":
                    continue
                if t.lstrip("-").isdigit() and int(t) < 0:      # a negative index ends the list (the runtimes' rule): "-1" = no device
                    break
                ids.append(t)
            n = min(n, len(ids))
    return n, "kfd topology"


def spawn_ranks(args, backend):
    """`python3 bench.py --gpus N` typed as is (no torch.distributed.run around it): start the N ranks as fresh child processes
    of this one -- which has made no GPU call and makes none -- with the rendezvous in the environment, relay rank 0's JSON
    line, return non-zero if any rank fails.  With fewer than N devices under the nccl backend (RCCL refuses two ranks on one
    device) print ONE record that says so instead of a measurement."""
    import socket
    import subprocess
    n = args.gpus
    have, how = visible_gpu_count()          # sysfs only: this process never loads the HIP runtime
    if have is None:
        have, how = torch.cuda.device_count(), "torch.cuda.device_count()"   # (no kfd node: not a ROCm box; counting does not initialise HIP)
    if backend == "nccl" and have < n:
        print(json.dumps({"metric": "ODE-steps/sec (fwd+bwd) on 16k-node graph, 64-d feats", "value": None, "unit": "ODE-steps/s",
                          "n_gpus": n, "steps": args.steps, "warmup": args.warmup, "skipped": True,
                          "reason": f"--gpus {n} needs {n} devices for one RCCL rank per GPU; this box has {have}",
                          "devices_visible": have, "devices_counted_by": how, "higher_is_better": True, "scaling": "weak", "dtype": "f32", "data": "synthetic"}))
        return 0
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True if r == 0 else None))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    for pr in procs[1:]:
        try:
            rcs.append(pr.wait(timeout=120 if rcs[0] == 0 else 5))
        except subprocess.TimeoutExpired:
            pr.kill()                           # (the exact child started above)
            rcs.append(pr.wait())
    sys.stdout.write(out0 or "")
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        print(f"bench.py: ranks failed (rank, exit code): {bad}", file=sys.stderr)
        return 1
    return 0


ROCPROF_KERNEL_OF_ROLE = {"fwd_persistent_solve": "node_fwd_persistent_kernel", "bwd_persistent_adjoint": "node_bwd_persistent_kernel"}


ROCPROF_CHILD_STEPS = 20     # (3 until round 6: of its 14 launches the first five and the two behind a pause ran cold -- profiles/r06_t_*)


def rocprof_roofline(out, role, algo_bytes):
    """roofline.frac from rocprofv3's own average for the dominant kernel: a CHILD process repeats ROCPROF_CHILD_STEPS headline steps under
    `rocprofv3 --kernel-trace --stats` (the command of tools/profile_round.sh, whose summary is committed under profiles/), its
    k_kernel_stats.csv gives the kernel's AverageNs, and `achieved` / `frac` are recomputed from that -- it reads 1-4.5 % longer
    than the library's dispatch events, so it is the conservative one.  The event figures stay in the record as *_events.  When
    rocprofv3 is missing or the child fails the event figures remain and `avg_launch_source` says so."""
    import csv
    import shutil
    import subprocess
    import tempfile
    r = out["roofline"]
    r["frac_events"], r["achieved_events"], r["avg_launch_us_events"] = r["frac"], r["achieved"], r["avg_launch_us"]
    events = ("the library's own dispatch events (hipExtLaunchKernelGGL start / stop, median of five passes)")
    exe, want = shutil.which("rocprofv3"), ROCPROF_KERNEL_OF_ROLE.get(role)
    if exe is None or want is None or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        r["avg_launch_source"] = events + "; no rocprofv3 pass (not installed, already under the profiler, or not a persistent plan)"
        return
    tmp = tempfile.mkdtemp(prefix="ngpde_rocprof_", dir="/tmp")
    try:
        env = dict(os.environ, TMPDIR="/tmp")
        cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "-o", "k", "--", sys.executable,
               os.path.abspath(__file__), "--steps", str(ROCPROF_CHILD_STEPS), "--warmup", "3", "--no-cpu-baseline", "--batched", "0", "--no-secondary"]
        subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
        found = None
        for base, _, files in os.walk(tmp):
            for f in files:
                if f.endswith("kernel_stats.csv"):
                    for row in csv.DictReader(open(os.path.join(base, f))):
                        if want in row["Name"]:
                            found = (float(row["AverageNs"]) * 1e-3, int(row["Calls"]))
        if found is None:
            raise RuntimeError("kernel not in k_kernel_stats.csv")
        us, calls = found
        achieved = algo_bytes / (us * 1e-6) / 1e9
        r["achieved"], r["frac"], r["avg_launch_us"] = round(achieved, 1), round(achieved / HBM_PEAK_GBS, 4), round(us, 3)
        r["avg_launch_source"] = (f"rocprofv3 --kernel-trace --stats AverageNs over {calls} launches of {want}, collected by a child run of "
                                  f"this script (--steps {ROCPROF_CHILD_STEPS} --warmup 3, the command of tools/profile_round.sh: a device that was idle runs its first four or "
                                  "five launches 3 - 20 % slower, so a pass of 3 steps read 5 % above the timed region's launches); *_events: " + events)
    except Exception as e:  # noqa: BLE001 -- the bench line must still come out
        r["avg_launch_source"] = events + f"; the rocprofv3 child pass failed ({type(e).__name__}: {e})"
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


CHILD_LAYER_REPS = 53      # (a prime: the layer's own kernels are the rows of the stats table whose call count is a multiple of it)


def child_layer(spec):
    """`bench.py --child-layer c4:fwd` (run by rocprof_layer_kernel_sum under rocprofv3): CHILD_LAYER_REPS calls of one layer leg and
    nothing else in a loop -- forward without autograd, or forward + backward -- so that the profiler's per-kernel totals divided by
    the repetitions are the device time of ONE call without the gaps between its launches."""
    name, mode = spec.split(":")
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    if name == "c4":
        layer, ps, st, x, _ = c4_layer(dev, 64, 0)
        _, ps = ng.optim.flatten_parameters(ng.to_device(ps, dev))
    else:
        layer, ps, st, x, _ = c5_layer(dev, float(name.split("_r")[1]))
        ps = ng.to_device(ps, dev)
        for v in _grad_leaves(ps):
            v.requires_grad_(True)
    if mode == "fwd":
        with torch.no_grad():
            for _ in range(CHILD_LAYER_REPS):
                layer(x, ps, st)
    else:
        x = x.detach().requires_grad_(True)
        leaves = [x] + _grad_leaves(ps)
        y0 = None
        for _ in range(CHILD_LAYER_REPS):
            for v in leaves:
                v.grad = None
            y = layer(x, ps, st)[0]
            if y0 is None:
                y0 = torch.ones_like(y)
            y.backward(y0)
    torch.cuda.synchronize()


def rocprof_layer_kernel_sum(name, mode):
    """ms of device time of one call of a layer leg = (sum over the layer's kernels of rocprofv3 --kernel-trace --stats'
    TotalDurationNs) / repetitions, from a child pass (child_layer).  None when the profiler is absent or the child fails."""
    import csv
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None
    tmp = tempfile.mkdtemp(prefix="ngpde_rocprof_", dir="/tmp")
    try:
        cmd = [exe, "--kernel-trace", "--stats", "--output-format", "csv", "-d", tmp, "-o", "k", "--", sys.executable, os.path.abspath(__file__),
               "--child-layer", f"{name}:{mode}"]
        subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp"), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300, check=True)
        total, kernels = 0.0, 0
        for base, _, files in os.walk(tmp):
            for f in files:
                if f.endswith("kernel_stats.csv"):
                    for row in csv.DictReader(open(os.path.join(base, f))):
                        calls = int(row["Calls"])
                        # the layer's launches: ngpde kernels launched a multiple of the repetitions (the handle's construction and
                        # the first call's one-off launches are not)
                        if "ngpde" in row["Name"] and calls >= CHILD_LAYER_REPS and calls % CHILD_LAYER_REPS == 0:
                            total += float(row["TotalDurationNs"])
                            kernels += calls // CHILD_LAYER_REPS
        return (total / CHILD_LAYER_REPS * 1e-6, kernels) if kernels else None
    except Exception:  # noqa: BLE001 -- the bench line must still come out
        return None
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _kernel_sum_roofline(rec, name, flop_fwd, allow):
    """roofline.frac of a matrix-pipe-bound layer leg from the rocprofv3 kernel sum of its forward (the conservative figure: it reads
    longer than an un-profiled graph replay, as the headline's does); the graph-replay fraction stays beside it"""
    r = rec["roofline"]
    r["frac_graph_replay"], r["achieved_graph_replay"] = r["frac"], r["achieved"]
    r["frac_source"] = "graph replay (no rocprofv3 child pass)"
    if not allow:
        return
    ks_f, ks_fb = rocprof_layer_kernel_sum(name, "fwd"), rocprof_layer_kernel_sum(name, "fwdbwd")
    if ks_f is None:
        return
    ach = flop_fwd / (ks_f[0] * 1e-3) / 1e12
    r["achieved"], r["frac"] = round(ach, 2), round(ach / FP32_MFMA_PEAK_TFS, 4)
    r["kernel_sum_ms_forward"], r["kernels_per_forward"] = round(ks_f[0], 4), ks_f[1]
    if ks_fb is not None:
        r["kernel_sum_ms_forward_backward"], r["kernels_per_forward_backward"] = round(ks_fb[0], 4), ks_fb[1]
        # forward + pullback = three times the forward's products (input and weight gradients of every Dense / contraction)
        r["frac_forward_backward"] = round(3 * flop_fwd / (ks_fb[0] * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFS, 4)
    r["frac_source"] = (f"rocprofv3 --kernel-trace --stats, child pass of {CHILD_LAYER_REPS} calls: sum of the launches' TotalDurationNs / calls "
                        "(no gaps between kernels; profiled runs read a few % longer than un-profiled ones)")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C3 / C4 / C5 layer measurements ('secondary')")
    ap.add_argument("--no-rocprof", action="store_true",
                    help="N=1: do not re-run a short headline pass under rocprofv3 --kernel-trace --stats for roofline.frac "
                         "(implied by --no-secondary; the fraction then comes from the library's dispatch events)")
    ap.add_argument("--batched", type=int, default=8,
                    help="N=1 only: after the BASELINE measurement, also time this many trajectories per GPU as one batched "
                         "graph (reported under 'batched', never in 'value'); 0 = skip")
    ap.add_argument("--child-layer", default=None, help=argparse.SUPPRESS)   # (internal: one layer leg in a loop, for the rocprofv3 child pass)
    args = ap.parse_args()
    if args.child_layer:
        child_layer(args.child_layer)
        return

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    # NGPDE_BENCH_BACKEND=gloo: rehearsal of the N > 1 path on a box with fewer GPUs than ranks (ranks share devices, the
    # collective goes through the host); the measured configuration is always nccl (= RCCL over xGMI), one rank per GPU
    backend = os.environ.get("NGPDE_BENCH_BACKEND", "nccl")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # plain `python3 bench.py --gpus N`: this process becomes the launcher.  It has not touched the GPU (devices are counted
        # from the kernel driver's sysfs topology, visible_gpu_count) and never will: the ranks are fresh child processes
        sys.exit(spawn_ranks(args, backend))
    if args.gpus != world:
        print(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        sys.exit(2)
    # Rehearsal on a shared device: two persistent solver launches in flight on one device can starve each other of residency
    # (include/ngpde.h, ngpde_node_flags).  Either the ranks take turns on the device through a file lock
    # (NGPDE_BENCH_SERIALISE=<lock file>: persistent plan + collective + Adam on one stream, one rank's solve at a time) or
    # they take the replayed plan
    lock_path = os.environ.get("NGPDE_BENCH_SERIALISE") if world > 1 else None
    if backend != "nccl" and not lock_path:
        os.environ.setdefault("NGPDE_NO_PERSISTENT", "1")
    dev_index = local_rank if backend == "nccl" else local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)   # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    # Under nccl the gradient all-reduce + Adam go through the library's own communicator (ngpde_comm_*, RCCL behind the C ABI: the
    # call a Julia host makes); NGPDE_BENCH_COMM=torch keeps torch.distributed's all_reduce + ngpde_adam_step.  Same collective, same
    # stream.  The ranks AGREE on the choice (a rank whose communicator failed would otherwise wait in a collective the others never
    # enter), and the line carries what RCCL itself reports: ncclCommCount / ncclCommUserRank and a checked sum over the ranks.
    native, comm_record = None, None
    if dist is not None:
        comm_record = {"impl": "torch.distributed all_reduce (" + backend + ")", "world": dist.get_world_size()}
    if dist is not None and backend == "nccl" and os.environ.get("NGPDE_BENCH_COMM", "native") == "native":
        err = None
        try:
            native = ng.dist.NativeComm.from_torch()
        except Exception as e:                        # (librccl.so missing, ncclCommInitRank refused, ...)
            err = f"{type(e).__name__}: {e}"
        okv = torch.tensor([0 if native is None else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(okv, op=dist.ReduceOp.MIN)
        if int(okv.item()) == 1:
            probe = torch.full((257,), float(rank + 1), dtype=torch.float32, device=dev)
            native.all_reduce(probe)                   # sum over the ranks of (rank + 1), on the bench's stream
            torch.cuda.synchronize()
            want = world * (world + 1) / 2.0
            cnt, urank = native.rccl_count_and_rank()
            comm_record = {"impl": "ngpde_grad_allreduce_adam (RCCL behind the C ABI)", "world": native.world, "rccl_comm_count": cnt,
                           "rccl_user_rank_of_rank0": urank, "probe_sum_ok": bool((probe == want).all().item())}
            if not comm_record["probe_sum_ok"] or cnt != world:
                raise RuntimeError(f"native communicator: {cnt} ranks met, probe sum {float(probe[0])} (expected {want})")
        else:
            if native is not None:
                native.close()
            native = None
            comm_record["native_fallback"] = err or "another rank could not create its communicator"
    s, t, u0_h, w1_h, b1_h, w2_h, b2_h = make_inputs(rank)
    g = ng.GNNGraph(s, t, num_nodes=N_NODES, index_base=0)
    lib = _lib.load()
    dv = lambda a: torch.as_tensor(a, device=dev)
    stream = torch.cuda.current_stream().cuda_stream
    p = _lib.ptr

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    class device_turn:
        """NGPDE_BENCH_SERIALISE rehearsal only: the ranks share ONE device, and a persistent solve needs all of it -- a rank
        holds the lock file from its first launch until its stream has drained.  A no-op in the measured configuration."""
        def __enter__(self):
            if lock_path:
                import fcntl
                self.f = open(lock_path, "a+")
                fcntl.flock(self.f, fcntl.LOCK_EX)
            return self

        def __exit__(self, *exc):
            if lock_path:
                import fcntl
                torch.cuda.synchronize()
                fcntl.flock(self.f, fcntl.LOCK_UN)
                self.f.close()
            return False

    def job(traj, n_steps, n_warmup):
        """`traj` independent trajectories on this GPU as ONE block-diagonal batched graph (traj = 1: the BASELINE workload).
        Returns (elapsed seconds of n_steps bench steps, max over ranks; forward / backward ms of one solve; plan)."""
        # traj > 1: a block-diagonal batch of `traj` graphs with ONE structure.  The persistent plan solves its members one
        # after the other on the member's handle; where it is not available the batch becomes one big derived graph.
        try:
            plan = _Plan(g.handle((True, None, False)), D, _lib.ACT["relu"], "tsit5", ODE_STEPS, DT, True, members=traj)
        except _lib.NgpdeError as e:
            if traj == 1 or e.code != _lib.ERR_UNSUPPORTED:
                raise
            plan = _Plan(ng.batch([g] * traj).handle((True, None, False)), D, _lib.ACT["relu"], "tsit5", ODE_STEPS, DT, True)
        u0 = dv(u0_h) if traj == 1 else torch.cat(
            [dv(S.normal(1000 + rank + 97 * k, D * N_NODES).reshape(N_NODES, D).astype(np.float32)) for k in range(traj)])
        # parameters as ONE flat vector [w1 | b1 | w2 | b2] (the reference's ComponentArray), gradients likewise
        pflat = torch.cat([dv(w1_h).reshape(-1), dv(b1_h), dv(w2_h).reshape(-1), dv(b2_h)]).contiguous()
        w1, b1 = pflat[:D * D], pflat[D * D:D * D + D]
        w2, b2 = pflat[D * D + D:2 * D * D + D], pflat[2 * D * D + D:]
        adam_m, adam_v = torch.zeros_like(pflat), torch.zeros_like(pflat)
        uT, du0 = torch.empty_like(u0), torch.empty_like(u0)
        seed_grad = torch.ones_like(u0)                       # d sum(u(T)) / d u(T)
        flat = torch.empty(2 * (D * D + D), dtype=torch.float32, device=dev)   # [dw1 | db1 | dw2 | db2]
        dw1, db1 = flat[:D * D], flat[D * D:D * D + D]
        dw2, db2 = flat[D * D + D:2 * D * D + D], flat[2 * D * D + D:]
        it = [0]

        def step():
            # one training step: solve, discrete adjoint, gradient all-reduce, fused Adam on the flat vector (1/world folded in)
            with device_turn():
                _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(uT), stream))
                _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed_grad), p(du0), p(dw1), p(db1), p(dw2), p(db2), stream))
            it[0] += 1
            if native is not None:
                native.all_reduce_adam(pflat, flat, adam_m, adam_v, 1e-5, 0.9, 0.999, 1e-8, it[0])
                return
            if dist is not None:
                dist.all_reduce(flat)
            _lib.check(lib.ngpde_adam_step(flat.numel(), p(pflat), p(flat), p(adam_m), p(adam_v), 1e-5, 0.9, 0.999, 1e-8, it[0],
                                           1.0 / world, stream))

        for _ in range(n_warmup):
            step()
        fence()
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(n_steps + 1)]   # per-step device time, for the median
        t0 = time.perf_counter()
        marks[0].record()
        for k in range(n_steps):
            step()
            marks[k + 1].record()
        fence()
        elapsed = time.perf_counter() - t0
        job.step_ms = sorted(marks[k].elapsed_time(marks[k + 1]) for k in range(n_steps))
        if dist is not None:
            tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            elapsed = float(tt.item())
        # forward / backward split of one solve, by HIP events on the launch stream (outside the timed region)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
        with device_turn():
            ev[0].record()
            _lib.check(lib.ngpde_node_gcn2_forward(plan.ptr, p(u0), p(w1), p(b1), p(w2), p(b2), p(uT), stream))
            ev[1].record()
            _lib.check(lib.ngpde_node_gcn2_backward(plan.ptr, p(seed_grad), p(du0), p(dw1), p(db1), p(dw2), p(db2), stream))
            ev[2].record()
        torch.cuda.synchronize()
        return elapsed, ev[0].elapsed_time(ev[1]), ev[1].elapsed_time(ev[2]), plan

    def pipeline_stats(plan):
        # the interleaved batch solve (two members per workgroup): how many (tile, member, phase) units of the last launches
        # found their halo rows gathered ahead of time, i.e. paid no exposed hand-off
        af, ab, tot = C.c_int64(), C.c_int64(), C.c_int64()
        _lib.check(lib.ngpde_node_pipeline_stats(plan.ptr, stream, C.byref(af), C.byref(ab), C.byref(tot)))
        return {"members_per_workgroup": 2 if (plan.members > 1 and os.environ.get("NGPDE_NO_INTERLEAVE") != "1") else 1,
                "slot_phases": int(tot.value), "gathered_ahead_forward": int(af.value), "gathered_ahead_backward": int(ab.value)}

    def launch_profile(plan):
        # per-launch DEVICE time of the four kernel roles, from start/stop events attached to the dispatches
        # (a persistent plan has ONE launch per direction, i.e. one sample per pass: the median of five passes)
        us = (C.c_float * 4)()
        cnt = (C.c_int32 * 4)()
        samples = []
        for _ in range(5 if "persistent_fwd" in plan.flags() else 1):
            with device_turn():
                _lib.check(lib.ngpde_node_profile(plan.ptr, 1, us, cnt, stream))
            samples.append([float(us[i]) for i in range(4)])
        med = np.median(np.asarray(samples), axis=0)
        for i in range(4):
            us[i] = float(med[i])
        return us, cnt

    def api_job(n_steps, n_warmup):
        """The same bench step through the PUBLIC API (graph_node.md:118-132: model call, loss, gradient, Optimisers.update):
        NeuralODE.__call__ (plan lookup, autograd node, output allocation) + .backward() + optim.update on the flat parameter
        vector.  Returns seconds per step (wall, this rank)."""
        rhs = ng.Chain(ng.GCNConv((D, D), "relu", initialgraph=g), ng.GCNConv((D, D), "relu", initialgraph=g))
        node = ng.NeuralODE(rhs, solver="tsit5", n_steps=ODE_STEPS, dt=DT)
        _, st = ng.setup(0, node)
        ps0 = {"layer_1": {"weight": torch.as_tensor(np.ascontiguousarray(w1_h.T)), "bias": torch.as_tensor(b1_h).reshape(D, 1)},
               "layer_2": {"weight": torch.as_tensor(np.ascontiguousarray(w2_h.T)), "bias": torch.as_tensor(b2_h).reshape(D, 1)}}
        flat, ps = ng.optim.flatten_parameters(ps0, device=dev)
        st_opt = ng.optim.setup(ng.optim.Adam(1e-5), flat)
        u0 = dv(u0_h).T                                    # (D x N), the reference's layout: a column-major view, no copy

        def step():
            flat.zero_grad()
            uT, _ = node(u0, ps, st)
            uT.sum().backward()
            ng.optim.update(st_opt, flat)
        for _ in range(n_warmup):
            step()
        fence()
        t0 = time.perf_counter()
        for _ in range(n_steps):
            step()
        fence()
        return (time.perf_counter() - t0) / n_steps

    elapsed, ms_fwd, ms_bwd, plan = job(1, args.steps, args.warmup)
    api_s = api_job(args.steps, args.warmup) if world == 1 else None
    step_ms = list(job.step_ms)      # this rank's per-step device times of the timed region (`value` stays K steps / wall time)

    def role_table(plan, traj):
        """kernel roles of a plan, their SURVEY 8(d) algorithmic bytes per launch and their measured device time per launch"""
        us, cnt = launch_profile(plan)
        evals = 6 * ODE_STEPS                                  # right-hand-side evaluations (= pullbacks) per solve
        if "persistent_fwd" in plan.flags():                   # ONE launch per direction: 2 layer evaluations per RHS evaluation
            roles = ["fwd_persistent_solve", None, "bwd_persistent_adjoint", None]
            algo = [2 * evals * BYTES_FWD_LAYER, 0.0, 2 * evals * BYTES_BWD_LAYER, 0.0]
        else:
            roles = ["fwd_layer1", "fwd_layer2_stage", "bwd_layer1", "bwd_stage_layer2"]
            algo = [BYTES_FWD_LAYER, BYTES_FWD_LAYER, BYTES_BWD_LAYER, BYTES_BWD_LAYER]
        algo = [traj * a for a in algo]
        kernels = {r: {"avg_us": round(float(us[i]), 3), "launches_per_solve": int(cnt[i]),
                       "algorithmic_MB": round(algo[i] / 1e6, 2),
                       "achieved_GBs": round(algo[i] / (float(us[i]) * 1e-6) / 1e9, 1) if us[i] > 0 else None}
                   for i, r in enumerate(roles) if r is not None}
        tot = [float(us[i]) * int(cnt[i]) if roles[i] else 0.0 for i in range(4)]
        dom = int(np.argmax(tot))
        return roles, algo, us, kernels, dom

    out = None
    if rank == 0:
        roles, algo, us, kernels, dom = role_table(plan, 1)
        achieved = algo[dom] / (float(us[dom]) * 1e-6) / 1e9
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")   # PMC-derived HBM bytes per launch, if collected
        if os.path.exists(tpath):
            try:
                traffic = json.load(open(tpath)).get(roles[dom])
            except Exception:
                traffic = None
        fl, bl = plan.launch_count()
        value = world * ODE_STEPS * args.steps / elapsed
        out = {
            "metric": "ODE-steps/sec (fwd+bwd) on 16k-node graph, 64-d feats",
            "value": round(value, 2), "unit": "ODE-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 4),
            "median_ms_per_step": round(float(np.median(step_ms)), 4),
            "median_value": round(world * ODE_STEPS / (float(np.median(step_ms)) * 1e-3), 2),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32",
            "data": "synthetic",
            "config": {"workload": "C2: 16384-node / 131072-edge closest-pairs radius graph, 64-d feats, "
                                   "RHS = 2 x GCNConv(64=>64, relu, self-loops), Tsit5 fixed dt=1/50 x 50 steps, "
                                   "forward + discrete adjoint of sum(u(T))",
                       "bench_step": "one full solve forward + backward (+ gradient all-reduce when N > 1) + fused Adam step",
                       "ode_steps_per_solve": ODE_STEPS, "rhs_evals_per_ode_step": 6,
                       "launches_per_solve": {"forward": fl, "backward": bl},
                       "ms_forward_solve": round(ms_fwd, 3), "ms_backward_solve": round(ms_bwd, 3),
                       "tape_GB": round(plan.tape_bytes() / 1e9, 3),
                       "parallelism": f"dp{world} (independent trajectories, all-reduce of 8320-float grads)",
                       "collective": comm_record if comm_record is not None else {"impl": "none (one rank)", "world": 1}},
            "roofline": {"bound": "hbm", "kernel": roles[dom], "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS,
                         "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
                         "traffic_source": "static: profiles/traffic.json, the rocprofv3 --pmc passes of tools/profile_round.sh "
                                           "(FETCH_SIZE x 2 + WRITE_SIZE per launch), not collected in this run",
                         "algorithmic_MB_per_launch": round(algo[dom] / 1e6, 2),
                         "avg_launch_us": round(float(us[dom]), 3),
                         "avg_launch_source": "the library's own dispatch events (hipExtLaunchKernelGGL start / stop, median of five passes); "
                                              "rocprofv3 --kernel-trace reads ~1-4.5 % longer for the same launch (profiles/r05_k_kernel_stats.csv: 2.953 ms, frac 0.444)"},
            "kernels": kernels,
            "plan": sorted(plan.flags()), "fault": bool(plan.fault()),
        }
        if api_s is not None:
            # the headline drives the C ABI directly (the call sequence a Julia shim would make); this is the same step through
            # the Python mirror of the reference's API -- plan pool, autograd, tensor allocation and loss included
            out["api_ms_per_step"] = round(1000.0 * api_s, 4)
            out["api_value"] = round(ODE_STEPS / api_s, 2)
        if world == 1 and args.batched > 1:
            # Secondary: the same kernels when a launch is no longer ONE wave of workgroups.  `traj` trajectories of the
            # BASELINE workload on this GPU as one block-diagonal batched GNNGraph (test/runtests.jl:89-102); each launch
            # then processes traj x the algorithmic bytes.  Not the headline: `value` above is one trajectory per GPU.
            plan = None
            nb = max(2, args.steps // 3)
            eb, fb, bb, planb = job(args.batched, nb, 1)
            rolesb, algob, usb, kernelsb, db = role_table(planb, args.batched)
            out["batched"] = {
                "trajectories_per_gpu": args.batched, "nodes": args.batched * N_NODES,
                "value": round(args.batched * ODE_STEPS * nb / eb, 1), "unit": "trajectory ODE-steps/s",
                "ms_forward_solve": round(fb, 3), "ms_backward_solve": round(bb, 3),
                "tape_GB": round(planb.tape_bytes() / 1e9, 3),
                "roofline": {"kernel": rolesb[db], "avg_launch_us": round(float(usb[db]), 3),
                             "achieved": round(algob[db] / (float(usb[db]) * 1e-6) / 1e9, 1), "unit": "GB/s",
                             "frac": round(algob[db] / (float(usb[db]) * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)},
                "kernels_avg_us": {r: round(float(usb[i]), 3) for i, r in enumerate(rolesb) if r is not None},
                "plan": sorted(planb.flags()),
                "pipeline": pipeline_stats(planb),
            }
            planb = None
        if world == 1 and args.batched > 1:
            # The same solve on a graph of TWICE the size (32 768 nodes / 262 144 edges, 1 024 tiles: more than the 512 workgroups
            # that can be resident): the persistent launches with two tiles per workgroup.  Not the headline; here to show what a
            # graph that is not the bench's own size runs at.
            n2 = 2 * N_NODES
            _, s2, t2 = S.closest_pairs_graph(n2, 2 * N_PAIRS, seed=GRAPH_SEED + 1)
            g2 = ng.GNNGraph(s2, t2, num_nodes=n2, index_base=0)
            plan2 = _Plan(g2.handle((True, None, False)), D, _lib.ACT["relu"], "tsit5", ODE_STEPS, DT, True)
            u2 = dv(S.normal(2000, D * n2).reshape(n2, D).astype(np.float32))
            uT2, du2, seed2 = torch.empty_like(u2), torch.empty_like(u2), torch.ones_like(u2)
            w1d, b1d, w2d, b2d = dv(w1_h), dv(b1_h), dv(w2_h), dv(b2_h)
            gw = [torch.empty_like(w1d), torch.empty_like(b1d), torch.empty_like(w2d), torch.empty_like(b2d)]

            def solve2():
                _lib.check(lib.ngpde_node_gcn2_forward(plan2.ptr, p(u2), p(w1d), p(b1d), p(w2d), p(b2d), p(uT2), stream))
                _lib.check(lib.ngpde_node_gcn2_backward(plan2.ptr, p(seed2), p(du2), p(gw[0]), p(gw[1]), p(gw[2]), p(gw[3]), stream))
            ms2 = _time_ms(solve2, 5)
            bytes2 = 2.0 * 2 * 6 * ODE_STEPS * (BYTES_FWD_LAYER + BYTES_BWD_LAYER)     # twice the nodes and edges of C2
            out["larger_graph"] = {"nodes": n2, "edges": int(s2.size), "tiles": n2 // 32, "value": round(ODE_STEPS / (ms2 * 1e-3), 1),
                                   "unit": "ODE-steps/s", "ms_per_solve_forward_backward": round(ms2, 3), "plan": sorted(plan2.flags()),
                                   "fault": bool(plan2.fault()),
                                   "roofline": {"bound": "hbm", "achieved": round(bytes2 / (ms2 * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                                "unit": "GB/s", "frac": round(bytes2 / (ms2 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                                                "what": "whole solve + adjoint: algorithmic bytes of 2 x C2 per right-hand-side evaluation"}}
            plan2 = None
            # ... and of FOUR times the size (65 536 nodes, 2 048 tiles): k tiles per workgroup taking turns ("tile rounds")
            n4 = 4 * N_NODES
            _, s4, t4 = S.closest_pairs_graph(n4, 4 * N_PAIRS, seed=GRAPH_SEED + 2)
            g4 = ng.GNNGraph(s4, t4, num_nodes=n4, index_base=0)
            plan4 = _Plan(g4.handle((True, None, False)), D, _lib.ACT["relu"], "tsit5", ODE_STEPS, DT, True)
            u4 = dv(S.normal(2001, D * n4).reshape(n4, D).astype(np.float32))
            uT4, du4, seed4 = torch.empty_like(u4), torch.empty_like(u4), torch.ones_like(u4)

            def solve4():
                _lib.check(lib.ngpde_node_gcn2_forward(plan4.ptr, p(u4), p(w1d), p(b1d), p(w2d), p(b2d), p(uT4), stream))
                _lib.check(lib.ngpde_node_gcn2_backward(plan4.ptr, p(seed4), p(du4), p(gw[0]), p(gw[1]), p(gw[2]), p(gw[3]), stream))
            ms4 = _time_ms(solve4, 3)
            bytes4 = 4.0 * 2 * 6 * ODE_STEPS * (BYTES_FWD_LAYER + BYTES_BWD_LAYER)
            out["larger_graph"]["x4"] = {"nodes": n4, "edges": int(s4.size), "tiles": n4 // 32, "value": round(ODE_STEPS / (ms4 * 1e-3), 1),
                                         "unit": "ODE-steps/s", "ms_per_solve_forward_backward": round(ms4, 3), "plan": sorted(plan4.flags()),
                                         "fault": bool(plan4.fault()),
                                         "roofline": {"bound": "hbm", "achieved": round(bytes4 / (ms4 * 1e-3) / 1e9, 1), "peak": HBM_PEAK_GBS,
                                                      "unit": "GB/s", "frac": round(bytes4 / (ms4 * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)}}
            plan4 = None
            # The C2 solve in the shapes that are not the headline's: d = 16 / 32 (zero-padded on the 64-wide persistent kernels) and a
            # graph WITH edge weights (GCNConv(use_edge_weight=true), src/layers.jl:206-231: tile-round kernels, slot weights in LDS)
            variants = {}
            ew = (0.25 + S.uniform01(77, int(s.size))).astype(np.float32)
            gwt = ng.GNNGraph(s, t, num_nodes=N_NODES, index_base=0, edge_weight=ew)
            for name, dd, handle in (("d16", 16, g.handle((True, None, False))), ("d32", 32, g.handle((True, None, False))),
                                     ("weighted_d64", D, gwt.handle((True, gwt.edge_weight, False)))):
                planv = _Plan(handle, dd, _lib.ACT["relu"], "tsit5", ODE_STEPS, DT, True)
                uv = dv(S.normal(3000 + dd, dd * N_NODES).reshape(N_NODES, dd).astype(np.float32))
                uTv, duv, seedv = torch.empty_like(uv), torch.empty_like(uv), torch.ones_like(uv)
                wv = [dv(S.glorot_uniform(50 + k, dd, dd).astype(np.float32)) for k in range(2)]
                bv = [torch.zeros(dd, device=dev) for _ in range(2)]
                gv = [torch.empty_like(wv[0]), torch.empty_like(bv[0]), torch.empty_like(wv[1]), torch.empty_like(bv[1])]

                def solvev():
                    _lib.check(lib.ngpde_node_gcn2_forward(planv.ptr, p(uv), p(wv[0]), p(bv[0]), p(wv[1]), p(bv[1]), p(uTv), stream))
                    _lib.check(lib.ngpde_node_gcn2_backward(planv.ptr, p(seedv), p(duv), p(gv[0]), p(gv[1]), p(gv[2]), p(gv[3]), stream))
                msv = _time_ms(solvev, 5)
                variants[name] = {"d": dd, "value": round(ODE_STEPS / (msv * 1e-3), 1), "unit": "ODE-steps/s",
                                  "ms_per_solve_forward_backward": round(msv, 3), "plan": sorted(planv.flags()), "fault": bool(planv.fault())}
                planv = None
            out["variants"] = variants
        if world == 1 and not args.no_cpu_baseline:
            cb, outs = cpu_baseline(s, t, u0_h, w1_h, b1_h, w2_h, b2_h)
            out["cpu_baseline"] = cb
        else:
            out["cpu_baseline"] = None
    plan = None
    if out is not None and world == 1 and not (args.no_rocprof or args.no_secondary):
        rocprof_roofline(out, roles[dom], algo[dom])
    if not args.no_secondary:
        sec = secondary(dev, world, rank, dist, rocprof=(world == 1 and not args.no_rocprof))          # every rank takes part (N > 1: the data-parallel C4 step)
        if out is not None:
            out["secondary"] = sec
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if out is not None:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
