"""NeuralODE -- the caller of the hot path in the reference's tutorial
(/root/reference/docs/src/tutorials/graph_node.md:44-66): `dudt(u, p, t) = model(u, p, st)` integrated
by an explicit Runge-Kutta scheme.  BASELINE configs fix the step count (Euler x 10, Tsit5 x 50), so
this integrator is fixed-step and its pullback is the discrete adjoint.

When the right-hand side is Chain(GCNConv(d => d, act), GCNConv(d => d, act)) on one graph (the
tutorial's `node_chain`, graph_node.md:78) the whole solve and its adjoint run device-resident from
HIP graphs (ngpde_node_gcn2_*).  Any other right-hand side (a GAT-style layer, VMHConv as in
docs/src/tutorials/VMH.md:85-89, ...) is stepped stage by stage through the layers' own kernels with
every Runge-Kutta combination -- forward and in the discrete adjoint -- as one ngpde_rk_stage_combine
launch (_NodeGenericFn): no torch element-wise kernel on the path.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from .graphs import GNNGraph
from .layers import AbstractExplicitLayer, Chain, GCNConv, rows_of
from .layers_mp import GATConv, VMHConv, _dense_stack, _node_data, _wt_b
from .batches import _padded_batch, _canonical_batch
from .plans import _rows_index, _check_plan_shapes, _Plan, _NodeGCN2Fn, _OdePlan, _ode_desc, _NodeOdeFn  # noqa: F401 (tools import _Plan from here)

_TSIT5_A = [
    [],
    [0.161],
    [-0.008480655492356989, 0.335480655492357],
    [2.8971530571054935, -6.359448489975075, 4.3622954328695815],
    [5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525],
    [5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383],
]
_TSIT5_B = [0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081,
            2.324710524099774]
TABLEAUS = {"euler": ([[]], [1.0]), "tsit5": (_TSIT5_A, _TSIT5_B)}


# ---- any right-hand side: explicit RK stepping with every combination as ONE library launch ---------------------------------


def _combine(base, c_self, terms, coefs, out=None):
    """out = c_self * base + sum_k coefs[k] * terms[k]  (ngpde_rk_stage_combine; row-major [N][D] float32 tensors of one size)"""
    lib = _lib.load()
    ref = base if base is not None else terms[0]
    if out is None:
        out = torch.empty_like(ref, memory_format=torch.contiguous_format)
    arr = (C.c_void_p * max(len(terms), 1))(*[t.data_ptr() for t in terms])
    cf = (C.c_float * max(len(coefs), 1))(*[float(c) for c in coefs])
    _lib.check(lib.ngpde_rk_stage_combine(ref.numel(), float(c_self), _lib.ptr(base), len(terms), arr, cf, _lib.ptr(out),
                                          _lib.current_stream()))
    return out


def _dense(t):
    """a view of `t` whose memory order is its index order (the cotangent of a (out x in) weight arrives as the transpose of
    the kernels' [in][out] array): element-wise combinations then need no copy"""
    if t.is_contiguous():
        return t
    if t.dim() == 2 and t.T.is_contiguous():
        return t.T
    return t.contiguous()


def _leaves(tree):
    out = []
    for v in tree.values():
        if isinstance(v, dict):
            out += _leaves(v)
        else:
            out.append(v)
    return out


def _graph_leaves(st):
    """the GNNGraph leaves of a state tree, in traversal order (the leaves updategraph replaces, src/utils.jl:24-31)"""
    out = []
    if isinstance(st, dict):
        for v in st.values():
            if isinstance(v, GNNGraph):
                out.append(v)
            elif isinstance(v, dict):
                out += _graph_leaves(v)
    return out


def _rebuild(tree, it):
    return {k: (_rebuild(v, it) if isinstance(v, dict) else next(it)) for k, v in tree.items()}


def _rk_forward(node, u, ps_in, st, needs, fresh=False):
    """u(T) of the fixed-step solve; with `needs` also the tape: per step and stage the (stage input, stage output) pair whose
    autograd closure is the layers' pullback.  fresh: the stage inputs get version counters of their own (a captured solve
    refills its static input buffer before every replay; the closures are only ever run at capture time)"""
    a, b = TABLEAUS[node.solver]
    dt, S = node.dt, len(b)
    ucur, tape, st_out = u, [], st
    k_save = node.save_every
    saves = [ucur] if (k_save and node.save_start) else []      # saveat: the states at t0 (+ j saveat), as DiffEq's sol.u
    with (torch.enable_grad() if needs else torch.no_grad()):
        for n_step in range(node.n_steps):
            ks, pairs = [], []
            for i in range(S):
                terms = [ks[j] for j in range(i) if a[i][j] != 0.0]
                coefs = [dt * a[i][j] for j in range(i) if a[i][j] != 0.0]
                U = _combine(ucur, 1.0, terms, coefs) if terms else ucur
                if needs:
                    U = (U.data if fresh else U.detach()).requires_grad_(True)
                k, st_out = node.model(U.T, ps_in, st_out)
                k = rows_of(k)
                if k.shape != ucur.shape:
                    raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH,
                                                 f"DimensionMismatch: the right-hand side maps {tuple(ucur.shape[::-1])} to "
                                                 f"{tuple(k.shape[::-1])}; du/dt must have the shape of u")
                ks.append(k.detach())
                pairs.append((U, k))
            ucur = _combine(ucur, 1.0, ks, [dt * bi for bi in b])
            if needs:
                tape.append(pairs)
            if k_save and (n_step + 1) % k_save == 0:
                saves.append(ucur)
    if k_save:      # [T][N][D]: the memory layout of the reference's (D x N x T) array; rows written by library launches
        out = torch.empty((len(saves),) + tuple(ucur.shape), dtype=ucur.dtype, device=ucur.device)
        for j, s_ in enumerate(saves):
            _combine(s_, 1.0, [], [], out=out[j])
        ucur = out
    return ucur, tape, st_out


def _accumulate_many(pairs):
    """acc += g for every (acc, g) pair of dense float32 tensors of equal size (ngpde_accumulate_many: one launch per 24 arrays)"""
    if not pairs:
        return
    for a_, g_ in pairs:
        if a_.numel() != g_.numel() or a_.dtype != torch.float32 or g_.dtype != torch.float32:
            raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, "parameter cotangent and its accumulator differ in size or type")
    n = len(pairs)
    accs = (C.c_void_p * n)(*[a_.data_ptr() for a_, _ in pairs])
    gs = (C.c_void_p * n)(*[g_.data_ptr() for _, g_ in pairs])
    cnt = (C.c_int64 * n)(*[a_.numel() for a_, _ in pairs])
    _lib.check(_lib.load().ngpde_accumulate_many(n, accs, gs, cnt, _lib.current_stream()))


def _rk_backward(node, tape, duT, params, retain=False):
    """discrete adjoint of _rk_forward: (du0, cotangents of `params`)"""
    a, b = TABLEAUS[node.solver]
    dt, S = node.dt, len(b)
    acc = [None] * len(params)
    k_save = node.save_every
    off = 0 if node.save_start else -1           # saved state j is u after (j - off') steps: index of u_n is n / k_save + off
    lam = duT[node.n_steps // k_save + off] if k_save else duT
    for n_step, pairs in zip(range(len(tape) - 1, -1, -1), reversed(tape)):
        ubar = [None] * S
        for i in reversed(range(S)):
            terms = [ubar[j] for j in range(i + 1, S) if a[j][i] != 0.0 and ubar[j] is not None]
            coefs = [dt * a[j][i] for j in range(i + 1, S) if a[j][i] != 0.0 and ubar[j] is not None]
            if b[i] == 0.0 and not terms:
                continue
            kbar = _combine(lam, dt * b[i], terms, coefs)
            U, k = pairs[i]
            grads = torch.autograd.grad(k, [U] + params, kbar, allow_unused=True, retain_graph=retain)
            if grads[0] is not None:
                ubar[i] = grads[0].contiguous()
            pairs_ag = []
            for n, g in enumerate(grads[1:]):
                if g is None:
                    continue
                if acc[n] is None:
                    # own the accumulator in a layout _dense can view without a copy (an expanded / strided cotangent would make
                    # _dense return a temporary, and the sum written into it would be lost)
                    acc[n] = g if (g.is_contiguous() or (g.dim() == 2 and g.T.is_contiguous())) else g.contiguous()
                elif acc[n].stride() == g.stride():
                    # same layout for every stage's cotangent of one parameter: add in memory order
                    pairs_ag.append((_dense(acc[n]), _dense(g)))
                else:       # layouts differ (one transposed, one not): index-wise sum in the accumulator's own layout
                    pairs_ag.append((_dense(acc[n]), _dense(g.contiguous() if acc[n].is_contiguous() else g.T.contiguous().T)))
            _accumulate_many(pairs_ag)      # ONE launch for all parameters of this stage (sixteen arrays in the VMH tutorial's model)
        live = [x for x in ubar if x is not None]
        if k_save and n_step % k_save == 0 and n_step // k_save + off >= 0:
            live.append(duT[n_step // k_save + off])          # lambda(t_n) also carries the cotangent of the state saved there
        if live:
            lam = _combine(lam, 1.0, live, [1.0] * len(live))
    return lam, acc


def _inner_params(ps, leaves, fresh=False):
    """aliases of the parameters that are leaves of the stages' autograd closures, and the tree holding them (fresh: with
    version counters of their own, so that an optimiser's in-place update between the captures does not invalidate them)"""
    inner = [(p.data if fresh else p.detach()).requires_grad_(p.requires_grad) if isinstance(p, torch.Tensor) else p for p in leaves]
    return inner, _rebuild(ps, iter(inner))


class _NodeGenericFn(torch.autograd.Function):
    """Fixed-step explicit Runge-Kutta solve of du/dt = model(u) for ANY model made of this package's layers, and its discrete
    adjoint.  Forward: the stage inputs u + dt sum_j a_ij k_j and the step update are one ngpde_rk_stage_combine launch each,
    the stages are the layers' own kernels; every stage keeps its (input, output) pair with the layer's pullback closure.
    Backward: per stage, in reverse, K-bar_i = dt b_i lambda + dt sum_{j>i} a_ji U-bar_j (one launch), U-bar_i and the parameter
    cotangents from the stage's pullback (the layers' backward kernels), parameter gradients accumulated by one launch each,
    lambda += sum_j U-bar_j (one launch).  No torch element-wise kernel takes part.
    [docs/src/tutorials/VMH.md:85-89 NeuralODE(VMHConv); graph_node.md:44-66; BASELINE config 3 "GAT as ODE RHS"]"""

    @staticmethod
    def forward(ctx, u, node, ps, st, *leaves):
        needs = any(ctx.needs_input_grad)
        inner, ps_in = _inner_params(ps, leaves) if needs else (list(leaves), ps)
        uT, tape, _ = _rk_forward(node, u.detach().contiguous(), ps_in, st, needs)
        ctx.node, ctx.tape, ctx.inner = node, tape if needs else None, inner
        return uT

    @staticmethod
    def backward(ctx, duT):
        params = [p for p in ctx.inner if isinstance(p, torch.Tensor) and p.requires_grad]
        lam, acc = _rk_backward(ctx.node, ctx.tape, duT.contiguous(), params)
        ctx.tape = None
        it = iter(acc)
        out = [next(it) if (isinstance(p, torch.Tensor) and p.requires_grad) else None for p in ctx.inner]
        return (lam, None, None, None, *out)


class _NoCyclicGC:
    """No cyclic garbage collection while a HIP graph is being captured: a collection in the middle of a capture may finalise an older
    captured solve, and destroying ITS graphs is an operation the capturing stream does not permit (the capture dies with
    hipErrorStreamCaptureUnsupported).  torch.cuda.graph collects once before the capture starts; this keeps it that way until it ends."""

    def __enter__(self):
        import gc
        self.was = gc.isenabled()
        gc.disable()

    def __exit__(self, *exc):
        import gc
        if self.was:
            gc.enable()
        return False


class _CapturedSolve:
    """The generic solve as two HIP graphs (NeuralODE(..., capture=True)): the forward stepping loop -- every layer launch
    and every Runge-Kutta combination of all steps -- is captured once and replayed per call, and so is the whole discrete
    adjoint the first time a backward pass asks for it.  Static buffers: the input is copied in, u(T) and the cotangents are
    the graphs' own buffers (valid until the next call of the same NeuralODE on the same arguments).  The captures bake in the
    addresses of the parameters and of the graph's derived arrays: a call with other parameter tensors, another graph or
    another input shape captures anew (NeuralODE keeps the most recent few)."""

    def __init__(self, node, u, ps, st, leaves, needs):
        self.node, self.needs = node, needs
        self.leaves = list(leaves)                         # keeps the captured addresses alive
        self.inner, self.ps_in = _inner_params(ps, leaves, fresh=True) if needs else (list(leaves), ps)
        self.params = [p for p in self.inner if isinstance(p, torch.Tensor) and p.requires_grad] if needs else []
        self.u_static = u.detach().clone()
        self.st = st
        # one eager solve on a side stream first: graph handles, workspaces and lazily built tables exist before the capture
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            _rk_forward(node, self.u_static, self.ps_in, st, False)
        torch.cuda.current_stream().wait_stream(side)
        self.fwd_graph = torch.cuda.CUDAGraph()
        # (relaxed: a finaliser that frees device memory in the middle of the capture must not invalidate it)
        with _NoCyclicGC(), torch.cuda.graph(self.fwd_graph, capture_error_mode="relaxed"):
            self.uT_static, self.tape, _ = _rk_forward(node, self.u_static, self.ps_in, st, needs, fresh=True)
        self.bwd_graph = None
        self.generation = 0                                # forward replays so far: the single static tape belongs to the last one

    def forward(self, u):
        self.u_static.copy_(u)
        self.fwd_graph.replay()
        self.generation += 1
        return self.uT_static

    def backward(self, duT, generation):
        if generation != self.generation:
            # `y1 = node(u1); y2 = node(u2); (y1 + y2).backward()`: the second replay has overwritten the first solve's tape
            raise _lib.NgpdeError(_lib.ERR_STATE, "NeuralODE(capture=True): another forward solve has replaced this solve's tape "
                                                  "(a captured solve holds ONE tape); run each backward before the next forward "
                                                  "on the same arguments, or use capture=False")
        if self.bwd_graph is None:
            self.duT_static = duT.detach().clone()
            self.bwd_graph = torch.cuda.CUDAGraph()
            with _NoCyclicGC(), torch.cuda.graph(self.bwd_graph, pool=self.fwd_graph.pool(), capture_error_mode="relaxed"):
                self.lam_static, self.acc_static = _rk_backward(self.node, self.tape, self.duT_static, self.params, retain=True)
        else:
            self.duT_static.copy_(duT)
        self.bwd_graph.replay()
        return self.lam_static, self.acc_static


class _NodeCapturedFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, u, solve, *leaves):
        ctx.solve = solve
        uT = solve.forward(u.detach().contiguous()).clone()
        ctx.generation = solve.generation
        return uT

    @staticmethod
    def backward(ctx, duT):
        lam, acc = ctx.solve.backward(duT.contiguous(), ctx.generation)
        it = iter(acc)
        out = [next(it) if (isinstance(p, torch.Tensor) and p.requires_grad) else None for p in ctx.solve.inner]
        return (lam.clone(), None, *[None if g is None else g.clone() for g in out])


class NeuralODE(AbstractExplicitLayer):
    """NeuralODE(model; solver="tsit5", tspan=(0, 1), n_steps=..., dt=None)

    A Lux container with the single field `model`, so `ps` and `st` are the model's own
    (graph_node.md:44-52, :59-66).  `solver` is "euler" or "tsit5"; the step is fixed:
    dt = (tspan[1] - tspan[0]) / n_steps unless given.  capture=True (an extension, for right-hand sides other than the
    two-GCNConv chain, which is always device-resident): the whole solve and its adjoint are captured into HIP graphs at
    the first call and replayed afterwards (_CapturedSolve).
    """

    def __init__(self, model, *, solver="tsit5", tspan=(0.0, 1.0), n_steps=10, dt=None, capture=False, saveat=None, save_start=True):
        solver = solver.lower()
        if solver not in TABLEAUS:
            raise _lib.ArgumentError(_lib.ERR_INVALID_ARGUMENT, f"unknown solver {solver!r}; one of {list(TABLEAUS)}")
        self.model, self.solver, self.tspan, self.n_steps = model, solver, tuple(tspan), int(n_steps)
        self.dt = float(dt) if dt is not None else (self.tspan[1] - self.tspan[0]) / self.n_steps
        # saveat (VMH.md:85 `NeuralODE(gnn, tspan, Tsit5(); saveat=dt_train)`): the output is the solution at t0, t0 + saveat, ..., T --
        # a (D x N x T) array -- instead of u(T).  The step is fixed, so saveat must be a whole number of steps.
        self.save_every, self.save_start = 0, bool(save_start)
        if saveat is not None:
            k = round(float(saveat) / self.dt)
            if k < 1 or abs(k * self.dt - float(saveat)) > 1e-6 * max(abs(float(saveat)), 1.0) or self.n_steps % k:
                raise _lib.ArgumentError(_lib.ERR_INVALID_ARGUMENT, f"NeuralODE: saveat = {saveat} must be a whole number of steps "
                                                                  f"(dt = {self.dt}) that divides n_steps = {self.n_steps}")
            self.save_every = k
        self._plans = {}
        self._no_member_plan = False
        self.capture = bool(capture)      # generic right-hand sides: replay the whole solve / adjoint from HIP graphs
        self._captured = {}
        self._gat_ok = {}                 # (id(graph handle), heads) -> (handle, does the device-resident GAT solver take it?)

    def initialparameters(self, rng):
        return self.model.initialparameters(rng)

    def initialstates(self, rng):
        return self.model.initialstates(rng)

    def statelength(self):
        return self.model.statelength()

    # -- is the right-hand side the tutorial's two-GCNConv chain on one graph? ---------------------
    max_plans = 2   # cached solver-plan keys per NeuralODE (forward-only and forward+backward of the current graph)
    max_outstanding = 8   # plans per key: one per solve whose backward is still outstanding (each owns a tape)

    def _gcn2(self, ps, st):
        m = self.model
        if not (isinstance(m, Chain) and len(m.chain) == 2 and all(isinstance(l, GCNConv) for l in m.chain)):
            return None
        l1, l2 = m.chain
        d = l1.in_chs
        same = (l1.out_chs == d and l2.in_chs == d and l2.out_chs == d and l1.act == l2.act
                and l1.add_self_loops == l2.add_self_loops and l1.use_edge_weight == l2.use_edge_weight
                and l1.bias == l2.bias)
        if not same or d not in (16, 32, 64, 128):
            return None
        g1, g2 = st["layer_1"]["graph"], st["layer_2"]["graph"]
        # copies made by wrapgraph / updategraph share one handle cache: same structure without comparing arrays
        if g1 is not g2 and g1._handles is not g2._handles and g1 != g2:
            return None
        if l1.use_edge_weight and (g1.edge_weight is None or g2.edge_weight is not g1.edge_weight):
            return None      # (the generic solver raises the layer's own error / handles two weight vectors)
        return l1, g1, d

    def plan_for(self, ps, st, with_backward):
        info = self._gcn2(ps, st)
        if info is None:
            return None
        l1, g, d = info
        # use_edge_weight=true: the graph's stored weights in the messages, the unweighted degree in the normalisation
        # (src/layers.jl:224 vs :230, as GCNConv.__call__ does); the plan's kernels read them from the handle's slot lists
        norm = (l1.add_self_loops, g.edge_weight if l1.use_edge_weight else None, False)
        # a batch of graphs that share ONE structure (batch([g, g, ...]) or copies of g with other features): the plan is built on
        # the member and solves the trajectories one after the other inside its persistent launches
        members = getattr(g, "_members", None)
        member_plan = (members is not None and len(members) > 1 and all(m._handles is members[0]._handles for m in members)
                       and not self._no_member_plan and not l1.use_edge_weight)
        handle = members[0].handle(norm) if member_plan else g.handle(norm)
        key = (id(handle), d, l1.act, bool(with_backward), len(members) if member_plan else 1)
        pool = self._plans.get(key)
        if pool is None:
            pool = self._plans[key] = []
            while len(self._plans) > self.max_plans:        # a plan owns its tape (GBs): keep only the most recent ones
                self._plans.pop(next(iter(self._plans)))     # (a training loop that swaps the graph every minibatch)
        else:
            self._plans[key] = self._plans.pop(key)          # most recently used last
        # a plan holds ONE solve's tape: `y1 = node(u1); y2 = node(u2); (y1 + y2).backward()` needs two
        for plan in pool:
            if not (with_backward and plan.busy()):
                return plan
        if len(pool) >= self.max_outstanding:
            raise _lib.NgpdeError(_lib.ERR_STATE, f"NeuralODE: {len(pool)} solves await their backward pass on this graph; "
                                                  "each holds a tape -- run backward (or raise NeuralODE.max_outstanding)")
        try:
            plan = _Plan(handle, d, l1.act, self.solver, self.n_steps, self.dt, with_backward,
                         members=len(members) if member_plan else 1)
        except _lib.NgpdeError as e:
            if not (member_plan and e.code == _lib.ERR_UNSUPPORTED):
                raise
            self._no_member_plan = True     # not a case of the persistent plan: one handle for the whole batch instead
            self._plans.pop(key, None)
            return self.plan_for(ps, st, with_backward)
        pool.append(plan)
        return plan

    def gat_plan_for(self, ps, st, u):
        """the device-resident plan when the right-hand side is ONE GAT-style layer in the one-launch shape (64 => heads x c = 64,
        concat, tiles fit the LDS halo) and the library takes it (ngpde_node_gat_supported); None otherwise"""
        m = self.model
        if isinstance(m, Chain) and len(m.chain) == 1 and isinstance(m.chain[0], GATConv):    # Chain(GATConv(...)): the same solve
            m, ps, st = m.chain[0], ps["layer_1"], st["layer_1"]
        if not (isinstance(m, GATConv) and m.concat and u.is_cuda and m.in_chs == 64 and m.heads * m.out_chs == 64):
            return None
        g = st["graph"]
        # a batch of graphs that share ONE structure: the plan is built on the member, two members per workgroup
        members = getattr(g, "_members", None)
        member_plan = members is not None and len(members) > 1 and all(mm._handles is members[0]._handles for mm in members)
        handle = m._graph(members[0] if member_plan else g).handle()
        with_backward = torch.is_grad_enabled() and (u.requires_grad or any(
            isinstance(v, torch.Tensor) and v.requires_grad for v in ps.values()))
        key = ("gat", id(handle), m.heads, m.act, m.negative_slope, bool(with_backward), len(members) if member_plan else 1)
        pool = self._plans.get(key)
        if pool is None:
            # asked once per (graph handle, head count) of this NeuralODE: the check compares the two directions' schedules on the
            # device and synchronises (the entry keeps the handle alive, so its id cannot be recycled)
            ok = self._gat_ok.get((id(handle), m.heads))
            if ok is None:
                ok = (handle, bool(_lib.load().ngpde_node_gat_supported(handle.ptr, 64, m.heads, m.out_chs)))
                self._gat_ok[(id(handle), m.heads)] = ok
                while len(self._gat_ok) > 8:
                    self._gat_ok.pop(next(iter(self._gat_ok)))
            if not ok[1]:
                return None
            pool = self._plans[key] = []
            while len(self._plans) > self.max_plans:
                self._plans.pop(next(iter(self._plans)))
        else:
            self._plans[key] = self._plans.pop(key)
        for plan in pool:
            if not (with_backward and plan.busy()):
                return plan
        if len(pool) >= self.max_outstanding:
            raise _lib.NgpdeError(_lib.ERR_STATE, f"NeuralODE: {len(pool)} solves await their backward pass on this graph; "
                                                  "each holds a tape -- run backward (or raise NeuralODE.max_outstanding)")
        plan = _OdePlan(handle, _ode_desc(_lib.RHS_GAT, self.solver, self.n_steps, self.dt, with_backward, len(members) if member_plan else 1, width=64,
                                          act=int(m.act), heads=int(m.heads), head_width=int(m.out_chs), negative_slope=float(m.negative_slope)), "gat")
        pool.append(plan)
        return plan

    def vmh_plan_for(self, ps, st, u, needs_grad):
        """the device-resident plan when the right-hand side is VMHConv(phi, gamma) on a scalar state in the shapes the library takes
        (ngpde_node_vmh_supported: docs/src/tutorials/VMH.md:75-89's model); (plan, weights and biases) or None"""
        m = self.model
        if not (isinstance(m, VMHConv) and u.is_cuda and u.dim() == 2 and u.shape[1] == 1):
            return None
        g = st["graph"]
        if list(g.ndata) != ["x"]:      # (a batched graph is one block-diagonal graph to this plan: its members' tiles never neighbour)
            return None
        remap = _canonical_batch(g, u.device)      # the members of an earlier batch in a new order: that batch's graph and plan serve
        nodemap, g_given = None, g
        if remap is not None:
            g, nodemap = remap
        try:
            phi, gam = _dense_stack(m.ϕ, ps["ϕ"], "ϕ"), _dense_stack(m.γ, ps["γ"], "γ")
        except _lib.NgpdeError:
            return None
        pos = _node_data(g, u.device)
        pd = pos.shape[1]
        wb, dims, acts = [], [], []
        for stack in (phi, gam):
            d, a = [], []
            for layer, p in stack:
                wt, b = _wt_b(p)
                wb += [wt, b]
                d.append(wt.shape[0])
                a.append(layer.act)
            d.append(rows_of(stack[-1][1]["weight"]).shape[1])
            dims.append(d)
            acts.append(a)
        aggr = _lib.AGGR.get(m.aggr)
        if aggr is None:
            return None
        # the plan's entries take pointers only: a parameter tree that does not chain (layer l's output width against layer l + 1's
        # input width, every bias against its layer, phi's input = [h_i; h_j - h_i; x_j - x_i], gamma's = [h_i; m_i]) would make the
        # kernels read past the weight arrays -- the reference fails in the matrix product with a DimensionMismatch
        k = 0
        for name, d in (("ϕ", dims[0]), ("γ", dims[1])):      # (that the stacks chain as the layer feeds them is ngpde_ode_create's check)
            _check_plan_shapes("NeuralODE(VMHConv)", u, u.shape[0], 1,
                               [(f"{name}.layer_{l + 1}.weight", wb[k + 2 * l], (d[l], d[l + 1])) for l in range(len(d) - 1)],
                               [(f"{name}.layer_{l + 1}.bias", wb[k + 2 * l + 1], d[l + 1]) for l in range(len(d) - 1)])
            k += 2 * (len(d) - 1)
        lib = _lib.load()
        ia = lambda v: (C.c_int32 * len(v))(*[int(t) for t in v])
        supported = lambda h: lib.ngpde_node_vmh_supported(h.ptr, 1, pd, len(acts[0]), ia(dims[0]), ia(acts[0]), len(acts[1]), ia(dims[1]),
                                                            ia(acts[1]), aggr)
        members = getattr(g, "_members", None)
        straddles = bool(members) and len(members) > 1 and any(mg.num_nodes % 32 for mg in members)
        index, pool, key, handle = None, None, None, None
        if not straddles:      # (a batch whose clouds share tiles goes to its padded form at once: no handle of the unpadded union is built)
            handle = g.handle()
            key = ("vmh", id(handle), tuple(dims[0]), tuple(acts[0]), tuple(dims[1]), tuple(acts[1]), aggr, bool(needs_grad))
            pool = self._plans.get(key)
        if pool is None and (straddles or not supported(handle)):
            # a batch of point clouds whose sizes are not multiples of the 32-row tile (VMH.md:120-134: 24 clouds of 3 000 points): a tile
            # that holds the end of one cloud and the start of the next stages two neighbourhoods and can overflow its halo, which takes
            # the persistent forms away from the whole handle.  The same batch with every cloud padded to whole tiles by isolated nodes
            # (no edges: no messages, a zero cotangent on their outputs) is served; u goes in and out through an index map.
            pad = _padded_batch(g, u.device)
            if pad is None:
                return None
            g, index = pad
            handle, pos = g.handle(), _node_data(g, u.device)
            key = ("vmh", id(handle), tuple(dims[0]), tuple(acts[0]), tuple(dims[1]), tuple(acts[1]), aggr, bool(needs_grad))
            pool = self._plans.get(key)
            if pool is None and not supported(handle):
                return None
        if pool is None:
            pool = self._plans[key] = []
            while len(self._plans) > self.max_plans:
                self._plans.pop(next(iter(self._plans)))
        else:
            self._plans[key] = self._plans.pop(key)
        if nodemap is not None:      # node of the given batch -> node of the earlier batch (-> its row among the padding nodes'), kept on the batch
            comp = getattr(g_given, "_vmh_comp", None)
            if comp is None or comp[0] is not index or comp[1] is not nodemap:
                comp = g_given._vmh_comp = (index, nodemap, nodemap if index is None else index.index_select(0, nodemap))
            index = comp[2]
        for plan in pool:
            if not (needs_grad and plan.busy()):
                return plan, wb, index
        if len(pool) >= self.max_outstanding:
            raise _lib.NgpdeError(_lib.ERR_STATE, f"NeuralODE: {len(pool)} solves await their backward pass on this graph; "
                                                  "each holds a tape -- run backward (or raise NeuralODE.max_outstanding)")
        # plans of this right-hand side on OTHER graphs are of no use any more (updategraph per minibatch, VMH.md:132-134): their tapes --
        # tens of GB at the tutorial's batch size -- go back to the library's pool before the new plan asks for its own
        # (only when THIS plan needs tapes, and only plans that hold tapes: a forward-only validation solve between training steps
        # must not evict the training graph's plan and make the next step rebuild it)
        if needs_grad:
            for old_key in [k for k in self._plans if k[0] == "vmh" and k[1] != id(handle) and k[-1]]:
                self._plans.pop(old_key)
        try:
            plan = _OdePlan(handle, _ode_desc(_lib.RHS_VMH, self.solver, self.n_steps, self.dt, needs_grad, width=1, pos_width=int(pd), aggr=int(aggr),
                                              pos=pos.data_ptr(), n_phi=len(acts[0]), phi_dims=dims[0], phi_acts=acts[0], n_gamma=len(acts[1]),
                                              gamma_dims=dims[1], gamma_acts=acts[1]), "vmh")
        except _lib.NgpdeError as e:
            # the tapes of a solve -- every layer's input rows and dz rows of every right-hand-side evaluation, 64 floats wide -- did not
            # fit the device (the library parks and re-uses the tapes of plans that went away, and frees them before it gives up): the
            # generic solver, which keeps O(stages) arrays, takes the solve
            if e.code != _lib.ERR_HIP:
                raise
            if not pool:
                self._plans.pop(key, None)
            return None
        pool.append(plan)
        return plan, wb, index

    def __call__(self, x, ps, st):
        u = rows_of(x)
        needs_grad = torch.is_grad_enabled() and (u.requires_grad or any(
            isinstance(v, torch.Tensor) and v.requires_grad for v in _leaves(ps)))      # (parameter trees of any depth: VMHConv's phi / gamma chains)
        plan = self.plan_for(ps, st, needs_grad) if not self.save_every else None
        if plan is not None:
            if not u.is_cuda:
                raise _lib.ArgumentError(_lib.ERR_INVALID_ARGUMENT, "NeuralODE: inputs must live on the GPU (no CPU fallback)")
            p1, p2 = ps["layer_1"], ps["layer_2"]
            b1 = p1["bias"].reshape(-1) if "bias" in p1 else None
            b2 = p2["bias"].reshape(-1) if "bias" in p2 else None
            w1, w2 = rows_of(p1["weight"]), rows_of(p2["weight"])
            # the plan's entries take no sizes (they walk the handle's rows): what GCNConv.__call__ would have checked
            # (check_num_nodes, the matrix product's own DimensionMismatch) is checked here, before any kernel runs
            d = self.model.chain[0].in_chs
            _check_plan_shapes("NeuralODE(GCNConv, GCNConv)", u, plan.n_nodes * plan.members, d,
                               [("layer_1.weight", w1, (d, d)), ("layer_2.weight", w2, (d, d))],
                               [("layer_1.bias", b1, d), ("layer_2.bias", b2, d)])
            uT = _NodeGCN2Fn.apply(u, w1, b1, w2, b2, plan)
            return uT.T, st
        gplan = self.gat_plan_for(ps, st, u) if not self.save_every else None
        if gplan is not None:
            gps = ps["layer_1"] if isinstance(self.model, Chain) else ps
            gm = self.model.chain[0] if isinstance(self.model, Chain) else self.model
            b = gps["bias"].reshape(-1) if "bias" in gps else None
            w, a = rows_of(gps["weight"]), rows_of(gps["a"])
            _check_plan_shapes("NeuralODE(GATConv)", u, gplan.n_nodes * gplan.members, 64, [("weight", w, (64, 64))], [("bias", b, 64)])
            if a.numel() != 2 * gm.out_chs * gm.heads:
                raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, f"DimensionMismatch: NeuralODE(GATConv): attention vector has "
                                             f"{a.numel()} entries, expected 2 x {gm.out_chs} x {gm.heads}")
            uT = _NodeOdeFn.apply(u, gplan, None, a, w, b)
            return uT.T, st
        vplan = self.vmh_plan_for(ps, st, u, needs_grad)
        if vplan is not None:
            plan_v, wb, index = vplan
            # a state whose node count is not the plan's graph's (a forgotten updategraph in the minibatch loop): the reference's
            # check_num_nodes DimensionMismatch, not N floats read and written through buffers of N'
            n_expected = plan_v.n_nodes if index is None else int(index.numel())
            if u.shape[0] != n_expected:
                raise _lib.DimensionMismatch(_lib.ERR_DIMENSION_MISMATCH, f"DimensionMismatch: NeuralODE(VMHConv): the state has {u.shape[0]} "
                                             f"nodes, the graph in the layer's state has {n_expected}")
            uin = u.reshape(-1)
            if index is not None:      # (a padded batch: the real nodes' rows among the isolated padding nodes')
                uin = _rows_index(uin, index, plan_v.n_nodes, True)
            if self.save_every:      # saveat: the (1 x N x T) array of the solution at t0 (+ j saveat)
                us = _NodeOdeFn.apply(uin, plan_v, (self.save_every, self.save_start), None, *wb)
                if index is not None:
                    us = _rows_index(us, index, plan_v.n_nodes, False)
                return us.T.unsqueeze(0), st
            uT = _NodeOdeFn.apply(uin, plan_v, None, None, *wb)
            if index is not None:
                uT = _rows_index(uT, index, plan_v.n_nodes, False)
            return uT.reshape(u.shape).T, st
        # any other right-hand side: explicit RK stepping through the layers' own kernels, every Runge-Kutta combination (and
        # every combination of the discrete adjoint) one library launch
        if not u.is_cuda:
            raise _lib.ArgumentError(_lib.ERR_INVALID_ARGUMENT, "NeuralODE: inputs must live on the GPU (no CPU fallback)")
        leaves = _leaves(ps)
        if self.capture:
            needs = torch.is_grad_enabled() and (u.requires_grad or any(isinstance(p, torch.Tensor) and p.requires_grad for p in leaves))
            # every GNNGraph leaf of the state tree (a container's graphs sit in st["layer_k"]["graph"]): the captures bake in
            # the addresses of their derived arrays.  The entry keeps `st` -- and with it the graphs -- alive, so an id cannot be
            # recycled while its key is cached.
            key = (tuple(u.shape), needs, tuple(id(g) for g in _graph_leaves(st)),
                   tuple(p.data_ptr() if isinstance(p, torch.Tensor) else id(p) for p in leaves))
            solve = self._captured.get(key)
            if solve is None:
                solve = self._captured[key] = _CapturedSolve(self, u, ps, st, leaves, needs)
                while len(self._captured) > self.max_plans:
                    self._captured.pop(next(iter(self._captured)))
            uT = _NodeCapturedFn.apply(u, solve, *leaves)
            return (uT.permute(2, 1, 0) if self.save_every else uT.T), st
        uT = _NodeGenericFn.apply(u, self, ps, st, *leaves)
        return (uT.permute(2, 1, 0) if self.save_every else uT.T), st      # saveat: (D x N x T), the reference's array of the solution
