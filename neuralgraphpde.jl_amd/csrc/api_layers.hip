// api_layers.hip -- layer-level C-ABI entries (declared in include/ngpde.h): ONE call evaluates an edge-function layer of the
// reference, one call its pullback.
//   ExplicitEdgeConv   /root/reference/src/layers.jl:94-112     VMHConv   :308-332     MPPDEConv   :390-422
// What the host side of rounds 1 - 4 orchestrated from Python (layers_mp.py / functional.py: the signed row blocks of phi's first
// weight, the node-level terms P / Q / E, the choice between the fused message launch and the primitives, the node update as a
// chain, twenty autograd nodes and their saved tensors) is done here, over the library's own primitives, with every temporary
// and every saved activation at a fixed offset of ONE caller-provided workspace.  A Julia host binds three functions per layer
// family (size query, forward, pullback) and writes one rrule.
//
// The arithmetic is exactly the primitives': this file only sequences the C entries of api_mp.hip / row_blocks.hip / optim.hip,
// so results are bit for bit those of the composed path (tests/test_layer_abi_gpu.py).
#include <algorithm>
#include <cstdlib>
#include <cstring>

#include <mutex>
#include <vector>

#include "common.h"

using namespace ngpde;

namespace {

constexpr int kMaxL = NGPDE_MLP_MAX_LAYERS;
constexpr int kMaxBlocks = 4;   // input blocks of one Dense (ngpde_dense_forward)

bool env_is(const char *name, char v) {
  const char *e = std::getenv(name);
  return e && e[0] == v;
}

// Bump allocator over the workspace in units of floats, 256-byte granules; with base == nullptr it only measures.
struct Arena {
  float *base = nullptr;
  size_t off = 0;
  float *take(size_t floats) {
    float *p = base ? base + off : nullptr;
    off += (floats + 63) & ~(size_t)63;
    return p;
  }
  void *take_bytes(size_t bytes) { return take((bytes + 3) / 4); }
};

constexpr size_t kArenaSlack = 256;   // the arena starts at the first 256-byte boundary inside the caller's buffer
inline float *arena_base(void *workspace) {
  return reinterpret_cast<float *>((reinterpret_cast<uintptr_t>(workspace) + (kArenaSlack - 1)) & ~(uintptr_t)(kArenaSlack - 1));
}

struct Blocks {   // a virtual vcat: the block table of one Dense call
  int n = 0;
  const float *ptr[kMaxBlocks] = {nullptr, nullptr, nullptr, nullptr};
  int32_t width[kMaxBlocks] = {0, 0, 0, 0};
  int32_t row_div[kMaxBlocks] = {1, 1, 1, 1};
  int state_of[kMaxBlocks] = {-1, -1, -1, -1};   // index of the state block this one is (gradient wanted), -1: a constant
  int total() const {
    int t = 0;
    for (int i = 0; i < n; ++i) t += width[i];
    return t;
  }
  bool add(const float *p, int w, int rd, int st) {
    if (w <= 0) return true;
    if (n == kMaxBlocks) return false;
    ptr[n] = p; width[n] = w; row_div[n] = rd; state_of[n] = st;
    ++n;
    return true;
  }
};

struct RowSpec {   // ngpde_row_blocks_gather / _scatter tables of phi's first weight
  int n_seg = 0, n_out = 0;
  int32_t out_index[16], dst0[16], src0[16], nrows[16], out_rows[4] = {0, 0, 0, 0};
  float sign[16];
  void block(int out, int rows, std::initializer_list<std::pair<int, float>> terms) {
    if (rows <= 0) return;
    for (auto &t : terms) {
      out_index[n_seg] = out; dst0[n_seg] = out_rows[out]; src0[n_seg] = t.first; nrows[n_seg] = rows; sign[n_seg] = t.second;
      ++n_seg;
    }
    out_rows[out] += rows;
  }
};

// Everything a forward / backward pair of one layer call needs to agree on: shapes, the path taken, and where each array lives.
// What the training forward decided, remembered per workspace so that the pullback can refuse a workspace it would misread: the plan
// is re-derived by every call from the descriptor, the graph and the environment switches (NGPDE_NO_FUSED_EDGE, NGPDE_DEEP_EDGE_BWD,
// NGPDE_GNO_MATERIALIZE, NGPDE_NO_GNO_GFORM ...), and the pullback reads saved activations at offsets of ITS plan -- a switch flipped or
// a descriptor changed between the two calls used to give wrong gradients without an error.  Host memory only (a few words per live
// workspace, bounded), nothing on the stream: the calls stay capture-safe.  A workspace the table does not know is trusted as before.
struct PlanStamp { const void *ws; uint64_t key; };
inline std::mutex &stamp_mu() { static std::mutex m; return m; }
inline std::vector<PlanStamp> &stamps() { static std::vector<PlanStamp> v; return v; }
inline void stamp_set(const void *ws, uint64_t key) {
  std::lock_guard<std::mutex> lk(stamp_mu());
  auto &v = stamps();
  for (auto &e : v)
    if (e.ws == ws) { e.key = key; return; }
  if (v.size() >= 256) v.erase(v.begin());   // (oldest first: a long-running host that never runs the pullback of old workspaces)
  v.push_back({ws, key});
}
inline bool stamp_agrees(const void *ws, uint64_t key) {
  std::lock_guard<std::mutex> lk(stamp_mu());
  for (auto &e : stamps())
    if (e.ws == ws) return e.key == key;
  return true;
}
inline uint64_t mix(uint64_t h, uint64_t v) { return (h ^ v) * 0x9E3779B97F4A7C15ull + (h >> 29); }

struct Plan {
  int64_t N = 0, E = 0;
  int G = 1, kind = 0, aggr = 1;
  int dh = 0, dd = 0, de = 0, dth = 0, h1 = 0, n_tail = 0, mw = 0, n_upd = 0, out_w = 0;
  int act1 = 0;
  Blocks A, B, U;          // target side, source side of phi's first layer; the node update's vcat
  int u_m = -1;            // index of the message block in U
  RowSpec rows;
  int w1_rows = 0;
  bool fused_msg = false, fused_bwd = false, need_dE = false, chain2 = false, chain2_fused = false, pair_shared = false;
  int32_t tail_dout[kMaxL], tail_act[kMaxL];
  const float *tail_w[kMaxL], *tail_b[kMaxL];
  // ---- forward region (kept until the pullback when training)
  float *wA = nullptr, *wB = nullptr, *wD = nullptr, *P = nullptr, *Q = nullptr, *Et = nullptr, *m = nullptr;
  float *save[kMaxL + 1];  // fused forward, primitives' pullback: pre-activations of every layer of phi, [E][w_l]
  float *z0 = nullptr, *a0 = nullptr, *ty[kMaxL], *tz[kMaxL];   // primitives: first-layer pre-activation / activation, tail outputs / pre-activations
  float *ua[kMaxL], *uz[kMaxL];                                  // update: layer outputs (the last one is y) and pre-activations
  size_t fwd_floats = 0;
  // ---- pullback scratch
  float *dm = nullptr, *dP = nullptr, *dQ = nullptr, *dE = nullptr, *dwA = nullptr, *dwB = nullptr, *dwD = nullptr;
  float *eg[2] = {nullptr, nullptr};   // [E][max width] ping-pong of the primitives' pullback
  float *ea = nullptr;                 // [E][max width]: an activation re-materialised from a saved pre-activation
  float *ng[2] = {nullptr, nullptr};   // [N][max width] ping-pong of the update's pullback
  float *dU[kMaxBlocks] = {nullptr, nullptr, nullptr, nullptr}, *dA[kMaxBlocks] = {nullptr, nullptr, nullptr, nullptr},
        *dB[kMaxBlocks] = {nullptr, nullptr, nullptr, nullptr};
  void *ws = nullptr;
  size_t ws_bytes = 0;
  size_t total_floats = 0;
};

int32_t check_mlp(const char *what, const ngpde_mlp_t &m, int min_layers) {
  NGPDE_REQUIRE(m.n_layers >= min_layers && m.n_layers <= kMaxL, NGPDE_ERR_UNSUPPORTED, "ngpde_edge_layer: %s has %d Dense layers (%d..%d supported)",
                what, m.n_layers, min_layers, kMaxL);
  for (int l = 0; l < m.n_layers; ++l) {
    NGPDE_REQUIRE(m.dims[l] > 0 && m.dims[l + 1] > 0, NGPDE_ERR_DIMENSION_MISMATCH, "DimensionMismatch: %s.layer_%d is (%d x %d)", what, l + 1,
                  m.dims[l + 1], m.dims[l]);
    NGPDE_REQUIRE(m.weight[l] != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_layer: %s.layer_%d.weight is NULL", what, l + 1);
    NGPDE_REQUIRE(m.act[l] >= NGPDE_ACT_IDENTITY && m.act[l] <= NGPDE_ACT_SOFTPLUS, NGPDE_ERR_INVALID_ARGUMENT,
                  "ngpde_edge_layer: %s.layer_%d: unknown activation code %d", what, l + 1, m.act[l]);
  }
  return NGPDE_OK;
}

// Fills the plan for (graph, descriptor, training).  a.base == nullptr: sizes only.
int32_t make_plan(const ngpde_graph *g, const ngpde_edge_layer_t &L, bool training, Arena &a, Plan &p) {
  NGPDE_REQUIRE(L.kind >= NGPDE_LAYER_EDGECONV && L.kind <= NGPDE_LAYER_MPPDE, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_layer: unknown kind %d", L.kind);
  NGPDE_REQUIRE(L.aggr >= NGPDE_AGGR_SUM && L.aggr <= NGPDE_AGGR_MUL, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_layer: unknown aggregation %d", L.aggr);
  NGPDE_REQUIRE(L.n_state >= 1 && L.n_state <= 4, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_layer: 1..4 state blocks, got %d", L.n_state);
  int32_t st;
  if ((st = check_mlp("phi", L.phi, 1))) return st;
  const bool has_update = L.kind != NGPDE_LAYER_EDGECONV;
  if ((st = check_mlp(L.kind == NGPDE_LAYER_MPPDE ? "psi" : "gamma", L.update, has_update ? 1 : 0))) return st;
  NGPDE_REQUIRE(has_update || L.update.n_layers == 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_layer: ExplicitEdgeConv has no node update");
  p.N = g->n_nodes; p.E = g->n_edges; p.G = std::max(g->n_graphs, 1);
  p.kind = L.kind; p.aggr = L.aggr;
  for (int k = 0; k < L.n_state; ++k) {
    NGPDE_REQUIRE(L.state_width[k] > 0 && L.state[k] != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_layer: state block %d is empty or NULL", k);
  }
  NGPDE_REQUIRE(L.node_feat_width >= 0 && L.pos_width >= 0 && L.edge_feat_width >= 0 && L.theta_width >= 0, NGPDE_ERR_DIMENSION_MISMATCH,
                "ngpde_edge_layer: negative feature width");
  NGPDE_REQUIRE((L.node_feat_width == 0 || L.node_feat) && (L.pos_width == 0 || L.pos) && (L.edge_feat_width == 0 || L.edge_feat) &&
                    (L.theta_width == 0 || L.theta), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_layer: a feature block with a width is NULL");
  const ngpde_mlp_t &phi = L.phi;
  p.h1 = phi.dims[1]; p.act1 = phi.act[0];
  p.n_tail = phi.n_layers - 1;
  p.mw = phi.dims[phi.n_layers];
  for (int l = 0; l < p.n_tail; ++l) {
    p.tail_dout[l] = phi.dims[l + 2]; p.tail_act[l] = phi.act[l + 1]; p.tail_w[l] = phi.weight[l + 1]; p.tail_b[l] = phi.bias[l + 1];
  }
  bool ok = true;
  if (L.kind == NGPDE_LAYER_MPPDE) {
    NGPDE_REQUIRE(L.n_state == 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_layer: MPPDEConv takes one state block (h)");
    p.dh = L.state_width[0]; p.dd = L.node_feat_width; p.de = L.edge_feat_width; p.dth = L.theta_width;
    NGPDE_REQUIRE(!(p.dth && (p.N % p.G || p.E % p.G)), NGPDE_ERR_DIMENSION_MISMATCH,
                  "DimensionMismatch: batched graphs must have the same structure (src/layers.jl:359-361)");
    const int rd = (int)(p.N / p.G);
    // [hi; hj; di - dj; e; theta]  (:409-410): target side [wa; wc; we] on [h, d, theta], source side [wb; -wc] on [h, d], edge features wd
    const int oa = 0, ob = p.dh, oc = 2 * p.dh, od = oc + p.dd, oe = od + p.de;
    p.w1_rows = oe + p.dth;
    ok = ok && p.A.add(L.state[0], p.dh, 1, 0) && p.A.add(L.node_feat, p.dd, 1, -1) && p.A.add(L.theta, p.dth, rd, -1);
    ok = ok && p.B.add(L.state[0], p.dh, 1, 0) && p.B.add(L.node_feat, p.dd, 1, -1);
    p.rows.block(0, p.dh, {{oa, 1.f}}); p.rows.block(0, p.dd, {{oc, 1.f}}); p.rows.block(0, p.dth, {{oe, 1.f}});
    p.rows.block(1, p.dh, {{ob, 1.f}}); p.rows.block(1, p.dd, {{oc, -1.f}});
    p.rows.n_out = 2;
    if (p.de) { p.rows.block(2, p.de, {{od, 1.f}}); p.rows.n_out = 3; }
    // psi([h; m; theta])  (:418)
    ok = ok && p.U.add(L.state[0], p.dh, 1, 0);
    p.u_m = p.U.n;
    ok = ok && p.U.add(nullptr, p.mw, 1, -1) && p.U.add(L.theta, p.dth, rd, -1);
  } else {
    NGPDE_REQUIRE(L.edge_feat_width == 0 && L.theta_width == 0, NGPDE_ERR_INVALID_ARGUMENT,
                  "ngpde_edge_layer: edge features / theta belong to MPPDEConv");
    for (int k = 0; k < L.n_state; ++k) p.dh += L.state_width[k];
    p.dh += L.node_feat_width;
    p.dd = L.pos_width;
    // EdgeConv [hi...; hj...; xj - xi] (:106): target [wa; -wc], source [wb; wc].  VMH [hi...; (hj - hi)...; xj - xi] (:316): target
    // [wa - wb; -wc], source [wb; wc].  Both sides read [state..., node_feat, pos].
    const int oa = 0, ob = p.dh, oc = 2 * p.dh;
    p.w1_rows = oc + p.dd;
    for (int side = 0; side < 2; ++side) {
      Blocks &S = side ? p.B : p.A;
      for (int k = 0; k < L.n_state; ++k) ok = ok && S.add(L.state[k], L.state_width[k], 1, k);
      ok = ok && S.add(L.node_feat, L.node_feat_width, 1, -1) && S.add(L.pos, p.dd, 1, -1);
    }
    if (L.kind == NGPDE_LAYER_VMH) p.rows.block(0, p.dh, {{oa, 1.f}, {ob, -1.f}});
    else p.rows.block(0, p.dh, {{oa, 1.f}});
    p.rows.block(0, p.dd, {{oc, -1.f}});
    p.rows.block(1, p.dh, {{ob, 1.f}}); p.rows.block(1, p.dd, {{oc, 1.f}});
    p.rows.n_out = 2;
    if (L.kind == NGPDE_LAYER_VMH) {   // gamma(vcat(values(x)..., m))  (:328)
      for (int k = 0; k < L.n_state; ++k) ok = ok && p.U.add(L.state[k], L.state_width[k], 1, k);
      p.u_m = p.U.n;
      ok = ok && p.U.add(nullptr, p.mw, 1, -1);
    }
  }
  NGPDE_REQUIRE(ok, NGPDE_ERR_UNSUPPORTED, "ngpde_edge_layer: more than %d blocks in one vcat", kMaxBlocks);
  NGPDE_REQUIRE(phi.dims[0] == p.w1_rows, NGPDE_ERR_DIMENSION_MISMATCH,
                "DimensionMismatch: first layer expects %d input features, the message has %d", phi.dims[0], p.w1_rows);
  p.n_upd = L.update.n_layers;
  if (p.n_upd) {
    NGPDE_REQUIRE(L.update.dims[0] == p.U.total(), NGPDE_ERR_DIMENSION_MISMATCH,
                  "DimensionMismatch: the node update expects %d input features, its vcat has %d", L.update.dims[0], p.U.total());
    p.out_w = L.update.dims[p.n_upd];
  } else {
    p.out_w = p.mw;
  }
  const size_t N = (size_t)p.N, E = (size_t)p.E;

  // ---- message path: one fused launch where the message MLP fits the fused kernel (widths <= 64, multiples of 4, <= 3 further
  // layers, tiles within the LDS halo; max / min only without gradients), the primitives otherwise
  // (max / min / *: with gradients only where the one-launch pullback takes it -- the primitives' pullbacks of those need the per-edge
  // messages, which the fused forward does not keep)
  const bool bwd_ok = !env_is("NGPDE_NO_FUSED_EDGE_BWD", '1') &&
                      ngpde_edge_mlp_backward_supported(g, p.h1, p.n_tail, p.n_tail ? p.tail_dout : nullptr, p.aggr) == 1;
  p.fused_msg = !env_is("NGPDE_NO_FUSED_EDGE", '1') && p.E > 0 && p.n_tail <= 3 &&
                (p.aggr == NGPDE_AGGR_SUM || p.aggr == NGPDE_AGGR_MEAN || !training || bwd_ok) &&
                ngpde_edge_mlp_supported(g, p.h1, p.n_tail, p.n_tail ? p.tail_dout : nullptr) == 1;
  if (p.fused_msg && training) {
    p.fused_bwd = bwd_ok;
    if (p.fused_bwd && p.n_tail >= 2) {
      // three / four-layer message MLPs: the one-launch pullback pays from ~32 k nodes up (one 4-wave workgroup per CU walks a long
      // chain per tile); NGPDE_DEEP_EDGE_BWD=1 / 0 forces it on / off
      p.fused_bwd = env_is("NGPDE_DEEP_EDGE_BWD", '1') || (!env_is("NGPDE_DEEP_EDGE_BWD", '0') && p.N >= 32768);
    }
  }

  // ---- forward region
  p.wA = a.take((size_t)p.rows.out_rows[0] * p.h1);
  p.wB = a.take((size_t)p.rows.out_rows[1] * p.h1);
  p.wD = p.de ? a.take((size_t)p.de * p.h1) : nullptr;
  p.P = a.take(N * p.h1);
  p.Q = a.take(N * p.h1);
  p.Et = p.de ? a.take(E * p.h1) : nullptr;
  for (int l = 0; l <= kMaxL; ++l) p.save[l] = nullptr;
  for (int l = 0; l < kMaxL; ++l) p.ty[l] = p.tz[l] = p.ua[l] = p.uz[l] = nullptr;
  if (p.fused_msg) {
    if (training && !p.fused_bwd) {
      p.save[0] = a.take(E * p.h1);
      for (int l = 0; l < p.n_tail; ++l) p.save[l + 1] = a.take(E * p.tail_dout[l]);
    }
  } else {
    p.a0 = a.take(E * p.h1);
    p.z0 = (training && p.act1 != 0) ? a.take(E * p.h1) : nullptr;
    for (int l = 0; l < p.n_tail; ++l) {
      p.ty[l] = a.take(E * p.tail_dout[l]);
      p.tz[l] = (training && p.tail_act[l] != 0) ? a.take(E * p.tail_dout[l]) : nullptr;
    }
  }
  // (ExplicitEdgeConv: the aggregate IS y -- kept here too where the pullback of max / min / * needs it)
  p.m = (p.n_upd || (training && !p.fused_msg && p.aggr >= NGPDE_AGGR_MAX)) ? a.take(N * p.mw) : nullptr;
  if (p.n_upd) {
    const ngpde_mlp_t &u = L.update;
    p.U.ptr[p.u_m] = p.m ? p.m : reinterpret_cast<const float *>((uintptr_t)256);   // (measuring pass: any aligned non-NULL address)
    p.chain2 = p.n_upd >= 2;
    if (p.chain2) {
      p.chain2_fused = ngpde_dense_chain2_fused(p.N, p.U.n, p.U.ptr, p.U.width, p.U.row_div, u.dims[1], u.dims[2]) == 1;
      p.ua[0] = (training || !p.chain2_fused) ? a.take(N * u.dims[1]) : nullptr;
    }
    for (int l = 0; l < p.n_upd; ++l) {
      if (l + 1 < p.n_upd && !(p.chain2 && l == 0)) p.ua[l] = a.take(N * u.dims[l + 1]);
      p.uz[l] = (training && u.act[l] != 0) ? a.take(N * u.dims[l + 1]) : nullptr;
    }
  }
  p.fwd_floats = a.off;
  if (!training) {
    p.total_floats = a.off;
    return NGPDE_OK;
  }

  // ---- pullback scratch
  int emax = p.h1, nmax = p.mw;
  for (int l = 0; l < p.n_tail; ++l) emax = std::max(emax, (int)p.tail_dout[l]);
  for (int l = 0; l < p.n_upd; ++l) nmax = std::max(nmax, (int)L.update.dims[l + 1]);
  size_t wsb = 0;
  auto dense_ws = [&](int64_t n, int din, int dout) { wsb = std::max(wsb, ngpde_dense_workspace_bytes(n, din, dout)); };
  p.dm = p.n_upd ? a.take(N * p.mw) : nullptr;
  if (p.n_upd) {
    p.ng[0] = a.take(N * nmax);
    p.ng[1] = a.take(N * nmax);
    for (int i = 0; i < p.U.n; ++i)
      if (p.U.state_of[i] >= 0) p.dU[i] = a.take(N * p.U.width[i]);
    dense_ws(p.N, p.U.total(), L.update.dims[1]);
    for (int l = 1; l < p.n_upd; ++l) dense_ws(p.N, L.update.dims[l], L.update.dims[l + 1]);
  }
  p.dP = a.take(N * p.h1);
  p.dQ = a.take(N * p.h1);
  if (p.fused_bwd) {
    p.need_dE = p.de > 0 || ngpde_edge_mlp_backward_needs_edge_buffer(g, p.h1, p.act1, p.de > 0, p.n_tail, p.n_tail ? p.tail_dout : nullptr,
                                                                       p.n_tail ? p.tail_act : nullptr, p.aggr) == 1;
    wsb = std::max(wsb, ngpde_edge_mlp_backward_workspace_bytes(g, p.h1, p.n_tail, p.n_tail ? p.tail_dout : nullptr));
    p.dE = p.need_dE ? a.take(E * p.h1) : nullptr;
  } else {
    p.eg[0] = a.take(E * emax);
    p.eg[1] = a.take(E * emax);
    p.ea = p.fused_msg ? a.take(E * emax) : nullptr;
    p.dE = a.take(E * p.h1);
    int prev = p.h1;
    for (int l = 0; l < p.n_tail; ++l) { dense_ws(p.E, prev, p.tail_dout[l]); prev = p.tail_dout[l]; }
  }
  p.dwA = a.take((size_t)p.rows.out_rows[0] * p.h1);
  p.dwB = a.take((size_t)p.rows.out_rows[1] * p.h1);
  p.dwD = p.de ? a.take((size_t)p.de * p.h1) : nullptr;
  if (p.de) dense_ws(p.E, p.de, p.h1);
  // the pair's pullback in one launch: both halves 64 wide, the shared leading block the only one that wants a gradient
  size_t pair_ws = 0;
  int wanted_other = 0;
  for (int i = 1; i < p.A.n; ++i) wanted_other += p.A.state_of[i] >= 0;
  if (p.h1 == 64 && p.A.ptr[0] == p.B.ptr[0] && wanted_other == 0)
    pair_ws = ngpde_dense_pair_backward_workspace_bytes(p.N, p.A.n, p.A.ptr, p.A.width, p.A.row_div, p.B.n, p.B.ptr, p.B.width, p.B.row_div, 64);
  p.pair_shared = pair_ws > 0;
  wsb = std::max(wsb, pair_ws);
  for (int i = 0; i < p.A.n; ++i)
    if (p.A.state_of[i] >= 0) {
      p.dA[i] = a.take(N * p.A.width[i]);
      if (!p.pair_shared) p.dB[i] = a.take(N * p.B.width[i]);
    }
  dense_ws(p.N, p.A.total(), p.h1);
  dense_ws(p.N, p.B.total(), p.h1);
  p.ws_bytes = wsb;
  p.ws = a.take_bytes(wsb);
  p.total_floats = a.off;
  return NGPDE_OK;
}

int32_t check_common(const char *fn, const ngpde_graph *g, const ngpde_edge_layer_t *L) {
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "%s: graph is NULL", fn);
  NGPDE_REQUIRE(L != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "%s: layer descriptor is NULL", fn);
  return NGPDE_OK;
}


// ---- GNOConv -------------------------------------------------------------------------------------------------------------------
// inv[i] = 1 / max(in-degree(i), 1): a mean aggregation's pullback multiplies a node's gradient by it once per node
__global__ void inv_in_degree_kernel(int n, const int32_t *__restrict__ rowptr, float *__restrict__ inv) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) inv[i] = 1.0f / fmaxf((float)(rowptr[i + 1] - rowptr[i]), 1.0f);
}

struct GnoPlan {
  int64_t N = 0, E = 0;
  int cin = 0, cout = 0, ds = 0, de = 0, h1 = 0, L = 0, kdim = 0, aggr = 1, act = 0, act1 = 0;
  bool reassoc = false, fused_msg = false, fused_agg = false, has_b2 = false;
  bool gform = false;      // aggregate-then-transform (gno_gform.hip): no [E][out] message, no T in the forward
  int nsplit = 1;
  int n_mid = 0;           // Dense layers of phi evaluated on [E] rows by the primitives (reassociated: 1 .. L - 2; literal: 1 .. L - 1)
  RowSpec rows;
  // forward region
  float *wa = nullptr, *wb = nullptr, *wd = nullptr, *wr = nullptr, *P = nullptr, *Q = nullptr, *Et = nullptr, *Bh = nullptr, *Wh = nullptr,
        *T = nullptr, *a = nullptr, *m = nullptr, *agg = nullptr, *z0 = nullptr, *a0 = nullptr, *zt = nullptr, *G = nullptr, *hs = nullptr,
        *slabs = nullptr;
  float *ty[kMaxL], *tz[kMaxL];
  // pullback scratch
  float *dz = nullptr, *dagg_s = nullptr, *inv = nullptr, *dT = nullptr, *dBh = nullptr, *dzE = nullptr, *dP = nullptr, *dQ = nullptr,
        *dm = nullptr, *eg[2] = {nullptr, nullptr}, *dwa = nullptr, *dwb = nullptr, *dwd = nullptr, *dwr = nullptr, *dxB = nullptr,
        *dxW = nullptr, *dxT = nullptr, *dxC = nullptr, *dsum = nullptr;
  void *ws = nullptr;
  size_t ws_bytes = 0, total_floats = 0;
};

int32_t make_gno_plan(const ngpde_graph *g, const ngpde_gno_layer_t &L, bool training, Arena &a, GnoPlan &p) {
  NGPDE_REQUIRE(L.in_chs > 0 && L.out_chs > 0, NGPDE_ERR_DIMENSION_MISMATCH, "ngpde_gno_layer: in_chs / out_chs must be positive");
  NGPDE_REQUIRE(L.aggr >= NGPDE_AGGR_SUM && L.aggr <= NGPDE_AGGR_MUL, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_layer: unknown aggregation %d", L.aggr);
  NGPDE_REQUIRE(L.act >= NGPDE_ACT_IDENTITY && L.act <= NGPDE_ACT_SOFTPLUS, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_layer: unknown activation %d", L.act);
  NGPDE_REQUIRE(L.h != nullptr && L.weight != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_layer: h or weight is NULL");
  NGPDE_REQUIRE(L.node_feat_width >= 0 && L.edge_feat_width >= 0 && (L.node_feat_width == 0 || L.node_feat) && (L.edge_feat_width == 0 || L.edge_feat),
                NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_layer: a feature block with a width is NULL");
  int32_t st;
  if ((st = check_mlp("phi", L.phi, 1))) return st;
  const ngpde_mlp_t &phi = L.phi;
  p.N = g->n_nodes; p.E = g->n_edges;
  p.cin = L.in_chs; p.cout = L.out_chs; p.ds = L.node_feat_width; p.de = L.edge_feat_width; p.aggr = L.aggr; p.act = L.act;
  p.L = phi.n_layers; p.h1 = phi.dims[1]; p.act1 = phi.act[0];
  NGPDE_REQUIRE(phi.dims[0] == 2 * p.ds + p.de && phi.dims[0] > 0, NGPDE_ERR_DIMENSION_MISMATCH,
                "DimensionMismatch: first layer expects %d input features, the message has %d", phi.dims[0], 2 * p.ds + p.de);
  NGPDE_REQUIRE(phi.dims[p.L] == p.cin * p.cout, NGPDE_ERR_DIMENSION_MISMATCH,
                "DimensionMismatch: phi must output in_chs*out_chs = %d rows, got %d", p.cin * p.cout, phi.dims[p.L]);
  // [si; sj; e]  (:523)
  int o = 0;
  if (p.ds) { p.rows.block(o++, p.ds, {{0, 1.f}}); p.rows.block(o++, p.ds, {{p.ds, 1.f}}); }
  if (p.de) p.rows.block(o++, p.de, {{2 * p.ds, 1.f}});
  p.rows.n_out = o;
  p.kdim = phi.dims[p.L - 1];
  p.reassoc = p.L >= 2 && phi.act[p.L - 1] == NGPDE_ACT_IDENTITY && !env_is("NGPDE_GNO_MATERIALIZE", '1') &&
              ngpde_gno_apply_supported(p.cout, p.kdim) == 1;
  p.has_b2 = p.reassoc && phi.bias[p.L - 1] != nullptr;
  p.fused_msg = p.reassoc && p.L == 2 && p.E > 0 && !env_is("NGPDE_NO_GNO_MFMA", '1') && (p.act1 == NGPDE_ACT_IDENTITY || p.act1 == NGPDE_ACT_RELU) &&
                ngpde_gno_message_supported(p.cout, p.kdim) == 1;
  p.fused_agg = p.fused_msg && (p.aggr == NGPDE_AGGR_SUM || p.aggr == NGPDE_AGGR_MEAN);
  // the edge index contracted first, per target: the forward needs neither T nor the [E][out] message (the pullback keeps the
  // by-source form and makes its own T)
  p.gform = p.fused_agg && ngpde_gno_gform_preferred(p.N, p.E, p.cin, p.kdim, p.cout, training ? 1 : 0) == 1;
  p.n_mid = p.fused_msg ? 0 : (p.reassoc ? p.L - 2 : p.L - 1);
  const size_t N = (size_t)p.N, E = (size_t)p.E;
  const size_t msg_w = (size_t)p.cout;
  for (int l = 0; l < kMaxL; ++l) p.ty[l] = p.tz[l] = nullptr;
  // ---- forward region
  if (p.ds) { p.wa = a.take((size_t)p.ds * p.h1); p.wb = a.take((size_t)p.ds * p.h1); }
  if (p.de) p.wd = a.take((size_t)p.de * p.h1);
  if (p.ds) { p.P = a.take(N * p.h1); p.Q = a.take(N * p.h1); }
  if (p.de) p.Et = a.take(E * p.h1);
  if (p.reassoc) {
    if (!p.gform || training) {
      p.wr = a.take((size_t)p.cin * p.cout * p.kdim);
      p.T = a.take(N * p.cout * p.kdim);
    }
    if (p.has_b2 && !p.gform) p.Bh = a.take(N * p.cout);
  }
  if (p.gform) {
    p.G = a.take(N * p.cin * p.kdim);
    if (p.has_b2) p.hs = a.take(N * p.cin);
    p.nsplit = ngpde_gno_gform_splits(p.N, p.cin, p.kdim, p.cout);
    p.slabs = a.take((size_t)(p.nsplit + 2) * N * p.cout);   // + the b2 term and W h, formed in the same launch
  } else {
    p.Wh = a.take(N * p.cout);
  }
  if (p.fused_msg) {
    p.a = training ? a.take(E * p.kdim) : nullptr;
  } else {
    p.a0 = a.take(E * p.h1);
    p.z0 = (training && p.act1 != 0) ? a.take(E * p.h1) : nullptr;
    for (int l = 0; l < p.n_mid; ++l) {
      p.ty[l] = a.take(E * phi.dims[l + 2]);
      p.tz[l] = (training && phi.act[l + 1] != 0) ? a.take(E * phi.dims[l + 2]) : nullptr;
    }
  }
  if (!p.gform) {
    p.m = a.take(E * msg_w);
    p.agg = a.take(N * msg_w);
  }
  p.zt = (training && p.act != 0) ? a.take(N * msg_w) : nullptr;
  if (!training) {
    p.total_floats = a.off;
    return NGPDE_OK;
  }
  // ---- pullback scratch
  size_t wsb = ngpde_bias_act_workspace_bytes(p.cout);
  auto dense_ws = [&](int64_t n, int din, int dout) { wsb = std::max(wsb, ngpde_dense_workspace_bytes(n, din, dout)); };
  p.dz = a.take(N * msg_w);
  if (p.fused_agg && p.aggr == NGPDE_AGGR_MEAN) { p.dagg_s = a.take(N * msg_w); p.inv = a.take(N); }
  if (p.reassoc) {
    p.dT = a.take(N * p.cout * p.kdim);
    p.dwr = a.take((size_t)p.cin * p.cout * p.kdim);
    p.dxT = a.take(N * p.cin);
    if (p.has_b2) { p.dBh = a.take(N * p.cout); p.dxB = a.take(N * p.cin); p.dsum = a.take(N * p.cin); }
    dense_ws(p.N, p.cin, p.cout * p.kdim);
  } else {
    p.dxC = a.take(N * p.cin);
    wsb = std::max(wsb, (size_t)p.E * p.cin * sizeof(float));
  }
  p.dxW = a.take(N * p.cin);
  dense_ws(p.N, p.cin, p.cout);
  p.dzE = a.take(E * p.h1);
  if (p.ds) { p.dP = a.take(N * p.h1); p.dQ = a.take(N * p.h1); p.dwa = a.take((size_t)p.ds * p.h1); p.dwb = a.take((size_t)p.ds * p.h1); dense_ws(p.N, p.ds, p.h1); }
  if (p.de) { p.dwd = a.take((size_t)p.de * p.h1); dense_ws(p.E, p.de, p.h1); }
  if (!p.fused_agg) p.dm = a.take(E * msg_w);
  if (!p.fused_msg) {
    int emax = p.h1;
    for (int l = 0; l <= p.n_mid; ++l) emax = std::max(emax, (int)phi.dims[l + 1]);
    p.eg[0] = a.take(E * emax);
    p.eg[1] = a.take(E * emax);
    for (int l = 0; l < p.n_mid; ++l) dense_ws(p.E, phi.dims[l + 1], phi.dims[l + 2]);
  } else if (!p.fused_agg) {
    p.eg[0] = a.take(E * p.kdim);
  }
  p.ws_bytes = wsb;
  p.ws = a.take_bytes(wsb);
  p.total_floats = a.off;
  return NGPDE_OK;
}

uint64_t edge_plan_key(const ngpde_graph *g, const Plan &p) {
  uint64_t h = mix(0x6e677064u, (uint64_t)(uintptr_t)g);
  h = mix(h, (uint64_t)p.N); h = mix(h, (uint64_t)p.E); h = mix(h, (uint64_t)p.kind * 16 + p.aggr);
  h = mix(h, ((uint64_t)p.h1 << 32) | (uint64_t)(p.n_tail * 64 + p.n_upd));
  h = mix(h, (uint64_t)p.fused_msg | (uint64_t)p.fused_bwd << 1 | (uint64_t)p.need_dE << 2 | (uint64_t)p.chain2 << 3 | (uint64_t)p.chain2_fused << 4 |
             (uint64_t)p.pair_shared << 5);
  return mix(h, (uint64_t)p.total_floats);
}
uint64_t gno_plan_key(const ngpde_graph *g, const GnoPlan &p) {
  uint64_t h = mix(0x676e6f00u, (uint64_t)(uintptr_t)g);
  h = mix(h, (uint64_t)p.N); h = mix(h, (uint64_t)p.E); h = mix(h, ((uint64_t)p.cin << 32) | (uint64_t)p.cout);
  h = mix(h, ((uint64_t)p.kdim << 32) | (uint64_t)(p.L * 64 + p.aggr * 8 + p.act1));
  h = mix(h, (uint64_t)p.reassoc | (uint64_t)p.fused_msg << 1 | (uint64_t)p.fused_agg << 2 | (uint64_t)p.has_b2 << 3 | (uint64_t)p.gform << 4 | (uint64_t)p.nsplit << 8);
  return mix(h, (uint64_t)p.total_floats);
}

}  // namespace

extern "C" {

size_t ngpde_edge_layer_workspace_bytes(const ngpde_graph_t *g, const ngpde_edge_layer_t *L, int32_t training) {
  if (!g || !L) return 0;
  Arena a;
  Plan p;
  if (make_plan(g, *L, training != 0, a, p) != NGPDE_OK) return 0;
  return p.total_floats * sizeof(float) + kArenaSlack;
}

int32_t ngpde_edge_layer_forward(const ngpde_graph_t *g, const ngpde_edge_layer_t *L, int32_t training, float *y, void *workspace,
                                 size_t workspace_bytes, ngpde_stream_t stream) {
  NGPDE_RANGE();
  int32_t st;
  if ((st = check_common("ngpde_edge_layer_forward", g, L))) return st;
  Arena a;
  a.base = arena_base(workspace);
  Plan p;
  if ((st = make_plan(g, *L, training != 0, a, p))) return st;
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(y != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_edge_layer_forward: y is NULL");
  NGPDE_REQUIRE(workspace != nullptr && workspace_bytes >= p.total_floats * sizeof(float) + kArenaSlack, NGPDE_ERR_WORKSPACE,
                "ngpde_edge_layer_forward: workspace of %zu bytes, %zu needed", workspace_bytes, p.total_floats * sizeof(float) + kArenaSlack);
  if (training) stamp_set(workspace, edge_plan_key(g, p));
  const ngpde_mlp_t &phi = L->phi;
  // the recombined first-layer weights in one launch
  float *outs[3] = {p.wA, p.wB, p.wD};
  if ((st = ngpde_row_blocks_gather(p.h1, p.w1_rows, phi.weight[0], p.rows.n_seg, p.rows.out_index, p.rows.dst0, p.rows.src0, p.rows.nrows,
                                    p.rows.sign, p.rows.n_out, outs, p.rows.out_rows, stream)))
    return st;
  // P (with phi's first bias) and Q: one pass over the shared leading block when the shapes allow
  if ((st = ngpde_dense_pair_forward(p.N, p.A.n, p.A.ptr, p.A.width, p.A.row_div, p.h1, NGPDE_ACT_IDENTITY, p.wA, phi.bias[0], p.P, nullptr,
                                     p.B.n, p.B.ptr, p.B.width, p.B.row_div, p.h1, NGPDE_ACT_IDENTITY, p.wB, nullptr, p.Q, nullptr, stream)))
    return st;
  if (p.de) {
    const float *eb[1] = {L->edge_feat};
    const int32_t ew[1] = {p.de}, er[1] = {1};
    if ((st = ngpde_dense_forward(p.E, 1, eb, ew, er, p.h1, NGPDE_ACT_IDENTITY, p.wD, nullptr, p.Et, nullptr, stream))) return st;
  }
  float *msg_out = p.m ? p.m : y;
  if (p.fused_msg) {
    if ((st = ngpde_edge_mlp_forward(g, p.h1, p.act1, p.P, p.Q, p.Et, p.n_tail, p.n_tail ? p.tail_dout : nullptr, p.n_tail ? p.tail_act : nullptr,
                                     p.n_tail ? p.tail_w : nullptr, p.n_tail ? p.tail_b : nullptr, p.aggr, msg_out, p.save, stream)))
      return st;
  } else {
    if ((st = ngpde_edge_combine_forward(g, p.h1, p.act1, p.P, p.Q, p.Et, p.a0, p.z0, stream))) return st;
    const float *cur = p.a0;
    int prev = p.h1;
    for (int l = 0; l < p.n_tail; ++l) {
      const float *b1[1] = {cur};
      const int32_t w1[1] = {prev}, r1[1] = {1};
      if ((st = ngpde_dense_forward(p.E, 1, b1, w1, r1, p.tail_dout[l], p.tail_act[l], p.tail_w[l], p.tail_b[l], p.ty[l], p.tz[l], stream))) return st;
      cur = p.ty[l];
      prev = p.tail_dout[l];
    }
    if ((st = ngpde_segment_reduce_forward(g, p.mw, p.aggr, cur, msg_out, stream))) return st;
  }
  if (!p.n_upd) return p.m ? ngpde_rk_stage_combine((int64_t)p.N * p.mw, 1.f, p.m, 0, nullptr, nullptr, y, stream) : NGPDE_OK;
  const ngpde_mlp_t &u = L->update;
  int l0 = 0;
  const float *cur = nullptr;
  if (p.chain2) {
    float *y2 = p.n_upd == 2 ? y : p.ua[1];
    if ((st = ngpde_dense_chain2_forward(p.N, p.U.n, p.U.ptr, p.U.width, p.U.row_div, u.dims[1], u.act[0], u.weight[0], u.bias[0], p.ua[0], p.uz[0],
                                         u.dims[2], u.act[1], u.weight[1], u.bias[1], y2, p.uz[1], stream)))
      return st;
    cur = y2;
    l0 = 2;
  } else {
    if ((st = ngpde_dense_forward(p.N, p.U.n, p.U.ptr, p.U.width, p.U.row_div, u.dims[1], u.act[0], u.weight[0], u.bias[0], y, p.uz[0], stream))) return st;
    return NGPDE_OK;
  }
  for (int l = l0; l < p.n_upd; ++l) {
    float *out = (l + 1 == p.n_upd) ? y : p.ua[l];
    const float *b1[1] = {cur};
    const int32_t w1[1] = {u.dims[l]}, r1[1] = {1};
    if ((st = ngpde_dense_forward(p.N, 1, b1, w1, r1, u.dims[l + 1], u.act[l], u.weight[l], u.bias[l], out, p.uz[l], stream))) return st;
    cur = out;
  }
  return NGPDE_OK;
}

int32_t ngpde_edge_layer_backward(const ngpde_graph_t *g, const ngpde_edge_layer_t *L, const float *dy, float *const *d_state,
                                  const ngpde_mlp_grad_t *dphi, const ngpde_mlp_grad_t *dupd, void *workspace, size_t workspace_bytes,
                                  ngpde_stream_t stream) {
  NGPDE_RANGE();
  int32_t st;
  if ((st = check_common("ngpde_edge_layer_backward", g, L))) return st;
  Arena a;
  a.base = arena_base(workspace);
  Plan p;
  if ((st = make_plan(g, *L, true, a, p))) return st;
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(dy != nullptr && dphi != nullptr && (p.n_upd == 0 || dupd != nullptr), NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_edge_layer_backward: dy or a gradient table is NULL");
  NGPDE_REQUIRE(stamp_agrees(workspace, edge_plan_key(g, p)), NGPDE_ERR_STATE,
                "ngpde_edge_layer_backward: this workspace was filled by a forward that chose another kernel sequence or layout (descriptor, graph or "
                "an NGPDE_* path switch changed between the two calls)");
  NGPDE_REQUIRE(workspace != nullptr && workspace_bytes >= p.total_floats * sizeof(float) + kArenaSlack, NGPDE_ERR_WORKSPACE,
                "ngpde_edge_layer_backward: workspace of %zu bytes, %zu needed", workspace_bytes, p.total_floats * sizeof(float) + kArenaSlack);
  const ngpde_mlp_t &phi = L->phi;
  for (int l = 0; l < phi.n_layers; ++l)
    NGPDE_REQUIRE(dphi->dweight[l] != nullptr && (phi.bias[l] == nullptr || dphi->dbias[l] != nullptr), NGPDE_ERR_INVALID_ARGUMENT,
                  "ngpde_edge_layer_backward: phi.layer_%d without its gradient buffers", l + 1);
  for (int l = 0; l < p.n_upd; ++l)
    NGPDE_REQUIRE(dupd->dweight[l] != nullptr && (L->update.bias[l] == nullptr || dupd->dbias[l] != nullptr), NGPDE_ERR_INVALID_ARGUMENT,
                  "ngpde_edge_layer_backward: update layer_%d without its gradient buffers", l + 1);
  auto want_state = [&](int k) { return d_state != nullptr && k >= 0 && d_state[k] != nullptr; };

  // ---- 1. the node update, last layer first
  const float *dmsg = dy;   // gradient of the aggregate
  if (p.n_upd) {
    const ngpde_mlp_t &u = L->update;
    p.U.ptr[p.u_m] = p.m;
    const float *cur = dy;
    int flip = 0;
    for (int l = p.n_upd - 1; l >= 1; --l) {   // layers 2 .. n, each on the previous layer's [N][dims[l]] output
      const float *in = p.ua[l - 1];
      const float *b1[1] = {in};
      const int32_t w1[1] = {u.dims[l]}, r1[1] = {1};
      float *dseg[1] = {p.ng[flip]};
      if ((st = ngpde_dense_backward(p.N, 1, b1, w1, r1, u.dims[l + 1], u.act[l], u.weight[l], p.uz[l], cur, dseg, dupd->dweight[l],
                                     u.bias[l] ? dupd->dbias[l] : nullptr, p.ws, p.ws_bytes, stream)))
        return st;
      cur = p.ng[flip];
      flip ^= 1;
    }
    float *dseg[kMaxBlocks] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < p.U.n; ++i) {
      if (i == p.u_m) dseg[i] = p.dm;
      else if (p.U.row_div[i] == 1 && want_state(p.U.state_of[i])) dseg[i] = p.dU[i];
    }
    if ((st = ngpde_dense_backward(p.N, p.U.n, p.U.ptr, p.U.width, p.U.row_div, u.dims[1], u.act[0], u.weight[0], p.uz[0], cur, dseg,
                                   dupd->dweight[0], u.bias[0] ? dupd->dbias[0] : nullptr, p.ws, p.ws_bytes, stream)))
      return st;
    dmsg = p.dm;
  }

  // ---- 2. the message path
  float *dtw[kMaxL], *dtb[kMaxL];
  for (int l = 0; l < p.n_tail; ++l) { dtw[l] = dphi->dweight[l + 1]; dtb[l] = p.tail_b[l] ? dphi->dbias[l + 1] : nullptr; }
  const float *dE = nullptr;
  if (p.fused_bwd) {
    if ((st = ngpde_edge_mlp_backward(g, p.h1, p.act1, p.P, p.Q, p.Et, p.n_tail, p.n_tail ? p.tail_dout : nullptr, p.n_tail ? p.tail_act : nullptr,
                                      p.n_tail ? p.tail_w : nullptr, p.n_tail ? p.tail_b : nullptr, p.aggr, dmsg, p.dP, p.dQ, p.dE,
                                      p.n_tail ? dtw : nullptr, p.n_tail ? dtb : nullptr, p.ws, p.ws_bytes, stream)))
      return st;
    dE = p.dE;
  } else {
    // gradient of the last per-edge array from the aggregate's, then the tail layers backwards, then the first layer's gather
    const float *last = p.fused_msg ? nullptr : (p.n_tail ? p.ty[p.n_tail - 1] : p.a0);
    const float *agg = p.fused_msg ? nullptr : p.m;   // (max / min / *: the primitives' forward kept both)
    int flip = 0;
    if ((st = ngpde_segment_reduce_backward(g, p.mw, p.aggr, last, agg, dmsg, p.eg[flip], stream))) return st;
    const float *cur = p.eg[flip];
    flip ^= 1;
    for (int l = p.n_tail - 1; l >= 0; --l) {
      const int din = l ? p.tail_dout[l - 1] : p.h1;
      const float *in;
      const float *z;
      if (p.fused_msg) {   // the fused forward kept pre-activations only: a_{l} = act(z_{l})
        const int act_in = l ? p.tail_act[l - 1] : p.act1;
        if ((st = ngpde_activation_forward((int64_t)p.E * din, act_in, p.save[l], p.ea, stream))) return st;
        in = p.ea;
        z = p.save[l + 1];
      } else {
        in = l ? p.ty[l - 1] : p.a0;
        z = p.tz[l];
      }
      const float *b1[1] = {in};
      const int32_t w1[1] = {din}, r1[1] = {1};
      float *dseg[1] = {p.eg[flip]};
      if ((st = ngpde_dense_backward(p.E, 1, b1, w1, r1, p.tail_dout[l], p.tail_act[l], p.tail_w[l], z, cur, dseg, dtw[l], dtb[l], p.ws, p.ws_bytes,
                                     stream)))
        return st;
      cur = p.eg[flip];
      flip ^= 1;
    }
    if ((st = ngpde_edge_combine_backward(g, p.h1, p.act1, cur, p.fused_msg ? p.save[0] : p.z0, p.dE, p.dP, p.dQ, stream))) return st;
    dE = p.dE;
  }

  // ---- 3. the edge features' term: only its weight block has a gradient
  if (p.de) {
    const float *eb[1] = {L->edge_feat};
    const int32_t ew[1] = {p.de}, er[1] = {1};
    float *dseg[1] = {nullptr};
    if ((st = ngpde_dense_backward(p.E, 1, eb, ew, er, p.h1, NGPDE_ACT_IDENTITY, p.wD, nullptr, dE, dseg, p.dwD, nullptr, p.ws, p.ws_bytes, stream))) return st;
  }

  // ---- 4. P and Q back to the state blocks and the recombined weights
  float *dbA = phi.bias[0] ? dphi->dbias[0] : nullptr;   // (phi's first bias rides on the target side)
  if (p.pair_shared) {
    // dx = dP WA^T + dQ WB^T (+ the node update's gradient w.r.t. the same block) in one launch
    const int s0 = p.A.state_of[0];
    float *dx = want_state(s0) ? d_state[s0] : p.dA[0];
    const float *addend = nullptr;
    for (int i = 0; i < p.U.n; ++i)
      if (p.n_upd && p.U.state_of[i] == s0 && want_state(s0)) addend = p.dU[i];
    if ((st = ngpde_dense_pair_backward(p.N, p.A.n, p.A.ptr, p.A.width, p.A.row_div, p.wA, p.dP, p.dwA, dbA, p.B.n, p.B.ptr, p.B.width, p.B.row_div,
                                        p.wB, p.dQ, p.dwB, nullptr, 64, dx, addend, p.ws, p.ws_bytes, stream)))
      return st;
  } else {
    float *dsegA[kMaxBlocks] = {nullptr, nullptr, nullptr, nullptr}, *dsegB[kMaxBlocks] = {nullptr, nullptr, nullptr, nullptr};
    for (int i = 0; i < p.A.n; ++i)
      if (want_state(p.A.state_of[i])) dsegA[i] = p.dA[i];
    for (int i = 0; i < p.B.n; ++i)
      if (want_state(p.B.state_of[i])) dsegB[i] = p.dB[i];
    if ((st = ngpde_dense_backward(p.N, p.A.n, p.A.ptr, p.A.width, p.A.row_div, p.h1, NGPDE_ACT_IDENTITY, p.wA, nullptr, p.dP, dsegA, p.dwA, dbA, p.ws,
                                   p.ws_bytes, stream)))
      return st;
    if ((st = ngpde_dense_backward(p.N, p.B.n, p.B.ptr, p.B.width, p.B.row_div, p.h1, NGPDE_ACT_IDENTITY, p.wB, nullptr, p.dQ, dsegB, p.dwB, nullptr, p.ws,
                                   p.ws_bytes, stream)))
      return st;
    // a state block's cotangent = target side + source side (+ the node update's): one combination launch per block
    for (int i = 0; i < p.A.n; ++i) {
      const int k = p.A.state_of[i];
      if (!want_state(k)) continue;
      // (the order of the composed layers' autograd graphs, so that the results are theirs bit for bit: MPPDEConv's node update
      // delivers its term into the target side's first, the others sum target, source, update)
      const float *upd = nullptr;
      for (int j = 0; j < p.U.n; ++j)
        if (p.n_upd && p.U.state_of[j] == k) upd = p.dU[j];
      const float *terms[2] = {p.dB[i], nullptr};
      const float coefs[2] = {1.f, 1.f};
      int nt = 1;
      if (upd) {
        if (p.kind == NGPDE_LAYER_MPPDE) { terms[0] = upd; terms[1] = p.dB[i]; }
        else terms[1] = upd;
        nt = 2;
      }
      if ((st = ngpde_rk_stage_combine((int64_t)p.N * p.A.width[i], 1.f, p.dA[i], nt, terms, coefs, d_state[k], stream))) return st;
    }
  }
  // ---- 5. the first weight's gradient from its recombined blocks'
  float *douts[3] = {p.dwA, p.dwB, p.dwD};
  return ngpde_row_blocks_scatter(p.h1, p.w1_rows, dphi->dweight[0], p.rows.n_seg, p.rows.out_index, p.rows.dst0, p.rows.src0, p.rows.nrows, p.rows.sign,
                                  p.rows.n_out, douts, p.rows.out_rows, stream);
}


size_t ngpde_gno_layer_workspace_bytes(const ngpde_graph_t *g, const ngpde_gno_layer_t *L, int32_t training) {
  if (!g || !L) return 0;
  Arena a;
  GnoPlan p;
  if (make_gno_plan(g, *L, training != 0, a, p) != NGPDE_OK) return 0;
  return p.total_floats * sizeof(float) + kArenaSlack;
}

int32_t ngpde_gno_layer_forward(const ngpde_graph_t *g, const ngpde_gno_layer_t *L, int32_t training, float *y, void *workspace,
                                size_t workspace_bytes, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr && L != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_layer_forward: graph or layer descriptor is NULL");
  Arena a;
  a.base = arena_base(workspace);
  GnoPlan p;
  int32_t st;
  if ((st = make_gno_plan(g, *L, training != 0, a, p))) return st;
  if (g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(y != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_layer_forward: y is NULL");
  NGPDE_REQUIRE(workspace != nullptr && workspace_bytes >= p.total_floats * sizeof(float) + kArenaSlack, NGPDE_ERR_WORKSPACE,
                "ngpde_gno_layer_forward: workspace of %zu bytes, %zu needed", workspace_bytes, p.total_floats * sizeof(float) + kArenaSlack);
  if (training) stamp_set(workspace, gno_plan_key(g, p));
  const ngpde_mlp_t &phi = L->phi;
  float *outs[3];
  int no = 0;
  if (p.ds) { outs[no++] = p.wa; outs[no++] = p.wb; }
  if (p.de) outs[no++] = p.wd;
  if ((st = ngpde_row_blocks_gather(p.h1, phi.dims[0], phi.weight[0], p.rows.n_seg, p.rows.out_index, p.rows.dst0, p.rows.src0, p.rows.nrows,
                                    p.rows.sign, p.rows.n_out, outs, p.rows.out_rows, stream)))
    return st;
  const int32_t one[1] = {1};
  if (p.de) {   // the edge features' term (it carries phi's first bias when there are no node features)
    const float *eb[1] = {L->edge_feat};
    const int32_t ew[1] = {p.de};
    if ((st = ngpde_dense_forward(p.E, 1, eb, ew, one, p.h1, NGPDE_ACT_IDENTITY, p.wd, p.ds ? nullptr : phi.bias[0], p.Et, nullptr, stream))) return st;
  }
  const float *w2 = phi.weight[p.L - 1], *b2 = phi.bias[p.L - 1];
  if (p.reassoc && !p.gform) {   // W2 as [in][out][k]: the transpose of [k][in * out]
    if ((st = ngpde_transpose(p.kdim, p.cin * p.cout, w2, p.wr, stream))) return st;
  }
  {   // the small node-level Dense layers -- P, Q on the node coordinates, B2 h, W h -- in ONE launch
    int64_t n[4];
    int32_t nseg[4], sw[4], srd[4], dout[4], act[4];
    const float *sp[4], *wt[4], *bs[4];
    float *ys[4], *zs[4];
    int q = 0;
    auto add = [&](const float *x, int xw, const float *w, const float *b, int d, float *yq) {
      n[q] = p.N; nseg[q] = 1; sp[q] = x; sw[q] = xw; srd[q] = 1; dout[q] = d; act[q] = NGPDE_ACT_IDENTITY; wt[q] = w; bs[q] = b; ys[q] = yq; zs[q] = nullptr;
      ++q;
    };
    if (p.ds) { add(L->node_feat, p.ds, p.wa, phi.bias[0], p.h1, p.P); add(L->node_feat, p.ds, p.wb, nullptr, p.h1, p.Q); }
    if (p.has_b2 && !p.gform) add(L->h, p.cin, b2, nullptr, p.cout, p.Bh);     // b2 read as the [in][out] matrix B2[i][o] = b2[o + out * i]
    if (!p.gform) add(L->h, p.cin, L->weight, nullptr, p.cout, p.Wh);
    if (q && (st = ngpde_dense_multi_forward(q, n, nseg, sp, sw, srd, dout, act, wt, bs, ys, zs, stream))) return st;
  }
  if (p.gform) {
    // G_i = Z_i^T H_i and the sum of the neighbours' h per target, then ONE node-level product against W2 as it lies in memory, the
    // b2 term on the summed rows, and the layer's tail in the slab reduction (:523-536)
    if ((st = ngpde_gno_gform_aggregate(g, p.cin, p.kdim, p.act1, p.aggr == NGPDE_AGGR_MEAN ? 1 : 0, p.P, p.Q, p.Et, L->h, p.G, p.has_b2 ? p.hs : nullptr, p.a,
                                        stream)))
      return st;
    return ngpde_gno_gform_transform(p.N, p.cin, p.kdim, p.cout, p.act, p.G, w2, p.has_b2 ? p.hs : nullptr, p.has_b2 ? b2 : nullptr, L->h, L->weight, L->bias, y,
                                     p.zt, p.slabs, p.nsplit, stream);
  }
  if (p.reassoc) {
    const float *hb[1] = {L->h};
    const int32_t hw[1] = {p.cin};
    if ((st = ngpde_dense_forward(p.N, 1, hb, hw, one, p.cout * p.kdim, NGPDE_ACT_IDENTITY, p.wr, nullptr, p.T, nullptr, stream))) return st;
  }
  if (p.fused_msg) {
    if ((st = ngpde_gno_message_forward(g, p.cout, p.kdim, p.act1, p.P, p.Q, p.Et, p.T, p.Bh, p.a, p.m, stream))) return st;
  } else {
    if ((st = ngpde_edge_combine_forward(g, p.h1, p.act1, p.P, p.Q, p.Et, p.a0, p.z0, stream))) return st;
    const float *cur = p.a0;
    for (int l = 0; l < p.n_mid; ++l) {
      const float *b1[1] = {cur};
      const int32_t w1[1] = {phi.dims[l + 1]};
      if ((st = ngpde_dense_forward(p.E, 1, b1, w1, one, phi.dims[l + 2], phi.act[l + 1], phi.weight[l + 1], phi.bias[l + 1], p.ty[l], p.tz[l], stream)))
        return st;
      cur = p.ty[l];
    }
    if (p.reassoc) st = ngpde_gno_apply_forward(g, p.cout, p.kdim, p.T, p.Bh, cur, p.m, stream);
    else st = ngpde_gno_contract_forward(g, p.cin, p.cout, cur, L->h, p.m, stream);
    if (st) return st;
  }
  if ((st = ngpde_segment_reduce_forward(g, p.cout, p.aggr, p.m, p.agg, stream))) return st;
  return ngpde_bias_act_forward(p.N, p.cout, p.act, p.agg, p.Wh, L->bias, y, p.zt, stream);
}

int32_t ngpde_gno_layer_backward(const ngpde_graph_t *g, const ngpde_gno_layer_t *L, const float *dy, float *dh, const ngpde_mlp_grad_t *dphi,
                                 float *dweight, float *dbias, void *workspace, size_t workspace_bytes, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr && L != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_gno_layer_backward: graph or layer descriptor is NULL");
  Arena a;
  a.base = arena_base(workspace);
  GnoPlan p;
  int32_t st;
  if ((st = make_gno_plan(g, *L, true, a, p))) return st;
  if (g->n_nodes == 0) return NGPDE_OK;
  const ngpde_mlp_t &phi = L->phi;
  NGPDE_REQUIRE(dy != nullptr && dphi != nullptr && dweight != nullptr && (L->bias == nullptr || dbias != nullptr), NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_gno_layer_backward: dy or a gradient buffer is NULL");
  for (int l = 0; l < p.L; ++l)
    NGPDE_REQUIRE(dphi->dweight[l] != nullptr && (phi.bias[l] == nullptr || dphi->dbias[l] != nullptr), NGPDE_ERR_INVALID_ARGUMENT,
                  "ngpde_gno_layer_backward: phi.layer_%d without its gradient buffers", l + 1);
  NGPDE_REQUIRE(stamp_agrees(workspace, gno_plan_key(g, p)), NGPDE_ERR_STATE,
                "ngpde_gno_layer_backward: this workspace was filled by a forward that chose another kernel sequence or layout (descriptor, graph or "
                "an NGPDE_* path switch changed between the two calls)");
  NGPDE_REQUIRE(workspace != nullptr && workspace_bytes >= p.total_floats * sizeof(float) + kArenaSlack, NGPDE_ERR_WORKSPACE,
                "ngpde_gno_layer_backward: workspace of %zu bytes, %zu needed", workspace_bytes, p.total_floats * sizeof(float) + kArenaSlack);
  const int32_t one[1] = {1};
  const size_t N = (size_t)p.N;
  // ---- the tail y = act.(agg + W h + b): dz is the gradient of the aggregate AND of W h
  const float *dz = dy;
  if (p.act != 0 || L->bias) {
    float *dzo = p.act != 0 ? p.dz : const_cast<float *>(dy);   // identity: the library only sums the columns for the bias
    if ((st = ngpde_bias_act_backward(p.N, p.cout, p.act, dy, p.zt, dzo, L->bias ? dbias : nullptr, p.ws, p.ws_bytes, stream))) return st;
    dz = dzo;
  }
  const float *hb[1] = {L->h};
  const int32_t hw[1] = {p.cin};
  // ---- the message: gradient of the aggregate -> dT, dBh (reassociated) / dh (literal), and the per-edge pre-activation's gradient
  const float *dlast = nullptr;   // gradient of the last per-edge array the primitives formed (their tail walks back from it)
  if (p.gform) {   // the forward contracted the edge index first and formed no T: the by-source pullback makes its own
    if ((st = ngpde_transpose(p.kdim, p.cin * p.cout, phi.weight[p.L - 1], p.wr, stream))) return st;
    if ((st = ngpde_dense_forward(p.N, 1, hb, hw, one, p.cout * p.kdim, NGPDE_ACT_IDENTITY, p.wr, nullptr, p.T, nullptr, stream))) return st;
  }
  if (p.fused_agg) {
    const float *dagg = dz;
    if (p.aggr == NGPDE_AGGR_MEAN) {   // a node's 1 / deg once per node, not once per edge inside the launch
      hipLaunchKernelGGL(inv_in_degree_kernel, dim3((unsigned)((p.N + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (int)p.N, g->by_t.rowptr, p.inv);
      if ((st = ngpde_rows_scale(p.N, p.cout, dz, p.inv, p.dagg_s, stream))) return st;
      dagg = p.dagg_s;
    }
    if ((st = ngpde_gno_message_backward_from_nodes(g, p.cout, p.kdim, NGPDE_AGGR_SUM, p.act1, p.T, p.a, dagg, p.dT, p.has_b2 ? p.dBh : nullptr,
                                                    (p.ds || p.de) ? p.dzE : nullptr, p.ds ? p.dQ : nullptr, stream)))
      return st;
    if (p.ds && (st = ngpde_segment_reduce_forward(g, p.kdim, NGPDE_AGGR_SUM, p.dzE, p.dP, stream))) return st;   // dP = sums of dz by target
  } else {
    if ((st = ngpde_segment_reduce_backward(g, p.cout, p.aggr, p.m, p.agg, dz, p.dm, stream))) return st;
    if (p.fused_msg) {
      if ((st = ngpde_gno_apply_backward(g, p.cout, p.kdim, p.T, p.a, p.dm, p.dT, p.has_b2 ? p.dBh : nullptr, p.eg[0], stream))) return st;
      if ((st = ngpde_edge_combine_backward(g, p.kdim, p.act1, p.eg[0], p.a, p.dzE, p.ds ? p.dP : nullptr, p.ds ? p.dQ : nullptr, stream))) return st;
    } else {
      const float *last = p.n_mid ? p.ty[p.n_mid - 1] : p.a0;
      if (p.reassoc) st = ngpde_gno_apply_backward(g, p.cout, p.kdim, p.T, last, p.dm, p.dT, p.has_b2 ? p.dBh : nullptr, p.eg[0], stream);
      else st = ngpde_gno_contract_backward(g, p.cin, p.cout, last, L->h, p.dm, p.eg[0], p.dxC, p.ws, p.ws_bytes, stream);
      if (st) return st;
      dlast = p.eg[0];
    }
  }
  if (!p.fused_msg) {
    int flip = 1;
    const float *cur = dlast;
    for (int l = p.n_mid - 1; l >= 0; --l) {   // phi's layers l + 2 on [E] rows, last first
      const float *in = l ? p.ty[l - 1] : p.a0;
      const float *b1[1] = {in};
      const int32_t w1[1] = {phi.dims[l + 1]};
      float *dseg[1] = {p.eg[flip]};
      if ((st = ngpde_dense_backward(p.E, 1, b1, w1, one, phi.dims[l + 2], phi.act[l + 1], phi.weight[l + 1], p.tz[l], cur, dseg, dphi->dweight[l + 1],
                                     phi.bias[l + 1] ? dphi->dbias[l + 1] : nullptr, p.ws, p.ws_bytes, stream)))
        return st;
      cur = p.eg[flip];
      flip ^= 1;
    }
    if ((st = ngpde_edge_combine_backward(g, p.h1, p.act1, cur, p.z0, p.dzE, p.ds ? p.dP : nullptr, p.ds ? p.dQ : nullptr, stream))) return st;
  }
  // ---- phi's first layer: the edge features' block, the two node-feature blocks (constants: weight gradients only)
  if (p.de) {
    const float *eb[1] = {L->edge_feat};
    const int32_t ew[1] = {p.de};
    float *dseg[1] = {nullptr};
    if ((st = ngpde_dense_backward(p.E, 1, eb, ew, one, p.h1, NGPDE_ACT_IDENTITY, p.wd, nullptr, p.dzE, dseg, p.dwd,
                                   (!p.ds && phi.bias[0]) ? dphi->dbias[0] : nullptr, p.ws, p.ws_bytes, stream)))
      return st;
  }
  if (p.ds) {
    const float *sb[1] = {L->node_feat};
    const int32_t sw[1] = {p.ds};
    float *dseg[1] = {nullptr};
    if ((st = ngpde_dense_backward(p.N, 1, sb, sw, one, p.h1, NGPDE_ACT_IDENTITY, p.wa, nullptr, p.dP, dseg, p.dwa, phi.bias[0] ? dphi->dbias[0] : nullptr,
                                   p.ws, p.ws_bytes, stream)))
      return st;
    if ((st = ngpde_dense_backward(p.N, 1, sb, sw, one, p.h1, NGPDE_ACT_IDENTITY, p.wb, nullptr, p.dQ, dseg, p.dwb, nullptr, p.ws, p.ws_bytes, stream))) return st;
  }
  float *douts[3];
  int no = 0;
  if (p.ds) { douts[no++] = p.dwa; douts[no++] = p.dwb; }
  if (p.de) douts[no++] = p.dwd;
  if ((st = ngpde_row_blocks_scatter(p.h1, phi.dims[0], dphi->dweight[0], p.rows.n_seg, p.rows.out_index, p.rows.dst0, p.rows.src0, p.rows.nrows,
                                     p.rows.sign, p.rows.n_out, douts, p.rows.out_rows, stream)))
    return st;
  // ---- the node-level Dense layers that read h: B2 h, W h (in the forward's order), T
  const bool want_h = dh != nullptr;
  if (p.has_b2) {
    float *dseg[1] = {want_h ? p.dxB : nullptr};
    if ((st = ngpde_dense_backward(p.N, 1, hb, hw, one, p.cout, NGPDE_ACT_IDENTITY, phi.bias[p.L - 1], nullptr, p.dBh, dseg, dphi->dbias[p.L - 1], nullptr,
                                   p.ws, p.ws_bytes, stream)))
      return st;
  }
  {
    float *dseg[1] = {want_h ? p.dxW : nullptr};
    if ((st = ngpde_dense_backward(p.N, 1, hb, hw, one, p.cout, NGPDE_ACT_IDENTITY, L->weight, nullptr, dz, dseg, dweight, nullptr, p.ws, p.ws_bytes, stream)))
      return st;
  }
  if (p.reassoc) {
    float *dseg[1] = {want_h ? p.dxT : nullptr};
    if ((st = ngpde_dense_backward(p.N, 1, hb, hw, one, p.cout * p.kdim, NGPDE_ACT_IDENTITY, p.wr, nullptr, p.dT, dseg, p.dwr, nullptr, p.ws, p.ws_bytes,
                                   stream)))
      return st;
    if ((st = ngpde_transpose(p.cin * p.cout, p.kdim, p.dwr, dphi->dweight[p.L - 1], stream))) return st;
  }
  if (!want_h) return NGPDE_OK;
  // dh: the composed layer's order -- T's term + (B2 h's + W h's) reassociated, the contraction's + W h's in the literal form
  const float coef[1] = {1.f};
  if (p.reassoc) {
    const float *inner = p.dxW;
    if (p.has_b2) {
      const float *t1[1] = {p.dxW};
      if ((st = ngpde_rk_stage_combine((int64_t)N * p.cin, 1.f, p.dxB, 1, t1, coef, p.dsum, stream))) return st;
      inner = p.dsum;
    }
    const float *t2[1] = {inner};
    return ngpde_rk_stage_combine((int64_t)N * p.cin, 1.f, p.dxT, 1, t2, coef, dh, stream);
  }
  const float *t3[1] = {p.dxW};
  return ngpde_rk_stage_combine((int64_t)N * p.cin, 1.f, p.dxC, 1, t3, coef, dh, stream);
}

}  // extern "C"
