// common.h -- internals shared by the translation units of libngpde_hip.so (not part of the ABI).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/ngpde.h"

namespace ngpde {

// thread-local message behind ngpde_last_error()
std::string &last_error();
int32_t fail(int32_t code, const char *fmt, ...);

// after a kernel launch inside a function that returns an ngpde status
#define NGPDE_LAUNCH_CHECK(name)                                                                    \
  do {                                                                                              \
    hipError_t _le = hipGetLastError();                                                             \
    if (_le != hipSuccess)                                                                          \
      return ::ngpde::fail(NGPDE_ERR_HIP, "%s launch failed: %s", name, hipGetErrorString(_le));    \
  } while (0)

#define NGPDE_HIP_CHECK(expr)                                                                     \
  do {                                                                                            \
    hipError_t _e = (expr);                                                                       \
    if (_e != hipSuccess)                                                                         \
      return ::ngpde::fail(NGPDE_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e),  \
                           __FILE__, __LINE__);                                                   \
  } while (0)

#define NGPDE_REQUIRE(cond, code, ...)                                                            \
  do {                                                                                            \
    if (!(cond)) return ::ngpde::fail(code, __VA_ARGS__);                                         \
  } while (0)

// One direction of the derived graph.  Row r lists the edges whose target (by_target) or source
// (by_source) is r, in the caller's COO order (stable counting sort), so segmented sums visit
// contributions in the same order as NNlib's serial scatter over the COO list.
struct Csr {
  int32_t *rowptr = nullptr;  // [n_nodes + 1]
  int32_t *col = nullptr;     // [n_edges] the node at the other end of each entry
  int32_t *eid = nullptr;     // [n_edges] position of the entry in the COO list
  int2 *ent = nullptr;        // [n_edges] {col, bits of GCN coefficient w_e * c[col]}; set by set_gcn_norm
  int4 *sched = nullptr;      // [n_sched] tile schedule entries {node, row start, degree, bits of c[node]}; node < 0 = padding
  int32_t *xpos = nullptr;    // [n_edges] position of each entry in the OTHER direction's list (same COO edge)
  int2 *ell = nullptr;        // [n_sched][kEllWidth] the first kEllWidth {col, coef} entries of each schedule row
                              // (zero padded), addressed by schedule POSITION: loadable without first reading `sched`
  // LDS-staged aggregation: per tile the UNIQUE rows its entries reference (own rows first: slot k = k-th row
  // of the tile), pre-scaled by c when staged; per schedule row 16 slot bytes (kHaloCap = the all-zero row).
  int2 *halo = nullptr;       // [n_tiles][kHaloCap] {node, bits of c[node]}
  int2 *tile_info = nullptr;  // [n_tiles] {halo count (0: tile uses the global gather), unused}
  uint8_t *slots = nullptr;   // [n_sched][kSlotWidth]
  float *slot_w = nullptr;    // [n_sched][kSlotWidth] per-edge weights w_e, or NULL for unweighted graphs
  bool halo_ok = false;       // every tile fits (<= kHaloCap distinct rows, degrees <= kSlotWidth)
  int32_t max_halo = 0;       // largest halo count over the tiles (sizes the LDS halo region of the fused edge kernel)
  std::vector<int32_t> h_rowptr, h_col, h_eid;
};

// Inverse of the FOREIGN part of the by-target halo lists, built on first use (graph_halo_inverse): the nodes that some OTHER tile's
// halo references (slot >= kTileRows), and for each of them the (tile, halo slot) pairs that hold it, as tile * stride + slot,
// ascending by tile.  Lets a kernel leave one partial row per (tile, foreign slot), write the partial row of an OWN slot straight
// to the node, and a second pass add the foreign rows -- a by-source sum without an [E][h] array (edge_mlp64.hip).  One stride per
// handle for its whole life (a published table is never freed before ngpde_graph_destroy).
struct HaloInverse {
  int32_t *node = nullptr;  // [n_listed] nodes with at least one foreign entry, ascending
  int32_t *ptr = nullptr;   // [n_listed + 1]
  int32_t *ent = nullptr;   // [ptr[n_listed]]
  int32_t n_listed = 0;
  int stride = 0;
};

constexpr int kTileRows = 32;  // node rows per workgroup of the fused kernels
constexpr int kHaloCap = 96;   // unique rows a tile may stage in LDS (24 KB at D = 64), plus one zero row
constexpr int kSlotWidth = 32;  // slot bytes per row of the LDS-staged aggregation (rows with more: global gather)
constexpr int kEllWidth = 16;  // entries per row held in the fixed-width block (rows with more spill to the CSR list)

}  // namespace ngpde

struct ngpde_graph {
  int64_t n_nodes = 0, n_edges = 0;
  int32_t n_graphs = 1;
  ngpde::Csr by_t, by_s;
  // GCN normalisation (src/layers.jl:210-226)
  bool has_norm = false;
  int32_t self_loops = 0;
  float *c = nullptr;  // [n_nodes] 1/sqrt(degree)
  float *w_coo = nullptr;  // [n_edges] the edge weights set_gcn_norm was given, in COO order (owned copy), or NULL: what the persistent
                           // solver's hub geometry builds its per-entry weight lists from (node_persistent_setup)
  int32_t max_in_degree = 0, max_out_degree = 0;
  // Locality schedule: a permutation of the nodes in which consecutive runs of kTileRows nodes are
  // graph-compact clusters (BFS-grown) and consecutive clusters are adjacent.  The fused kernels give
  // each workgroup one run, and each XCD a contiguous range of runs, so the rows a workgroup gathers
  // are mostly the rows its own XCD wrote/read last: L2-resident instead of cross-XCD traffic.
  std::vector<int32_t> h_order;
  int32_t *order = nullptr;   // the same permutation on the device
  bool device_built = false;  // built by ngpde_graph_create_device: no host copies of the CSR lists
  int32_t n_sched = 0;  // n_tiles * kTileRows
  // derived on first use, under lazy_mu (the handle stays shareable between host threads)
  mutable std::mutex lazy_mu;
  mutable int sched_same = -1;   // do the by-target and by-source schedules name the same node at every position? (-1: not asked yet; gat_fused.hip)
  mutable ngpde::HaloInverse halo_inv;
};

namespace ngpde {

// ---- device-side handle construction (graph_device.hip)
template <class I>
int32_t graph_create_device(int64_t n_nodes, int64_t n_edges, const I *s, const I *t, int index_base, int32_t n_graphs,
                            const int32_t *order_dev, hipStream_t stream, ngpde_graph **out);
int32_t set_gcn_norm_device(ngpde_graph *g, int add_self_loops, const float *w_dev, int weighted_degree, hipStream_t stream);
// the inverse of the by-target halo lists (built once per handle, synchronously, on the host; graph.hip)
int32_t graph_halo_inverse(const ngpde_graph *g, int stride, const HaloInverse **out);

// ---- launchers implemented in gcn_kernels.hip ---------------------------------------------------

// Weighted combination of node-local rows that a fused kernel evaluates in its epilogue/prologue:
//   v[row] = coef_self * (value held in registers) + sum_k coef[k] * ptr[k][row]
struct Comb {
  int n = 0;
  const float *ptr[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  float coef[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  float coef_self = 0.f;
};

struct OwnFirst;   // a solver plan's own-first slot tables (below)
struct FusedFwdArgs {
  const ngpde_graph *g = nullptr;
  int d = 0, act = 0;
  const float *x = nullptr;      // [N][d] gathered operand
  const float *wt = nullptr;     // [d][d] row-major [in][out]  (= Julia (out x in) column-major)
  const float *bias = nullptr;   // [d] or null
  float *y = nullptr;            // [N][d] act(z)
  float *save_agg = nullptr;     // [N][d] or null
  float *save_z = nullptr;       // [N][d] or null
  uint8_t *save_mask = nullptr;  // [fused_mask_bytes] or null: 4 sign bits of z per thread and row (relu' for the pullback)
  bool pre = false;              // pre-scaled pipeline: x holds c .* x, y / comb_out are written as c .* y (see fused_prescaled_supported)
  const OwnFirst *of = nullptr;  // a solver plan's own-first slot tables (halo path only), or NULL: the handle's lists
  // optional Runge-Kutta stage combination evaluated on the freshly computed rows of y
  bool has_comb = false;
  Comb comb;
  float *comb_out = nullptr;
  // optional start/stop events attached to the dispatch itself (profiling pass only)
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};
bool fused_supported(int din, int dout);
bool fused_prescaled_supported(const ngpde_graph *g, int d);
int32_t launch_fused_fwd(const FusedFwdArgs &a, hipStream_t stream);

struct FusedBwdArgs {
  const ngpde_graph *g = nullptr;
  int d = 0, act = 0;
  bool aggregate = false;        // true: T = A^T-aggregate(g_in) rows; false: T = rows of g_in
  const float *g_in = nullptr;   // [N][d]
  // combination stage (adjoint of the RK stage sums)
  bool has_comb = false;
  Comb comb;                     // V = comb.coef_self * T + sum_k ...
  float *store_t = nullptr;      // [N][d] or null: T rows (the stage adjoint U-bar_i)
  float *store_v = nullptr;      // [N][d] or null: V rows (lambda update)
  float v_scale = 1.f;           // K-bar = v_scale * V
  // dense part (skipped when do_dense == false)
  bool do_dense = true;
  const float *z = nullptr;      // [N][d] saved pre-activation (or y for relu/identity); unused when mask is given
  const uint8_t *mask = nullptr; // sign bits written by the forward launch (same tile / thread layout), relu only
  bool pre = false;              // pre-scaled pipeline: g_in holds c .* G, T is dL/d(c .* x), g_out is written as c .* G
  const OwnFirst *of = nullptr;  // as in FusedFwdArgs (the by-source tables)
  const float *saved_agg = nullptr;  // [N][d]
  const float *wt = nullptr;     // [d][d]
  float *g_out = nullptr;        // [N][d]  dZ * W
  float *slab_dw = nullptr;      // [n_blocks][d*d] accumulated (+=)
  float *slab_db = nullptr;      // [n_blocks][d]   accumulated (+=)
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};
int fused_tile_rows();
int fused_num_blocks(int64_t n_nodes);
size_t fused_mask_bytes(int64_t n_nodes, int d);   // bytes of one sign-bit mask (FusedFwdArgs::save_mask)
int fused_num_slabs(int64_t n_nodes, int d);   // <= fused_num_blocks: slabs the backward launches actually write
int32_t launch_fused_bwd(const FusedBwdArgs &a, hipStream_t stream);
// ct > 0: dW slabs in MFMA-fragment order (ct = D/16 column tiles) -> row-major [in][out]; ct == 0: plain sum
int32_t launch_reduce_slabs(const float *slab, int n_slabs, int len, int ct, float *out, hipStream_t stream);
int32_t launch_zero(void *ptr, size_t bytes, hipStream_t stream);   // graph-capture-safe replacement of hipMemsetAsync(ptr, 0, bytes)
// ---- persistent solver launches (node_persistent.hip): the whole forward solve / adjoint of the 2 x GCNConv(64 => 64) plan as
// ONE launch each, tiles synchronised by per-tile phase flags
// Own-first slot tables of a solver plan over GCNConv layers (node_persistent.hip: own_first_tables_build).  A 32-row tile's own rows
// are in LDS when a phase of the persistent solver starts; what it waits for are the rows of other tiles.  Per direction the plan
// holds a COPY of the handle's slot bytes in which every row lists the slots of its tile's own rows first, padded with the all-zero
// row to a whole number of 4-slot rounds common to the four rows of a wave, then the slots of foreign rows -- and the schedule
// entries with the padded length in place of the degree, so that every kernel of the plan (persistent in all its forms, replayed)
// walks the same rounds in the same order and the results stay bitwise comparable -- plus, per wave, the number of own rounds,
// which the one-tile persistent kernels sum BEFORE they wait for their neighbours.  The handle's own lists are untouched (the
// edge-function, GAT and VMH kernels index per-edge arrays by a row's CSR position).
struct OwnFirst {
  uint8_t *slots[2] = {nullptr, nullptr};   // [n_sched][kSlotWidth]            (0: by target, 1: by source)
  int4 *sched[2] = {nullptr, nullptr};      // [n_sched] {node, row start, PADDED length, bits of c}
  float *slot_w[2] = {nullptr, nullptr};    // [n_sched][kSlotWidth] or NULL (unweighted)
  uint8_t *pre[2] = {nullptr, nullptr};     // [n_tiles][8] own rounds of wave w's rows 4 w .. 4 w + 3 (0: that wave keeps the handle's order)
};
int32_t own_first_tables_build(const ngpde_graph *g, OwnFirst *of, hipStream_t stream);
void own_first_tables_free(OwnFirst *of);

struct NodePersist {
  int n_tiles = 0;
  int pair_wgs = 0;            // tile-pair mode: workgroups of a launch (each holds tiles t and t + pair_wgs), else 0
  int *nbr = nullptr;          // [n_tiles][64] wait lists, -1 padded
  unsigned *sync = nullptr;    // [2 n_tiles + 1] 128-byte lines: phase flag per tile for slot 0, for slot 1, then the abort word
  size_t sync_bytes = 0;
  unsigned *fault = nullptr;   // sticky: some persistent launch of this plan gave up waiting (device pointer of fault_host)
  volatile unsigned *fault_host = nullptr;   // the same word in pinned host memory: readable without a synchronisation
  int *stats = nullptr;        // [n_tiles][2]: slot-phases of the last forward / adjoint launch gathered ahead of time (interleaved kernels)
  float *coef = nullptr;       // device tables indexed by the stage: forward [42], adjoint [48] (layout: node.hip)
  // hub geometry (node_persistent.hip: graphs whose tiles do not fit the handle's 96-row halo lists): per direction (0 by target,
  // 1 by source) the tile lists the HUB kernels read, built by node_persistent_setup; `nbr` is then [n_tiles][256]
  bool hub = false;
  struct HubLists {
    int32_t *halo = nullptr;   // [n_tiles][256]
    uint8_t *slots = nullptr;  // [n_tiles][4096]
    int2 *rows = nullptr;      // [n_sched]
    int4 *info = nullptr;      // [n_tiles]
    uint8_t *longs = nullptr;  // [n_tiles][32]
    float *w = nullptr;        // [n_tiles][4096] the entries' edge weights beside their slot bytes, or NULL (unweighted)
    int4 *sched = nullptr;     // [n_sched] the geometry's own tile partition (both directions hold the same one)
  } hub_lists[2];
};
struct NodePersistFwd {
  const ngpde_graph *g = nullptr;
  const NodePersist *ps = nullptr;
  int n_steps = 0, S = 0, act = 0, n_members = 1;
  const float *u_in = nullptr;
  float *u_out = nullptr, *bufA = nullptr, *bufB = nullptr;
  const float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
  float *tape = nullptr;
  uint8_t *masks = nullptr;
  float *ztape = nullptr;      // pre-activations (adjoint of an activation other than relu), same shape as tape
  size_t row_elems = 0, mask_bytes = 0;
  bool interleave = false;     // two members of a batch at a time per workgroup: bufA / bufB hold two [N][64] arrays each
  bool pair = false;           // two TILES of the one member per workgroup (graphs of more tiles than co-resident workgroups)
  int k_tiles = 0;             // > 0: tile rounds -- k_tiles tiles per workgroup taking turns, their state in `state` [7][N][64]
  float *state = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  bool no_latch = false;       // the caller latches the fault word in its next kernel (node.hip: the exit scaling)
  const OwnFirst *of = nullptr;   // the plan's own-first slot tables, or NULL (the handle's lists)
};
struct NodePersistBwd {
  const ngpde_graph *g = nullptr;
  const NodePersist *ps = nullptr;
  int n_steps = 0, S = 0, n_members = 1, act = NGPDE_ACT_RELU;
  float *lam = nullptr, *g1 = nullptr, *g2 = nullptr;
  const float *w1 = nullptr, *w2 = nullptr, *tape = nullptr, *ztape = nullptr;
  const uint8_t *masks = nullptr;
  size_t row_elems = 0, mask_bytes = 0;
  float *slab_dw1 = nullptr, *slab_db1 = nullptr, *slab_dw2 = nullptr, *slab_db2 = nullptr;
  bool interleave = false;     // two members at a time: g1 / g2 hold two [N][64] arrays each, ubar = [2][5][N][64] scratch
  bool pair = false;           // two tiles of the one member per workgroup (ubar = [5][N][64] scratch)
  int k_tiles = 0;             // > 0: tile rounds (ubar = [5][N][64], lam holds lambda between the turns)
  float *ubar = nullptr;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
  bool no_latch = false;       // the caller latches the fault word in its next kernel (node.hip: the slab reduction)
  const OwnFirst *of = nullptr;   // the plan's own-first slot tables, or NULL (the handle's lists)
};
const unsigned *node_persistent_abort_word(const NodePersist *ps);   // the abort word of the plan's persistent launches
bool node_persistent_interleave_env();
bool node_persistent_disabled_env();
bool node_persistent_supported(const ngpde_graph *g, int d, int act, bool with_bwd);
int node_persistent_mode(const ngpde_graph *g, int d, int act, bool with_bwd);   // 0 none, 1 one tile per workgroup, 2 tile pairs, 3 tile rounds
// graphs the modes above refuse because a tile does not fit the handle's halo lists (hubs): can the hub geometry be tried?  (Whether
// every tile fits ITS caps is found out by node_persistent_setup(..., hub = true), which returns NGPDE_ERR_UNSUPPORTED otherwise.)
bool node_persistent_hub_possible(const ngpde_graph *g, int d);
int node_persistent_rounds(const ngpde_graph *g);                               // mode 3: tiles per workgroup
int32_t node_persistent_setup(const ngpde_graph *g, const float *coef_host /* [90] */, NodePersist *ps, bool pair = false, bool hub = false);
void node_persistent_free(NodePersist *ps);
int32_t launch_node_fwd_persistent(const NodePersistFwd &a, hipStream_t stream);
int32_t launch_node_bwd_persistent(const NodePersistBwd &a, hipStream_t stream);
// Persistent launches of one process take turns per device (node_persistent.hip): enter() takes the device's turnstile lock and makes
// `stream` wait for the previous persistent launch on the device, leave() records this one and releases the lock; the lock is
// held from enter to leave, so two host threads cannot both pass the wait and then launch side by side.  The destructor releases
// a lock that an error path left behind.  On a stream that is being captured into a HIP graph the event wait / record is skipped
// (a captured event would poison later eager launches): ordering inside the captured graph is the capture's own.
struct PersistentTurn {
  int dev = -1;
  bool held = false;
  hipStream_t stream = nullptr;
  int32_t enter(hipStream_t s);
  int32_t leave();
  ~PersistentTurn();
  PersistentTurn() = default;
  PersistentTurn(const PersistentTurn &) = delete;
  PersistentTurn &operator=(const PersistentTurn &) = delete;
};
// the GAT-style layer (64 => heads x c = 64) as ODE right-hand side, device-resident (gat_fused.hip; plan: node_gat.hip)
struct GatNodeFwd {
  const ngpde_graph *g = nullptr;
  const NodePersist *ps = nullptr;   // wait lists, flags, abort / fault words (node_persistent_setup)
  int heads = 0, act = 0, n_steps = 0, S = 0;
  float slope = 0.2f;
  bool taped = false;
  const float *u_in = nullptr, *wt = nullptr, *a = nullptr, *bias = nullptr;
  float *u_out = nullptr, *xs = nullptr, *yz = nullptr, *alpha = nullptr, *kbuf = nullptr;
  const float *cf = nullptr;         // device table [(S + 1)][8]
  int n_members = 1;                 // > 1: block-diagonal batch of identical structures; arrays [members][...] with the strides below
  size_t xs_stride = 0, yz_stride = 0, alpha_stride = 0;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};
struct GatNodeBwd {
  const ngpde_graph *g = nullptr;
  const NodePersist *ps = nullptr;
  int heads = 0, act = 0, n_steps = 0, S = 0;
  float slope = 0.2f;
  const float *wt = nullptr, *a = nullptr, *xs = nullptr, *yz = nullptr, *alpha = nullptr, *duT = nullptr;
  float *lam = nullptr, *ubar = nullptr, *dzbuf = nullptr, *dscore = nullptr, *dal = nullptr;
  float *slab_db = nullptr, *slab_dw = nullptr, *slab_u = nullptr;
  const int *xpad = nullptr;
  const float *cb = nullptr;         // device table [S][8]
  float *dwt = nullptr, *da = nullptr, *db = nullptr;   // outputs (db nullable)
  int n_members = 1;
  size_t xs_stride = 0, yz_stride = 0, alpha_stride = 0;
  hipEvent_t ev_start = nullptr, ev_stop = nullptr;
};
bool gat_node_persistent_supported(const ngpde_graph *g, int heads, int c);
size_t gat_node_dscore_elems(const ngpde_graph *g);                       // floats of ONE padded dscore buffer
int32_t launch_gat_node_xpad(const ngpde_graph *g, int *pad_of_p, int *xpad, hipStream_t stream);
int32_t launch_gat_node_fwd(const GatNodeFwd &a, hipStream_t stream);
int32_t launch_gat_node_bwd(const GatNodeBwd &a, hipStream_t stream);
// NeuralODE(VMHConv(phi, gamma)) device-resident (node_vmh.hip; plan: node.hip)
constexpr int kVmhMaxL = 4;      // Dense layers per MLP
struct VmhShape {
  int hd = 1, pd = 2, aggr = 1, n_phi = 0, n_gam = 0;
  int phi_dims[kVmhMaxL + 1] = {0, 0, 0, 0, 0}, gam_dims[kVmhMaxL + 1] = {0, 0, 0, 0, 0};   // layer l: dims[l] => dims[l + 1]
  int phi_act[kVmhMaxL] = {0, 0, 0, 0}, gam_act[kVmhMaxL] = {0, 0, 0, 0};
};
struct VmhLaunch {
  const ngpde_graph *g = nullptr;
  const NodePersist *ps = nullptr;
  VmhShape shape;
  int n_steps = 0, S = 0;
  const float *pos = nullptr;
  const float *phi_w[kVmhMaxL] = {nullptr, nullptr, nullptr, nullptr}, *phi_b[kVmhMaxL] = {nullptr, nullptr, nullptr, nullptr};
  const float *gam_w[kVmhMaxL] = {nullptr, nullptr, nullptr, nullptr}, *gam_b[kVmhMaxL] = {nullptr, nullptr, nullptr, nullptr};
  const float *u_in = nullptr;
  float *u_out = nullptr, *x0 = nullptr, *x1 = nullptr, *tape_phi = nullptr, *tape_gam = nullptr;
  float *lam = nullptr, *dz_phi = nullptr, *dz_gam = nullptr, *dsrc = nullptr;   // dsrc: [2][E]
  const float *cf = nullptr, *cb = nullptr;
  // saveat (docs/src/tutorials/VMH.md:85): the state after every save_every steps goes to save[j][N], j = step / save_every - 1 + save_off
  // (save_off = 1: slot 0 holds u0); the adjoint adds dsave[j] to lambda at that time
  float *state = nullptr;        // tile rounds: [16][N] per-node state between a half tile's turns (forward 6 rows, adjoint 8)
  const int *srcpos = nullptr, *srcdeg = nullptr;   // launch_vmh_srcpos
  float *save = nullptr;
  const float *dsave = nullptr;
  int save_every = 0, save_off = 0;
};
bool node_vmh_supported(const ngpde_graph *g, const VmhShape &s);
int32_t launch_node_vmh_fwd(const VmhLaunch &a, hipStream_t stream);
int32_t launch_node_vmh_bwd(const VmhLaunch &a, hipStream_t stream);
int32_t launch_vmh_copy_block(const float *src, int sp, float *dst, int dp, int rows, int cols, hipStream_t stream);
// srcpos[r][kSlotWidth] / srcdeg[r] for every row r of the by-target schedule: the positions (in the by-target edge order) of the out-edges
// of the row's node -- the by-source gather's addresses, laid out by schedule row so that a tile reads them with one coalesced load
int32_t launch_vmh_srcpos(const ngpde_graph *g, int *srcpos, int *srcdeg, hipStream_t stream);
bool gat_fused_supported(const ngpde_graph *g, int heads, int c);
int32_t launch_gat_fused_fwd(const ngpde_graph *g, int heads, int c, float slope, const float *wx, const float *al, const float *ar,
                             float *out, float *alpha, hipStream_t stream);

// generic (any feature width) building blocks
int32_t launch_spmm_generic(const ngpde_graph *g, bool by_source, bool gcn_norm, int d, int aggr,
                            const float *x, const float *edge_weight, float *out, hipStream_t stream);
// gradient w.r.t. GCNConv's edge_weight argument (gcn_generic.hip); nd_scratch: n_nodes floats
int32_t launch_gcn_edge_weight_grad(const ngpde_graph *g, int d, const float *g3, const float *x3, const float *bias_or_null, const float *dxp,
                                    const float *xp, float *nd_scratch, float *dw, hipStream_t stream);
// dz = dy * act'(z)
int32_t launch_act_bwd(int64_t count, int act, const float *dy, const float *z, float *dz, hipStream_t stream);
// out[o] = sum_n a[n][o]
int32_t launch_colsum(int64_t n, int d, const float *a, float *out, hipStream_t stream);

// ---- message-passing primitives (mp_kernels.hip) -------------------------------------------------------
struct SegTable {   // virtual concatenation [X1 | X2 | ...] of up to 4 row-major blocks
  int n = 0;
  const float *ptr[4] = {nullptr, nullptr, nullptr, nullptr};
  int width[4] = {0, 0, 0, 0};
  int row_div[4] = {1, 1, 1, 1};   // block row = row / row_div (per-graph features repeated over a graph's rows)
  int offset[5] = {0, 0, 0, 0, 0};
  int vec[4] = {0, 0, 0, 0};       // block may be read with 16-byte loads (offset, width multiples of 4, base aligned)
};
struct SegGrad {
  int n = 0;
  float *ptr[4] = {nullptr, nullptr, nullptr, nullptr};
  int width[4] = {0, 0, 0, 0};
  int offset[5] = {0, 0, 0, 0, 0};
};
int32_t launch_dense_multi_fwd(int count, const int64_t *n, const SegTable *segs, const int *din, const int *dout, const int *act,
                               const float *const *wt, const float *const *bias, float *const *y, float *const *save_z, hipStream_t stream);
int32_t launch_dense_seg_fwd(int64_t n, const SegTable &segs, int din, int dout, int act, const float *wt,
                             const float *bias, float *y, float *save_z, hipStream_t stream);
int dense_fwd_splits(int64_t n, int din, int dout);
// two Dense layers in one streaming launch (dense_mfma.hip): two outputs from one 64-wide input block / a two-layer chain
bool dense_pair_fwd_applicable(int64_t n, const SegTable &ta, int dina, int douta, const SegTable &tb, int dinb, int doutb);
int32_t launch_dense_pair_fwd(int64_t n, const SegTable &ta, int dina, int douta, int acta, const float *wta, const float *ba, float *ya,
                              float *za, const SegTable &tb, int dinb, int doutb, int actb, const float *wtb, const float *bb, float *yb,
                              float *zb, hipStream_t stream);
bool dense_chain_fwd_applicable(int64_t n, const SegTable &t1, int din1, int dmid, int dout2);
int32_t launch_dense_chain_fwd(int64_t n, const SegTable &t1, int din1, int act1, const float *wt1, const float *b1, float *a1, float *z1,
                               int dout2, int act2, const float *wt2, const float *b2, float *y, float *z2, hipStream_t stream);
int dense_fwd_split_count(int din, int nsplit);
int32_t launch_dense_seg_fwd_splitk(int64_t n, const SegTable &segs, int din, int dout, const float *wt, float *partial, int nsplit,
                                    hipStream_t stream);
int32_t launch_sum_partials(int64_t count, int nparts, size_t stride, float *x, hipStream_t stream);
int32_t launch_spmm_gcn_tail(const ngpde_graph *g, int d, const float *x, const float *bias, int act, float *out, float *save_z,
                             hipStream_t stream);
int32_t launch_dense_dz(int64_t count, int act, const float *dy, const float *z, float *dz, hipStream_t stream);
int32_t launch_dense_seg_bwd_input(int64_t n, const SegGrad &segs, int din, int dout, const float *dz, const float *wt,
                                   hipStream_t stream);
int dense_weight_chunks(int64_t n, int din, int dout);
// dense_stream_bwd.hip: the whole Dense pullback (dz, input and weight pullbacks, bias) as one streaming launch; grid 0 = not applicable
int dense_stream_bwd_grid(int64_t n, const SegTable &t, int din, int dout, float *const *dseg);
int dense_pair_bwd_grid(int64_t n, const SegTable &ta, int dina, const SegTable &tb, int dinb);
size_t dense_pair_bwd_workspace(int grid, int dina, int dinb);
int32_t launch_dense_pair_bwd(int64_t n, const SegTable &ta, int dina, const float *wta, const float *dya, float *dwta, float *dba,
                              const SegTable &tb, int dinb, const float *wtb, const float *dyb, float *dwtb, float *dbb, float *dx,
                              const float *dx_add, void *workspace, int grid, hipStream_t stream);
int32_t launch_dense_stream_bwd(int64_t n, const SegTable &t, int din, int act, const float *wt, const float *z, const float *dy,
                                float *const *dseg, float *dwt, float *dbias, float *slabs, int grid, hipStream_t stream);
int dense_bwd_input_splits(int64_t n, int din, int dout);
size_t dense_bwd_input_split_bytes(int64_t n, int din, int dout);
int32_t launch_dense_bwd_input_splitk(int64_t n, float *dx, int din, int dout, const float *dz, const float *wt, float *part,
                                      hipStream_t stream);
int32_t launch_dense_seg_bwd_weight(int64_t n, const SegTable &segs, int din, int dout, const float *dz, float *dwt,
                                    float *db, float *partial, hipStream_t stream);
int32_t launch_edge_permute(const ngpde_graph *g, int d, bool inverse, const float *src, float *dst, hipStream_t stream);
int32_t launch_edge_combine_fwd(const ngpde_graph *g, int h, int act, const float *P, const float *Q, const float *Eterm,
                                float *a_out, float *z_out, hipStream_t stream);
int32_t launch_edge_combine_bwd(const ngpde_graph *g, int h, int act, const float *da, const float *z, float *dz, float *dP,
                                float *dQ, hipStream_t stream);
int32_t launch_edge_sum_by_source(const ngpde_graph *g, int h, const float *per_edge, float *out, hipStream_t stream);
int32_t launch_segment_reduce_fwd(const ngpde_graph *g, int d, int aggr, const float *M, float *out, hipStream_t stream);
int32_t launch_segment_reduce_bwd(const ngpde_graph *g, int d, int aggr, const float *M, const float *out, const float *dout,
                                  float *dM, hipStream_t stream);
int32_t launch_gno_contract_fwd(const ngpde_graph *g, int cin, int cout, const float *K, const float *h, float *m,
                                hipStream_t stream);
int32_t launch_gno_contract_bwd(const ngpde_graph *g, int cin, int cout, const float *K, const float *h, const float *dm,
                                float *dK, float *dhe, hipStream_t stream);
bool gno_apply_supported(int cout, int kdim);
int32_t launch_gno_apply_fwd(const ngpde_graph *g, int cout, int kdim, const float *T, const float *Bh, const float *z,
                             float *m, hipStream_t stream);
int32_t launch_gno_apply_bwd(const ngpde_graph *g, int cout, int kdim, const float *T, const float *z, const float *dm,
                             float *dT, float *dBh, float *dz, hipStream_t stream);
bool gno_apply_mfma_supported(int cout, int kdim);   // gno_mfma.hip: the same message on the matrix pipe
int32_t launch_gno_apply_mfma_fwd(const ngpde_graph *g, int cout, int kdim, const float *T, const float *Bh, const float *z, float *m,
                                  hipStream_t stream);
int32_t launch_gno_message_mfma_fwd(const ngpde_graph *g, int cout, int kdim, int act1, const float *P, const float *Q, const float *E,
                                    const float *T, const float *Bh, float *z_out, float *m, hipStream_t stream);
// dagg != NULL: dm is not read; dm_e = dagg[t_e] (* 1 / deg(t_e) if mean) is formed in the launch, dz leaves multiplied by
// act1'(z) (identity / relu on the activated z) and dq [N][k] (nullable) = its sums by source  (gno_mfma.hip, GnoNodeGrad)
int32_t launch_gno_apply_mfma_bwd(const ngpde_graph *g, int cout, int kdim, const float *T, const float *z, const float *dm, float *dT,
                                  float *dBh, float *dz, hipStream_t stream, const float *dagg = nullptr, int mean = 0, int act1 = 0,
                                  float *dq = nullptr);
int32_t launch_gat_scores(int64_t n, int heads, int c, const float *wx, const float *a, float *al, float *ar,
                          hipStream_t stream);
int32_t launch_gat_fwd(const ngpde_graph *g, int heads, int c, float slope, const float *wx, const float *al, const float *ar,
                       float *out, float *alpha, hipStream_t stream);
int32_t launch_gat_bwd(const ngpde_graph *g, int heads, int c, float slope, const float *wx, const float *a, const float *al,
                       const float *ar, const float *alpha, const float *dout, float *dscore, float *dal, float *dar,
                       float *dwx, float *da, hipStream_t stream);
int32_t launch_spectral_weights(int64_t n_edges, float nn, const float *e, float *w, hipStream_t stream);
// ---- the whole GAT-style layer (gat_fused.hip)
bool gat_layer_fused_supported(const ngpde_graph *g, int din, int heads, int c);
size_t gat_layer_workspace_bytes(const ngpde_graph *g, int heads);
int32_t launch_gat_layer_fwd(const ngpde_graph *g, int heads, float slope, int act, const float *x, const float *wt, const float *a,
                             const float *bias, float *y, float *alpha, float *save_z, hipStream_t stream);
int32_t launch_gat_layer_bwd(const ngpde_graph *g, int heads, float slope, int act, const float *x, const float *wt, const float *a,
                             const float *yz, const float *alpha, const float *dy, float *dx, float *dwt, float *da, float *db,
                             void *workspace, size_t workspace_bytes, hipStream_t stream);
constexpr int kColsumChunks = 128;
int32_t launch_colsum2(int64_t n, int d, const float *a, float *partial, float *out, hipStream_t stream);
// y = act(a + addend + bias) (addend, bias nullable), 16-byte accesses when d % 4 == 0
int32_t launch_bias_act2(int64_t n, int d, int act, const float *a, const float *addend, const float *bias, float *y, float *save_z,
                         hipStream_t stream);

// ---- fused edge-MLP forward (edge_mlp_fused.hip) ----------------------------------------------------------
struct EdgeMlpArgs {
  int h1 = 0, act1 = 0, aggr = 0, n_tail = 0;
  const float *P = nullptr, *Q = nullptr, *Eterm = nullptr;
  int din[3] = {0, 0, 0}, dout[3] = {0, 0, 0}, act[3] = {0, 0, 0};
  const float *wt[3] = {nullptr, nullptr, nullptr}, *bias[3] = {nullptr, nullptr, nullptr};
  float *out = nullptr;
  float *save_z[4] = {nullptr, nullptr, nullptr, nullptr};
};
struct EdgeMlpBwdArgs {
  int h1 = 0, act1 = 0, aggr = 0, n_tail = 0, dw = 0, act2 = 0;
  const float *P = nullptr, *Q = nullptr, *Eterm = nullptr, *wt = nullptr, *bias = nullptr, *dout = nullptr;
  float *dP = nullptr, *dQ = nullptr, *dE = nullptr, *dwt = nullptr, *dbias = nullptr;
  void *workspace = nullptr;
  size_t workspace_bytes = 0;
};
// edge_mlp_deep_bwd.hip: the same pullback for 2 or 3 Dense layers after the first (message MLPs of 3 / 4 layers)
struct EdgeMlpDeepBwdArgs {
  int h1 = 0, act1 = 0, aggr = 0, n_tail = 0;
  int dout[3] = {0, 0, 0}, act[3] = {0, 0, 0};
  const float *P = nullptr, *Q = nullptr, *Eterm = nullptr, *dout_grad = nullptr;
  const float *wt[3] = {nullptr, nullptr, nullptr}, *bias[3] = {nullptr, nullptr, nullptr};
  float *dP = nullptr, *dQ = nullptr, *dE = nullptr;
  float *dwt[3] = {nullptr, nullptr, nullptr}, *dbias[3] = {nullptr, nullptr, nullptr};
  void *workspace = nullptr;
  size_t workspace_bytes = 0;
};
bool edge_mlp_deep_bwd_supported(const ngpde_graph *g, int h1, int n_tail, const int *dout, int aggr);
size_t edge_mlp_deep_bwd_workspace(const ngpde_graph *g, int h1, int n_tail, const int *dout);
int32_t launch_edge_mlp_deep_bwd(const ngpde_graph *g, const EdgeMlpDeepBwdArgs &a, hipStream_t stream);
bool edge_mlp_fused_bwd_supported(const ngpde_graph *g, int h1, int n_tail, int dw, int aggr);
size_t edge_mlp_fused_bwd_workspace(const ngpde_graph *g, int h1, int n_tail, int dw);
int32_t launch_edge_mlp_fused_bwd(const ngpde_graph *g, const EdgeMlpBwdArgs &a, hipStream_t stream);
int32_t launch_dense_weight_reduce(int nchunk, int din, int dout, const float *partial, float *dwt, float *db, hipStream_t stream);
// the whole pullback of a Dense of at most 64 x 64 in one launch + the slab reduction (dense_small_bwd.hip)
int dense_small_bwd_grid(int64_t n, int din, int dout);
int dense_small_fwd_grid(int64_t n, int din, int dout);
int32_t launch_dense_small_fwd(int64_t n, const SegTable &t, int din, int dout, int act, const float *wt, const float *bias, float *y,
                               float *save_z, int grid, hipStream_t stream);
int32_t launch_dense_small_bwd(int64_t n, const SegTable &t, int din, int dout, int act, const float *wt, const float *z, const float *dy,
                               float *const *dseg, float *dwt, float *dbias, float *slabs, int grid, hipStream_t stream);
int32_t launch_edge_sum_by_source(const ngpde_graph *g, int h, const float *per_edge, float *out, hipStream_t stream);
bool edge_mlp_fused_supported(const ngpde_graph *g, const EdgeMlpArgs &a);
int32_t launch_edge_mlp_fused_fwd(const ngpde_graph *g, const EdgeMlpArgs &a, hipStream_t stream);
int32_t launch_activation_fwd(int64_t count, int act, const float *z, float *a, hipStream_t stream);
// edge_mlp64.hip: software-pipelined specialisations for the 64-wide two-layer message MLP (BASELINE config 4's shape)
bool edge_mlp64_fwd_applicable(const ngpde_graph *g, const EdgeMlpArgs &a);
int32_t launch_edge_mlp64_fwd(const ngpde_graph *g, const EdgeMlpArgs &a, hipStream_t stream);
bool edge_mlp64_bwd_applicable(const ngpde_graph *g, const EdgeMlpBwdArgs &a);
bool edge_mlp64_bwd_dq_in_launch(const ngpde_graph *g, const EdgeMlpBwdArgs &a);   // the by-source sum inside the launch: no [E][64] buffer needed
size_t edge_mlp64_bwd_workspace(const ngpde_graph *g);
int32_t launch_edge_mlp64_bwd(const ngpde_graph *g, const EdgeMlpBwdArgs &a, hipStream_t stream);


// ---- ROCTx ranges around the C-ABI entries (tracing).  rocprofv3 --marker-trace then attributes the
// launches of a trace to the layer call that made them.  The marker library (rocprofiler-sdk-roctx) is looked up lazily at the
// first entry and the ranges are no-ops when it is not there: the product library has no link-time dependency on a profiler.
struct RoctxApi {
  int (*push)(const char *) = nullptr;
  int (*pop)() = nullptr;
};
const RoctxApi &roctx_api();   // graph.hip
struct RoctxRange {
  bool on;
  explicit RoctxRange(const char *name) : on(roctx_api().push != nullptr) {
    if (on) roctx_api().push(name);
  }
  ~RoctxRange() {
    if (on) roctx_api().pop();
  }
  RoctxRange(const RoctxRange &) = delete;
  RoctxRange &operator=(const RoctxRange &) = delete;
};
#define NGPDE_RANGE() ::ngpde::RoctxRange ngpde_roctx_range_(__func__)

}  // namespace ngpde
