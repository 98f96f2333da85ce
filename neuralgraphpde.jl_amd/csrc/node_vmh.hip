// node_vmh.hip -- NeuralODE(VMHConv(phi, gamma)) device-resident: the fixed-step solve and its discrete adjoint as ONE persistent
// launch each (+ one weight-pullback GEMM per Dense layer behind the adjoint).
// [caller in the reference: docs/src/tutorials/VMH.md:75-89 (NeuralODE(VMHConv(phi, gamma), tspan, Tsit5(); ...)); layer:
//  src/layers.jl:308-332: m_i = mean_j phi([h_i; h_j - h_i; x_j - x_i]), h' = gamma([h_i; m_i])]
//
// The tutorial's graph is small (3 000 nodes, 18 000 edges) and its MLPs are deep (4 Dense layers each, 60 wide): through the
// generic solver a right-hand side + pullback is 38 dependent launches of one under-filled wave of workgroups each (~250 us).  Here:
//   * a workgroup (8 waves, one workgroup per CU) owns 16 target nodes -- half of one 32-row tile of the handle's
//     locality schedule -- and all their in-edges for the whole solve;
//   * every Dense layer of phi and gamma is held in LDS zero-padded to 64 x 64, UNPADDED in stride and XOR-swizzled (16 KB per
//     matrix: eight matrices are 128 KB; gcn_tile.h, mfma_rows_times_bswz64 has the read pattern);
//   * the message MLP runs per 16-edge wave slice as a chain of TRANSPOSED fp32 MFMA products in registers (edge_mlp_fused.hip),
//     the node MLP on the 16 rows with its four output-column blocks dealt to the four waves;
//   * tiles exchange ONE float per node per right-hand-side evaluation (the scalar state h): write-through stores, per-workgroup
//     phase flags, bounded spins, abort word (node_persistent.hip's protocol);
//   * the forward tapes every layer's INPUT rows; the adjoint reads them back (activation derivatives come from the taped
//     outputs: identity / relu / tanh / sigmoid), walks both MLPs backwards, tapes every layer's dz, and exchanges one float per
//     EDGE (the gradient towards the edge's source: the by-source sum crosses tiles) per evaluation;
//   * parameter gradients are NOT accumulated in the launch (eight 64 x 64 accumulators per workgroup fit neither registers nor
//     LDS): dW_l = A_l^T dZ_l is one large weight-pullback GEMM per layer over the tapes of all evaluations afterwards
//     (dense_mfma.hip), at full-chip efficiency.
//   * tape rows (layer inputs forward, dz rows in the adjoint) of a one-round half tile stay in registers and are stored ONE EVALUATION
//     LATE, layer by layer in front of the layer's MFMAs: no burst of ~90 KB per workgroup at the hand-off; the tape rows the adjoint
//     reads are asked for a phase (gamma, phi's last hidden layer) or a layer ahead;
//   * graphs of more half tiles than the device keeps resident -- the tutorial's minibatch is 24 clouds of 3 000 points as one graph,
//     VMH.md:120-134 -- run in TILE ROUNDS (template parameter ROUNDS): whole 32-row tiles, K per workgroup, walked in the same order in
//     every phase with the weights resident, the rows' Runge-Kutta state in memory between turns; the adjoint's sweep ph runs, per tile,
//     the second half of phase ph - 1 and at once the first half of phase ph.
// State width 1 (a scalar field, as in the tutorial), positions of 1-3 coordinates, MLPs of 2-4 layers up to 64 wide with
// identity output layers; up to kVmhMaxTurns tiles per resident workgroup.  Anything else keeps the generic solver.
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <tuple>

#include "common.h"
#include "device_utils.h"
#include "persistent_mem.h"

namespace ngpde {

namespace {

constexpr int VT = 512;               // threads per workgroup: 8 waves, two per SIMD (the tutorial graph's 96 edges per workgroup are one round)
constexpr int VROUND = (VT / 64) * 16;   // edges per round: one 16-edge slice per wave
constexpr int VW = 64;                // padded layer width
constexpr int VR = 16;                // target rows of a half tile (the unit of the one-tile form; tile rounds: whole tiles, 2 VR)
constexpr int VTS = VW + 4;           // staging tile stride
constexpr int VNbr = 64;
constexpr int kVmhMaxTurns = 64;       // tile rounds: tiles per workgroup (64 x 256 CUs x 32 rows = 524 288 nodes)

#ifdef NGPDE_STAMPS
// diagnostic build only (tools/stamps_vmh.py): shader-clock stamps of thread 0 at 8 points of the first g_vst_max phases
unsigned long long *g_vst_base = nullptr;
int g_vst_max = 0;
#define NGPDE_VST_FIELD unsigned long long *stamps; int stamps_max;
#define NGPDE_VST(m, ph, k)                                                                                   \
  do {                                                                                                        \
    if (threadIdx.x == 0 && (m).stamps && (ph) <= (m).stamps_max)                                             \
      (m).stamps[((size_t)blockIdx.x * (m).stamps_max + ((ph) - 1)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define NGPDE_VST_FIELD
#define NGPDE_VST(m, ph, k)
#endif

struct VmhMeta {
  const int4 *sched_t;
  const int2 *halo_t, *info_t;
  const uint8_t *slots_t;
  const int *rowptr_s, *xpos_s;
  const int *nbr;                    // [n_tiles][64] tile-level wait lists (node_persistent_setup)
  unsigned *flags, *abort_word;      // one 128-byte line per WORKGROUP (2 per tile)
  int n_tiles, n_nodes;
  const float *pos;                  // [N][pd]
  int pd, aggr;
  int n_phi, n_gam;
  int phi_din[kVmhMaxL], phi_dout[kVmhMaxL], phi_act[kVmhMaxL];
  int gam_din[kVmhMaxL], gam_dout[kVmhMaxL], gam_act[kVmhMaxL];
  const float *phi_w[kVmhMaxL], *phi_b[kVmhMaxL], *gam_w[kVmhMaxL], *gam_b[kVmhMaxL];
  size_t n_edges;
  const int *srcpos, *srcdeg;        // by schedule row: positions of the node's out-edges in the by-target order, their count (launch_vmh_srcpos)
  int evals;                         // right-hand-side evaluations of a solve: the tapes are [layer][evals][rows][64] (a layer's rows contiguous)
  int s_rows;                        // rows of the staging tile: 64, 96 or 128 (what the LDS left by the weights allows)
  NGPDE_VST_FIELD
};

// derivative of an activation from its OUTPUT y = act(z)
__device__ __forceinline__ float dact_out(int act, float y) {
  switch (act) {
    case NGPDE_ACT_RELU: return y > 0.f ? 1.0f : 0.0f;
    case NGPDE_ACT_TANH: return 1.0f - y * y;
    case NGPDE_ACT_SIGMOID: return y * (1.0f - y);
    default: return 1.0f;
  }
}
template <int ACT, int N>
__device__ __forceinline__ void dact_out_n(float4 (&y)[N]) {
#pragma unroll
  for (int i = 0; i < N; ++i) y[i] = make_float4(dact_out(ACT, y[i].x), dact_out(ACT, y[i].y), dact_out(ACT, y[i].z), dact_out(ACT, y[i].w));
}
template <int N>
__device__ __forceinline__ void f4n_dact_out(int act, float4 (&y)[N]) {   // y <- act'(.) from the outputs y, one uniform switch
  switch (act) {
    case NGPDE_ACT_RELU: dact_out_n<NGPDE_ACT_RELU>(y); break;
    case NGPDE_ACT_TANH: dact_out_n<NGPDE_ACT_TANH>(y); break;
    case NGPDE_ACT_SIGMOID: dact_out_n<NGPDE_ACT_SIGMOID>(y); break;
    default: dact_out_n<NGPDE_ACT_IDENTITY>(y); break;
  }
}

struct VCtx {
  int tid, lane, wave, g16, q, ei, kq;
  int wg, tile, half, hcount, total;
  bool row_valid;          // (threads of group g16: row g16 is a node)
  int node;                // node of row g16
};

// static tables of the workgroup in LDS
struct VTabs {
  float *W;                // [(n_phi + n_gam)][64][64] swizzled
  float *S;                // [64][VTS] staging
  float *bias;             // [8][64]
  float *hh;               // [96] h of the halo nodes
  float *px;               // [96][4] positions of the halo nodes
  int *hnode;              // [96]
  int *off;                // [17]
  int *rs;                 // [16]
  int *rnode;              // [16]
  float *inv;              // [16]
  unsigned short *edge;    // [rows x kSlotWidth]  r | slot << 8
  float *misc;             // [64]: coefficients etc.
  int *s_ok;
};

__device__ __forceinline__ size_t vtabs_dyn_floats(int n_mats) { return (size_t)n_mats * VW * VW + (size_t)VW * VTS; }

// W (row-major [din][dout]) -> LDS 64 x 64 zero-padded, row j / quad k4 at quad k4 ^ (j & 15).
// transpose = true: row j = OUTPUT j, columns = inputs (the forward's A operand rows: W^T); false: row j = INPUT j (W itself)
__device__ __forceinline__ void stage_weight(const float *w, int din, int dout, float *dst, int tid, bool transpose) {
  for (int idx = tid; idx < VW * VW / 4; idx += VT) {
    const int j = idx >> 4, k4 = idx & 15;
    float v[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k = 4 * k4 + r;
      if (transpose) v[r] = (w && k < din && j < dout) ? w[(size_t)k * dout + j] : 0.f;
      else v[r] = (w && j < din && k < dout) ? w[(size_t)j * dout + k] : 0.f;
    }
    *reinterpret_cast<float4 *>(&dst[j * VW + 4 * (k4 ^ (j & 15))]) = make_float4(v[0], v[1], v[2], v[3]);
  }
}

// rows (block*16 + ei) of a staged matrix, quad (4 cb + kq): the A-operand fragment of one transposed MFMA k-block
__device__ __forceinline__ float4 wfrag(const float *mat, int block, int cb, int ei, int kq) {
  return *reinterpret_cast<const float4 *>(&mat[(block * 16 + ei) * VW + 4 * ((4 * cb + kq) ^ ei)]);
}

// out[ob] += sum_ib (block (ob, ib) of a staged matrix) x in[ib] on a wave's 16 columns (edges or rows); out comes in holding the
// initial accumulators (the bias, or zeros).  The NOUT accumulator chains advance side by side -- a chain's MFMAs are NOUT issue slots
// apart, so none waits out the 40-cycle dependent latency -- and the fragments of the next input block are on their way from LDS while
// the MFMAs of this one issue.  Every chain adds its products in the same order as a plain loop over the inputs.  Blocks beyond a
// layer's real widths are zero in the staged matrix: running them changes nothing, skipping them (NOUT / NIN) saves their MFMAs.
template <int NOUT, int NIN>
__device__ __forceinline__ void slice_matmul(const float *mat, const float4 (&in)[4], f32x4 (&out)[4], int ei, int kq) {
  float4 w[NOUT], wn[NOUT];
#pragma unroll
  for (int ob = 0; ob < NOUT; ++ob) w[ob] = wfrag(mat, ob, 0, ei, kq);
#pragma unroll
  for (int ib = 0; ib < NIN; ++ib) {
    if (ib + 1 < NIN) {
#pragma unroll
      for (int ob = 0; ob < NOUT; ++ob) wn[ob] = wfrag(mat, ob, ib + 1, ei, kq);
    }
    __builtin_amdgcn_sched_barrier(0);   // (the scheduler would otherwise sink each read to its first use to save registers)
#pragma unroll
    for (int ob = 0; ob < NOUT; ++ob) out[ob] = mfma16(w[ob].x, in[ib].x, out[ob]);
#pragma unroll
    for (int ob = 0; ob < NOUT; ++ob) out[ob] = mfma16(w[ob].y, in[ib].y, out[ob]);
#pragma unroll
    for (int ob = 0; ob < NOUT; ++ob) out[ob] = mfma16(w[ob].z, in[ib].z, out[ob]);
#pragma unroll
    for (int ob = 0; ob < NOUT; ++ob) out[ob] = mfma16(w[ob].w, in[ib].w, out[ob]);
    __builtin_amdgcn_sched_barrier(0);
    if (ib + 1 < NIN) {
#pragma unroll
      for (int ob = 0; ob < NOUT; ++ob) w[ob] = wn[ob];
    }
  }
}

// the context of unit h: half tile 2 tile + half (VRT = 16: the workgroup's only one), or whole tile h (VRT = 32: the one whose turn it
// is in the tile rounds -- twelve 16-edge slices at degree 6, three per SIMD, where a half tile's six leave two SIMDs half idle)
template <int VRT>
__device__ __forceinline__ void vctx_init(const VmhMeta &m, VCtx &c, const VTabs &t, int h) {
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  c.g16 = c.tid >> 4;
  c.q = c.tid & 15;
  c.ei = c.lane & 15;
  c.kq = c.lane >> 4;
  c.wg = VRT == 32 ? 2 * h : h;   // (the unit's first flag line)
  c.tile = c.wg >> 1;
  c.half = c.wg & 1;
  const int4 sc = m.sched_t[(size_t)c.tile * kTileRows + c.half * VR + min(c.g16, VRT - 1)];   // (lane groups beyond VRT have no row)
  c.row_valid = sc.x >= 0 && c.g16 < VRT;
  c.node = max(sc.x, 0);
  c.hcount = __builtin_amdgcn_readfirstlane(m.info_t[c.tile].x);
  if (c.q == 0 && c.g16 < VRT) {
    const int d = sc.x >= 0 ? sc.z : 0;
    t.off[c.g16 + 1] = d;
    t.rs[c.g16] = sc.y;
    t.rnode[c.g16] = sc.x;
    t.inv[c.g16] = m.aggr == NGPDE_AGGR_MEAN ? (d > 0 ? 1.0f / (float)d : 0.f) : 1.0f;
    if (c.g16 == 0) t.off[0] = 0;
  }
  if (c.tid < kHaloCap) {
    const int nd = m.halo_t[(size_t)c.tile * kHaloCap + c.tid].x;
    t.hnode[c.tid] = nd;
    for (int k = 0; k < 4; ++k) t.px[c.tid * 4 + k] = (c.tid < c.hcount && k < m.pd) ? m.pos[(size_t)nd * m.pd + k] : 0.f;
    t.hh[c.tid] = 0.f;
  }
  if (c.tid == 0) *t.s_ok = 1;
  __syncthreads();
  if (c.tid < VRT) {
    int v = t.off[c.tid + 1];
#pragma unroll
    for (int o = 1; o < VRT; o <<= 1) {
      const int u = __shfl_up(v, o);
      if (c.tid >= o) v += u;
    }
    t.off[c.tid + 1] = v;
  }
  __syncthreads();
  c.total = t.off[VRT];
  if (c.g16 < VRT) {   // edge table: k -> (row, halo slot of the source)
    const int lo = t.off[c.g16], hi = t.off[c.g16 + 1];
    const uint8_t *sl = m.slots_t + ((size_t)c.tile * kTileRows + c.half * VR + c.g16) * kSlotWidth;
    for (int k = lo + c.q; k < hi; k += 16) t.edge[k] = (unsigned short)(c.g16 | ((unsigned)sl[k - lo] << 8));
  }
  __syncthreads();
}

// TILE ROUNDS: what a unit's tables are built from, asked for a turn ahead (registers; the loads travel under the turn before):
// part A needs only the unit's index, part B (positions of the halo nodes, the rows' state) needs what A brought.
struct VPre {
  int4 sc;            // schedule row of lane group g16
  int hcount, hnode;  // rows staged; halo node of thread tid < kHaloCap
  unsigned slots2;    // slot bytes q and q + 16 of row g16
  int sp0, sp1, sdeg; // (adjoint) by-source positions q, q + 16 and their count of row g16
  int rnode;          // node of row tid (tid < VRT)
  float px[3];
  float st[8];        // the rows' state (forward: u, k_0 .. k_4; adjoint: lambda, U-bar_0 .. 5, own sum)
};
template <int VRT>
__device__ __forceinline__ void vctx_fetch_a(const VmhMeta &m, int h, VPre &pre, bool adjoint) {
  const int tid = threadIdx.x, g16 = tid >> 4, q = tid & 15;
  const int line = VRT == 32 ? 2 * h : h, tile = line >> 1, half = line & 1;
  const size_t row = (size_t)tile * kTileRows + half * VR + min(g16, VRT - 1);
  pre.sc = m.sched_t[row];
  pre.hcount = m.info_t[tile].x;
  pre.hnode = tid < kHaloCap ? m.halo_t[(size_t)tile * kHaloCap + tid].x : 0;
  const uint8_t *sl = m.slots_t + row * kSlotWidth;
  pre.slots2 = (unsigned)sl[q] | ((unsigned)sl[q + 16] << 8);
  pre.rnode = tid < VRT ? m.sched_t[(size_t)tile * kTileRows + half * VR + tid].x : -1;
  pre.sp0 = pre.sp1 = pre.sdeg = 0;
  if (adjoint) {
    pre.sp0 = m.srcpos[row * kSlotWidth + q];
    pre.sp1 = m.srcpos[row * kSlotWidth + q + 16];
    pre.sdeg = m.srcdeg[row];
  }
}
__device__ __forceinline__ void vctx_fetch_px(const VmhMeta &m, VPre &pre) {
  const int tid = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 3; ++k) pre.px[k] = (tid < pre.hcount && tid < kHaloCap && k < m.pd) ? m.pos[(size_t)pre.hnode * m.pd + k] : 0.f;
}
// the tables of unit h from what vctx_fetch_a / _px brought: vctx_init without its loads
template <int VRT>
__device__ __forceinline__ void vctx_commit(const VmhMeta &m, VCtx &c, const VTabs &t, int h, const VPre &pre) {
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.wave = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  c.g16 = c.tid >> 4;
  c.q = c.tid & 15;
  c.ei = c.lane & 15;
  c.kq = c.lane >> 4;
  c.wg = VRT == 32 ? 2 * h : h;
  c.tile = c.wg >> 1;
  c.half = c.wg & 1;
  const int4 sc = pre.sc;
  c.row_valid = sc.x >= 0 && c.g16 < VRT;
  c.node = max(sc.x, 0);
  c.hcount = __builtin_amdgcn_readfirstlane(pre.hcount);
  const int d = sc.x >= 0 ? sc.z : 0;
  if (c.q == 0 && c.g16 < VRT) {
    t.off[c.g16 + 1] = d;
    t.rs[c.g16] = sc.y;
    t.rnode[c.g16] = sc.x;
    t.inv[c.g16] = m.aggr == NGPDE_AGGR_MEAN ? (d > 0 ? 1.0f / (float)d : 0.f) : 1.0f;
    if (c.g16 == 0) t.off[0] = 0;
  }
  if (c.tid < kHaloCap) {
    t.hnode[c.tid] = pre.hnode;
    for (int k = 0; k < 4; ++k) t.px[c.tid * 4 + k] = k < 3 ? pre.px[k] : 0.f;
    t.hh[c.tid] = 0.f;
  }
  if (c.tid == 0) *t.s_ok = 1;
  __syncthreads();
  if (c.tid < VRT) {
    int v = t.off[c.tid + 1];
#pragma unroll
    for (int o = 1; o < VRT; o <<= 1) {
      const int u = __shfl_up(v, o);
      if (c.tid >= o) v += u;
    }
    t.off[c.tid + 1] = v;
  }
  __syncthreads();
  c.total = t.off[VRT];
  if (c.g16 < VRT) {   // edge table: k -> (row, halo slot of the source); a row has at most kSlotWidth = 32 entries, two per lane
    const int lo = t.off[c.g16];
    if (c.q < d) t.edge[lo + c.q] = (unsigned short)(c.g16 | ((pre.slots2 & 0xffu) << 8));
    if (c.q + 16 < d) t.edge[lo + c.q + 16] = (unsigned short)(c.g16 | ((pre.slots2 >> 8) << 8));
  }
  __syncthreads();
}

// wave 0 polls: lane l < 63 watches both halves of tile nbr[l], lane 0 also this tile's other half, lane 63 the abort word
__device__ __forceinline__ bool vmh_wait(const VmhMeta &m, const VCtx &c, int need, int *s_ok) {
  if (need <= 0) return true;
  if (c.wave == 0) {
    const int nb = c.lane < 63 ? m.nbr[(size_t)c.tile * VNbr + c.lane] : -1;
    const unsigned *a0 = (c.lane == 63) ? m.abort_word : (nb >= 0 ? m.flags + 32 * (2 * nb) : nullptr);
    const unsigned *a1 = (c.lane < 63 && nb >= 0) ? m.flags + 32 * (2 * nb + 1) : nullptr;
    const unsigned *a2 = (c.lane == 0) ? m.flags + 32 * (c.wg ^ 1) : nullptr;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    for (unsigned it = 1;; ++it) {
      unsigned f0 = (c.lane == 63) ? 0u : (unsigned)need, f1 = (unsigned)need, f2 = (unsigned)need;
      if (a0) f0 = __hip_atomic_load(a0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a1) f1 = __hip_atomic_load(a1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (a2) f2 = __hip_atomic_load(a2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__any((int)(c.lane == 63 && f0 != 0))) { ok = false; break; }
      if (__all((int)(c.lane == 63 || (f0 >= (unsigned)need && f1 >= (unsigned)need && f2 >= (unsigned)need)))) break;
      if ((it & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {
        if (c.lane == 0) __hip_atomic_store(m.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (c.lane == 0) *s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return *s_ok != 0;
}
__device__ __forceinline__ void vmh_publish(const VmhMeta &m, const VCtx &c, int ph, bool whole_tile) {
  wait_vmcnt0();
  __syncthreads();
  if (c.tid == 0) __hip_atomic_store(m.flags + 32 * c.wg, (unsigned)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (whole_tile && c.tid == 1) __hip_atomic_store(m.flags + 32 * (c.wg + 1), (unsigned)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ float4 ld4_nt(const float *p) {   // a tape row: read once
  const f4v v = __builtin_nontemporal_load(reinterpret_cast<const NGPDE_GLOBAL_AS f4v *>(reinterpret_cast<uintptr_t>(p)));
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ float ld_sc1(const float *p) {
  return __uint_as_float(__hip_atomic_load(reinterpret_cast<const unsigned *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
}
__device__ __forceinline__ void st_sc1(float *p, float v) {
  __hip_atomic_store(reinterpret_cast<unsigned *>(p), __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------------
// forward solve
// ---------------------------------------------------------------------------------------------------------------------
struct VmhFwdK {
  VmhMeta m;
  int n_steps, S;
  const float *u_in;
  float *u_out, *x0, *x1;     // the exchanged stage input [N], ping-pong
  float *save;                // saveat: [T][N] or null
  int save_every, save_off;
  float *state;               // tile rounds: [6][N] u and k_0 .. k_4 of the own nodes between a half tile's turns
  float *tape_phi;            // [n_phi][evals][E][64] inputs of phi's layers (p order), or null (forward-only plan)
  float *tape_gam;            // [n_gam][evals][N][64] inputs of gamma's layers
  const float *cf;            // [42] forward coefficient table (node_persistent.hip's layout)
};

template <bool ROUNDS>
__global__ __launch_bounds__(VT, 1) void node_vmh_fwd_kernel(const VmhFwdK p) {
  constexpr int VRT = ROUNDS ? 2 * VR : VR;      // rows of a unit: a half tile, or (tile rounds) a whole tile
  constexpr int VMaxET = VRT * kSlotWidth;
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  __shared__ __attribute__((aligned(16))) float s_bias[2 * kVmhMaxL * VW];
  __shared__ float s_hh[kHaloCap], s_px[kHaloCap * 4], s_inv[VRT], s_misc[64];
  __shared__ int s_hnode[kHaloCap], s_off[VRT + 1], s_rs[VRT], s_rnode[VRT], s_okw[2];
  __shared__ unsigned short s_edge[VMaxET];
  const VmhMeta &m = p.m;
  const int n_mats = m.n_phi + m.n_gam;
  VTabs t;
  t.W = dyn; t.S = dyn + (size_t)n_mats * VW * VW; t.bias = s_bias; t.hh = s_hh; t.px = s_px; t.hnode = s_hnode; t.off = s_off; t.rs = s_rs;
  t.rnode = s_rnode; t.inv = s_inv; t.edge = s_edge; t.misc = s_misc; t.s_ok = s_okw;
  VCtx c;
  // TILE ROUNDS: a graph of more half tiles than the device keeps resident gives every workgroup K of them -- b, b + G, b + 2 G, ... --
  // which it walks in that order in every phase; the weights stay in LDS for all of them, a half tile's tables are rebuilt at its
  // turn and the Runge-Kutta state of its rows waits in memory.  By the time a half tile's turn comes again its neighbours' rows of
  // the previous phase have long arrived: the hand-off that bounds the one-tile form costs nothing here.
  // (ROUNDS is a template parameter: the one-half-tile form keeps its register allocation, the rounds form carries no rows across phases)
  const int nh = ROUNDS ? m.n_tiles : 2 * m.n_tiles, G = gridDim.x, K = ROUNDS ? (nh + G - 1) / G : 1;
  vctx_init<VRT>(m, c, t, blockIdx.x);
  // two waves share a SIMD (waves w and w + 4): the first runs at high priority, so the pair does not march in lockstep through
  // MFMA chain and activation code alike -- the second fills the matrix pipe while the first is in its VALU stretches
  if (c.wave < 4) __builtin_amdgcn_s_setprio(3);
  for (int l = 0; l < m.n_phi; ++l) stage_weight(m.phi_w[l], m.phi_din[l], m.phi_dout[l], t.W + (size_t)l * VW * VW, c.tid, true);
  for (int l = 0; l < m.n_gam; ++l) stage_weight(m.gam_w[l], m.gam_din[l], m.gam_dout[l], t.W + (size_t)(m.n_phi + l) * VW * VW, c.tid, true);
  if (c.tid < VW) {
    for (int l = 0; l < m.n_phi; ++l) t.bias[l * VW + c.tid] = (m.phi_b[l] && c.tid < m.phi_dout[l]) ? m.phi_b[l][c.tid] : 0.f;
    for (int l = 0; l < m.n_gam; ++l) t.bias[(kVmhMaxL + l) * VW + c.tid] = (m.gam_b[l] && c.tid < m.gam_dout[l]) ? m.gam_b[l][c.tid] : 0.f;
  }
  if (c.tid < 42) t.misc[c.tid] = p.cf[c.tid];
  __syncthreads();
  const int ei = c.ei, kq = c.kq;
  int n_rounds = (c.total + VROUND - 1) / VROUND;
  const size_t E = m.n_edges, N = (size_t)m.n_nodes;
  const int rg = min(c.g16, VRT - 1);   // the row of this lane group (groups 16.. idle in the row-wise steps)
  const bool has_row = c.g16 < VRT;
  // the 16 row lanes (tid < 16 <-> row tid) keep the Runge-Kutta state of their node
  float su = 0.f, sk0 = 0.f, sk1 = 0.f, sk2 = 0.f, sk3 = 0.f, sk4 = 0.f;
  int my_node = c.tid < VRT ? t.rnode[c.tid] : -1;
  if (my_node >= 0) su = p.u_in[my_node];
  const int last_ph = p.n_steps * p.S;
  VPre pre;          // (tile rounds) the next turn's tables and state, on their way
  int pre_h = -1, pre_ph = 0;
  // With ONE round (the usual case) the tape rows of a slice stay in registers and leave one evaluation LATE: layer l's rows of the
  // previous evaluation are stored right before this evaluation's overwrite them -- four stores per lane in front of every layer's 64
  // MFMAs instead of ~100 KB per workgroup in one burst, and the drain in front of the flag waits for the 16 state values alone
  const bool defer = !ROUNDS && p.tape_phi != nullptr && n_rounds == 1;
  float4 ta[kVmhMaxL][4];
  size_t pe_keep = 0, ev_prev = 0;
  bool valid_keep = false, have_prev = false;
  auto store_tape_rows = [&](int l, size_t evx) {
    const int n_ct = (m.phi_din[l] + 15) >> 4;
    float *row = p.tape_phi + (((size_t)l * m.evals + evx) * E + pe_keep) * VW + 4 * kq;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
      if (ct < n_ct) __builtin_nontemporal_store((f4v){ta[l][ct].x, ta[l][ct].y, ta[l][ct].z, ta[l][ct].w}, reinterpret_cast<NGPDE_GLOBAL_AS f4v *>(reinterpret_cast<uintptr_t>(row + 16 * ct)));
  };
  bool ok = true;
  int ph = 0;
  for (int n = 0; n < p.n_steps && ok; ++n) {
    for (int i = 0; i < p.S && ok; ++i) {
      ++ph;
      const float *X = ph == 1 ? p.u_in : (((ph - 1) & 1) ? p.x1 : p.x0);
      float *Xn = (ph & 1) ? p.x1 : p.x0;
      const size_t ev = (size_t)(n * p.S + i);
      for (int s = 0; s < K; ++s) {
      if constexpr (ROUNDS) {   // this turn's tile: tables, rows' state -- from what the turn before asked for, when there was one
        const int h = blockIdx.x + s * G;
        if (h >= nh) break;
        if (pre_h == h) {
          vctx_commit<VRT>(m, c, t, h, pre);
          my_node = c.tid < VRT ? pre.rnode : -1;
          su = pre.st[0]; sk0 = pre.st[1]; sk1 = pre.st[2]; sk2 = pre.st[3]; sk3 = pre.st[4]; sk4 = pre.st[5];
        } else {
          vctx_init<VRT>(m, c, t, h);
          my_node = c.tid < VRT ? t.rnode[c.tid] : -1;
          if (my_node >= 0) {
            su = ph == 1 ? p.u_in[my_node] : p.state[my_node];
            sk0 = ph == 1 ? 0.f : p.state[N + my_node]; sk1 = ph == 1 ? 0.f : p.state[2 * N + my_node]; sk2 = ph == 1 ? 0.f : p.state[3 * N + my_node];
            sk3 = ph == 1 ? 0.f : p.state[4 * N + my_node]; sk4 = ph == 1 ? 0.f : p.state[5 * N + my_node];
          }
        }
        n_rounds = (c.total + VROUND - 1) / VROUND;
        // the next turn's unit (this sweep's next tile, or the first one of the next sweep): part A of its tables is asked for now
        int hn = h + G;
        pre_ph = ph;
        if (hn >= nh) { hn = blockIdx.x; pre_ph = ph + 1; }
        pre_h = (hn != h && pre_ph <= last_ph) ? hn : -1;      // (a workgroup's only tile keeps its state in registers: nothing to fetch)
        if (pre_h >= 0) vctx_fetch_a<VRT>(m, pre_h, pre, false);
      }
      NGPDE_VST(m, ph, 0);
      if (!vmh_wait(m, c, ph - 1, t.s_ok)) { ok = false; break; }
      NGPDE_VST(m, ph, 1);
      if (c.tid < c.hcount) t.hh[c.tid] = ph == 1 ? X[t.hnode[c.tid]] : ld_sc1(X + t.hnode[c.tid]);
      __syncthreads();
      NGPDE_VST(m, ph, 2);
      // ---- message MLP per 16-edge wave slice, messages summed per target through the staging tile
      float4 racc = f4_zero();
      const int lo = t.off[rg], hi = has_row ? t.off[rg + 1] : t.off[rg];
      for (int rd = 0; rd < n_rounds; ++rd) {
        const int c0 = rd * VROUND;
        const bool wave_on = c0 + c.wave * 16 < c.total;   // wave-uniform
        const int k = c0 + c.wave * 16 + ei;
        const bool valid = k < c.total;
        float4 msg[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
        if (wave_on) {
          const unsigned ew = t.edge[valid ? k : 0];
          const int r = ew & 0xff, slot = ew >> 8, trow = c.half * VR + r;
          const size_t pe = (size_t)(t.rs[r] + (k - t.off[r]));
          const float hi_ = t.hh[trow], hj = t.hh[slot];
          float feat[8] = {hi_, hj - hi_, t.px[slot * 4] - t.px[trow * 4], t.px[slot * 4 + 1] - t.px[trow * 4 + 1],
                           t.px[slot * 4 + 2] - t.px[trow * 4 + 2], 0.f, 0.f, 0.f};
          if (m.pd < 3) feat[4] = 0.f;
          if (m.pd < 2) feat[3] = 0.f;
          float4 a[4];
          a[0] = valid ? (kq == 0 ? make_float4(feat[0], feat[1], feat[2], feat[3]) : (kq == 1 ? make_float4(feat[4], 0.f, 0.f, 0.f) : f4_zero()))
                       : f4_zero();
          a[1] = a[2] = a[3] = f4_zero();
          // (the layer loop is unrolled: the kept rows have static register indices and a layer's shape words are loaded once)
#pragma unroll
          for (int l = 0; l < kVmhMaxL; ++l) {
            if (l >= m.n_phi) break;
            const int din = m.phi_din[l], dw = m.phi_dout[l];
            const int n_ct = (din + 15) >> 4, n_mt = (dw + 15) >> 4;   // uniform
            if (defer) {
              if (have_prev && valid_keep) store_tape_rows(l, ev_prev);
#pragma unroll
              for (int ct = 0; ct < 4; ++ct) ta[l][ct] = a[ct];
              if (l + 1 == m.n_phi) {   // (a lane's edge is the same in every evaluation of a one-round solve)
                pe_keep = pe;
                valid_keep = valid;
              }
            } else if (p.tape_phi && valid) {
              float *row = p.tape_phi + (((size_t)l * m.evals + ev) * E + pe) * VW + 4 * kq;
#pragma unroll
              for (int ct = 0; ct < 4; ++ct)
                if (ct < n_ct) __builtin_nontemporal_store((f4v){a[ct].x, a[ct].y, a[ct].z, a[ct].w}, reinterpret_cast<NGPDE_GLOBAL_AS f4v *>(reinterpret_cast<uintptr_t>(row + 16 * ct)));
            }
            const float *mat = t.W + (size_t)l * VW * VW;
            f32x4 acc[4];   // (the accumulators start from the bias: z = b + W a)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
              const float4 b4 = *reinterpret_cast<const float4 *>(&t.bias[l * VW + 16 * mt + 4 * kq]);
              acc[mt] = (f32x4){b4.x, b4.y, b4.z, b4.w};
            }
            if (n_ct == 1) slice_matmul<4, 1>(mat, a, acc, ei, kq);
            else if (n_mt <= 3) slice_matmul<3, 4>(mat, a, acc, ei, kq);
            else slice_matmul<4, 4>(mat, a, acc, ei, kq);
            float4 z[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) z[mt] = make_float4(acc[mt][0], acc[mt][1], acc[mt][2], acc[mt][3]);
            if (l + 1 < m.n_phi) {
              f4n_act<4>(m.phi_act[l], z);
#pragma unroll
              for (int mt = 0; mt < 4; ++mt) a[mt] = (valid && 16 * mt + 4 * kq < dw) ? z[mt] : f4_zero();
            } else {
#pragma unroll
              for (int mt = 0; mt < 4; ++mt) msg[mt] = (valid && 16 * mt + 4 * kq < dw) ? z[mt] : f4_zero();
            }
          }
        }
        NGPDE_VST(m, ph, 3);
        // messages through the staging tile, as many waves at a time as it has 16-row blocks (all, when the LDS left by the
        // weights allows); lane group r sums the rows of target r in edge order
        const int sw = m.s_rows >> 4;
        for (int w0 = 0; w0 < VT / 64; w0 += sw) {
          const int cs = c0 + 16 * w0;
          if (cs >= c.total) break;   // (uniform)
          if (c.wave >= w0 && c.wave < w0 + sw) {
            float *mine = t.S + (size_t)((c.wave - w0) * 16) * VTS;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<float4 *>(&mine[ei * VTS + 16 * mt + 4 * kq]) = msg[mt];
          }
          __syncthreads();
          if (rd == 0 && w0 == 0) NGPDE_VST(m, ph, 7);   // (diagnostic build: every wave's messages of the first step are staged)
          {
            const float *base = t.S + 4 * c.q - cs * VTS;
            const int ce = min(cs + 16 * sw, c0 + VROUND);   // (the last wave group of a round may be a partial one)
            for (int kk = max(lo, cs); kk < min(hi, ce); ++kk) racc = f4_add(racc, *reinterpret_cast<const float4 *>(base + kk * VTS));
          }
          __syncthreads();
        }
      }
      NGPDE_VST(m, ph, 4);
      if constexpr (ROUNDS) {   // part B of the next turn's prefetch: the halo nodes' positions and the rows' state (part A has landed by now)
        if (pre_h >= 0) {
          vctx_fetch_px(m, pre);
          const int nd = c.tid < VRT ? pre.rnode : -1;
#pragma unroll
          for (int j = 0; j < 6; ++j) pre.st[j] = 0.f;
          if (nd >= 0) {
            pre.st[0] = pre_ph == 1 ? p.u_in[nd] : p.state[nd];
            if (pre_ph > 1) {
#pragma unroll
              for (int j = 1; j < 6; ++j) pre.st[j] = p.state[(size_t)j * N + nd];
            }
          }
        }
      }
      // ---- node MLP on the 16 rows: input [h_i; m_i; 0 ...] in tile A (rows 0..15 of the staging area), layers ping-pong A <-> B
      float *tA = t.S, *tB = t.S + (size_t)VRT * VTS;
      if (has_row) {
        const float iv = t.inv[rg];
        const float4 mm = f4_scale(iv, racc);
        float *row = tA + rg * VTS;
        // columns 1 + 4 q .. 4 + 4 q (the message sits behind the state value); columns beyond the message width are zero already
        if (1 + 4 * c.q < VW) row[1 + 4 * c.q] = mm.x;
        if (2 + 4 * c.q < VW) row[2 + 4 * c.q] = mm.y;
        if (3 + 4 * c.q < VW) row[3 + 4 * c.q] = mm.z;
        if (4 + 4 * c.q < VW) row[4 + 4 * c.q] = mm.w;
        if (c.q == 0) row[0] = t.hh[c.half * VR + rg];
      }
      __syncthreads();
#pragma unroll
      for (int l = 0; l < kVmhMaxL; ++l) {
        if (l >= m.n_gam) break;
        const int din = m.gam_din[l], dw = m.gam_dout[l];
        const int n_ct = (din + 15) >> 4, n_mt = (dw + 15) >> 4;
        const float *tin = (l & 1) ? tB : tA;
        float *tout = (l & 1) ? tA : tB;
        if (p.tape_gam && c.row_valid && 4 * c.q < 16 * n_ct)
          *reinterpret_cast<float4 *>(p.tape_gam + (((size_t)l * m.evals + ev) * N + c.node) * VW + 4 * c.q) = *reinterpret_cast<const float4 *>(&tin[rg * VTS + 4 * c.q]);
        const int mt = c.wave & 3, rgp = c.wave >> 2;   // the wave's block of output columns and its 16 rows (a half tile: waves 4.. idle)
        const bool g_on = rgp < VRT / 16;
        float4 zo = f4_zero();
        if (mt < n_mt && g_on) {
          const float *mat = t.W + (size_t)(m.n_phi + l) * VW * VW;
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
          // (all four input blocks in flight at once; the columns of the tile beyond the layer's input width are zeros, and so are
          // the staged weights there)
          float4 w4[4], av[4];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            w4[ct] = wfrag(mat, mt, ct, ei, kq);
            av[ct] = *reinterpret_cast<const float4 *>(&tin[(16 * rgp + ei) * VTS + 16 * ct + 4 * kq]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            if (ct < n_ct) {
              acc = mfma16(w4[ct].x, av[ct].x, acc);
              acc = mfma16(w4[ct].y, av[ct].y, acc);
              acc = mfma16(w4[ct].z, av[ct].z, acc);
              acc = mfma16(w4[ct].w, av[ct].w, acc);
            }
          }
          const float4 b4 = *reinterpret_cast<const float4 *>(&t.bias[(kVmhMaxL + l) * VW + 16 * mt + 4 * kq]);
          const float4 z = make_float4(acc[0] + b4.x, acc[1] + b4.y, acc[2] + b4.z, acc[3] + b4.w);
          zo = (16 * mt + 4 * kq < dw) ? f4_act(m.gam_act[l], z) : f4_zero();
          // (padded columns inside a quad: the staged weights and biases are zero there, act(0) = 0 for the supported activations
          // except sigmoid -- the next layer's weight rows for them are zero, so they never feed a real column)
        }
        if (g_on) *reinterpret_cast<float4 *>(&tout[(16 * rgp + ei) * VTS + 16 * mt + 4 * kq]) = zo;
        __syncthreads();
      }
      NGPDE_VST(m, ph, 5);
      // ---- stage derivative k_i = gamma's output (column 0); the next stage input (or the step update) of the own nodes
      const float *tfin = (m.n_gam & 1) ? tB : tA;
      if (c.tid < VRT) {
        const float yv = my_node >= 0 ? tfin[c.tid * VTS] : 0.f;
        sk0 = i == 0 ? yv : sk0; sk1 = i == 1 ? yv : sk1; sk2 = i == 2 ? yv : sk2; sk3 = i == 3 ? yv : sk3; sk4 = i == 4 ? yv : sk4;
        float v = t.misc[36 + i] * yv;
        v = fmaf(1.0f, su, v);
        v = fmaf(t.misc[i * 6 + 0], sk0, v); v = fmaf(t.misc[i * 6 + 1], sk1, v); v = fmaf(t.misc[i * 6 + 2], sk2, v);
        v = fmaf(t.misc[i * 6 + 3], sk3, v); v = fmaf(t.misc[i * 6 + 4], sk4, v);
        if (i == p.S - 1) {
          su = v;
          if (p.save && (n + 1) % p.save_every == 0 && my_node >= 0) p.save[(size_t)((n + 1) / p.save_every - 1 + p.save_off) * N + my_node] = v;
        }
        if (my_node >= 0) st_sc1(Xn + my_node, v);
      }
      vmh_publish(m, c, ph, ROUNDS);
      NGPDE_VST(m, ph, 6);
      if (ROUNDS && my_node >= 0) {
        p.state[my_node] = su; p.state[N + my_node] = sk0; p.state[2 * N + my_node] = sk1; p.state[3 * N + my_node] = sk2;
        p.state[4 * N + my_node] = sk3; p.state[5 * N + my_node] = sk4;
        if (ph == last_ph) {
          if (p.u_out) p.u_out[my_node] = su;
          if (p.save && p.save_off) p.save[my_node] = p.u_in[my_node];
        }
      }
      }   // turns
      have_prev = true;
      ev_prev = ev;
    }
  }
  if constexpr (ROUNDS) {
    if (!ok) {   // a wait gave up: every output row of this workgroup's half tiles says so
      const float bad = __int_as_float(0x7fc00000);
      for (int s = 0; s < K; ++s) {
        const int h = blockIdx.x + s * G;
        if (h >= nh) break;
        const int nd = c.tid < VRT ? m.sched_t[(size_t)h * kTileRows + c.tid].x : -1;
        if (nd >= 0) {
          if (p.u_out) p.u_out[nd] = bad;
          if (p.save)
            for (int j = 0; j < p.n_steps / p.save_every + p.save_off; ++j) p.save[(size_t)j * N + nd] = bad;
        }
      }
    }
    return;
  }
  if (defer && have_prev && valid_keep) {   // the last evaluation's rows
#pragma unroll
    for (int l = 0; l < kVmhMaxL; ++l)
      if (l < m.n_phi) store_tape_rows(l, ev_prev);
  }
  // (u_out may alias u_in: a workgroup writes its rows only after every reader of its u0 rows is past its first phase -- the
  // host takes this plan for solves of at least two right-hand-side evaluations)
  if (my_node >= 0) {
    if (p.u_out) p.u_out[my_node] = ok ? su : __int_as_float(0x7fc00000);
    if (p.save) {
      if (p.save_off) p.save[my_node] = ok ? p.u_in[my_node] : __int_as_float(0x7fc00000);
      if (!ok)
        for (int j = 0; j < p.n_steps / p.save_every; ++j) p.save[(size_t)(j + p.save_off) * N + my_node] = __int_as_float(0x7fc00000);
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// discrete adjoint
// ---------------------------------------------------------------------------------------------------------------------
struct VmhBwdK {
  VmhMeta m;
  int n_steps, S;
  float *lam;                 // [N] in: dL/du(T); out: dL/du0
  const float *tape_phi, *tape_gam;
  float *dz_phi, *dz_gam;     // same shapes: every layer's dz
  float *dsrc0, *dsrc1;       // [E] the per-edge gradient towards the edge's SOURCE (p order), ping-pong by phase parity
  const float *cb;            // [S][8]: cb[i][i] = dt b_i, cb[i][j] (j > i) = dt a[j][i]
  const float *dsave;         // saveat: the cotangents of the saved states [T][N] (lam comes in holding the last one's), or null
  int save_every, save_off;
  float *state;               // tile rounds: [8][N] lambda, U-bar_0 .. U-bar_5 and the first half's own-row sums between a half tile's turns
};

template <bool ROUNDS>
__global__ __launch_bounds__(VT, 1) void node_vmh_bwd_kernel(const VmhBwdK p) {
  constexpr int VRT = ROUNDS ? 2 * VR : VR;
  constexpr int VMaxET = VRT * kSlotWidth;
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  __shared__ float s_hh[kHaloCap], s_px[kHaloCap * 4], s_inv[VRT], s_misc[64], s_es[VMaxET], s_row[VRT * 4];
  __shared__ int s_hnode[kHaloCap], s_off[VRT + 1], s_rs[VRT], s_rnode[VRT], s_okw[2], s_srcpos[VRT * kSlotWidth], s_srcdeg[VRT];
  __shared__ unsigned short s_edge[VMaxET];
  const VmhMeta &m = p.m;
  const int n_mats = m.n_phi + m.n_gam;
  VTabs t;
  t.W = dyn; t.S = dyn + (size_t)n_mats * VW * VW; t.bias = nullptr; t.hh = s_hh; t.px = s_px; t.hnode = s_hnode; t.off = s_off; t.rs = s_rs;
  t.rnode = s_rnode; t.inv = s_inv; t.edge = s_edge; t.misc = s_misc; t.s_ok = s_okw;
  VCtx c;
  // TILE ROUNDS (see the forward kernel): K half tiles per workgroup.  A phase of the adjoint has its hand-off in the middle, so a
  // workgroup makes TWO passes over its half tiles: pass 1 walks every one back to the per-edge gradients and publishes, pass 2 gathers by
  // source -- a half tile waiting for a later one of the same workgroup would otherwise wait forever.
  // (ROUNDS is a template parameter: the one-half-tile form keeps its register allocation, the rounds form carries no rows across phases)
  const int nh = ROUNDS ? m.n_tiles : 2 * m.n_tiles, G = gridDim.x, K = ROUNDS ? (nh + G - 1) / G : 1;
  vctx_init<VRT>(m, c, t, blockIdx.x);
  // two waves share a SIMD (waves w and w + 4): the first runs at high priority, so the pair does not march in lockstep through
  // MFMA chain and activation code alike -- the second fills the matrix pipe while the first is in its VALU stretches
  if (c.wave < 4) __builtin_amdgcn_s_setprio(3);
  for (int l = 0; l < m.n_phi; ++l) stage_weight(m.phi_w[l], m.phi_din[l], m.phi_dout[l], t.W + (size_t)l * VW * VW, c.tid, false);
  for (int l = 0; l < m.n_gam; ++l) stage_weight(m.gam_w[l], m.gam_din[l], m.gam_dout[l], t.W + (size_t)(m.n_phi + l) * VW * VW, c.tid, false);
  if (c.tid < p.S * 8 && c.tid < 64) t.misc[c.tid] = p.cb[c.tid];
  auto fill_srcpos = [&]() {   // positions, in the by-target order, of the out-edges of the own nodes: the by-source gather's addresses
    if (c.g16 < VRT) {         // (one coalesced read of the plan's schedule-ordered copy: no node -> row pointer -> position chain per turn)
      const size_t row = (size_t)c.tile * kTileRows + c.half * VR + c.g16;
      if (c.q == 0) s_srcdeg[c.g16] = m.srcdeg[row];
      for (int j = c.q; j < kSlotWidth; j += 16) s_srcpos[c.g16 * kSlotWidth + j] = m.srcpos[row * kSlotWidth + j];
    }
  };
  fill_srcpos();
  __syncthreads();
  const int ei = c.ei, kq = c.kq;
  const int Mw = m.phi_dout[m.n_phi - 1];
  int n_rounds = (c.total + VROUND - 1) / VROUND;
  const size_t E = m.n_edges, N = (size_t)m.n_nodes;
  const int S = p.S;
  const int rg = min(c.g16, VRT - 1);
  const bool has_row = c.g16 < VRT;
  // the 16 row lanes keep lambda and the stage adjoints of their node
  int my_node = c.tid < VRT ? t.rnode[c.tid] : -1;
  float lam = my_node >= 0 ? p.lam[my_node] : 0.f;
  float ub[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const int last_ph = p.n_steps * S;
  auto load_state = [&](int ph) {   // (tile rounds) lambda and the stage adjoints of this turn's rows
    if (my_node >= 0) {
      lam = ph == 1 ? p.lam[my_node] : p.state[my_node];
#pragma unroll
      for (int j = 0; j < 6; ++j) ub[j] = ph == 1 ? 0.f : p.state[(size_t)(1 + j) * N + my_node];
    } else {
      lam = 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j) ub[j] = 0.f;
    }
  };
  auto store_state = [&]() {
    if (my_node >= 0) {
      p.state[my_node] = lam;
#pragma unroll
      for (int j = 0; j < 6; ++j) p.state[(size_t)(1 + j) * N + my_node] = ub[j];
    }
  };
  float *tA = t.S, *tB = t.S + (size_t)VRT * VTS;
  // With ONE round a lane's edge is the same in every phase, and what a phase reads from the tapes does not depend on the exchange:
  // the outputs of gamma's hidden layers and of phi's last hidden layer are fetched a phase ahead, behind the publish, and land while
  // the workgroup waits for its neighbours; phi's lower layers are fetched one layer ahead, under the layer's MFMAs.
  const bool one_round = !ROUNDS && n_rounds == 1;
  const int k1 = c.wave * 16 + ei;
  const bool valid1 = one_round && k1 < c.total;
  const int r1 = t.edge[valid1 ? k1 : 0] & 0xff;
  const size_t pe1 = (size_t)(t.rs[r1] + (k1 - t.off[r1]));
  float4 ytop[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()}, yg[kVmhMaxL - 1];
  auto fetch_phi = [&](int l, size_t ev, float4 (&y)[4]) {   // the output of phi's layer l: the input tape of layer l + 1
    const int n_mt = (m.phi_dout[l] + 15) >> 4;
    const float *yrow = p.tape_phi + (((size_t)(l + 1) * m.evals + ev) * E + pe1) * VW + 4 * kq;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) y[mt] = (valid1 && mt < n_mt) ? ld4_nt(yrow + 16 * mt) : f4_zero();
  };
  auto fetch_phase = [&](size_t ev) {
#pragma unroll
    for (int l = 0; l < kVmhMaxL - 1; ++l) {
      yg[l] = f4_zero();
      if (l + 1 < m.n_gam && c.row_valid && 4 * c.q < 16 * ((m.gam_dout[l] + 15) >> 4))
        yg[l] = ld4_nt(p.tape_gam + (((size_t)(l + 1) * m.evals + ev) * N + c.node) * VW + 4 * c.q);
    }
    if (one_round && m.n_phi >= 2) fetch_phi(m.n_phi - 2, ev, ytop);
  };
  fetch_phase((size_t)(p.n_steps * S - 1));
  // the dz rows of a one-round solve leave one evaluation late, layer by layer in front of the layer's MFMAs (see the forward kernel)
  const bool defer = one_round;
  float4 tz[kVmhMaxL][4];
  size_t pe_keep = pe1, ev_prev = 0;
  bool valid_keep = valid1, have_prev = false;
  auto store_dz_rows = [&](int l, size_t evx) {
    const int n_mt = (m.phi_dout[l] + 15) >> 4;
    float *zrow = p.dz_phi + (((size_t)l * m.evals + evx) * E + pe_keep) * VW + 4 * kq;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
      if (mt < n_mt) __builtin_nontemporal_store((f4v){tz[l][mt].x, tz[l][mt].y, tz[l][mt].z, tz[l][mt].w}, reinterpret_cast<NGPDE_GLOBAL_AS f4v *>(reinterpret_cast<uintptr_t>(zrow + 16 * mt)));
  };
  bool ok = true;
  // the second half of a phase: the by-source gather behind the hand-off, the stage adjoint (false: the wait gave up)
  auto pass2 = [&](int ph2, int i2, const float *dsrc2) -> bool {
    if (!vmh_wait(m, c, ph2, t.s_ok)) return false;
    NGPDE_VST(m, ph2, 6);
    {   // the by-source sum: the out-edges of row g16 (16 lanes, two entries each), then a fixed-order lane reduction
      const int dg = has_row ? s_srcdeg[rg] : 0;
      float a = 0.f;
      if (c.q < dg) a = ld_sc1(dsrc2 + s_srcpos[rg * kSlotWidth + c.q]);
      if (c.q + 16 < dg) a += ld_sc1(dsrc2 + s_srcpos[rg * kSlotWidth + c.q + 16]);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) a += __shfl_xor(a, o);
      if (c.q == 0 && has_row) s_row[2 * VRT + rg] = a;
    }
    __syncthreads();
    if (c.tid < VRT) {
      const float ubar = my_node >= 0 ? s_row[VRT + c.tid] + s_row[2 * VRT + c.tid] : 0.f;
#pragma unroll
      for (int j = 0; j < 6; ++j) ub[j] = (j == i2) ? ubar : ub[j];
      if (i2 == 0) {
        float v = lam;
        for (int j = 0; j < S; ++j) v += ub[j];
        lam = v;
      }
    }
    return true;
  };
  // TILE ROUNDS sweep the half tiles evals + 1 times: a half tile's turn in sweep ph is the second half of phase ph - 1 -- its neighbours
  // published that phase during the sweep before -- followed at once by the first half of phase ph: one rebuild of its tables and one
  // round trip of its rows' state per phase
  const int n_sweeps = last_ph + (ROUNDS ? 1 : 0);
  VPre pre;          // (tile rounds) the next turn's tables, by-source positions and state, on their way
  int pre_h = -1, pre_ph = 0;
  auto fetch_b = [&]() {   // part B of the next turn's prefetch: the halo nodes' positions, the rows' state
    if (pre_h < 0) return;
    vctx_fetch_px(m, pre);
    const int nd = c.tid < VRT ? pre.rnode : -1;
#pragma unroll
    for (int j = 0; j < 8; ++j) pre.st[j] = 0.f;
    if (nd >= 0) {
      pre.st[0] = pre_ph == 1 ? p.lam[nd] : p.state[nd];
      if (pre_ph > 1) {
#pragma unroll
        for (int j = 1; j < 8; ++j) pre.st[j] = p.state[(size_t)j * N + nd];
      }
    }
  };
  float4 yall[kVmhMaxL - 1][4];   // (tile rounds) the outputs of phi's hidden layers of the lane's edge in the coming round
  auto fetch_y = [&](int l, size_t ev, int rd) {   // yall[l] <- the output of phi's layer l (the input tape of layer l + 1) for round rd
    const int k = rd * VROUND + c.wave * 16 + ei;
    const bool valid = l + 1 < m.n_phi && k < c.total;
    const int r = t.edge[valid ? k : 0] & 0xff;
    const size_t pe = (size_t)(t.rs[r] + (k - t.off[r]));
    const int n_mt = (m.phi_dout[l] + 15) >> 4;
    const float *yrow = p.tape_phi + (((size_t)(l + 1) * m.evals + ev) * E + pe) * VW + 4 * kq;
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) yall[l][mt] = (valid && mt < n_mt) ? ld4_nt(yrow + 16 * mt) : f4_zero();
  };
  for (int ph = 1; ph <= n_sweeps && ok; ++ph) {
    {
      const bool do1 = ph <= last_ph;
      const int idx = min(ph, last_ph) - 1;
      const int n = p.n_steps - 1 - idx / S, i = S - 1 - idx % S;
      const size_t ev = (size_t)(n * S + i);
      float *dsrc = (ph & 1) ? p.dsrc1 : p.dsrc0;
      // ======== the workgroup's half tiles: K-bar, gamma and phi backwards, the per-edge gradients, publish
      for (int s = 0; s < K; ++s) {
      if constexpr (ROUNDS) {
        const int h = blockIdx.x + s * G;
        if (h >= nh) break;
        NGPDE_VST(m, ph, 7);
        float own_sum = 0.f;
        if (pre_h == h) {      // tables, by-source positions and the rows' state from what the turn before asked for
          vctx_commit<VRT>(m, c, t, h, pre);
          if (c.g16 < VRT) {
            if (c.q == 0) s_srcdeg[c.g16] = pre.sdeg;
            s_srcpos[c.g16 * kSlotWidth + c.q] = pre.sp0;
            s_srcpos[c.g16 * kSlotWidth + c.q + 16] = pre.sp1;
          }
          my_node = c.tid < VRT ? pre.rnode : -1;
          lam = pre.st[0];
#pragma unroll
          for (int j = 0; j < 6; ++j) ub[j] = pre.st[1 + j];
          own_sum = pre.st[7];
        } else {
          vctx_init<VRT>(m, c, t, h);
          fill_srcpos();
          my_node = c.tid < VRT ? t.rnode[c.tid] : -1;
          load_state(ph);
          if (ph > 1 && my_node >= 0) own_sum = p.state[(size_t)7 * N + my_node];
        }
        n_rounds = (c.total + VROUND - 1) / VROUND;
        {   // the next turn's unit: part A of its tables is asked for now
          int hn = h + G;
          pre_ph = ph;
          if (hn >= nh) { hn = blockIdx.x; pre_ph = ph + 1; }
          pre_h = (hn != h && pre_ph <= n_sweeps) ? hn : -1;
          if (pre_h >= 0) vctx_fetch_a<VRT>(m, pre_h, pre, true);
        }
        if (do1) {   // this phase's tape rows: they land under the second half of the phase before and K-bar / gamma
          fetch_phase(ev);
#pragma unroll
          for (int l = 0; l < kVmhMaxL - 1; ++l) fetch_y(l, ev, 0);   // (the first round's; a later round's rows are asked for a round ahead)
        }
        NGPDE_VST(m, ph, 5);
        if (ph > 1) {
          if (my_node >= 0) s_row[VRT + c.tid] = own_sum;
          __syncthreads();
          const int idx2 = ph - 2;
          if (!pass2(ph - 1, S - 1 - idx2 % S, ((ph - 1) & 1) ? p.dsrc1 : p.dsrc0)) { ok = false; break; }
          if (!do1) {   // the sweep behind the last phase: the rows' lambda is dL/du0
            if (p.dsave && p.save_off && my_node >= 0) lam += p.dsave[my_node];
            if (my_node >= 0) p.lam[my_node] = lam;
            fetch_b();
            __syncthreads();
            continue;
          }
        }
      }
      // a state saved after step n + 1 hands its cotangent to lambda before step n + 1 is walked back
      if (i == S - 1 && p.dsave && n + 1 < p.n_steps && (n + 1) % p.save_every == 0 && my_node >= 0)
        lam += p.dsave[(size_t)((n + 1) / p.save_every - 1 + p.save_off) * N + my_node];
      NGPDE_VST(m, ph, 0);
      // ---- K-bar_i of the own nodes -> the gradient of gamma's output (column 0 of tile A)
      if (c.tid < VRT) {
        float kbar = t.misc[i * 8 + i] * lam;
#pragma unroll
        for (int j = 0; j < 6; ++j)
          if (j > i && j < S) kbar = fmaf(t.misc[i * 8 + j], ub[j], kbar);
        s_row[c.tid] = my_node >= 0 ? kbar : 0.f;
      }
      __syncthreads();
      if (has_row) *reinterpret_cast<float4 *>(&tA[rg * VTS + 4 * c.q]) = (c.q == 0) ? make_float4(s_row[rg], 0.f, 0.f, 0.f) : f4_zero();
      __syncthreads();
      // ---- gamma backwards: g (tile) -> dz_l = g . act'(output of layer l) -> tape; g <- W_l dz_l
#pragma unroll
      for (int li = 0; li < kVmhMaxL; ++li) {
        const int l = kVmhMaxL - 1 - li;
        if (l >= m.n_gam) continue;
        const int din = m.gam_din[l], dw = m.gam_dout[l];
        const int n_ct = (din + 15) >> 4, n_mt = (dw + 15) >> 4;
        float *tg = ((m.n_gam - 1 - l) & 1) ? tB : tA;       // holds g (gradient of layer l's output), becomes dz in place
        float *tn = ((m.n_gam - 1 - l) & 1) ? tA : tB;
        if (has_row) {
          float4 g = *reinterpret_cast<const float4 *>(&tg[rg * VTS + 4 * c.q]);
          if (l + 1 < m.n_gam && c.row_valid && 4 * c.q < 16 * n_mt) {
            float4 dy[1] = {yg[l < kVmhMaxL - 1 ? l : 0]};
            f4n_dact_out<1>(m.gam_act[l], dy);
            g = f4_mul(g, dy[0]);
          }
          if (!(c.row_valid && 4 * c.q < dw)) g = f4_zero();
          *reinterpret_cast<float4 *>(&tg[rg * VTS + 4 * c.q]) = g;
          if (c.row_valid && 4 * c.q < 16 * n_mt) *reinterpret_cast<float4 *>(p.dz_gam + (((size_t)l * m.evals + ev) * N + c.node) * VW + 4 * c.q) = g;
        }
        __syncthreads();
        const int ct = c.wave & 3, rgp = c.wave >> 2;   // the wave's block of INPUT columns and its 16 rows
        const bool g_on = rgp < VRT / 16;
        float4 go = f4_zero();
        if (ct < n_ct && g_on) {
          const float *mat = t.W + (size_t)(m.n_phi + l) * VW * VW;
          f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
          float4 w4[4], dv[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            w4[mt] = wfrag(mat, ct, mt, ei, kq);
            dv[mt] = *reinterpret_cast<const float4 *>(&tg[(16 * rgp + ei) * VTS + 16 * mt + 4 * kq]);
          }
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            if (mt < n_mt) {
              acc = mfma16(w4[mt].x, dv[mt].x, acc);
              acc = mfma16(w4[mt].y, dv[mt].y, acc);
              acc = mfma16(w4[mt].z, dv[mt].z, acc);
              acc = mfma16(w4[mt].w, dv[mt].w, acc);
            }
          }
          go = make_float4(acc[0], acc[1], acc[2], acc[3]);
        }
        if (g_on) *reinterpret_cast<float4 *>(&tn[(16 * rgp + ei) * VTS + 16 * ct + 4 * kq]) = go;
        __syncthreads();
      }
      // d(gamma's input) = [dh_i; dm_i]: tile tgin, row r; the message gradient is scaled by 1 / deg (mean)
      const float *tgin = (m.n_gam & 1) ? tB : tA;
      NGPDE_VST(m, ph, 1);
      // ---- phi backwards per 16-edge wave slice
      for (int k = c.tid; k < c.total; k += VT) s_es[k] = 0.f;
      __syncthreads();
      for (int rd = 0; rd < n_rounds; ++rd) {
        const int c0 = rd * VROUND;
        const bool wave_on = c0 + c.wave * 16 < c.total;
        const int k = c0 + c.wave * 16 + ei;
        const bool valid = k < c.total;
        if (wave_on) {
          const unsigned ew = t.edge[valid ? k : 0];
          const int r = ew & 0xff;
          const size_t pe = (size_t)(t.rs[r] + (k - t.off[r]));
          const float inv = valid ? t.inv[r] : 0.f;
          float4 g[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int f = 16 * mt + 4 * kq;
            const float *src = tgin + r * VTS + 1 + f;     // (the message sits behind the state value in gamma's input)
            g[mt] = (valid && f < Mw) ? make_float4(inv * src[0], f + 1 < Mw ? inv * src[1] : 0.f, f + 2 < Mw ? inv * src[2] : 0.f, f + 3 < Mw ? inv * src[3] : 0.f)
                                      : f4_zero();
          }
          float4 ycur[4], ynext[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) ycur[mt] = ynext[mt] = ytop[mt];
#pragma unroll
          for (int li = 0; li < kVmhMaxL; ++li) {
            const int l = kVmhMaxL - 1 - li;
            if (l >= m.n_phi) continue;
            const int din = m.phi_din[l], dw = m.phi_dout[l];
            const int n_ct = (din + 15) >> 4, n_mt = (dw + 15) >> 4;
            if (l + 1 < m.n_phi) {
              const float *yrow = p.tape_phi + (((size_t)(l + 1) * m.evals + ev) * E + pe) * VW + 4 * kq;
              float4 dy[4];
#pragma unroll
              for (int mt = 0; mt < 4; ++mt)
                dy[mt] = one_round ? ycur[mt]
                                   : (ROUNDS ? yall[l < kVmhMaxL - 1 ? l : 0][mt]
                                             : ((valid && mt < n_mt) ? *reinterpret_cast<const float4 *>(yrow + 16 * mt) : f4_zero()));
              if (ROUNDS && rd + 1 < n_rounds) fetch_y(l < kVmhMaxL - 1 ? l : 0, ev, rd + 1);   // the registers are free again: the next round's rows
              f4n_dact_out<4>(m.phi_act[l], dy);
#pragma unroll
              for (int mt = 0; mt < 4; ++mt) g[mt] = f4_mul(g[mt], dy[mt]);
            }
            if (one_round && l >= 1 && l + 1 < m.n_phi) fetch_phi(l - 1, ev, ynext);   // for the layer below, under this layer's MFMAs
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
              if (!(valid && 16 * mt + 4 * kq < dw)) g[mt] = f4_zero();
            if (defer) {
              if (have_prev && valid_keep) store_dz_rows(l, ev_prev);
#pragma unroll
              for (int mt = 0; mt < 4; ++mt) tz[l][mt] = g[mt];
            } else if (valid) {
              float *zrow = p.dz_phi + (((size_t)l * m.evals + ev) * E + pe) * VW + 4 * kq;
#pragma unroll
              for (int mt = 0; mt < 4; ++mt)
                if (mt < n_mt) __builtin_nontemporal_store((f4v){g[mt].x, g[mt].y, g[mt].z, g[mt].w}, reinterpret_cast<NGPDE_GLOBAL_AS f4v *>(reinterpret_cast<uintptr_t>(zrow + 16 * mt)));
            }
            const float *mat = t.W + (size_t)l * VW * VW;
            f32x4 gn[4] = {(f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}, (f32x4){0.f, 0.f, 0.f, 0.f}};
            if (n_ct == 1) slice_matmul<1, 4>(mat, g, gn, ei, kq);
            else if (n_mt <= 3) slice_matmul<4, 3>(mat, g, gn, ei, kq);
            else slice_matmul<4, 4>(mat, g, gn, ei, kq);
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
              g[ct] = make_float4(gn[ct][0], gn[ct][1], gn[ct][2], gn[ct][3]);
              ycur[ct] = ynext[ct];
            }
          }
          // d(input) = [d h_i, d (h_j - h_i), ...]: lane kq = 0 holds features 0..3
          if (valid && kq == 0) {
            s_es[k] = g[0].x - g[0].y;          // towards the target
            st_sc1(dsrc + pe, g[0].y);          // towards the source: gathered by its owner after the hand-off
          }
        }
      }
      NGPDE_VST(m, ph, 2);
      if constexpr (ROUNDS) fetch_b();   // (part A of the next turn's prefetch has landed by now)
      __syncthreads();
      NGPDE_VST(m, ph, 3);
      if (c.tid < VRT) {   // what the own rows get from their own edges and from gamma
        float a = tgin[c.tid * VTS];
        for (int k = t.off[c.tid]; k < t.off[c.tid + 1]; ++k) a += s_es[k];
        s_row[VRT + c.tid] = a;
      }
      vmh_publish(m, c, ph, ROUNDS);
      NGPDE_VST(m, ph, 4);
      if constexpr (ROUNDS) {
        store_state();
        if (my_node >= 0) p.state[(size_t)7 * N + my_node] = s_row[VRT + c.tid];          // the rows' own sums wait for the next sweep
        __syncthreads();   // the tables are rebuilt for the next half tile
      }
      }   // half tiles
      if (!ok) break;
      if constexpr (!ROUNDS) {
        have_prev = true;
        ev_prev = ev;
        if (ev > 0) fetch_phase(ev - 1);
        NGPDE_VST(m, ph, 5);
        if (!pass2(ph, i, dsrc)) { ok = false; break; }
      }
    }
  }
  if constexpr (ROUNDS) {
    if (!ok) {   // a wait gave up: every output row of this workgroup's half tiles says so
      for (int s = 0; s < K; ++s) {
        const int h = blockIdx.x + s * G;
        if (h >= nh) break;
        const int nd = c.tid < VRT ? m.sched_t[(size_t)h * kTileRows + c.tid].x : -1;
        if (nd >= 0) p.lam[nd] = __int_as_float(0x7fc00000);
      }
    }
    return;
  }
  if (defer && have_prev && valid_keep) {   // the last evaluation's dz rows
#pragma unroll
    for (int l = 0; l < kVmhMaxL; ++l)
      if (l < m.n_phi) store_dz_rows(l, ev_prev);
  }
  if (p.dsave && p.save_off && my_node >= 0) lam += p.dsave[my_node];
  if (my_node >= 0) p.lam[my_node] = ok ? lam : __int_as_float(0x7fc00000);
}

__global__ void vmh_set_word_kernel(unsigned *w, unsigned v) {
  if (threadIdx.x == 0) *w = v;
}
__global__ void vmh_latch_fault_kernel(const unsigned *abort_word, unsigned *fault) {
  if (threadIdx.x == 0 && *abort_word != 0) *fault = 1u;
}
__global__ void vmh_copy_block_kernel(const float *src, int sp, float *dst, int dp, int rows, int cols) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x;
  if (idx < rows * cols) dst[(idx / cols) * dp + idx % cols] = src[(idx / cols) * sp + idx % cols];
}

size_t vmh_lds_bytes(int n_mats, int s_rows) { return ((size_t)n_mats * VW * VW + (size_t)s_rows * VTS) * sizeof(float); }
// How a graph of n_tiles tiles runs: every half tile its own workgroup when all of them are resident at once (rounds = false), else
// whole tiles in tile rounds on as many workgroups as are resident.  s_rows: the staging tile, as many of the workgroup's waves as the
// LDS holds beside the weights and the instantiation's static arrays.  grid = 0: neither form fits.
struct VmhPlanGeo {
  bool rounds = false;
  int grid = 0, turns = 0, s_rows = 64;
  size_t lds = 0;
};
template <bool ROUNDS>
static VmhPlanGeo vmh_geo_of(int n_mats, int units, int cus) {
  VmhPlanGeo geo;
  geo.rounds = ROUNDS;
  hipFuncAttributes fa{}, ba{};
  if (hipFuncGetAttributes(&fa, reinterpret_cast<const void *>(node_vmh_fwd_kernel<ROUNDS>)) != hipSuccess) return geo;
  if (hipFuncGetAttributes(&ba, reinterpret_cast<const void *>(node_vmh_bwd_kernel<ROUNDS>)) != hipSuccess) return geo;
  const size_t fixed = std::max(fa.sharedSizeBytes, ba.sharedSizeBytes), cap = 160 * 1024;
  geo.s_rows = 0;
  for (int rows = VROUND; rows >= 64; rows -= 32)
    if (vmh_lds_bytes(n_mats, rows) + fixed <= cap) { geo.s_rows = rows; break; }
  if (geo.s_rows == 0) return geo;
  geo.lds = vmh_lds_bytes(n_mats, geo.s_rows);
  int occ = 1 << 30;
  auto take = [&](auto kernel) {
    int o = 0;
    if (hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)geo.lds) != hipSuccess) o = 0;
    else if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kernel, VT, geo.lds) != hipSuccess) o = 0;
    occ = std::min(occ, o);
  };
  take(node_vmh_fwd_kernel<ROUNDS>); take(node_vmh_bwd_kernel<ROUNDS>);
  if (occ < 1 || units < 1) return geo;
  geo.grid = std::min(units, cus * occ);
  geo.turns = (units + geo.grid - 1) / geo.grid;
  return geo;
}
static VmhPlanGeo vmh_geo_uncached(int n_mats, int n_tiles, bool no_rounds) {
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return VmhPlanGeo();
  VmhPlanGeo one = vmh_geo_of<false>(n_mats, 2 * n_tiles, cus);
  if (one.grid > 0 && one.turns == 1) return one;
  if (no_rounds) return VmhPlanGeo();
  VmhPlanGeo many = vmh_geo_of<true>(n_mats, n_tiles, cus);
  if (many.grid > 0 && many.turns <= kVmhMaxTurns) return many;
  return VmhPlanGeo();
}
// (asked at every launch: the attribute / occupancy queries behind it are remembered per shape and device)
static VmhPlanGeo vmh_geo(int n_mats, int n_tiles) {
  const char *nr = std::getenv("NGPDE_NO_VMH_ROUNDS");   // (tests and A/B runs: graphs beyond the resident half tiles to the generic solver)
  const bool no_rounds = nr && nr[0] == '1';
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return VmhPlanGeo();
  static std::mutex mu;
  static std::map<std::tuple<int, int, int, bool>, VmhPlanGeo> seen;
  const auto key = std::make_tuple(dev, n_mats, n_tiles, no_rounds);
  std::lock_guard<std::mutex> lock(mu);
  auto it = seen.find(key);
  if (it != seen.end()) return it->second;
  if (seen.size() > 256) seen.clear();
  const VmhPlanGeo geo = vmh_geo_uncached(n_mats, n_tiles, no_rounds);
  seen[key] = geo;
  return geo;
}

}  // namespace

bool node_vmh_supported(const ngpde_graph *g, const VmhShape &s) {
  const char *off = std::getenv("NGPDE_NO_VMH_NODE");
  if (off && off[0] == '1') return false;
  if (!g || !g->has_norm || !g->by_t.halo_ok || !g->by_s.halo_ok) return false;
  if (s.hd != 1 || s.pd < 1 || s.pd > 3) return false;
  if (s.n_phi < 2 || s.n_phi > kVmhMaxL || s.n_gam < 2 || s.n_gam > kVmhMaxL) return false;
  if (s.aggr != NGPDE_AGGR_SUM && s.aggr != NGPDE_AGGR_MEAN) return false;
  auto act_ok = [](int a) { return a == NGPDE_ACT_IDENTITY || a == NGPDE_ACT_RELU || a == NGPDE_ACT_TANH || a == NGPDE_ACT_SIGMOID; };
  if (s.phi_dims[0] != 2 * s.hd + s.pd || s.gam_dims[0] != s.hd + s.phi_dims[s.n_phi] || s.gam_dims[s.n_gam] != s.hd) return false;
  for (int l = 0; l <= s.n_phi; ++l)
    if (s.phi_dims[l] < 1 || s.phi_dims[l] > VW) return false;
  for (int l = 0; l <= s.n_gam; ++l)
    if (s.gam_dims[l] < 1 || s.gam_dims[l] > VW) return false;
  if (s.phi_dims[s.n_phi] + s.hd > VW) return false;
  for (int l = 0; l < s.n_phi; ++l)
    if (!act_ok(s.phi_act[l])) return false;
  for (int l = 0; l < s.n_gam; ++l)
    if (!act_ok(s.gam_act[l])) return false;
  if (s.phi_act[s.n_phi - 1] != NGPDE_ACT_IDENTITY || s.gam_act[s.n_gam - 1] != NGPDE_ACT_IDENTITY) return false;   // (output layers)
  if (g->max_in_degree > kSlotWidth || g->max_out_degree > kSlotWidth) return false;
  return vmh_geo(s.n_phi + s.n_gam, g->n_sched / kTileRows).grid > 0;
}

static void fill_meta(VmhMeta &m, const VmhLaunch &a) {
  const ngpde_graph *g = a.g;
  m.sched_t = g->by_t.sched; m.halo_t = g->by_t.halo; m.info_t = g->by_t.tile_info; m.slots_t = g->by_t.slots;
  m.rowptr_s = g->by_s.rowptr; m.xpos_s = g->by_s.xpos;
  m.nbr = a.ps->nbr;
  m.n_tiles = g->n_sched / kTileRows;
  m.flags = a.ps->sync; m.abort_word = a.ps->sync + (size_t)(2 * m.n_tiles) * 32;
  m.n_nodes = (int)g->n_nodes; m.pos = a.pos; m.pd = a.shape.pd; m.aggr = a.shape.aggr;
  m.n_phi = a.shape.n_phi; m.n_gam = a.shape.n_gam;
  for (int l = 0; l < kVmhMaxL; ++l) {
    m.phi_din[l] = l < m.n_phi ? a.shape.phi_dims[l] : 0; m.phi_dout[l] = l < m.n_phi ? a.shape.phi_dims[l + 1] : 0;
    m.phi_act[l] = l < m.n_phi ? a.shape.phi_act[l] : 0;
    m.gam_din[l] = l < m.n_gam ? a.shape.gam_dims[l] : 0; m.gam_dout[l] = l < m.n_gam ? a.shape.gam_dims[l + 1] : 0;
    m.gam_act[l] = l < m.n_gam ? a.shape.gam_act[l] : 0;
    m.phi_w[l] = l < m.n_phi ? a.phi_w[l] : nullptr; m.phi_b[l] = l < m.n_phi ? a.phi_b[l] : nullptr;
    m.gam_w[l] = l < m.n_gam ? a.gam_w[l] : nullptr; m.gam_b[l] = l < m.n_gam ? a.gam_b[l] : nullptr;
  }
  m.n_edges = (size_t)g->n_edges;
  m.srcpos = a.srcpos; m.srcdeg = a.srcdeg;
  m.evals = a.n_steps * a.S;
#ifdef NGPDE_STAMPS
  m.stamps = g_vst_base; m.stamps_max = g_vst_max;
#endif
}

int32_t launch_node_vmh_fwd(const VmhLaunch &a, hipStream_t stream) {
  const NodePersist &ps = *a.ps;
  int32_t st;
  PersistentTurn turn;
  if ((st = turn.enter(stream))) return st;
  if ((st = launch_zero(ps.sync, ps.sync_bytes, stream))) return st;
  VmhFwdK k;
  fill_meta(k.m, a);
  {
    const char *fa = std::getenv("NGPDE_DEBUG_FORCE_ABORT");
    if (fa && fa[0] == '1') hipLaunchKernelGGL(vmh_set_word_kernel, dim3(1), dim3(64), 0, stream, k.m.abort_word, 1u);
  }
  k.n_steps = a.n_steps; k.S = a.S; k.u_in = a.u_in; k.u_out = a.u_out; k.x0 = a.x0; k.x1 = a.x1;
  k.save = a.save; k.save_every = a.save_every; k.save_off = a.save_off;
  k.state = a.state;
  k.tape_phi = a.tape_phi; k.tape_gam = a.tape_gam; k.cf = a.cf;
  const VmhPlanGeo geo = vmh_geo(k.m.n_phi + k.m.n_gam, k.m.n_tiles);
  NGPDE_REQUIRE(geo.grid > 0, NGPDE_ERR_UNSUPPORTED, "node_vmh_fwd_kernel: the graph does not fit the device-resident plan");
  k.m.s_rows = geo.s_rows;
  // (the attribute is per function, and another shape's geometry may have lowered it since)
  if (geo.rounds) {
    NGPDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(node_vmh_fwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)geo.lds));
    hipLaunchKernelGGL(node_vmh_fwd_kernel<true>, dim3(geo.grid), dim3(VT), geo.lds, stream, k);
  } else {
    NGPDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(node_vmh_fwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)geo.lds));
    hipLaunchKernelGGL(node_vmh_fwd_kernel<false>, dim3(geo.grid), dim3(VT), geo.lds, stream, k);
  }
  NGPDE_LAUNCH_CHECK("node_vmh_fwd_kernel");
  hipLaunchKernelGGL(vmh_latch_fault_kernel, dim3(1), dim3(64), 0, stream, k.m.abort_word, ps.fault);
  NGPDE_LAUNCH_CHECK("latch_fault_kernel");
  return turn.leave();
}

int32_t launch_node_vmh_bwd(const VmhLaunch &a, hipStream_t stream) {
  const NodePersist &ps = *a.ps;
  int32_t st;
  PersistentTurn turn;
  if ((st = turn.enter(stream))) return st;
  if ((st = launch_zero(ps.sync, ps.sync_bytes, stream))) return st;
  VmhBwdK k;
  fill_meta(k.m, a);
  k.dsave = a.dsave; k.save_every = a.save_every; k.save_off = a.save_off;
  k.state = a.state;
  k.n_steps = a.n_steps; k.S = a.S; k.lam = a.lam; k.tape_phi = a.tape_phi; k.tape_gam = a.tape_gam; k.dz_phi = a.dz_phi; k.dz_gam = a.dz_gam;
  k.dsrc0 = a.dsrc; k.dsrc1 = a.dsrc + k.m.n_edges; k.cb = a.cb;
  const VmhPlanGeo geo = vmh_geo(k.m.n_phi + k.m.n_gam, k.m.n_tiles);
  NGPDE_REQUIRE(geo.grid > 0, NGPDE_ERR_UNSUPPORTED, "node_vmh_bwd_kernel: the graph does not fit the device-resident plan");
  k.m.s_rows = geo.s_rows;
  if (geo.rounds) {
    NGPDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(node_vmh_bwd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)geo.lds));
    hipLaunchKernelGGL(node_vmh_bwd_kernel<true>, dim3(geo.grid), dim3(VT), geo.lds, stream, k);
  } else {
    NGPDE_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(node_vmh_bwd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)geo.lds));
    hipLaunchKernelGGL(node_vmh_bwd_kernel<false>, dim3(geo.grid), dim3(VT), geo.lds, stream, k);
  }
  NGPDE_LAUNCH_CHECK("node_vmh_bwd_kernel");
  hipLaunchKernelGGL(vmh_latch_fault_kernel, dim3(1), dim3(64), 0, stream, k.m.abort_word, ps.fault);
  NGPDE_LAUNCH_CHECK("latch_fault_kernel");
  return turn.leave();
}

__global__ void vmh_srcpos_kernel(const int4 *__restrict__ sched_t, int n_sched, const int *__restrict__ rowptr_s, const int *__restrict__ xpos_s,
                                  int *__restrict__ srcpos, int *__restrict__ srcdeg) {
  const int idx = blockIdx.x * blockDim.x + threadIdx.x, r = idx / kSlotWidth, j = idx % kSlotWidth;
  if (r >= n_sched) return;
  const int nd = sched_t[r].x;
  const int rp = nd >= 0 ? rowptr_s[nd] : 0, dg = nd >= 0 ? min(rowptr_s[nd + 1] - rp, kSlotWidth) : 0;
  srcpos[idx] = j < dg ? xpos_s[rp + j] : 0;
  if (j == 0) srcdeg[r] = dg;
}

int32_t launch_vmh_srcpos(const ngpde_graph *g, int *srcpos, int *srcdeg, hipStream_t stream) {
  const int n = g->n_sched;
  if (n == 0) return NGPDE_OK;
  hipLaunchKernelGGL(vmh_srcpos_kernel, dim3((n * kSlotWidth + 255) / 256), dim3(256), 0, stream, g->by_t.sched, n, g->by_s.rowptr, g->by_s.xpos, srcpos, srcdeg);
  NGPDE_LAUNCH_CHECK("vmh_srcpos_kernel");
  return NGPDE_OK;
}

int32_t launch_vmh_copy_block(const float *src, int sp, float *dst, int dp, int rows, int cols, hipStream_t stream) {
  if (rows * cols == 0) return NGPDE_OK;
  hipLaunchKernelGGL(vmh_copy_block_kernel, dim3((rows * cols + 255) / 256), dim3(256), 0, stream, src, sp, dst, dp, rows, cols);
  NGPDE_LAUNCH_CHECK("vmh_copy_block_kernel");
  return NGPDE_OK;
}

}  // namespace ngpde

#ifdef NGPDE_STAMPS
extern "C" int32_t ngpde_debug_set_vmh_stamps(unsigned long long *dev_buf, int32_t max_phases) {
  ngpde::g_vst_base = dev_buf;
  ngpde::g_vst_max = max_phases;
  return 0;
}
#endif
