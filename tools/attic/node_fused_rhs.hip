// node_fused_rhs.hip -- the persistent solver with ONE hand-off per right-hand-side evaluation: both GCNConv layers of
//   du/dt = Chain(GCNConv(64 => 64, relu), GCNConv(64 => 64, relu))(u)
// evaluated per tile from a 2-hop halo.
// [caller of the hot path in the reference: docs/src/tutorials/graph_node.md:44-66, :78; layer: src/layers.jl:200-239]
//
// Why.  node_persistent.hip exchanges rows between tiles after EVERY layer: 600 hand-offs per direction of a 50-step Tsit5 solve,
// and a hand-off (drain -> flag -> detect -> gather across the fabric) costs ~2.4 us that nothing of the same tile can run under
// (DESIGN.md 5.1, 9 item 4).  Here a tile gathers the stage input x on its 2-HOP halo H2 (every row its 1-hop halo rows
// reference: <= 160 rows, ~90 on the BASELINE graph), evaluates layer 1 on its 1-hop halo H1 (own rows + the <= 64 rows they
// reference: the rows layer 2 needs), keeps those layer-1 outputs in LDS, and evaluates layer 2 on its own rows -- one exchange per
// right-hand side (300 per direction), for ~1.8 x the layer-1 work per tile.  The adjoint is the mirror image: the exchanged array
// is c .* (dZ2 W2^T); a tile gathers it on H2, forms dL/dy1 and dZ1 on H1 (the relu sign bits of foreign rows come from the
// owners' forward masks, which are indexed by NODE here), G1 = dZ1 W1^T on H1, then the stage adjoint, K-bar and layer 2's dense
// half on its own rows.  Parameter gradients only ever sum a tile's OWN rows.
//
// LDS is what the 2-hop form needs (40 KB of halo rows + a 96-row operand tile), so W leaves it: in the forward both W^T are B
// fragments in registers for the whole launch (16 + 16 per lane), in the adjoint the fragments of the product at hand are fetched
// from the cache hierarchy (16 KB per W, resident) right before it.
//
// Arithmetic is that of node_persistent.hip operation for operation -- a row's neighbours are summed in the row's own CSR order
// whichever tile does it, the MFMA products contract in the same order -- so u(T), du0 and the parameter gradients are BITWISE
// equal to the one-hop persistent plan (tests/test_gcn_gpu.py), which is the test of the synchronisation.
//
// Synchronisation: as node_persistent.hip (per-tile phase flags, write-through row stores, sc1 LDS-DMA gathers, bounded spins,
// abort word), with the wait lists taken over the 2-hop halos and the exchanged array ping-ponging between two buffers: phase ph
// reads buffer (ph - 1) & 1 and writes buffer ph & 1, and a tile that has seen its neighbours finish phase ph - 1 knows they are
// done reading the buffer it is about to overwrite.
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <unordered_map>

#include "common.h"
#include "device_utils.h"
#include "gcn_tile.h"
#include "persistent_mem.h"

namespace ngpde {

namespace {

constexpr int PD = 64;
constexpr int kH2Cap = kHop2Cap;          // rows of a tile's 2-hop halo (common.h)
constexpr int kRegRows = kH2Cap + 6;      // LDS row region: [zero row][kH2Cap slots] + room for two 32 x 68 operand tiles behind row 96
constexpr int kRegF = kRegRows * PD;
constexpr int kTS = PD + 4;               // operand tile row stride (gcn_tile.h: Geo<64>::TS)
constexpr int kTF = kHaloCap * kTS;       // the 96-row operand tile (layer-1 inputs / dZ1)
constexpr int kXT = 97 * PD;              // adjoint: offset of the two own-row operand tiles (A1, A2) in the row region
constexpr int kNbr = 64;                  // wait-list stride per tile; lane 63 of the polling wave watches the abort word
constexpr int kTabF = kHaloCap * 8 + kH2Cap + kHaloCap + 2 * PD + 48 + 8;   // slot words, H2 node ids, c of the H1 rows, biases, coefficients, flags
static_assert(kXT + 2 * kTM * kTS <= kRegF, "operand tiles fit the row region");
static_assert((kRegF + kTF + kTabF) * 4 <= 80 * 1024 - 64, "two workgroups per CU");
static_assert(Geo<PD>::TS == kTS && Geo<PD>::LPR == 16 && Geo<PD>::GROUPS == kTM, "one 16-lane group per tile row");

#ifdef NGPDE_STAMPS
// diagnostic build only (tools/stamps_fused.py): shader-clock stamps of thread 0 at 8 points of the first stamps_max phases
#define NGPDE_FUSED_STAMP_FIELD unsigned long long *stamps; int stamps_max;
#define NGPDE_FST(m, ph, k)                                                                                   \
  do {                                                                                                        \
    if (threadIdx.x == 0 && (m).stamps && (ph) <= (m).stamps_max)                                             \
      (m).stamps[((size_t)blockIdx.x * (m).stamps_max + ((ph) - 1)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define NGPDE_FUSED_STAMP_FIELD
#define NGPDE_FST(m, ph, k)
#endif

struct FusedMeta {            // one direction of the graph (by target: forward; by source: adjoint)
  const unsigned *slots2;     // [n_tiles][96][8] slot bytes of the H1 rows: H2 slot + 1, 0 = the all-zero row
  const int *hnode;           // [n_tiles][kH2Cap] node of every H2 slot (own rows, then H1, then the rest), 0 padded
  const int2 *info;           // [n_tiles] {rows of H1, rows of H2}
  const uint8_t *deg;         // [n_tiles][96] list length of every H1 row
  const int2 *halo;           // the handle's 1-hop halo list {node, bits of c}: c of the H1 rows
  const int4 *sched;
  const int *nbr;             // [n_tiles][64] wait lists over the 2-hop halos of both directions
  unsigned *flags, *abort_word;
  int n_tiles;
  NGPDE_FUSED_STAMP_FIELD
};

struct Ctx {
  int tid, lane, wave_u, grp, q, tile, node, h1, h2, my_nbr;
  int wmax[3];                // per pass of 32 H1 rows: longest list among the wave's four rows (wave-uniform)
  bool valid;
  float ci;
  unsigned own;               // byte offset of this thread's 16 bytes of its own row in a [N][64] array
};

struct Tabs {                 // the tile's tables in LDS for the whole launch
  unsigned *slots;            // [96][8]
  int *hnode;                 // [kH2Cap]
  float *c1;                  // [96] c of the H1 rows (0 beyond H1 and for padding rows)
  float *bias;                // [2][64] (forward)
  float *coef;                // [48]
  int *s_ok;
};

__device__ __forceinline__ Tabs carve_tabs(float *base) {
  Tabs t;
  t.slots = reinterpret_cast<unsigned *>(base);
  t.hnode = reinterpret_cast<int *>(base + kHaloCap * 8);
  t.c1 = base + kHaloCap * 8 + kH2Cap;
  t.bias = t.c1 + kHaloCap;
  t.coef = t.bias + 2 * PD;
  t.s_ok = reinterpret_cast<int *>(t.coef + 48);
  return t;
}

__device__ __forceinline__ void ctx_init(const FusedMeta &m, Ctx &c, const Tabs &t) {
  c.tid = threadIdx.x;
  c.lane = c.tid & 63;
  c.wave_u = __builtin_amdgcn_readfirstlane(c.tid >> 6);
  c.grp = c.tid >> 4;
  c.q = c.tid & 15;
  c.tile = xcd_tile(blockIdx.x, m.n_tiles);
  const int4 sc = m.sched[(size_t)c.tile * kTM + c.grp];
  c.valid = sc.x >= 0;
  c.node = max(sc.x, 0);
  c.ci = c.valid ? __int_as_float(sc.w) : 0.f;
  c.own = (unsigned)c.node * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
  const int2 inf = m.info[c.tile];
  c.h1 = __builtin_amdgcn_readfirstlane(inf.x);
  c.h2 = __builtin_amdgcn_readfirstlane(inf.y);
  for (int idx = c.tid; idx < kHaloCap * 8; idx += kThreads) t.slots[idx] = m.slots2[(size_t)c.tile * kHaloCap * 8 + idx];
  if (c.tid < kH2Cap) t.hnode[c.tid] = m.hnode[(size_t)c.tile * kH2Cap + c.tid];
  if (c.tid < kHaloCap) {
    const int2 he = m.halo[(size_t)c.tile * kHaloCap + c.tid];
    t.c1[c.tid] = c.tid < c.h1 ? __int_as_float(he.y) : 0.f;   // (padding rows of the last tile carry {0, 0})
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int r = 32 * k + c.grp;
    int wm = r < c.h1 ? (int)m.deg[(size_t)c.tile * kHaloCap + r] : 0;
    wm = max(wm, __shfl_xor(wm, 16));
    wm = max(wm, __shfl_xor(wm, 32));
    c.wmax[k] = __builtin_amdgcn_readfirstlane(wm);
  }
  c.my_nbr = m.nbr[(size_t)c.tile * kNbr + c.lane];
  if (c.tid == 0) *t.s_ok = 1;
}

// Wait until every tile of the wait list has finished phase ph - 1 (node_persistent.hip: tile_wait).  Bounded: abort word + ~2 s.
__device__ __forceinline__ bool fused_wait(const FusedMeta &m, const Ctx &c, int ph, int *s_ok) {
  if (ph <= 1) return true;
  if (c.wave_u == 0) {
    const unsigned need = (unsigned)(ph - 1);
    const unsigned *addr = (c.lane == 63) ? m.abort_word : (c.my_nbr >= 0 ? m.flags + 32 * c.my_nbr : nullptr);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    for (unsigned it = 1;; ++it) {
      unsigned f = need;
      if (addr) f = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__any((int)(c.lane == 63 && f != 0))) { ok = false; break; }
      if (__all((int)(c.lane == 63 || f >= need))) break;
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

// every storing wave drains, the workgroup meets, ONE lane publishes
__device__ __forceinline__ void fused_publish(const FusedMeta &m, const Ctx &c, int ph) {
  wait_vmcnt0();
  __syncthreads();
  if (c.tid == 0) __hip_atomic_store(m.flags + 32 * c.tile, (unsigned)ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// rows of OTHER tiles in this tile's 2-hop halo: memory -> LDS rows 1 + 32 .., sc1 (stored write-through by their owners in the
// previous phase).  A wave's four 16-lane groups stage four consecutive slots: its 1 KiB lands lane-linearly.
__device__ __forceinline__ void gather_h2(const Ctx &c, const Tabs &t, const float *X, float *reg) {
  float4 *R4 = reinterpret_cast<float4 *>(reg);
#pragma unroll
  for (int k = 1; k < kH2Cap / 32; ++k) {
    if (4 * c.wave_u + 32 * k < c.h2) {   // wave-uniform
      const unsigned off = (unsigned)t.hnode[c.grp + 32 * k] * (unsigned)(PD * 4) + (unsigned)(c.q * 16);
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(X) + off),
                                       (__attribute__((address_space(3))) void *)(R4 + (1 + c.grp + 32 * k) * 16 + c.q), 16, 0, 16);
    }
  }
  wait_vmcnt0();
  __syncthreads();
}

// sum of row r's neighbours (its slot bytes, CSR order, four at a time) + row r itself (self loop), all from the LDS row region:
// node_persistent.hip's tile_aggregate, for any row of H1
__device__ __forceinline__ float4 agg_row(const Tabs &t, const float *reg, int r, int q, int wmax) {
  const float4 *R4 = reinterpret_cast<const float4 *>(reg);
  float4 a = f4_zero();
#pragma unroll
  for (int jw = 0; jw < 8; ++jw) {
    if (jw * 4 < wmax) {   // wave-uniform
      const unsigned w = t.slots[r * 8 + jw];
      float4 v[4];
#pragma unroll
      for (int jb = 0; jb < 4; ++jb) v[jb] = R4[((w >> (8 * jb)) & 0xff) * 16 + q];
      a = f4_add(a, f4_add(f4_add(v[0], v[1]), f4_add(v[2], v[3])));
    }
  }
  return f4_add(a, R4[(1 + r) * 16 + q]);
}

// one 16 x 16 output block: rows rb * 16 .. + 15 of the operand tile times the wave's 16 columns of B, B as fragments in
// registers (b[4 kb + r] = B[k = 16 kb + 4 kq + r][column]): gcn_tile.h's mfma_rows_times_bt with the same contraction order
__device__ __forceinline__ f32x4 mfma_block(const float *tile, int rb, int lane, const float (&b)[16]) {
  const int i = lane & 15, kq = lane >> 4;
  const float *pa = tile + (rb * 16 + i) * kTS + 4 * kq;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 a_cur = *reinterpret_cast<const float4 *>(pa);
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    float4 a_nxt = a_cur;
    if (kb + 1 < 4) a_nxt = *reinterpret_cast<const float4 *>(pa + (kb + 1) * 16);
    acc = mfma16(a_cur.x, b[4 * kb + 0], acc);
    acc = mfma16(a_cur.y, b[4 * kb + 1], acc);
    acc = mfma16(a_cur.z, b[4 * kb + 2], acc);
    acc = mfma16(a_cur.w, b[4 * kb + 3], acc);
    a_cur = a_nxt;
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------------------------------
// forward solve
// ---------------------------------------------------------------------------------------------------------------------
struct PFwdF {
  FusedMeta m;          // lists by TARGET
  int n_steps, S, n_members;
  const float *u_in;    // [n_members][N][64]  c .* u0
  float *u_out;         // [n_members][N][64]  c .* u(T)
  float *buf0, *buf1;   // the exchanged stage input, ping-pong
  const float *w1, *b1, *w2, *b2;
  float *tape;          // [n_members][n_steps][S][2][N][64] aggregated layer inputs of the OWN rows, or null (forward-only plan)
  uint8_t *masks;       // [n_members][n_steps][S][2][mask_bytes] relu sign bits, 8 bytes per NODE
  size_t row_elems, mask_bytes;
  const float *cf;      // forward coefficient table [42] (node_persistent.hip)
};

template <bool TAPE>
__global__ __launch_bounds__(kThreads, 4) void node_fwd_fused_kernel(const PFwdF p) {
  __shared__ __attribute__((aligned(16))) float lds[kRegF + kTF + kTabF];
  float *reg = lds, *ldsT = lds + kRegF;
  const Tabs t = carve_tabs(ldsT + kTF);
  Ctx c;
  ctx_init(p.m, c, t);
  if (c.tid < 42) t.coef[c.tid] = p.cf[c.tid];
  if (c.tid < PD) t.bias[c.tid] = p.b1 ? p.b1[c.tid] : 0.f;
  else if (c.tid < 2 * PD) t.bias[c.tid] = p.b2 ? p.b2[c.tid - PD] : 0.f;
  float4 *R4 = reinterpret_cast<float4 *>(reg);
  if (c.grp == 0) R4[c.q] = f4_zero();   // row 0: the all-zero row
  // the wave's 16 output columns of both W^T as B fragments: lane (i, kq) holds W[in = 16 kb + 4 kq + r][out = 16 ct + i]
  const int i16 = c.lane & 15, kq = c.lane >> 4, ct = c.wave_u & 3, rb0 = c.wave_u >> 2;
  float bw1[16], bw2[16];
#pragma unroll
  for (int kb = 0; kb < 4; ++kb)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      bw1[4 * kb + r] = p.w1[(16 * kb + 4 * kq + r) * PD + 16 * ct + i16];
      bw2[4 * kb + r] = p.w2[(16 * kb + 4 * kq + r) * PD + 16 * ct + i16];
    }
  __syncthreads();
  const float bias1 = t.bias[16 * ct + i16];
  const float4 bias2 = reinterpret_cast<const float4 *>(t.bias + PD)[c.q];
  const int nrb = (c.h1 + 15) >> 4;        // 16-row blocks of the layer-1 product (2 .. 6)
  bool ok = true;
  int ph = 0;   // phases count on across the members
  for (int mb = 0; mb < p.n_members; ++mb) {
  const float *u_in = p.u_in + (size_t)mb * p.row_elems;
  const size_t ev0 = (size_t)mb * p.n_steps * p.S * 2;
  float4 u = f4_sel(c.valid && ok, ld4_g(u_in, c.own), f4_zero());
  float4 k0 = f4_zero(), k1 = f4_zero(), k2 = f4_zero(), k3 = f4_zero(), k4 = f4_zero(), k5 = f4_zero();
  R4[(1 + c.grp) * 16 + c.q] = u;   // (nobody reads the row region between a publish and the next gather's barrier)
  for (int n = 0; n < p.n_steps && ok; ++n) {
    for (int i = 0; i < p.S && ok; ++i) {
      ++ph;
      const float *X = (n == 0 && i == 0) ? u_in : ((ph - 1) & 1 ? p.buf1 : p.buf0);
      float *Xn = (ph & 1) ? p.buf1 : p.buf0;
      const size_t ev = ev0 + (size_t)(n * p.S + i) * 2;
      NGPDE_FST(p.m, ph, 0);
      if (!fused_wait(p.m, c, ph, t.s_ok)) { ok = false; break; }
      NGPDE_FST(p.m, ph, 1);
      gather_h2(c, t, X, reg);
      NGPDE_FST(p.m, ph, 2);
      // ---- layer 1 on H1: a_r = c_r * (sum of the stored, pre-scaled rows)
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        if (32 * k + 4 * c.wave_u < 16 * nrb) {   // wave-uniform
          const int r = 32 * k + c.grp;
          const float4 a = f4_scale(t.c1[r], agg_row(t, reg, r, c.q, c.wmax[k]));
          const float4 a1 = f4_sel(r < c.h1, a, f4_zero());
          *reinterpret_cast<float4 *>(&ldsT[r * kTS + 4 * c.q]) = a1;
          if (TAPE && k == 0 && c.valid) st4_stream_g(p.tape + ev * p.row_elems, c.own, a1);
        }
      }
      __syncthreads();
      NGPDE_FST(p.m, ph, 3);
      // product + epilogue in the accumulator layout (lane (i, kq): rows 4 kq + reg of the block, column 16 ct + i): bias, relu,
      // c_r; the layer-1 outputs of H1 replace the stage input in the row region (every wave is past its aggregation)
      for (int rb = rb0; rb < nrb; rb += 2) {
        const f32x4 acc = mfma_block(ldsT, rb, c.lane, bw1);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int row = rb * 16 + 4 * kq + rg;
          const float z = acc[rg] + bias1;
          if (row < c.h1) reg[(1 + row) * PD + 16 * ct + i16] = t.c1[row] * fmaxf(z, 0.f);
          if (TAPE && rb < 2) {   // own rows: 16 sign bits per (row, column tile), 8 bytes per node
            const unsigned long long bal = __ballot(z > 0.f);
            const int nd = t.hnode[row];   // (-1: padding row of the last tile)
            if (i16 == 0 && nd >= 0)
              *reinterpret_cast<NGPDE_GLOBAL_AS unsigned short *>(reinterpret_cast<uintptr_t>(p.masks + ev * p.mask_bytes) + (unsigned)nd * 8u +
                                                                   2u * ct) = (unsigned short)(bal >> (16 * kq));
          }
        }
      }
      __syncthreads();
      NGPDE_FST(p.m, ph, 4);
      // ---- layer 2 on the own rows
      {
        const float4 a2 = f4_scale(c.ci, agg_row(t, reg, c.grp, c.q, c.wmax[0]));
        *reinterpret_cast<float4 *>(&ldsT[c.grp * kTS + 4 * c.q]) = a2;
        if (TAPE && c.valid) st4_stream_g(p.tape + (ev + 1) * p.row_elems, c.own, a2);
      }
      __syncthreads();
      NGPDE_FST(p.m, ph, 5);
      {
        const f32x4 acc = mfma_block(ldsT, rb0, c.lane, bw2);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) ldsT[(kTM + rb0 * 16 + 4 * kq + rg) * kTS + 16 * ct + i16] = acc[rg];
      }
      __syncthreads();
      NGPDE_FST(p.m, ph, 6);
      const float4 z = f4_add(*reinterpret_cast<const float4 *>(&ldsT[(kTM + c.grp) * kTS + 4 * c.q]), bias2);
      const unsigned sign_bits = (z.x > 0.f ? 1u : 0u) | (z.y > 0.f ? 2u : 0u) | (z.z > 0.f ? 4u : 0u) | (z.w > 0.f ? 8u : 0u);
      const float4 yv = f4_sel(c.valid, f4_scale(c.ci, f4_act(NGPDE_ACT_RELU, z)), f4_zero());
      // k_i = yv; next stage input (or the step update) = u + sum_j cf[i][j] k_j, in the replayed plan's order
      k0 = f4_sel(i == 0, yv, k0); k1 = f4_sel(i == 1, yv, k1); k2 = f4_sel(i == 2, yv, k2);
      k3 = f4_sel(i == 3, yv, k3); k4 = f4_sel(i == 4, yv, k4); k5 = f4_sel(i == 5, yv, k5);
      float4 v = f4_scale(t.coef[36 + i], yv);
      v = f4_fma(1.0f, u, v);
      v = f4_fma(t.coef[i * 6 + 0], k0, v); v = f4_fma(t.coef[i * 6 + 1], k1, v); v = f4_fma(t.coef[i * 6 + 2], k2, v);
      v = f4_fma(t.coef[i * 6 + 3], k3, v); v = f4_fma(t.coef[i * 6 + 4], k4, v);
      if (i == p.S - 1) u = v;
      if (c.valid) store_sc1(Xn, c.own, v);
      R4[(1 + c.grp) * 16 + c.q] = v;
      NGPDE_FST(p.m, ph, 7);
      fused_publish(p.m, c, ph);
      if (TAPE) {   // layer 2's sign bits, same node-indexed layout; only the adjoint launch reads them
        const unsigned hi = __shfl_down(sign_bits, 1);
        if (c.valid && (c.q & 1) == 0) stu8_g(p.masks + (ev + 1) * p.mask_bytes, (unsigned)c.node * 8u + (unsigned)(c.q >> 1), (uint8_t)(sign_bits | (hi << 4)));
      }
    }
  }
  // (a tile writes its rows of u(T) only after all readers of its u0 rows are past that member's first phase: the host
  // takes this plan only for solves of at least two right-hand-side evaluations)
  if (c.valid) st4_g(p.u_out + (size_t)mb * p.row_elems, c.own, f4_sel(ok, u, f4_nan()));
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// discrete adjoint
// ---------------------------------------------------------------------------------------------------------------------
struct PBwdF {
  FusedMeta m;          // lists by SOURCE
  int n_steps, S, n_members;
  float *lam;           // [n_members][N][64] in: dL/du~(T); out: dL/du~0
  float *g0, *g1;       // the exchanged array c .* (dZ2 W2^T), ping-pong
  const float *w1, *w2;
  const float *tape;
  const uint8_t *masks;
  size_t row_elems, mask_bytes;
  float *slab_dw1, *slab_db1, *slab_dw2, *slab_db2;   // [n_tiles][...] written once, at the end
  const float *cb;      // adjoint coefficient table [48] (node_persistent.hip)
};

__global__ __launch_bounds__(kThreads, 4) void node_bwd_fused_kernel(const PBwdF p) {
  __shared__ __attribute__((aligned(16))) float lds[kRegF + kTF + kTabF];
  float *reg = lds, *ldsD = lds + kRegF;
  float *ldsX1 = reg + kXT, *ldsX2 = ldsX1 + kTM * kTS;        // own rows of the saved layer inputs (behind row 96 of the row region)
  float *ldsDZ2 = ldsD + kTM * kTS, *ldsGo = ldsD + 2 * kTM * kTS;   // rows 32..63: dZ2; rows 64..95: layer 2's product
  const Tabs t = carve_tabs(ldsD + kTF);
  Ctx c;
  ctx_init(p.m, c, t);
  if (c.tid < 48) t.coef[c.tid] = p.cb[c.tid];
  float4 *R4 = reinterpret_cast<float4 *>(reg);
  if (c.grp == 0) R4[c.q] = f4_zero();
  const int i16 = c.lane & 15, kq = c.lane >> 4, ct = c.wave_u & 3, rb0 = c.wave_u >> 2;
  constexpr int NT = Geo<PD>::CT * Geo<PD>::CT, DWT = Geo<PD>::DWT, WAVES = Geo<PD>::WAVES, DBP = Geo<PD>::DBP;
  f32x4 dw1[DWT], dw2[DWT];
#pragma unroll
  for (int mm = 0; mm < DWT; ++mm) dw1[mm] = dw2[mm] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float db1 = 0.f, db2 = 0.f;
  const int dbc = c.tid / DBP, dbpart = c.tid % DBP;
  const int S = p.S;
  const int nrb = (c.h1 + 15) >> 4;
  __syncthreads();

  // the wave's 16 columns of a straight W (B^T[j = in][k = out] = W[j][k]) as fragments: four 16-byte loads per lane from the
  // cache hierarchy; `w` passes through an empty asm so that the loads stay inside the phase (hoisted, the 32 registers of both
  // W would live across the whole launch)
  auto w_frags = [&](const float *w, float (&b)[16]) {
    asm volatile("" : "+s"(w));
    const float4 *w4 = reinterpret_cast<const float4 *>(w + (16 * ct + i16) * PD + 4 * kq);
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const float4 v = w4[4 * kb];
      b[4 * kb + 0] = v.x; b[4 * kb + 1] = v.y; b[4 * kb + 2] = v.z; b[4 * kb + 3] = v.w;
    }
  };
  // dWt[i][o] += sum_n A[n][i] dZ[n][o] over the tile's 32 own rows; db += column sums of dZ (node_persistent.hip)
  auto dw_products = [&](const float *tX, const float *tDZ, f32x4 (&dwl)[DWT], float &dbl) {
#pragma unroll
    for (int mm = 0; mm < DWT; ++mm) {
      const int tt = c.wave_u + WAVES * mm;
      if (tt < NT) {   // wave-uniform
        const int mt = tt / Geo<PD>::CT, nt = tt % Geo<PD>::CT;
#pragma unroll
        for (int kh = 0; kh < 2; ++kh) {
          float a[kTM / 8], b[kTM / 8];
#pragma unroll
          for (int ks = 0; ks < kTM / 8; ++ks) {
            a[ks] = tX[(4 * (ks + 4 * kh) + kq) * kTS + mt * 16 + i16];
            b[ks] = tDZ[(4 * (ks + 4 * kh) + kq) * kTS + nt * 16 + i16];
          }
#pragma unroll
          for (int ks = 0; ks < kTM / 8; ++ks) dwl[mm] = mfma16(a[ks], b[ks], dwl[mm]);
        }
      }
    }
    float s = 0.f;
#pragma unroll
    for (int nn = dbpart; nn < kTM; nn += DBP) s += tDZ[nn * kTS + dbc];
#pragma unroll
    for (int o = 1; o < DBP; o <<= 1) s += __shfl_xor(s, o);
    dbl += s;
  };
  // four relu sign bits of (node, columns 4 q .. 4 q + 3) of one (evaluation, layer)
  auto mask_of = [&](size_t ev, int node) -> unsigned {
    return (ldu8_g(p.masks + ev * p.mask_bytes, (unsigned)node * 8u + (unsigned)(c.q >> 1)) >> (4 * (c.q & 1))) & 0xfu;
  };
  auto masked = [&](unsigned mk, float4 kb) {
    return make_float4((mk & 1u) ? kb.x : 0.f, (mk & 2u) ? kb.y : 0.f, (mk & 4u) ? kb.z : 0.f, (mk & 8u) ? kb.w : 0.f);
  };
  // layer 2's dense half on the own rows: dZ2 = relu' .* (c K-bar), G2 = dZ2 W2^T, c .* G2 stored for the neighbours, published;
  // then the parameter-gradient products of this phase (nobody waits for them)
  float4 gown = f4_zero();
  auto dense2 = [&](int ph, float4 kbar, unsigned mk2, float4 x2, bool with_dw1) {
    const float4 dz = c.valid ? masked(mk2, f4_scale(c.ci, kbar)) : f4_zero();
    *reinterpret_cast<float4 *>(&ldsDZ2[c.grp * kTS + 4 * c.q]) = dz;
    *reinterpret_cast<float4 *>(&ldsX2[c.grp * kTS + 4 * c.q]) = f4_sel(c.valid, x2, f4_zero());
    float bw[16];
    w_frags(p.w2, bw);
    __syncthreads();
    NGPDE_FST(p.m, ph, 5);
    {
      const f32x4 acc = mfma_block(ldsDZ2, rb0, c.lane, bw);
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) ldsGo[(rb0 * 16 + 4 * kq + rg) * kTS + 16 * ct + i16] = acc[rg];
    }
    __syncthreads();
    NGPDE_FST(p.m, ph, 6);
    const float4 gv = f4_sel(c.valid, f4_scale(c.ci, *reinterpret_cast<const float4 *>(&ldsGo[c.grp * kTS + 4 * c.q])), f4_zero());
    if (c.valid) store_sc1((ph & 1) ? p.g1 : p.g0, c.own, gv);
    gown = gv;
    NGPDE_FST(p.m, ph, 7);
    fused_publish(p.m, c, ph);
    if (with_dw1) dw_products(ldsX1, ldsD, dw1, db1);
    dw_products(ldsX2, ldsDZ2, dw2, db2);
  };

  bool ok = true;
  int ph = 0;   // phases count on across the members (the parameter-gradient accumulators too)
  for (int mb = 0; mb < p.n_members; ++mb) {
  float *lam_g = p.lam + (size_t)mb * p.row_elems;
  const size_t ev0 = (size_t)mb * p.n_steps * S * 2;
  float4 lam = f4_sel(c.valid && ok, ld4_g(lam_g, c.own), f4_zero());
  float4 ub1 = f4_zero(), ub2 = f4_zero(), ub3 = f4_zero(), ub4 = f4_zero(), ub5 = f4_zero();
  if (ok) {   // first phase of a member: K-bar of the last stage of the last step = dt b_S lambda, layer 2's dense half only
    ++ph;
    const size_t ev = ev0 + (size_t)((p.n_steps - 1) * S + (S - 1)) * 2 + 1;
    const unsigned mk2 = mask_of(ev, c.node);
    const float4 x2 = ld4_stream_g(p.tape + ev * p.row_elems, c.own);
    if (!fused_wait(p.m, c, ph, t.s_ok)) ok = false;   // (the previous member's last readers of the buffer this phase writes)
    else dense2(ph, f4_scale(t.coef[S - 1], lam), mk2, x2, false);
  }
  for (int n = p.n_steps - 1; n >= 0 && ok; --n) {
    for (int i = S - 1; i >= 0 && ok; --i) {
      ++ph;
      const bool last = (i == 0 && n == 0);
      const size_t ev1 = ev0 + (size_t)(n * S + i) * 2;
      const size_t ev2 = ev0 + ((i >= 1) ? (size_t)(n * S + i - 1) * 2 + 1 : (size_t)((max(n, 1) - 1) * S + (S - 1)) * 2 + 1);
      NGPDE_FST(p.m, ph, 0);
      // loads that need no neighbour, in flight during the wait: the own tape rows, the sign bits of every H1 row of this thread
      unsigned mks = 0;   // four bits per H1 row of this thread (passes 0..2), then layer 2's of the own row
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int r = 32 * k + c.grp;
        const unsigned mv = mask_of(ev1, k == 0 ? c.node : (r < c.h1 ? t.hnode[r] : 0));   // unconditional load, clamped address
        mks |= ((r < c.h1 && (k > 0 || c.valid)) ? mv : 0u) << (4 * k);
      }
      const float4 x1 = ld4_stream_g(p.tape + ev1 * p.row_elems, c.own);
      float4 x2 = f4_zero();
      if (!last) {
        mks |= mask_of(ev2, c.node) << 12;
        x2 = ld4_stream_g(p.tape + ev2 * p.row_elems, c.own);
      }
      if (!fused_wait(p.m, c, ph, t.s_ok)) { ok = false; break; }
      NGPDE_FST(p.m, ph, 1);
      R4[(1 + c.grp) * 16 + c.q] = gown;   // own rows of the array published last phase
      gather_h2(c, t, ((ph - 1) & 1) ? p.g1 : p.g0, reg);
      NGPDE_FST(p.m, ph, 2);
      // ---- layer 2's aggregation pullback on H1: dL/dy1_r = sum of the gathered rows; dZ1 = relu' .* (c_r dL/dy1_r)
#pragma unroll 1
      for (int k = 0; k < 3; ++k) {
        if (32 * k + 4 * c.wave_u < 16 * nrb) {   // wave-uniform
          const int r = 32 * k + c.grp;
          const float4 kb = f4_scale(t.c1[r], agg_row(t, reg, r, c.q, k == 0 ? c.wmax[0] : (k == 1 ? c.wmax[1] : c.wmax[2])));
          *reinterpret_cast<float4 *>(&ldsD[r * kTS + 4 * c.q]) = masked(mks >> (4 * k), kb);   // (no bits for rows beyond H1 / padding rows)
        }
      }
      float bw[16];
      w_frags(p.w1, bw);
      __syncthreads();
      NGPDE_FST(p.m, ph, 3);
      // (rows 97.. of the row region are free from here on: the gathered rows have been summed)
      *reinterpret_cast<float4 *>(&ldsX1[c.grp * kTS + 4 * c.q]) = f4_sel(c.valid, x1, f4_zero());
      // G1 = dZ1 W1^T on H1, c_r .* G1 into the row region (H1 slots)
      for (int rb = rb0; rb < nrb; rb += 2) {
        const f32x4 acc = mfma_block(ldsD, rb, c.lane, bw);
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
          const int row = rb * 16 + 4 * kq + rg;
          if (row < c.h1) reg[(1 + row) * PD + 16 * ct + i16] = t.c1[row] * acc[rg];
        }
      }
      __syncthreads();
      NGPDE_FST(p.m, ph, 4);
      // ---- own rows: U-bar_i = A^T g1; K-bar of the stage evaluated before it (or the lambda update)
      const float4 tt = agg_row(t, reg, c.grp, c.q, c.wmax[0]);
      float4 kbar;
      if (i >= 1) {
        ub1 = f4_sel(i == 1, tt, ub1); ub2 = f4_sel(i == 2, tt, ub2); ub3 = f4_sel(i == 3, tt, ub3);
        ub4 = f4_sel(i == 4, tt, ub4); ub5 = f4_sel(i == 5, tt, ub5);
        float4 v = f4_scale(t.coef[42 + i], tt);
        v = f4_fma(t.coef[i - 1], lam, v);
        v = f4_fma(t.coef[6 + i * 6 + 2], ub2, v); v = f4_fma(t.coef[6 + i * 6 + 3], ub3, v);
        v = f4_fma(t.coef[6 + i * 6 + 4], ub4, v); v = f4_fma(t.coef[6 + i * 6 + 5], ub5, v);
        kbar = v;
      } else {
        float4 v = f4_scale(1.0f, tt);
        v = f4_fma(1.0f, lam, v);
        v = f4_fma(1.0f, ub1, v); v = f4_fma(1.0f, ub2, v); v = f4_fma(1.0f, ub3, v);
        v = f4_fma(1.0f, ub4, v); v = f4_fma(1.0f, ub5, v);
        lam = v;
        kbar = f4_scale(t.coef[S - 1], v);
      }
      if (last) {   // nothing to exchange; the flag still goes out (the next member's first phase waits for it)
        fused_publish(p.m, c, ph);
        dw_products(ldsX1, ldsD, dw1, db1);
        break;
      }
      dense2(ph, kbar, mks >> 12, x2, true);
    }
  }
  if (c.valid) st4_g(lam_g, c.own, f4_sel(ok, lam, f4_nan()));
  }
  // the tile's contribution to the parameter gradients: one slab per tile, summed by reduce_slabs_kernel
  const float bad = __int_as_float(0x7fc00000);
  auto write_slab = [&](const f32x4 (&dwl)[DWT], float dbl, float *slab_dw, float *slab_db) {
    float4 *slab4 = reinterpret_cast<float4 *>(slab_dw + (size_t)blockIdx.x * PD * PD);
#pragma unroll
    for (int mm = 0; mm < DWT; ++mm) {
      const int tt = c.wave_u + WAVES * mm;
      if (tt < NT) slab4[tt * 64 + c.lane] = f4_sel(ok, make_float4(dwl[mm][0], dwl[mm][1], dwl[mm][2], dwl[mm][3]), f4_nan());
    }
    if (dbpart == 0) slab_db[(size_t)blockIdx.x * PD + dbc] = ok ? dbl : bad;
  };
  write_slab(dw1, db1, p.slab_dw1, p.slab_db1);
  write_slab(dw2, db2, p.slab_dw2, p.slab_db2);
}

}  // namespace

// ---- host side -----------------------------------------------------------------------------------------------------------

namespace {

struct HostDir {            // what the builder needs of one direction, on the host
  std::vector<int32_t> rowptr, col;
  std::vector<int4> sched;
  std::vector<int2> halo, info;
};

int32_t fetch_dir(const ngpde_graph *g, const Csr &c, HostDir &h) {
  const size_t n = (size_t)g->n_nodes, m = (size_t)g->n_edges, ns = (size_t)g->n_sched, nt = ns / kTileRows;
  if (!c.h_rowptr.empty()) {
    h.rowptr = c.h_rowptr;
    h.col = c.h_col;
  } else {
    h.rowptr.resize(n + 1);
    h.col.resize(std::max<size_t>(m, 1));
    NGPDE_HIP_CHECK(hipMemcpy(h.rowptr.data(), c.rowptr, (n + 1) * 4, hipMemcpyDeviceToHost));
    if (m) NGPDE_HIP_CHECK(hipMemcpy(h.col.data(), c.col, m * 4, hipMemcpyDeviceToHost));
  }
  h.sched.resize(ns);
  h.halo.resize(nt * kHaloCap);
  h.info.resize(nt);
  NGPDE_HIP_CHECK(hipMemcpy(h.sched.data(), c.sched, ns * sizeof(int4), hipMemcpyDeviceToHost));
  NGPDE_HIP_CHECK(hipMemcpy(h.halo.data(), c.halo, h.halo.size() * sizeof(int2), hipMemcpyDeviceToHost));
  NGPDE_HIP_CHECK(hipMemcpy(h.info.data(), c.tile_info, nt * sizeof(int2), hipMemcpyDeviceToHost));
  return NGPDE_OK;
}

struct Hop2Host {
  std::vector<uint32_t> slots2;   // [nt][96][8]
  std::vector<int32_t> hnode;     // [nt][kH2Cap]
  std::vector<int2> info;         // [nt]
  std::vector<uint8_t> deg;       // [nt][96]
};

// The 2-hop halo of every tile: H2 = H1 (the handle's halo list, own rows first) followed by every other row the H1 rows
// reference, in order of first appearance walking the H1 rows in slot order and each row's list in CSR order; per H1 row its
// list as H2 slots + 1 (0: unused).  Returns false when a tile does not fit.
bool build_hop2(const ngpde_graph *g, const HostDir &h, Hop2Host &o) {
  const int nt = g->n_sched / kTileRows;
  o.slots2.assign((size_t)nt * kHaloCap * 8, 0u);
  o.hnode.assign((size_t)nt * kH2Cap, 0);
  o.info.assign((size_t)nt, make_int2(0, 0));
  o.deg.assign((size_t)nt * kHaloCap, 0);
  std::vector<int32_t> slot_of((size_t)g->n_nodes, -1), stamp((size_t)g->n_nodes, -1);
  for (int tl = 0; tl < nt; ++tl) {
    const int h1 = h.info[tl].x;
    if (h1 < kTileRows || h1 > kHaloCap) return false;
    int count = h1;
    for (int r = 0; r < h1; ++r) {
      const bool own = r < kTileRows;
      if (own && h.sched[(size_t)tl * kTileRows + r].x < 0) {   // padding row of the last tile
        o.hnode[(size_t)tl * kH2Cap + r] = -1;
        continue;
      }
      const int32_t v = h.halo[(size_t)tl * kHaloCap + r].x;
      stamp[v] = tl;
      slot_of[v] = r;
      o.hnode[(size_t)tl * kH2Cap + r] = v;
    }
    for (int r = 0; r < h1; ++r) {
      if (r < kTileRows && h.sched[(size_t)tl * kTileRows + r].x < 0) continue;
      const int32_t v = h.halo[(size_t)tl * kHaloCap + r].x;
      const int32_t rs = h.rowptr[v], dg = h.rowptr[v + 1] - rs;
      if (dg > kSlotWidth) return false;
      o.deg[(size_t)tl * kHaloCap + r] = (uint8_t)dg;
      uint8_t *bytes = reinterpret_cast<uint8_t *>(&o.slots2[((size_t)tl * kHaloCap + r) * 8]);
      for (int j = 0; j < dg; ++j) {
        const int32_t u = h.col[rs + j];
        if (stamp[u] != tl) {
          if (count >= kH2Cap) return false;
          stamp[u] = tl;
          slot_of[u] = count;
          o.hnode[(size_t)tl * kH2Cap + count] = u;
          ++count;
        }
        bytes[j] = (uint8_t)(slot_of[u] + 1);
      }
    }
    o.info[tl] = make_int2(h1, count);
  }
  return true;
}

template <class T>
int32_t upload_vec(T **dst, const std::vector<T> &v) {
  *dst = nullptr;
  NGPDE_HIP_CHECK(hipMalloc((void **)dst, std::max<size_t>(v.size(), 1) * sizeof(T)));
  if (!v.empty()) NGPDE_HIP_CHECK(hipMemcpy(*dst, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return NGPDE_OK;
}

// Opt-in (NGPDE_FUSED_RHS=1): measured SLOWER than the one-hop plan on the BASELINE graph -- the redundant
// layer-1 work on the 1-hop halo and the spread of that work between tiles cost more than the saved hand-off
bool fused_disabled_env() {
  const char *e = std::getenv("NGPDE_FUSED_RHS");
  return !(e && e[0] == '1');
}

#ifdef NGPDE_STAMPS
unsigned long long *g_fst_base = nullptr;
int g_fst_max = 0;
#endif

FusedMeta make_fmeta(const Csr &c, const Hop2Dev &d, const NodePersist &ps) {
  FusedMeta m;
  m.slots2 = d.slots2; m.hnode = d.hnode; m.info = d.info; m.deg = d.deg; m.halo = c.halo; m.sched = c.sched;
  m.nbr = ps.nbr2; m.flags = ps.sync; m.abort_word = ps.sync + (size_t)ps.n_tiles * 64; m.n_tiles = ps.n_tiles;
#ifdef NGPDE_STAMPS
  m.stamps = g_fst_base; m.stamps_max = g_fst_max;
#endif
  return m;
}

__global__ void fused_set_word_kernel(unsigned *w, unsigned v) {
  if (threadIdx.x == 0) *w = v;
}
__global__ void fused_latch_fault_kernel(const unsigned *abort_word, unsigned *fault) {
  if (threadIdx.x == 0 && *abort_word != 0) *fault = 1u;
}

}  // namespace

#ifdef NGPDE_STAMPS
extern "C" int32_t ngpde_debug_set_fused_stamps(unsigned long long *dev_buf, int32_t max_phases) {
  g_fst_base = dev_buf;   // [n_tiles][max_phases][8], or NULL
  g_fst_max = max_phases;
  return NGPDE_OK;
}
#endif

// Can the plan run with one hand-off per right-hand side?  The conditions of the one-tile-per-workgroup persistent plan
// (node_persistent_mode == 1: checked by the caller), relu, and every workgroup of BOTH fused kernels co-resident.
bool node_fused_rhs_possible(const ngpde_graph *g, int act, int n_evals) {
  if (fused_disabled_env() || !g || act != NGPDE_ACT_RELU || n_evals < 2 || g->by_t.slot_w || g->by_s.slot_w) return false;   // (unweighted graphs)
  int dev = 0, cus = 0;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  int occ = 1 << 30;
  auto take = [&](auto kernel) {
    int o = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kernel, kThreads, 0) != hipSuccess) o = 0;
    occ = std::min(occ, o);
  };
  take(node_fwd_fused_kernel<true>);
  take(node_fwd_fused_kernel<false>);
  take(node_bwd_fused_kernel);
  const int nt = g->n_sched / kTileRows;
  return nt >= 1 && nt <= cus * occ;
}

// The 2-hop tables of both directions and the wait lists over them.  NGPDE_ERR_UNSUPPORTED when a tile's 2-hop halo exceeds
// kHop2Cap rows or a wait list 63 tiles: the caller keeps the one-hop plan.
int32_t node_fused_setup(const ngpde_graph *g, NodePersist *ps) {
  const int nt = g->n_sched / kTileRows;
  HostDir hd[2];
  int32_t st;
  if ((st = fetch_dir(g, g->by_t, hd[0])) || (st = fetch_dir(g, g->by_s, hd[1]))) return st;
  std::vector<int32_t> tile_of((size_t)g->n_nodes, 0);
  for (int pos = 0; pos < g->n_sched; ++pos) {
    const int v = hd[0].sched[pos].x;
    if (v >= 0) tile_of[v] = pos / kTileRows;
  }
  std::vector<std::vector<int>> nb(nt);
  Hop2Host hh[2];
  for (int d = 0; d < 2; ++d) {
    NGPDE_REQUIRE(build_hop2(g, hd[d], hh[d]), NGPDE_ERR_UNSUPPORTED, "fused right-hand side: a tile's 2-hop halo exceeds %d rows", kH2Cap);
    for (int tl = 0; tl < nt; ++tl)
      for (int k = kTileRows; k < hh[d].info[tl].y; ++k) {
        const int u = tile_of[hh[d].hnode[(size_t)tl * kH2Cap + k]];
        if (u == tl) continue;
        nb[tl].push_back(u);
        nb[u].push_back(tl);
      }
  }
  std::vector<int> lists((size_t)nt * kNbr, -1);
  for (int tl = 0; tl < nt; ++tl) {
    std::sort(nb[tl].begin(), nb[tl].end());
    nb[tl].erase(std::unique(nb[tl].begin(), nb[tl].end()), nb[tl].end());
    NGPDE_REQUIRE((int)nb[tl].size() <= kNbr - 1, NGPDE_ERR_UNSUPPORTED, "fused right-hand side: a tile's wait list exceeds %d tiles", kNbr - 1);
    for (size_t k = 0; k < nb[tl].size(); ++k) lists[(size_t)tl * kNbr + k] = nb[tl][k];
  }
  for (int d = 0; d < 2; ++d) {
    Hop2Dev &dv = ps->hop2[d];
    if ((st = upload_vec(&dv.slots2, hh[d].slots2)) || (st = upload_vec(&dv.hnode, hh[d].hnode)) || (st = upload_vec(&dv.info, hh[d].info)) ||
        (st = upload_vec(&dv.deg, hh[d].deg)))
      return st;
  }
  return upload_vec(&ps->nbr2, lists);
}

void node_fused_free(NodePersist *ps) {
  for (int d = 0; d < 2; ++d) {
    Hop2Dev &dv = ps->hop2[d];
    if (dv.slots2) (void)hipFree(dv.slots2);
    if (dv.hnode) (void)hipFree(dv.hnode);
    if (dv.info) (void)hipFree(dv.info);
    if (dv.deg) (void)hipFree(dv.deg);
    dv = Hop2Dev();
  }
  if (ps->nbr2) (void)hipFree(ps->nbr2);
  ps->nbr2 = nullptr;
}

int32_t launch_node_fwd_fused(const NodePersistFwd &a, hipStream_t stream) {
  const ngpde_graph *g = a.g;
  const NodePersist &ps = *a.ps;
  NGPDE_REQUIRE(ps.nbr2 && a.act == NGPDE_ACT_RELU && !a.interleave && !a.pair && a.k_tiles == 0, NGPDE_ERR_STATE,
                "fused right-hand-side launch without its setup");
  NGPDE_REQUIRE(!a.tape || a.masks, NGPDE_ERR_INVALID_ARGUMENT, "fused forward with a tape needs the sign-bit masks");
  NGPDE_REQUIRE(!a.tape || a.mask_bytes >= (size_t)g->n_nodes * 8, NGPDE_ERR_INVALID_ARGUMENT, "fused forward: sign-bit masks too small");
  int32_t st;
  PersistentTurn turn;
  if ((st = turn.enter(stream))) return st;
  if ((st = launch_zero(ps.sync, ps.sync_bytes, stream))) return st;
  PFwdF k;
  k.m = make_fmeta(g->by_t, ps.hop2[0], ps);
  {
    const char *fa = std::getenv("NGPDE_DEBUG_FORCE_ABORT");
    if (fa && fa[0] == '1') hipLaunchKernelGGL(fused_set_word_kernel, dim3(1), dim3(64), 0, stream, k.m.abort_word, 1u);
  }
  k.n_steps = a.n_steps; k.S = a.S; k.n_members = a.n_members;
  k.u_in = a.u_in; k.u_out = a.u_out; k.buf0 = a.bufA; k.buf1 = a.bufB;
  k.w1 = a.w1; k.b1 = a.b1; k.w2 = a.w2; k.b2 = a.b2;
  k.tape = a.tape; k.masks = a.masks; k.row_elems = a.row_elems; k.mask_bytes = a.mask_bytes;
  k.cf = ps.coef;
  const dim3 grid(ps.n_tiles), block(kThreads);
  if (a.tape) {
    if (a.ev_start) hipExtLaunchKernelGGL(node_fwd_fused_kernel<true>, grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k);
    else hipLaunchKernelGGL(node_fwd_fused_kernel<true>, grid, block, 0, stream, k);
  } else {
    if (a.ev_start) hipExtLaunchKernelGGL(node_fwd_fused_kernel<false>, grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k);
    else hipLaunchKernelGGL(node_fwd_fused_kernel<false>, grid, block, 0, stream, k);
  }
  NGPDE_LAUNCH_CHECK("node_fwd_fused_kernel");
  hipLaunchKernelGGL(fused_latch_fault_kernel, dim3(1), dim3(64), 0, stream, k.m.abort_word, ps.fault);
  NGPDE_LAUNCH_CHECK("latch_fault_kernel");
  return turn.leave();
}

int32_t launch_node_bwd_fused(const NodePersistBwd &a, hipStream_t stream) {
  const ngpde_graph *g = a.g;
  const NodePersist &ps = *a.ps;
  NGPDE_REQUIRE(ps.nbr2 && a.act == NGPDE_ACT_RELU && !a.interleave && !a.pair && a.k_tiles == 0 && a.masks, NGPDE_ERR_STATE,
                "fused right-hand-side adjoint launch without its setup");
  int32_t st;
  PersistentTurn turn;
  if ((st = turn.enter(stream))) return st;
  if ((st = launch_zero(ps.sync, ps.sync_bytes, stream))) return st;
  PBwdF k;
  k.m = make_fmeta(g->by_s, ps.hop2[1], ps);
  k.n_steps = a.n_steps; k.S = a.S; k.n_members = a.n_members;
  k.lam = a.lam; k.g0 = a.g1; k.g1 = a.g2; k.w1 = a.w1; k.w2 = a.w2;
  k.tape = a.tape; k.masks = a.masks; k.row_elems = a.row_elems; k.mask_bytes = a.mask_bytes;
  k.slab_dw1 = a.slab_dw1; k.slab_db1 = a.slab_db1; k.slab_dw2 = a.slab_dw2; k.slab_db2 = a.slab_db2;
  k.cb = ps.coef + 42;
  const dim3 grid(ps.n_tiles), block(kThreads);
  if (a.ev_start) hipExtLaunchKernelGGL(node_bwd_fused_kernel, grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k);
  else hipLaunchKernelGGL(node_bwd_fused_kernel, grid, block, 0, stream, k);
  NGPDE_LAUNCH_CHECK("node_bwd_fused_kernel");
  hipLaunchKernelGGL(fused_latch_fault_kernel, dim3(1), dim3(64), 0, stream, k.m.abort_word, ps.fault);
  NGPDE_LAUNCH_CHECK("latch_fault_kernel");
  return turn.leave();
}

}  // namespace ngpde
