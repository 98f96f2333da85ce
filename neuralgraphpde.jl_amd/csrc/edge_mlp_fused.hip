// edge_mlp_fused.hip -- fused forward of the edge-function layers' message path
//   m_i = aggr_{e: t_e = i}  phi( [h_i; h_j; ...] )      (/root/reference/src/layers.jl:103-111, :313-326, :402-416)
// for message MLPs phi = Dense, Dense, ... up to 64 wide: one launch does
//   gather (LDS-staged distinct source rows of the tile) -> z1_e = P[t_e] + Q[s_e] + E_e -> act ->
//   the remaining Dense layers on fp32 MFMA with the weights resident in LDS -> in-tile segmented reduction,
// so no [E][h] array is written unless the caller asks for the pre-activations (training).  The reference
// materialises gather(x, t), gather(x, s), their vcat and every layer's activations over all E edges.
//
// A workgroup (8 waves) owns one 32-row tile of the locality schedule and walks the tile's edges in chunks of
// 64 (4 MFMA row tiles); per chunk and Dense layer each wave issues 32 v_mfma_f32_16x16x4_f32.  Row r of the
// tile is summed by lane group r in edge (= COO) order: no atomics, bitwise reproducible.
#include <algorithm>

#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

constexpr int kT = 512, kW = 64, kTS = kW + 4, kChunk = 128, kGroups = 32;   // kChunk = 8 waves x 16 edges

#ifdef NGPDE_STAMPS
// diagnostic build only (tools/stamps_edge.py): phase timestamps of each workgroup's FIRST tile, [n_blocks][16] words
unsigned long long *g_edge_stamps = nullptr;
#define EDGE_STAMP(k)                                                                                     \
  do {                                                                                                    \
    if (threadIdx.x == 0 && p.stamps && first_tile) {                                                     \
      p.stamps[(size_t)blockIdx.x * 16 + (k)] = clock64();                                                \
      if ((k) == 0 || (k) == 7) p.stamps[(size_t)blockIdx.x * 16 + 8 + ((k) ? 1 : 0)] = wall_clock64();  \
    }                                                                                                     \
  } while (0)
#else
#define EDGE_STAMP(k)
#endif

struct EdgeMlpK {
  const int4 *sched;
  const int2 *halo;
  const uint8_t *slots;
  int n_tiles, h1, act1, aggr, halo_rows;
  const float *P, *Q, *Eterm;
  int n_tail;
  int din[3], dout[3], act[3];
  const float *wt[3], *bias[3];
  float *out;
  float *save_z[4];   // [0]: z1 [E][h1]; [k]: pre-activation of tail layer k [E][dout_k]; nullable
#ifdef NGPDE_STAMPS
  unsigned long long *stamps;
#endif
};

__device__ __forceinline__ int xcd_tile(int b, int nb) {
  const int x = b % 8, k = b / 8;
  const int q = nb / 8, r = nb % 8;
  return x * q + min(x, r) + k;
}

__device__ __forceinline__ float4 load4_guard(const float *base, size_t row, int width, int q) {
  return (4 * q < width) ? *reinterpret_cast<const float4 *>(base + row * width + 4 * q) : f4_zero();
}

// tile metadata / rows of one tile held in registers between the moment they are fetched (under the previous tile's
// arithmetic) and the moment they are staged into LDS
struct TileMeta {
  int4 sc;
  uint4 s0, s1;
  int2 he[3];
};
struct TileRows {
  float4 prow, hv[3];
};

// Register-chained message MLP.  A wave owns 16 edges of the chunk; lane (i = lane & 15, kq = lane >> 4) holds, for edge i,
// the features {16 ct + 4 kq + r : ct = 0..3, r = 0..3} of the current activation as four float4.  A Dense layer is
// evaluated TRANSPOSED, z^T = W^T a^T, with v_mfma_f32_16x16x4_f32(A = W^T tile from LDS, B = a^T from registers): the D
// layout of that product (row = output feature 4 kq + r, column = edge i) is again exactly this register layout, so the
// layers chain in registers -- no LDS round trip and no barrier between assemble, Dense layers, bias and activation.
// MFMA step (ct, r) contracts, in lane kq, feature 16 ct + 4 kq + r of both operands (the order of a sum's terms is free).
template <int NTAIL>
__global__ __launch_bounds__(kT, (NTAIL <= 1 ? 4 : 2)) void edge_mlp_fused_fwd_kernel(const EdgeMlpK p) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  float *ldsQ = dyn;                                              // [halo_rows + 1][kTS]  rows of Q the tile references
  float *ldsP = ldsQ + (size_t)(p.halo_rows + 1) * kTS;           // [32][kTS]             P rows of the tile's targets
  float *ldsWt = ldsP + kGroups * kTS;                            // [NTAIL][64 out][kTS]  tail weights, W^T
  float *ldsMsg = ldsWt + (size_t)NTAIL * kW * kTS;               // [kChunk][kTS]         messages of the chunk
  __shared__ int ldsOff[kGroups + 1], ldsRs[kGroups];
  __shared__ __attribute__((aligned(16))) unsigned ldsSlots[kGroups * 8];   // 32 slot bytes per row
  __shared__ uint8_t ldsRowOf[kGroups * kSlotWidth];                        // tile edge k -> row of the tile
  __shared__ __attribute__((aligned(16))) float ldsBias[(NTAIL > 0 ? NTAIL : 1) * kW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = tid >> 4, q = tid & 15;           // staging / reduction role: group g <-> row g, lane q <-> features 4q..4q+3
  const int ei = lane & 15, kq = lane >> 4;         // MFMA role: edge ei of the wave's 16, k-quarter kq
  const int h1 = p.h1, zero_slot = p.halo_rows;

  // persistent workgroup: XCD x = blockIdx % 8 owns a contiguous range of tiles; its workgroups stride through it
  const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
  const int range_len = p.n_tiles / 8 + (xcd < p.n_tiles % 8 ? 1 : 0);
  const int range_lo = xcd * (p.n_tiles / 8) + min(xcd, p.n_tiles % 8);

  auto fetch_meta = [&](int tile, TileMeta &m) {
    m.sc = p.sched[(size_t)tile * kTileRows + grp];
    m.s0 = reinterpret_cast<const uint4 *>(p.slots)[((size_t)tile * kTileRows + grp) * 2];
    m.s1 = reinterpret_cast<const uint4 *>(p.slots)[((size_t)tile * kTileRows + grp) * 2 + 1];
#pragma unroll
    for (int k = 0; k < 3; ++k) m.he[k] = p.halo[(size_t)tile * kHaloCap + min(grp + k * kGroups, kHaloCap - 1)];
  };
  auto fetch_rows = [&](const TileMeta &m, TileRows &r) {
    r.prow = p.P ? load4_guard(p.P, max(m.sc.x, 0), h1, q) : f4_zero();
#pragma unroll
    for (int k = 0; k < 3; ++k)
      r.hv[k] = (p.Q && grp + k * kGroups < p.halo_rows) ? load4_guard(p.Q, m.he[k].x, h1, q) : f4_zero();
  };

  // ---- once per workgroup: tail weights as W^T rows (output j, contiguous inputs) and biases in LDS
  // (all loads of all layers first, unconditional from clamped addresses, pinned, then selected and written: a `cond ? load : 0` is an
  // exec-masked branch per load, and with one tile per workgroup -- a 3 000-node graph -- this prologue is most of the launch)
  {
    const int j = tid % kW, kg0 = tid / kW;   // output column j, input quads kg0 and kg0 + 8
    float t[NTAIL > 0 ? NTAIL : 1][2][4];
#pragma unroll
    for (int l = 0; l < NTAIL; ++l) {
      const int dinl = p.din[l], doutl = p.dout[l];
      const float *w = p.wt[l] + min(j, doutl - 1);
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int r = 0; r < 4; ++r) t[l][ps][r] = w[min(4 * (kg0 + 8 * ps) + r, dinl - 1) * doutl];
    }
#pragma unroll
    for (int l = 0; l < NTAIL; ++l)
#pragma unroll
      for (int ps = 0; ps < 2; ++ps)
#pragma unroll
        for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(t[l][ps][r]));
#pragma unroll
    for (int l = 0; l < NTAIL; ++l) {
      const int dinl = p.din[l], doutl = p.dout[l];
#pragma unroll
      for (int ps = 0; ps < 2; ++ps) {
        const int k = 4 * (kg0 + 8 * ps);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (k + r < dinl && j < doutl) ? t[l][ps][r] : 0.f;
        *reinterpret_cast<float4 *>(&ldsWt[l * kW * kTS + j * kTS + k]) = make_float4(v[0], v[1], v[2], v[3]);
      }
      if (tid < kW) ldsBias[l * kW + tid] = (p.bias[l] && tid < doutl) ? p.bias[l][tid] : 0.f;
    }
  }
  if (grp == 0) *reinterpret_cast<float4 *>(&ldsQ[zero_slot * kTS + 4 * q]) = f4_zero();   // the all-zero row

  TileMeta meta;
  TileRows rows;
  int jt = wg_in_xcd;
  if (jt < range_len) {
    fetch_meta(range_lo + jt, meta);
    fetch_rows(meta, rows);
  }
  const int last_w = (NTAIL > 0) ? p.dout[NTAIL - 1] : h1;

  for (; jt < range_len; jt += wgs_per_xcd) {
#ifdef NGPDE_STAMPS
    const bool first_tile = (jt == wg_in_xcd + 4 * wgs_per_xcd);   // a tile in steady state (the fifth of the workgroup)
#endif
    EDGE_STAMP(0);
    // ---- stage this tile (fetched under the previous tile's arithmetic)
    const int4 sc = meta.sc;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int hh = grp + k * kGroups;
      if (hh < p.halo_rows) *reinterpret_cast<float4 *>(&ldsQ[hh * kTS + 4 * q]) = rows.hv[k];
    }
    *reinterpret_cast<float4 *>(&ldsP[grp * kTS + 4 * q]) = rows.prow;
    if (q == 0) {
      ldsOff[grp + 1] = sc.x >= 0 ? sc.z : 0;   // degrees; turned into offsets below
      ldsRs[grp] = sc.y;
      if (grp == 0) ldsOff[0] = 0;
    }
    if (q < 8) {
      const unsigned w[8] = {meta.s0.x, meta.s0.y, meta.s0.z, meta.s0.w, meta.s1.x, meta.s1.y, meta.s1.z, meta.s1.w};
      unsigned v = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) v = (q == j) ? w[j] : v;
      ldsSlots[grp * 8 + q] = v;
    }
    // ---- next tile: metadata now (one L2 round trip, lands during the prefix sums), rows after the first chunk
    const int jn = jt + wgs_per_xcd;
    const bool has_next = jn < range_len;       // workgroup-uniform
    if (has_next) fetch_meta(range_lo + jn, meta);
    __syncthreads();
    if (tid < kGroups) {   // inclusive scan of the 32 degrees inside wave 0 (DPP shuffles, no LDS round trips)
      int v = ldsOff[tid + 1];
#pragma unroll
      for (int o = 1; o < kGroups; o <<= 1) {
        const int u = __shfl_up(v, o);
        if (tid >= o) v += u;
      }
      ldsOff[tid + 1] = v;
    }
    __syncthreads();
    const int total = ldsOff[kGroups];
    const int my_lo = ldsOff[grp], my_hi = ldsOff[grp + 1];
    for (int k = my_lo + q; k < my_hi; k += 16) ldsRowOf[k] = (uint8_t)grp;   // deg <= kSlotWidth: total <= 1024

    float4 racc;
    if (p.aggr == NGPDE_AGGR_MAX) racc = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    else if (p.aggr == NGPDE_AGGR_MIN) racc = make_float4(INFINITY, INFINITY, INFINITY, INFINITY);
    else if (p.aggr == NGPDE_AGGR_MUL) racc = make_float4(1.f, 1.f, 1.f, 1.f);   // scatter(*): the neutral element (an empty neighbourhood gives 1)
    else racc = f4_zero();
    __syncthreads();
    EDGE_STAMP(1);

    bool rows_fetched = false;
    for (int c0 = 0; c0 < total; c0 += kChunk) {
      // ---- this lane's edge: row, halo slot, position in p order (a wave past the tile's last edge only joins the barriers)
      const bool wave_on = c0 + wave * 16 < total;   // wave-uniform
      const int k = c0 + wave * 16 + ei;
      const bool valid = k < total;
      int r = 0, slot = zero_slot;
      size_t pe = 0;
      if (valid) {
        r = ldsRowOf[k];
        const int j = k - ldsOff[r];
        slot = (ldsSlots[r * 8 + (j >> 2)] >> (8 * (j & 3))) & 0xff;
        pe = (size_t)(ldsRs[r] + j);
      }
      // ---- a1 = act1(P[t] + Q[s] + E), features 16 ct + 4 kq .. + 3
      float4 a[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
      if (wave_on) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const int f = 16 * ct + 4 * kq;
        float4 z = f4_add(*reinterpret_cast<const float4 *>(&ldsP[r * kTS + f]), *reinterpret_cast<const float4 *>(&ldsQ[slot * kTS + f]));
        if (p.Eterm && valid && f < h1) z = f4_add(z, *reinterpret_cast<const float4 *>(p.Eterm + pe * h1 + f));
        if (p.save_z[0] && valid && f < h1) *reinterpret_cast<float4 *>(p.save_z[0] + pe * h1 + f) = z;
        a[ct] = z;
      }
      f4n_act<4>(p.act1, a);                           // one uniform activation switch for the 16 values
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
        if (!(valid && 16 * ct + 4 * kq < h1)) a[ct] = f4_zero();
      }
      if (c0 == 0) EDGE_STAMP(2);
      if (has_next && !rows_fetched) {   // the next tile's rows: in flight across this tile's MFMAs
        fetch_rows(meta, rows);
        rows_fetched = true;
      }
      // ---- remaining Dense layers, transposed product on MFMA, chained in registers
      if (wave_on) {
#pragma unroll
      for (int l = 0; l < NTAIL; ++l) {
        const int dw = p.dout[l];
        const int n_ct = (p.din[l] + 15) >> 4, n_mt = (dw + 15) >> 4;   // uniform
        const float *wl = ldsWt + l * kW * kTS + ei * kTS + 4 * kq;
        f32x4 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          acc[mt] = (f32x4){0.f, 0.f, 0.f, 0.f};
          if (mt < n_mt) {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) {
              if (ct < n_ct) {
                const float4 w4 = *reinterpret_cast<const float4 *>(wl + mt * 16 * kTS + 16 * ct);
                acc[mt] = mfma16(w4.x, a[ct].x, acc[mt]);
                acc[mt] = mfma16(w4.y, a[ct].y, acc[mt]);
                acc[mt] = mfma16(w4.z, a[ct].z, acc[mt]);
                acc[mt] = mfma16(w4.w, a[ct].w, acc[mt]);
              }
            }
          }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int f = 16 * mt + 4 * kq;
          const float4 b4 = *reinterpret_cast<const float4 *>(&ldsBias[l * kW + f]);
          const float4 z = make_float4(acc[mt][0] + b4.x, acc[mt][1] + b4.y, acc[mt][2] + b4.z, acc[mt][3] + b4.w);
          if (p.save_z[l + 1] && valid && f < dw) *reinterpret_cast<float4 *>(p.save_z[l + 1] + pe * dw + f) = z;
          a[mt] = z;
        }
        f4n_act<4>(p.act[l], a);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
          if (!(valid && 16 * mt + 4 * kq < dw)) a[mt] = f4_zero();
      }
      }
      if (c0 == 0) EDGE_STAMP(3);
      // ---- messages of the chunk -> LDS, then lane group g sums the messages of row g in edge order
#pragma unroll
      for (int mt = 0; mt < 4; ++mt)
        *reinterpret_cast<float4 *>(&ldsMsg[(wave * 16 + ei) * kTS + 16 * mt + 4 * kq]) = a[mt];
      __syncthreads();
      if (c0 == 0) EDGE_STAMP(4);
      {
        const int lo = max(my_lo, c0), hi = min(my_hi, c0 + kChunk);
        for (int kk = lo; kk < hi; ++kk) {
          const float4 m = *reinterpret_cast<const float4 *>(&ldsMsg[(kk - c0) * kTS + 4 * q]);
          if (p.aggr == NGPDE_AGGR_MAX) racc = make_float4(fmaxf(racc.x, m.x), fmaxf(racc.y, m.y), fmaxf(racc.z, m.z), fmaxf(racc.w, m.w));
          else if (p.aggr == NGPDE_AGGR_MIN) racc = make_float4(fminf(racc.x, m.x), fminf(racc.y, m.y), fminf(racc.z, m.z), fminf(racc.w, m.w));
          else if (p.aggr == NGPDE_AGGR_MUL) racc = f4_mul(racc, m);
          else racc = f4_add(racc, m);
        }
      }
      __syncthreads();
      if (c0 == 0) EDGE_STAMP(5);
    }
    EDGE_STAMP(6);
    if (has_next && !rows_fetched) fetch_rows(meta, rows);   // a tile without edges
    if (sc.x >= 0 && 4 * q < last_w) {
      const int deg = my_hi - my_lo;
      if (p.aggr == NGPDE_AGGR_MEAN) racc = deg > 0 ? f4_scale(1.0f / (float)deg, racc) : f4_zero();
      *reinterpret_cast<float4 *>(p.out + (size_t)sc.x * last_w + 4 * q) = racc;
    }
    EDGE_STAMP(7);
  }
}

// ---- fused pullback of the message path (0 or 1 Dense layer after the first) -------------------------------------------------
// Same tiling, staging and register layout as the forward kernel.  Per 16-edge wave slice, all in registers:
//   z1 = P[t] + Q[s] + E, a1 = act1(z1)                         (recomputed: nothing per-edge was saved by the forward)
//   z2^T = W2^T a1^T + b2            (MFMA, transposed product)   dz2 = g[t] * act2'(z2),  g = dout / deg (mean) or dout (+)
//   da1^T = W2 dz2^T                 (MFMA, transposed product)   dz1 = da1 * act1'(z1)
//   dW2 += a1^T dz2                  (MFMA over the wave's 16 edges; the two operands are transposed through a wave-private
//                                     4 KB LDS tile; 16 accumulator tiles per wave live for the whole kernel)
// dz1 goes to HBM once ([E][h1], p order: it is dE and the input of the by-source sum that gives dQ) and through LDS into
// the in-tile segmented sum that gives dP.  At the end the 8 waves fold their dW2 / db2 accumulators into one slab per
// workgroup in a fixed order; a reduce kernel sums the slabs.  No atomics.
struct EdgeMlpBwdK {
  const int4 *sched;
  const int2 *halo;
  const uint8_t *slots;
  int n_tiles, h1, act1, aggr, halo_rows, n_tail, dw, act2;   // dw = width of the tail layer's output (NTAIL = 1)
  const float *P, *Q, *Eterm, *wt, *bias, *dout;
  float *dP, *dE, *partial;   // partial: [n_workgroups][(h1 + 1)][dw]  (row h1 = bias gradient)
};

template <int NTAIL>
__global__ __launch_bounds__(kT, 2) void edge_mlp_fused_bwd_kernel(const EdgeMlpBwdK p) {
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  float *ldsQ = dyn;                                              // [halo_rows + 1][kTS]
  float *ldsP = ldsQ + (size_t)(p.halo_rows + 1) * kTS;           // [32][kTS]
  float *ldsG = ldsP + kGroups * kTS;                             // [32][kTS]  incoming gradient rows (already / deg for mean)
  float *ldsS = ldsG + kGroups * kTS;                             // [kChunk][kTS]  wave-private transposes, then dz1 of the chunk
  float *ldsWf = ldsS + kChunk * kTS;                             // [64 out][kTS]  W2^T   (NTAIL = 1)
  float *ldsWb = ldsWf + (NTAIL ? kW * kTS : 0);                  // [64 in][kTS]   W2
  float *ldsZc = ldsWb + (NTAIL ? kW * kTS : 0);                  // [32][kTS]  aggr = *: how many of the target's messages are zero, per feature;
                                                                  //            max / min: the target's extremum
  __shared__ int ldsOff[kGroups + 1], ldsRs[kGroups];
  __shared__ __attribute__((aligned(16))) unsigned ldsSlots[kGroups * 8];
  __shared__ uint8_t ldsRowOf[kGroups * kSlotWidth];
  __shared__ __attribute__((aligned(16))) float ldsBias[kW];

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = tid >> 4, q = tid & 15;
  const int ei = lane & 15, kq = lane >> 4;
  const int h1 = p.h1, zero_slot = p.halo_rows, dw = NTAIL ? p.dw : p.h1;

  const int xcd = blockIdx.x & 7, wg_in_xcd = blockIdx.x >> 3, wgs_per_xcd = gridDim.x >> 3;
  const int range_len = p.n_tiles / 8 + (xcd < p.n_tiles % 8 ? 1 : 0);
  const int range_lo = xcd * (p.n_tiles / 8) + min(xcd, p.n_tiles % 8);

  auto fetch_meta = [&](int tile, TileMeta &m) {
    m.sc = p.sched[(size_t)tile * kTileRows + grp];
    m.s0 = reinterpret_cast<const uint4 *>(p.slots)[((size_t)tile * kTileRows + grp) * 2];
    m.s1 = reinterpret_cast<const uint4 *>(p.slots)[((size_t)tile * kTileRows + grp) * 2 + 1];
#pragma unroll
    for (int k = 0; k < 3; ++k) m.he[k] = p.halo[(size_t)tile * kHaloCap + min(grp + k * kGroups, kHaloCap - 1)];
  };

  if (NTAIL) {
    const int j = tid % kW, kg0 = tid / kW;
#pragma unroll
    for (int ps = 0; ps < 2; ++ps) {
      const int k = 4 * (kg0 + 8 * ps);
      float t[4], u[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        t[r] = (k + r < h1 && j < dw) ? p.wt[(size_t)(k + r) * dw + j] : 0.f;        // W2^T row j (output), inputs k..k+3
        u[r] = (j < h1 && k + r < dw) ? p.wt[(size_t)j * dw + k + r] : 0.f;          // W2 row j (input), outputs k..k+3
      }
      *reinterpret_cast<float4 *>(&ldsWf[j * kTS + k]) = make_float4(t[0], t[1], t[2], t[3]);
      *reinterpret_cast<float4 *>(&ldsWb[j * kTS + k]) = make_float4(u[0], u[1], u[2], u[3]);
    }
    if (tid < kW) ldsBias[tid] = (p.bias && tid < dw) ? p.bias[tid] : 0.f;
  }
  if (grp == 0) *reinterpret_cast<float4 *>(&ldsQ[zero_slot * kTS + 4 * q]) = f4_zero();

  // dW2 accumulators of this wave: tile (ct, mt) <-> rows 16 ct .. + 15 (inputs) x columns 16 mt .. + 15 (outputs)
  f32x4 accW[NTAIL ? 4 : 1][NTAIL ? 4 : 1];
  float4 dbacc[NTAIL ? 4 : 1];
#pragma unroll
  for (int a = 0; a < (NTAIL ? 4 : 1); ++a) {
    dbacc[a] = f4_zero();
#pragma unroll
    for (int b = 0; b < (NTAIL ? 4 : 1); ++b) accW[a][b] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  // the per-edge forward the pullback recomputes: z1 = P_i + Q_j + E_e, a1 = act1(z1) (zero on padded features / invalid edges) ...
  auto first_layer = [&](int r, int slot, size_t pe, bool valid, float4 (&z1)[4], float4 (&a1)[4]) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int f = 16 * ct + 4 * kq;
      float4 z = f4_add(*reinterpret_cast<const float4 *>(&ldsP[r * kTS + f]), *reinterpret_cast<const float4 *>(&ldsQ[slot * kTS + f]));
      if (p.Eterm && valid && f < h1) z = f4_add(z, *reinterpret_cast<const float4 *>(p.Eterm + pe * h1 + f));
      z1[ct] = z;
      a1[ct] = z;
    }
    f4n_act<4>(p.act1, a1);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
      if (!(valid && 16 * ct + 4 * kq < h1)) a1[ct] = f4_zero();
  };
  // ... and z2 = W2^T a1 + b2 (transposed product: the D layout is the operand layout of the next product)
  auto second_layer = [&](const float4 (&a1)[4], float4 (&z2)[4]) {
    const int n_ct = (h1 + 15) >> 4, n_mt = (dw + 15) >> 4;   // uniform
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      z2[mt] = f4_zero();
      if (mt < n_mt) {
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
        const float *wl = ldsWf + (mt * 16 + ei) * kTS + 4 * kq;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          if (ct < n_ct) {
            const float4 w4 = *reinterpret_cast<const float4 *>(wl + 16 * ct);
            acc = mfma16(w4.x, a1[ct].x, acc);
            acc = mfma16(w4.y, a1[ct].y, acc);
            acc = mfma16(w4.z, a1[ct].z, acc);
            acc = mfma16(w4.w, a1[ct].w, acc);
          }
        }
        const float4 b4 = *reinterpret_cast<const float4 *>(&ldsBias[16 * mt + 4 * kq]);
        z2[mt] = make_float4(acc[0] + b4.x, acc[1] + b4.y, acc[2] + b4.z, acc[3] + b4.w);
      }
    }
  };
  const bool mul = p.aggr == NGPDE_AGGR_MUL, ext = p.aggr == NGPDE_AGGR_MAX || p.aggr == NGPDE_AGGR_MIN, want_max = p.aggr == NGPDE_AGGR_MAX;

  TileMeta meta;
  int jt = wg_in_xcd;
  if (jt < range_len) fetch_meta(range_lo + jt, meta);

  for (; jt < range_len; jt += wgs_per_xcd) {
    const int4 sc = meta.sc;
    // ---- stage the tile: Q halo rows, P rows, gradient rows (the loads of a tile are issued here; the next tile's metadata
    // is prefetched below)
    {
      const int node = max(sc.x, 0);
      const float4 prow = (p.P && 4 * q < h1) ? *reinterpret_cast<const float4 *>(p.P + (size_t)node * h1 + 4 * q) : f4_zero();
      float4 hv[3];
#pragma unroll
      for (int k = 0; k < 3; ++k)
        hv[k] = (p.Q && grp + k * kGroups < p.halo_rows && 4 * q < h1)
                    ? *reinterpret_cast<const float4 *>(p.Q + (size_t)meta.he[k].x * h1 + 4 * q) : f4_zero();
      float4 grow = (sc.x >= 0 && 4 * q < dw) ? *reinterpret_cast<const float4 *>(p.dout + (size_t)node * dw + 4 * q) : f4_zero();
      if (p.aggr == NGPDE_AGGR_MEAN) grow = sc.z > 0 ? f4_scale(1.0f / (float)sc.z, grow) : f4_zero();
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        const int hh = grp + k * kGroups;
        if (hh < p.halo_rows) *reinterpret_cast<float4 *>(&ldsQ[hh * kTS + 4 * q]) = hv[k];
      }
      *reinterpret_cast<float4 *>(&ldsP[grp * kTS + 4 * q]) = prow;
      *reinterpret_cast<float4 *>(&ldsG[grp * kTS + 4 * q]) = grow;
    }
    if (q == 0) {
      ldsOff[grp + 1] = sc.x >= 0 ? sc.z : 0;
      ldsRs[grp] = sc.y;
      if (grp == 0) ldsOff[0] = 0;
    }
    if (q < 8) {
      const unsigned w[8] = {meta.s0.x, meta.s0.y, meta.s0.z, meta.s0.w, meta.s1.x, meta.s1.y, meta.s1.z, meta.s1.w};
      unsigned v = 0;
#pragma unroll
      for (int j = 0; j < 8; ++j) v = (q == j) ? w[j] : v;
      ldsSlots[grp * 8 + q] = v;
    }
    const int jn = jt + wgs_per_xcd;
    if (jn < range_len) fetch_meta(range_lo + jn, meta);
    __syncthreads();
    if (tid < kGroups) {
      int v = ldsOff[tid + 1];
#pragma unroll
      for (int o = 1; o < kGroups; o <<= 1) {
        const int u = __shfl_up(v, o);
        if (tid >= o) v += u;
      }
      ldsOff[tid + 1] = v;
    }
    __syncthreads();
    const int total = ldsOff[kGroups];
    const int my_lo = ldsOff[grp], my_hi = ldsOff[grp + 1];
    for (int k = my_lo + q; k < my_hi; k += 16) ldsRowOf[k] = (uint8_t)grp;
    float4 racc = f4_zero();
    __syncthreads();

    if (mul || ext) {
      // ---- aggr = *: a first pass over the tile's edges recomputes the messages and leaves, per target and feature, the product of the
      // nonzero ones (folded into the gradient row) and the number of zeros; max / min: the target's extremum (the second pass gives the
      // gradient to every message equal to it, as NNlib's pullback of scatter(max) does)
      float4 pacc = make_float4(1.f, 1.f, 1.f, 1.f), zacc = f4_zero();
      const float e0 = want_max ? -INFINITY : INFINITY;
      float4 eacc = make_float4(e0, e0, e0, e0);
      for (int c0 = 0; c0 < total; c0 += kChunk) {
        const bool wave_on = c0 + wave * 16 < total;   // wave-uniform
        const int k = c0 + wave * 16 + ei;
        const bool valid = k < total;
        int r = 0, slot = zero_slot;
        size_t pe = 0;
        if (valid) {
          r = ldsRowOf[k];
          const int j = k - ldsOff[r];
          slot = (ldsSlots[r * 8 + (j >> 2)] >> (8 * (j & 3))) & 0xff;
          pe = (size_t)(ldsRs[r] + j);
        }
        float *mine = ldsS + (size_t)(wave * 16) * kTS;
        float4 m[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
        if (wave_on) {
          float4 z1[4], a1[4];
          first_layer(r, slot, pe, valid, z1, a1);
          if (NTAIL) {
            second_layer(a1, m);
            f4n_act<4>(p.act2, m);
          } else {
#pragma unroll
            for (int ct = 0; ct < 4; ++ct) m[ct] = a1[ct];
          }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<float4 *>(&mine[ei * kTS + 16 * mt + 4 * kq]) = m[mt];
        __syncthreads();
        {
          const int lo = max(my_lo, c0), hi = min(my_hi, c0 + kChunk);
          for (int kk = lo; kk < hi; ++kk) {
            const float4 v = *reinterpret_cast<const float4 *>(&ldsS[(kk - c0) * kTS + 4 * q]);
            pacc = make_float4(pacc.x * (v.x != 0.f ? v.x : 1.f), pacc.y * (v.y != 0.f ? v.y : 1.f), pacc.z * (v.z != 0.f ? v.z : 1.f),
                               pacc.w * (v.w != 0.f ? v.w : 1.f));
            zacc = make_float4(zacc.x + (v.x == 0.f ? 1.f : 0.f), zacc.y + (v.y == 0.f ? 1.f : 0.f), zacc.z + (v.z == 0.f ? 1.f : 0.f),
                               zacc.w + (v.w == 0.f ? 1.f : 0.f));
            eacc = want_max ? make_float4(fmaxf(eacc.x, v.x), fmaxf(eacc.y, v.y), fmaxf(eacc.z, v.z), fmaxf(eacc.w, v.w))
                            : make_float4(fminf(eacc.x, v.x), fminf(eacc.y, v.y), fminf(eacc.z, v.z), fminf(eacc.w, v.w));
          }
        }
        __syncthreads();
      }
      if (mul) {
        float4 *gr = reinterpret_cast<float4 *>(&ldsG[grp * kTS + 4 * q]);
        *gr = f4_mul(*gr, pacc);
      }
      *reinterpret_cast<float4 *>(&ldsZc[grp * kTS + 4 * q]) = mul ? zacc : eacc;
      __syncthreads();
    }

    for (int c0 = 0; c0 < total; c0 += kChunk) {
      const bool wave_on = c0 + wave * 16 < total;   // wave-uniform
      const int k = c0 + wave * 16 + ei;
      const bool valid = k < total;
      int r = 0, slot = zero_slot;
      size_t pe = 0;
      if (valid) {
        r = ldsRowOf[k];
        const int j = k - ldsOff[r];
        slot = (ldsSlots[r * 8 + (j >> 2)] >> (8 * (j & 3))) & 0xff;
        pe = (size_t)(ldsRs[r] + j);
      }
      float *mine = ldsS + (size_t)(wave * 16) * kTS;          // this wave's 16 rows of the staging tile
      float4 dz1[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
      if (wave_on) {
        float4 z1[4], a1[4];
        first_layer(r, slot, pe, valid, z1, a1);
        f4n_dact<4>(p.act1, z1);                       // z1 <- act1'(z1): only the derivative is needed from here on
        float4 gz[4];                                          // NTAIL = 0: g itself; NTAIL = 1: dz2 = g * act2'(z2)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
          const int f = 16 * mt + 4 * kq;
          gz[mt] = (valid && f < dw) ? *reinterpret_cast<const float4 *>(&ldsG[r * kTS + f]) : f4_zero();
        }
        // aggr = *: dL/dm_e = g_i . (product of the target's OTHER messages) = (g_i . product of its nonzero messages) / m_e where
        // none of them is zero; the one zero message of a row gets the product of the others; two zeros leave nothing
        auto others = [&](float4 (&gv)[4], const float4 (&m)[4]) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int f = 16 * mt + 4 * kq;
            const float4 zc = (valid && f < dw) ? *reinterpret_cast<const float4 *>(&ldsZc[r * kTS + f]) : make_float4(2.f, 2.f, 2.f, 2.f);
            if (ext) {   // (zc holds the extremum; invalid edges / padded features carry a zero gradient already)
              gv[mt] = make_float4(m[mt].x == zc.x ? gv[mt].x : 0.f, m[mt].y == zc.y ? gv[mt].y : 0.f, m[mt].z == zc.z ? gv[mt].z : 0.f,
                                   m[mt].w == zc.w ? gv[mt].w : 0.f);
              continue;
            }
            auto one = [](float g, float mm, float z) { return mm != 0.f ? (z == 0.f ? g / mm : 0.f) : (z == 1.f ? g : 0.f); };
            gv[mt] = make_float4(one(gv[mt].x, m[mt].x, zc.x), one(gv[mt].y, m[mt].y, zc.y), one(gv[mt].z, m[mt].z, zc.z), one(gv[mt].w, m[mt].w, zc.w));
          }
        };
        if (!NTAIL && (mul || ext)) others(gz, a1);
        if (NTAIL) {
          const int n_ct = (h1 + 15) >> 4, n_mt = (dw + 15) >> 4;   // uniform
          // ---- z2 (transposed product), dz2
          float4 z2[4];
          second_layer(a1, z2);
          if (mul || ext) {
            float4 m2[4] = {z2[0], z2[1], z2[2], z2[3]};
            f4n_act<4>(p.act2, m2);
            others(gz, m2);
          }
          f4n_dact<4>(p.act2, z2);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            gz[mt] = f4_mul(gz[mt], z2[mt]);                  // g is zero for invalid edges / padded features
            dbacc[mt] = f4_add(dbacc[mt], gz[mt]);
          }
          // ---- dW2 += a1^T dz2 over this wave's 16 edges: both operands transposed through the wave's LDS rows
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<float4 *>(&mine[ei * kTS + 16 * ct + 4 * kq]) = a1[ct];
          float a1T[4][4];
#pragma unroll
          for (int ct = 0; ct < 4; ++ct)
#pragma unroll
            for (int sI = 0; sI < 4; ++sI) a1T[ct][sI] = mine[(4 * sI + kq) * kTS + 16 * ct + ei];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) *reinterpret_cast<float4 *>(&mine[ei * kTS + 16 * mt + 4 * kq]) = gz[mt];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            if (mt < n_mt) {
              float dzT[4];
#pragma unroll
              for (int sI = 0; sI < 4; ++sI) dzT[sI] = mine[(4 * sI + kq) * kTS + 16 * mt + ei];
#pragma unroll
              for (int ct = 0; ct < 4; ++ct) {
                if (ct < n_ct) {
#pragma unroll
                  for (int sI = 0; sI < 4; ++sI) accW[ct][mt] = mfma16(a1T[ct][sI], dzT[sI], accW[ct][mt]);
                }
              }
            }
          }
          // ---- da1 (transposed product with W2), dz1
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            if (ct < n_ct) {
              f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
              const float *wl = ldsWb + (ct * 16 + ei) * kTS + 4 * kq;
#pragma unroll
              for (int mt = 0; mt < 4; ++mt) {
                if (mt < n_mt) {
                  const float4 w4 = *reinterpret_cast<const float4 *>(wl + 16 * mt);
                  acc = mfma16(w4.x, gz[mt].x, acc);
                  acc = mfma16(w4.y, gz[mt].y, acc);
                  acc = mfma16(w4.z, gz[mt].z, acc);
                  acc = mfma16(w4.w, gz[mt].w, acc);
                }
              }
              dz1[ct] = f4_mul(make_float4(acc[0], acc[1], acc[2], acc[3]), z1[ct]);
            }
          }
        } else {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) dz1[ct] = f4_mul(gz[ct], z1[ct]);
        }
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const int f = 16 * ct + 4 * kq;
          if (!(valid && f < h1)) dz1[ct] = f4_zero();
          else if (p.dE) *reinterpret_cast<float4 *>(p.dE + pe * h1 + f) = dz1[ct];
        }
      }
      // ---- dz1 of the chunk -> LDS, lane group g sums the rows of target g in edge order (= dP)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) *reinterpret_cast<float4 *>(&mine[ei * kTS + 16 * ct + 4 * kq]) = dz1[ct];
      __syncthreads();
      {
        const int lo = max(my_lo, c0), hi = min(my_hi, c0 + kChunk);
        for (int kk = lo; kk < hi; ++kk) racc = f4_add(racc, *reinterpret_cast<const float4 *>(&ldsS[(kk - c0) * kTS + 4 * q]));
      }
      __syncthreads();
    }
    if (p.dP && sc.x >= 0 && 4 * q < h1) *reinterpret_cast<float4 *>(p.dP + (size_t)sc.x * h1 + 4 * q) = racc;
  }

  // ---- fold the waves' dW2 / db2 accumulators into this workgroup's slab, wave by wave (fixed order), then write it out
  if (NTAIL) {
    float *slab = ldsS;                                            // [(h1 + 1)][dw], needs 65 * 64 floats <= kChunk * kTS
    __syncthreads();
    for (int idx = tid; idx < (h1 + 1) * dw; idx += kT) slab[idx] = 0.f;
    // db: sum the 16 edge lanes of each k-quarter inside the wave first
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
      float v[4] = {dbacc[mt].x, dbacc[mt].y, dbacc[mt].z, dbacc[mt].w};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
#pragma unroll
        for (int o = 1; o < 16; o <<= 1) v[c] += __shfl_xor(v[c], o);
      }
      dbacc[mt] = make_float4(v[0], v[1], v[2], v[3]);
    }
    __syncthreads();
    for (int w = 0; w < kT / 64; ++w) {
      if (wave == w) {
#pragma unroll
        for (int ct = 0; ct < 4; ++ct)
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              const int kin = 16 * ct + 4 * kq + r, o = 16 * mt + ei;
              if (kin < h1 && o < dw) slab[kin * dw + o] += accW[ct][mt][r];
            }
        if (ei == 0) {
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) {
            const int o = 16 * mt + 4 * kq;
            if (o < dw) {
              slab[h1 * dw + o] += dbacc[mt].x; slab[h1 * dw + o + 1] += dbacc[mt].y;
              slab[h1 * dw + o + 2] += dbacc[mt].z; slab[h1 * dw + o + 3] += dbacc[mt].w;
            }
          }
        }
      }
      __syncthreads();
    }
    float *dst = p.partial + (size_t)blockIdx.x * (h1 + 1) * dw;
    for (int idx = tid; idx < (h1 + 1) * dw; idx += kT) dst[idx] = slab[idx];
  }
}

__global__ void activation_fwd_kernel(int64_t count, int act, const float *__restrict__ z, float *__restrict__ a) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    a[i] = act_apply(act, z[i]);
}

}  // namespace

bool edge_mlp_fused_supported(const ngpde_graph *g, const EdgeMlpArgs &a) {
  if (!g || !g->has_norm || !g->by_t.halo_ok) return false;   // needs the tile schedule, halo lists and slot bytes
  if (a.h1 <= 0 || a.h1 > kW || a.h1 % 4) return false;
  if (a.n_tail < 0 || a.n_tail > 3) return false;
  int prev = a.h1;
  for (int l = 0; l < a.n_tail; ++l) {
    if (a.din[l] != prev || a.dout[l] <= 0 || a.dout[l] > kW || a.dout[l] % 4) return false;
    prev = a.dout[l];
  }
  return true;
}

int32_t launch_edge_mlp_fused_fwd(const ngpde_graph *g, const EdgeMlpArgs &a, hipStream_t stream) {
  NGPDE_REQUIRE(edge_mlp_fused_supported(g, a), NGPDE_ERR_UNSUPPORTED,
                "fused edge-MLP path needs widths <= 64 and multiples of 4, <= 3 layers after the first, and a graph whose "
                "tiles fit the LDS halo (degree <= %d, <= %d distinct sources per 32-row tile)", kSlotWidth, kHaloCap);
  if (g->n_nodes == 0) return NGPDE_OK;
  if (edge_mlp64_fwd_applicable(g, a)) return launch_edge_mlp64_fwd(g, a, stream);
  EdgeMlpK k;
  k.sched = g->by_t.sched; k.halo = g->by_t.halo; k.slots = g->by_t.slots;
  k.n_tiles = (int)(g->n_sched / kTileRows); k.h1 = a.h1; k.act1 = a.act1; k.aggr = a.aggr;
  k.P = a.P; k.Q = a.Q; k.Eterm = a.Eterm; k.n_tail = a.n_tail;
  for (int l = 0; l < 3; ++l) {
    k.din[l] = a.din[l]; k.dout[l] = a.dout[l]; k.act[l] = a.act[l]; k.wt[l] = a.wt[l]; k.bias[l] = a.bias[l];
  }
  k.out = a.out;
  for (int l = 0; l < 4; ++l) k.save_z[l] = a.save_z[l];
#ifdef NGPDE_STAMPS
  k.stamps = g_edge_stamps;
#endif
  // LDS: the halo region is sized by the largest halo of this graph's tiles; persistent workgroups, a multiple of the 8 XCDs
  k.halo_rows = std::max<int>(kTileRows, std::min<int>(kHaloCap, g->by_t.max_halo));
  const size_t lds = ((size_t)(k.halo_rows + 1) * kTS + (size_t)kGroups * kTS + (size_t)a.n_tail * kW * kTS + (size_t)kChunk * kTS) * sizeof(float);
  const int wgs_per_cu = (lds + 2048 <= 80 * 1024) ? 2 : 1;
  const int per_xcd = std::max(1, std::min(32 * wgs_per_cu, (k.n_tiles + 7) / 8));
  const dim3 grid(8 * per_xcd), block(kT);
  auto launch = [&](auto kernel) -> hipError_t {
    // more than 64 KB of dynamic LDS has to be requested explicitly
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, k);
    return hipSuccess;
  };
  hipError_t le;
  switch (a.n_tail) {
    case 0: le = launch(edge_mlp_fused_fwd_kernel<0>); break;
    case 1: le = launch(edge_mlp_fused_fwd_kernel<1>); break;
    case 2: le = launch(edge_mlp_fused_fwd_kernel<2>); break;
    default: le = launch(edge_mlp_fused_fwd_kernel<3>); break;
  }
  if (le != hipSuccess) return fail(NGPDE_ERR_HIP, "edge_mlp_fused_fwd_kernel: LDS request of %zu bytes refused: %s", lds, hipGetErrorString(le));
  NGPDE_LAUNCH_CHECK("edge_mlp_fused_fwd_kernel");
  return NGPDE_OK;
}

#ifdef NGPDE_STAMPS
}  // namespace ngpde
extern "C" int32_t ngpde_debug_set_edge_stamps(unsigned long long *buf) {
  ngpde::g_edge_stamps = buf;
  return 0;
}
namespace ngpde {
#endif

static int edge_bwd_grid(const ngpde_graph *g) {
  const int n_tiles = (int)(g->n_sched / kTileRows);
  return 8 * std::max(1, std::min(32, (n_tiles + 7) / 8));   // one persistent workgroup per CU, a multiple of the 8 XCDs
}

bool edge_mlp_fused_bwd_supported(const ngpde_graph *g, int h1, int n_tail, int dw, int aggr) {
  if (!g || !g->has_norm || !g->by_t.halo_ok) return false;
  if (h1 <= 0 || h1 > kW || h1 % 4) return false;
  if (n_tail < 0 || n_tail > 1) return false;                 // deeper message MLPs take the primitives' pullback
  if (n_tail == 1 && (dw <= 0 || dw > kW || dw % 4)) return false;
  return aggr >= NGPDE_AGGR_SUM && aggr <= NGPDE_AGGR_MUL;   // (max / min / * on this kernel only: the 64-wide and deep ones take + / mean)
}

size_t edge_mlp_fused_bwd_workspace(const ngpde_graph *g, int h1, int n_tail, int dw) {
  const size_t general = n_tail ? (size_t)edge_bwd_grid(g) * (h1 + 1) * dw * sizeof(float) + 256 : 256;
  return std::max(general, (h1 == kW && n_tail == 1 && dw == kW) ? edge_mlp64_bwd_workspace(g) : (size_t)0);   // (edge_mlp64.hip)
}

int32_t launch_edge_mlp_fused_bwd(const ngpde_graph *g, const EdgeMlpBwdArgs &a, hipStream_t stream) {
  NGPDE_REQUIRE(edge_mlp_fused_bwd_supported(g, a.h1, a.n_tail, a.dw, a.aggr), NGPDE_ERR_UNSUPPORTED,
                "fused edge-MLP pullback needs widths <= 64 and multiples of 4, at most one layer after the first and a graph whose tiles fit "
                "the LDS halo");
  if (g->n_nodes == 0) return NGPDE_OK;
  const size_t need = edge_mlp_fused_bwd_workspace(g, a.h1, a.n_tail, a.dw);
  NGPDE_REQUIRE(a.workspace && a.workspace_bytes >= need, NGPDE_ERR_WORKSPACE, "fused edge-MLP pullback: workspace too small (%zu < %zu bytes)",
                a.workspace_bytes, need);
  if (edge_mlp64_bwd_applicable(g, a)) return launch_edge_mlp64_bwd(g, a, stream);   // (checks dE itself: not needed when it sums by source in the launch)
  NGPDE_REQUIRE(a.dE != nullptr || g->n_edges == 0, NGPDE_ERR_INVALID_ARGUMENT, "fused edge-MLP pullback: the [E][h1] buffer dE is required");
  EdgeMlpBwdK k;
  k.sched = g->by_t.sched; k.halo = g->by_t.halo; k.slots = g->by_t.slots;
  k.n_tiles = (int)(g->n_sched / kTileRows); k.h1 = a.h1; k.act1 = a.act1; k.aggr = a.aggr; k.n_tail = a.n_tail;
  k.dw = a.n_tail ? a.dw : a.h1; k.act2 = a.act2;
  k.halo_rows = std::max<int>(kTileRows, std::min<int>(kHaloCap, g->by_t.max_halo));
  k.P = a.P; k.Q = a.Q; k.Eterm = a.Eterm; k.wt = a.wt; k.bias = a.bias; k.dout = a.dout;
  k.dP = a.dP; k.dE = a.dE; k.partial = (float *)a.workspace;
  const size_t lds = ((size_t)(k.halo_rows + 1) * kTS + 2 * (size_t)kGroups * kTS + (size_t)kChunk * kTS + (a.n_tail ? 2 * (size_t)kW * kTS : 0) +
                      (a.aggr >= NGPDE_AGGR_MAX ? (size_t)kGroups * kTS : 0)) * sizeof(float);
  const dim3 grid(edge_bwd_grid(g)), block(kT);
  auto launch = [&](auto kernel) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, grid, block, lds, stream, k);
    return hipSuccess;
  };
  const hipError_t le = a.n_tail ? launch(edge_mlp_fused_bwd_kernel<1>) : launch(edge_mlp_fused_bwd_kernel<0>);
  if (le != hipSuccess) return fail(NGPDE_ERR_HIP, "edge_mlp_fused_bwd_kernel: LDS request of %zu bytes refused: %s", lds, hipGetErrorString(le));
  NGPDE_LAUNCH_CHECK("edge_mlp_fused_bwd_kernel");
  int32_t st;
  if (a.n_tail && (st = launch_dense_weight_reduce((int)grid.x, a.h1, a.dw, k.partial, a.dwt, a.dbias, stream))) return st;
  if (a.dQ && (st = launch_edge_sum_by_source(g, a.h1, a.dE, a.dQ, stream))) return st;
  return NGPDE_OK;
}

int32_t launch_activation_fwd(int64_t count, int act, const float *z, float *a, hipStream_t stream) {
  if (count == 0) return NGPDE_OK;
  const int blocks = (int)std::min<int64_t>((count + 255) / 256, 4096);
  hipLaunchKernelGGL(activation_fwd_kernel, dim3(blocks), dim3(256), 0, stream, count, act, z, a);
  NGPDE_LAUNCH_CHECK("activation_fwd_kernel");
  return NGPDE_OK;
}

}  // namespace ngpde
