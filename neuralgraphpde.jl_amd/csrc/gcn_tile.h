// gcn_tile.h -- tile geometry and device helpers shared by the fused GCN kernels (gcn_fused.hip) and the persistent
// solver kernels (node_persistent.hip): one workgroup of 512 threads per 32-row tile, D/4 lanes per feature row,
// fp32 MFMA tile products from LDS.  Internal (anonymous namespace: every translation unit gets its own copy).
#pragma once

#include "common.h"
#include "device_utils.h"

namespace ngpde {
namespace {

constexpr int kThreads = 512;
constexpr int kTM = kTileRows;
constexpr int kXcds = 8;

template <int D>
struct Geo {
  static constexpr int LPR = D / 4;                          // lanes per feature row (float4 each)
  static constexpr int GROUPS = kThreads / LPR;              // row groups per workgroup
  static constexpr int R = (kTM + GROUPS - 1) / GROUPS;      // rows per group
  static constexpr int U = (16 / R < LPR) ? 16 / R : LPR;    // neighbour rows in flight per row
  static constexpr bool ELL = (LPR <= kEllWidth);            // first entry chunk from the position-indexed block
  static constexpr int TS = D + 4;                           // LDS row stride (floats): 16-B aligned rows, b128 reads of 16 rows cover 64 banks
  static constexpr int W4 = (D * D / 4 + kThreads - 1) / kThreads;  // float4 of W per thread (straight copy)
  static constexpr int KGP = kThreads / D;                   // k-groups (4 consecutive k) per pass of the transposing loader
  static constexpr int NPASS = (D / 4 + KGP - 1) / KGP;
  static constexpr int RT = kTM / 16;                        // 16-row MFMA tiles per workgroup
  static constexpr int CT = D / 16;                          // 16-col MFMA tiles
  static constexpr int WAVES = kThreads / 64;
  static constexpr int CGRP = WAVES / RT;                    // waves sharing one row tile
  static constexpr int CPW = (CT + CGRP - 1) / CGRP;         // column tiles per wave
  static constexpr int DWT = (CT * CT + WAVES - 1) / WAVES;  // dW tiles per wave
  static constexpr int DBP = kThreads / D;                   // row-partials per column in the db reduction
  static constexpr bool HALO = (D <= 64);                    // LDS-staged aggregation (a 128-wide halo would not fit beside the tiles)
  static constexpr int HI = (kHaloCap + GROUPS - 1) / GROUPS; // halo rows staged per group
  static constexpr int XH = HALO ? (kHaloCap + 1) * D : 0;   // floats of the halo region (+1: the all-zero row)
};

// blockIdx -> tile, bijective for any grid size: blocks b, b+8, b+16, ... (dispatched to one XCD in
// practice) get consecutive tiles.  Placement only changes speed, never results.
__device__ __forceinline__ int xcd_tile(int b, int nb) {
  const int x = b % kXcds, k = b / kXcds;
  const int q = nb / kXcds, r = nb % kXcds;
  return x * q + min(x, r) + k;
}

// Streaming (touch-once) accesses of the tape -- the aggregated input saved by forward, the saved
// activations read back by backward -- are marked non-temporal so they do not evict what the next launch
// re-reads from the XCD's L2 (gathered rows, schedule/entry arrays, dW slabs).
__device__ __forceinline__ void store_stream4(float4 *p, float4 v) {
  __builtin_nontemporal_store(v.x, &p->x);
  __builtin_nontemporal_store(v.y, &p->y);
  __builtin_nontemporal_store(v.z, &p->z);
  __builtin_nontemporal_store(v.w, &p->w);
}
__device__ __forceinline__ float4 load_stream4(const float4 *p) {
  typedef float f4v __attribute__((ext_vector_type(4)));
  const f4v v = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));
  return make_float4(v.x, v.y, v.z, v.w);
}

// ---- fp32 MFMA tile products from LDS ------------------------------------------------------------------
// Out[kTM][D] = A[kTM][D] x B, with A row-major (stride TS) and B stored TRANSPOSED, Bt[col][k] (stride
// TS), so lane (i = l&15, kq = l>>4) feeds four consecutive k-steps of v_mfma_f32_16x16x4_f32 from ONE
// ds_read_b128 per operand: k-step (kb, r) contracts k = 16 kb + 4 kq + r on both operands.  Operand
// registers are double-buffered across kb so LDS latency hides under the MFMAs.
template <int D>
__device__ __forceinline__ void mfma_rows_times_bt(const float *ldsA, const float *ldsBt, float *ldsOut, int wave_u,
                                                   int lane) {
  using G = Geo<D>;
  const int rt = wave_u % G::RT;
  const int cg = wave_u / G::RT;
  if (cg >= G::CT) return;   // fewer column tiles than wave groups (D <= 32); wave-uniform
  const int i = lane & 15, kq = lane >> 4;
  const float *pa = ldsA + (rt * 16 + i) * G::TS + 4 * kq;
  const float *pb[G::CPW];
  f32x4 acc[G::CPW];
#pragma unroll
  for (int m = 0; m < G::CPW; ++m) {
    pb[m] = ldsBt + ((cg + G::CGRP * m) * 16 + i) * G::TS + 4 * kq;
    acc[m] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  float4 a_cur = *reinterpret_cast<const float4 *>(pa);
  float4 b_cur[G::CPW];
#pragma unroll
  for (int m = 0; m < G::CPW; ++m) b_cur[m] = *reinterpret_cast<const float4 *>(pb[m]);
#pragma unroll
  for (int kb = 0; kb < D / 16; ++kb) {
    float4 a_nxt = a_cur, b_nxt[G::CPW];
#pragma unroll
    for (int m = 0; m < G::CPW; ++m) b_nxt[m] = b_cur[m];
    if (kb + 1 < D / 16) {
      a_nxt = *reinterpret_cast<const float4 *>(pa + (kb + 1) * 16);
#pragma unroll
      for (int m = 0; m < G::CPW; ++m) b_nxt[m] = *reinterpret_cast<const float4 *>(pb[m] + (kb + 1) * 16);
    }
    const float av[4] = {a_cur.x, a_cur.y, a_cur.z, a_cur.w};
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int m = 0; m < G::CPW; ++m) {
        const float bv[4] = {b_cur[m].x, b_cur[m].y, b_cur[m].z, b_cur[m].w};
        acc[m] = mfma16(av[r], bv[r], acc[m]);
      }
    }
    a_cur = a_nxt;
#pragma unroll
    for (int m = 0; m < G::CPW; ++m) b_cur[m] = b_nxt[m];
  }
#pragma unroll
  for (int m = 0; m < G::CPW; ++m) {
    const int ct = cg + G::CGRP * m;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) ldsOut[(rt * 16 + 4 * kq + reg) * G::TS + ct * 16 + i] = acc[m][reg];
  }
}

}  // namespace
}  // namespace ngpde
