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
  static constexpr bool HALO = (D <= 128);                   // LDS-staged aggregation (128 wide: 50 KB of halo rows, one workgroup per CU either way)
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

// Row fetch with a scalar base and a 32-bit byte offset (saddr + voffset addressing: one VGPR per
// in-flight load instead of a 64-bit pointer pair).  Callers guarantee n_nodes * D * 4 < 2^32.
template <int LPR>
__device__ __forceinline__ float4 load_row4(const float4 *__restrict__ X4, int row, int q) {
  const unsigned off = (unsigned)row * (unsigned)(LPR * 16) + (unsigned)(q * 16);
  return *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(X4) + off);
}

// ---- LDS-staged aggregation ---------------------------------------------------------------------------------
// The gather above moves ~280 KB per CU per launch through the L1/TA port (16 row slots x 64 rows x 256 B),
// although a 32-row cluster tile references only ~57 distinct rows.  Here the workgroup stages those rows
// once (3 x 16-byte loads per lane instead of 16), pre-scaled by c[node], and every row then sums its
// neighbours out of LDS.  The chain stays flat: halo list, slot bytes and schedule entries are all
// position-indexed (round 1), the halo rows are round 2.  Each lane reads its row's 16 slot bytes itself (a
// 16-byte broadcast load), so no lane shuffles; unused slots name the all-zero row, so no masking.
template <int D>
struct HaloRegs {
  int2 he[Geo<D>::HI];
  uint4 sl[Geo<D>::R][2];
  float4 sw[Geo<D>::R][8];
  float4 hv[Geo<D>::HI];
};

// round 1: position-indexed halo entries and slot bytes (padded lists: no count, nothing to wait for first)
template <int D>
__device__ __forceinline__ void halo_round1(const int2 *__restrict__ halo, const uint4 *__restrict__ slots16,
                                            const float4 *__restrict__ slot_w4, int tile, int grp, bool active,
                                            HaloRegs<D> &h) {
  using G = Geo<D>;
  // unconditional loads from clamped positions (a `cond ? load : 0` becomes an exec-masked branch with its
  // own full wait); inactive groups / padding slots are neutralised later
#pragma unroll
  for (int k = 0; k < G::HI; ++k) {
    const int hh = min(grp + k * G::GROUPS, kHaloCap - 1);
    h.he[k] = halo[(size_t)tile * kHaloCap + hh];
  }
#pragma unroll
  for (int r = 0; r < G::R; ++r) {
    const size_t pos = (size_t)tile * kTM + min(grp * G::R + r, kTM - 1);
    h.sl[r][0] = slots16[pos * 2];
    h.sl[r][1] = slots16[pos * 2 + 1];
    if (slot_w4) {   // uniform
#pragma unroll
      for (int j = 0; j < 8; ++j) h.sw[r][j] = slot_w4[pos * 8 + j];
    }
  }
}

// schedule entries of this thread's rows, same rule: load always, neutralise afterwards
template <int D>
__device__ __forceinline__ void load_sched(const int4 *__restrict__ sched, int tile, int grp, bool active,
                                           int4 (&sc)[Geo<D>::R]) {
  using G = Geo<D>;
#pragma unroll
  for (int r = 0; r < G::R; ++r) {
    const int4 v = sched[(size_t)tile * kTM + min(grp * G::R + r, kTM - 1)];
    sc[r].x = active ? v.x : -1;
    sc[r].y = active ? v.y : 0;
    sc[r].z = active ? v.z : 0;
    sc[r].w = active ? v.w : 0;
  }
}

// round 2: the tile's distinct rows.  Issue this BEFORE any other load of the kernel that is consumed later:
// vmcnt retires in order, so an older, slower load (HBM tape) would otherwise sit in front of these.
//
// DMA: the rows go memory -> LDS directly (global_load_lds_dwordx4: wave-uniform LDS base + 16 B per lane; a wave's
// 64 / LPR groups stage consecutive halo slots, so its 1 KiB lands contiguously at Xh4[hh * LPR + q]).  No VGPR round trip
// and no ds_write_b128 (13 cycles per wave-instruction on the store path: two workgroups' 25 KB cost ~650 cycles there).
// Only for rows that need no scaling on the way in (PRE: the producer stored them already multiplied by c[node]).
// AUX: cache-policy bits of the DMA (16 = sc1: rows another workgroup of THIS launch stored write-through, gat persistent solver)
template <int D, bool DMA, int AUX = 0>
__device__ __forceinline__ void halo_round2(const float4 *__restrict__ X4, int q, int grp, float *ldsXh, HaloRegs<D> &h) {
  using G = Geo<D>;
#pragma unroll
  for (int k = 0; k < G::HI; ++k) {
    if constexpr (DMA) {
      const int hh = grp + k * G::GROUPS;
      if (hh < kHaloCap) {   // wave-uniform: kHaloCap is a multiple of the groups per wave
        const unsigned off = (unsigned)h.he[k].x * (unsigned)(G::LPR * 16) + (unsigned)(q * 16);
        __builtin_amdgcn_global_load_lds(
            (const __attribute__((address_space(1))) void *)(reinterpret_cast<const char *>(X4) + off),
            (__attribute__((address_space(3))) void *)(reinterpret_cast<float4 *>(ldsXh) + hh * G::LPR + q), 16, 0, AUX);
      }
    } else {
      h.hv[k] = load_row4<G::LPR>(X4, h.he[k].x, q);
    }
  }
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

// The same product with B^T in LDS UNPADDED (row stride 64: 16 KB instead of 17) and XOR-swizzled so that the fragment reads stay
// conflict-free: the four-float quad k4 of row j sits at quad k4 ^ (j & 15).  Same contraction order, same bits.  D = 64 only.
__device__ __forceinline__ void mfma_rows_times_bswz64(const float *ldsA, const float *ldsBswz, float *ldsOut, int wave_u, int lane) {
  using G = Geo<64>;
  const int rt = wave_u % G::RT, ct = wave_u / G::RT;
  const int i = lane & 15, kq = lane >> 4;
  const float *pa = ldsA + (rt * 16 + i) * G::TS + 4 * kq;
  const float *pb = ldsBswz + (ct * 16 + i) * 64;
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < 4; ++kb) {
    const float4 a = *reinterpret_cast<const float4 *>(pa + kb * 16);
    const float4 b = *reinterpret_cast<const float4 *>(pb + 4 * ((4 * kb + kq) ^ i));
    acc = mfma16(a.x, b.x, acc);
    acc = mfma16(a.y, b.y, acc);
    acc = mfma16(a.z, b.z, acc);
    acc = mfma16(a.w, b.w, acc);
  }
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) ldsOut[(rt * 16 + 4 * kq + reg) * G::TS + ct * 16 + i] = acc[reg];
}

// The same product with B as fragments in REGISTERS (b[4 kb + r] = Bt[column tile's column i][16 kb + 4 kq + r]): no copy of the
// matrix in LDS.  Same contraction order, bit for bit the same result.  D = 64 only (one 16 x 16 block per wave).
__device__ __forceinline__ void mfma_rows_times_bfrag64(const float *ldsA, const float (&b)[16], float *ldsOut, int wave_u, int lane) {
  using G = Geo<64>;
  const int rt = wave_u % G::RT, ct = wave_u / G::RT;
  const int i = lane & 15, kq = lane >> 4;
  const float *pa = ldsA + (rt * 16 + i) * G::TS + 4 * kq;
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
#pragma unroll
  for (int reg = 0; reg < 4; ++reg) ldsOut[(rt * 16 + 4 * kq + reg) * G::TS + ct * 16 + i] = acc[reg];
}

}  // namespace
}  // namespace ngpde
