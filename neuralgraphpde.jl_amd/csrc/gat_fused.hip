// gat_fused.hip -- a whole GAT-style layer (64 => heads x c = 64, heads in {1, 2, 4}) as ONE forward launch and
// two pullback launches (+ one reduction) on the tile / halo machinery of the fused GCN kernels.
//   y_i = act( ||_k sum_{e: t_e = i} alpha_{e,k} W_k x_{s_e} + b ),   alpha = softmax_e leakyrelu(a_l,k . W_k x_i + a_r,k . W_k x_s)
// [GraphNeuralNetworks.jl GATConv on softmax_edge_neighbors, the primitive the reference re-exports at
//  /root/reference/src/NeuralGraphPDE.jl:7; BASELINE config 3: GATConv 4 heads x 16 on the 16k-node graph.]
//
// Nothing per node is written but the output, and the per-edge work is kept off the VALU as far as it goes (round 3; round 2
// took scores and aggregates from the INPUT rows -- v = W^T a, per-head aggregates, one product afterwards -- and was VALU-bound:
// a 64-float dot reduced over 16 lanes for every score, 16 fma per entry and lane):
//   * W x of the tile's STAGED rows (own + halo, <= 96) comes off the matrix pipe once per tile ([rows][64] x [64][64], <= 48
//     v_mfma_f32_16x16x4_f32 per wave, B fragments of W straight from memory), into LDS; no W x array in memory;
//   * scores:  a_l,k . (W x_i)_k as a 4-float dot per lane summed over the C / 4 lanes that hold head k (quad permutes + mirrors);
//   * messages:  lane q of a row aggregates ITS head's coefficient times its quad of the entries' W x rows: one fma4 per entry.
// Saved for the pullback: alpha ([E][heads], p order) with the sign bit carrying leakyrelu's branch (alpha >= 0).
// Pullback:  (1) by target: dz = dy . act', d alpha_{e,k} = dz_{i,k} . (W x_s)_k -- W x of the staged rows on MFMA, four-lane dots --,
// softmax + leakyrelu pullback -> dscore [E][heads], dal [N][heads], db slabs;  (2) by source: dWx_j = sum_e alpha_e dz_{t_e} +
// dal_j a_l + dar_j a_r (dar_j = sum of dscore over the outgoing edges), dx = dWx W^T and dW += x^T dWx on MFMA with the
// accumulators of a workgroup's tiles kept in registers, u_l = sum_j dal_j x_j, u_r likewise;  (3) slabs -> dW, db, and
// da_l,k = W_k u_l,k.  No atomics: every output has one writer and a fixed summation order.
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "device_utils.h"
#include "gcn_tile.h"

namespace ngpde {

namespace {

constexpr int GD = 64;                 // input width = heads * c
using GG = Geo<GD>;                    // 16 lanes per row, 32 row groups, one row per group
constexpr int kATS = 4 * GD + 4;       // row stride (floats) of the per-head [32][heads * 64] tiles
constexpr int kSrcTiles = 2;           // tiles per workgroup of the by-source pullback launch (dW accumulators stay in registers)

// sum / max over the 16 lanes of a DPP row (= one row group): rotate-and-add, every lane ends with the total
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, false));
}
__device__ __forceinline__ float row_sum16(float v) {
  v += dpp_mov<0x128>(v);   // row_ror:8
  v += dpp_mov<0x124>(v);   // row_ror:4
  v += dpp_mov<0x122>(v);   // row_ror:2
  v += dpp_mov<0x121>(v);   // row_ror:1
  return v;
}
__device__ __forceinline__ float row_max16(float v) {
  v = fmaxf(v, dpp_mov<0x128>(v));
  v = fmaxf(v, dpp_mov<0x124>(v));
  v = fmaxf(v, dpp_mov<0x122>(v));
  v = fmaxf(v, dpp_mov<0x121>(v));
  return v;
}
__device__ __forceinline__ float dot4(float4 a, float4 b) { return fmaf(a.x, b.x, fmaf(a.y, b.y, fmaf(a.z, b.z, a.w * b.w))); }
__device__ __forceinline__ float sel4(const float (&v)[4], int k) { return k == 0 ? v[0] : k == 1 ? v[1] : k == 2 ? v[2] : v[3]; }
__device__ __forceinline__ int slot_byte(const unsigned (&w)[8], int jw, int jb) { return (int)((w[jw] >> (8 * jb)) & 0xff); }

// longest row of the wave (uniform): whole 4-slot words beyond it are skipped by the slot loops
__device__ __forceinline__ int wave_max_deg(int deg) {
  int w = max(deg, __shfl_xor(deg, 16));
  w = max(w, __shfl_xor(w, 32));
  return __builtin_amdgcn_readfirstlane(w);
}

// the tile's position-indexed metadata (round 1) and its staged rows (round 2, memory -> LDS by DMA)
struct TileMeta {
  int4 sc;              // {node (< 0: padding), row start in the direction's list, degree, -}
  unsigned w[8];        // the row's 32 slot bytes (every lane of the group holds them)
  int my[2];            // slot byte of list entries q and q + 16 of the row (this lane's two entries)
};
// round 1 alone (static per tile: the persistent solver keeps the result), round 2 alone (per phase), and both (one-launch layer)
__device__ __forceinline__ void tile_meta_load(const int2 *halo, const uint8_t *slots, const int4 *sched, int tile, int grp, int q,
                                               HaloRegs<GD> &hr, TileMeta &m) {
  halo_round1<GD>(halo, reinterpret_cast<const uint4 *>(slots), nullptr, tile, grp, true, hr);
  const size_t pos = (size_t)tile * kTM + grp;
  m.sc = sched[pos];
  m.my[0] = slots[pos * kSlotWidth + q];
  m.my[1] = slots[pos * kSlotWidth + 16 + q];
}
__device__ __forceinline__ void tile_meta_words(const HaloRegs<GD> &hr, TileMeta &m) {
  m.w[0] = hr.sl[0][0].x; m.w[1] = hr.sl[0][0].y; m.w[2] = hr.sl[0][0].z; m.w[3] = hr.sl[0][0].w;
  m.w[4] = hr.sl[0][1].x; m.w[5] = hr.sl[0][1].y; m.w[6] = hr.sl[0][1].z; m.w[7] = hr.sl[0][1].w;
}
__device__ __forceinline__ void tile_meta(const int2 *halo, const uint8_t *slots, const int4 *sched, const float *rows, int tile,
                                          int grp, int q, float *ldsXh, TileMeta &m) {
  HaloRegs<GD> hr;
  tile_meta_load(halo, slots, sched, tile, grp, q, hr, m);
  halo_round2<GD, true>(reinterpret_cast<const float4 *>(rows), q, grp, ldsXh, hr);
  tile_meta_words(hr, m);
}

// Out[32][head blocks] = A[32][heads * KIN] (x) per-head blocks of B, on v_mfma_f32_16x16x4_f32:
// wave tile (rt, ct): rows rt*16.., output columns ct*16..; contraction over KDIM starting at a_off / b_off.
template <int KDIM>
__device__ __forceinline__ f32x4 mfma_block(const float *pa, const float *pb) {
  f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int kb = 0; kb < KDIM / 16; ++kb) {
    const float4 a = *reinterpret_cast<const float4 *>(pa + kb * 16);
    const float4 b = *reinterpret_cast<const float4 *>(pb + kb * 16);
    acc = mfma16(a.x, b.x, acc);
    acc = mfma16(a.y, b.y, acc);
    acc = mfma16(a.z, b.z, acc);
    acc = mfma16(a.w, b.w, acc);
  }
  return acc;
}

// ---------------------------------------------------------------------------------------------------
// forward
// ---------------------------------------------------------------------------------------------------
#ifdef NGPDE_STAMPS
// diagnostic build only (tools/stamps_gat.py): shader-clock stamps of thread 0 of every workgroup
unsigned long long *g_gat_stamps = nullptr;
#define NGPDE_GST(k) do { if (threadIdx.x == 0 && p.stamps) { p.stamps[(size_t)blockIdx.x * 16 + (k)] = clock64(); if ((k) == 0 || (k) == 10) p.stamps[(size_t)blockIdx.x * 16 + 11 + (k) / 10] = wall_clock64(); } } while (0)
#define NGPDE_GST_FIELD unsigned long long *stamps;
#define NGPDE_GST_SET(kk) kk.stamps = g_gat_stamps;
#define NGPDE_GSTP(ptr, k) do { if (threadIdx.x == 0 && (ptr)) (ptr)[(size_t)blockIdx.x * 16 + (k)] = clock64(); } while (0)
#define NGPDE_GSTP_OF(kk) (kk).stamps
#else
#define NGPDE_GST(k)
#define NGPDE_GST_FIELD
#define NGPDE_GST_SET(kk)
#define NGPDE_GSTP(ptr, k)
#define NGPDE_GSTP_OF(kk) nullptr
#endif

struct GatFwdK {
  NGPDE_GST_FIELD
  const float *x, *wt, *a, *bias;
  const int4 *sched;
  const int2 *halo;
  const int2 *tile_info;   // [n_tiles] {rows staged, -}
  const uint8_t *slots;
  int n_tiles, act;
  float slope;
  float *y, *alpha, *save_z;
};

// The layer's forward in pieces, so that the one-launch layer and the persistent solver (gat_node_fwd_persistent_kernel) run the
// SAME code: per-launch constants (v vectors, W fragments), the per-tile computation from the staged rows to the pre-activation.
struct GatFwdLds {
  float *Xh;    // [(kHaloCap + 1)][GD]   staged input rows (the halo region comes first: an LDS-DMA destination is a 16-bit offset)
  float *S;     // [kTM][kSlotWidth][4]   alpha per (row, entry, head); later the output tile
  float *A;     // [kTM][kATS]            per-head aggregates
  float *Ar;    // [(kHaloCap + 1)][4]
  float *V;     // [2][4][GD]             v_l, v_r per head
};
constexpr int kFwdXhF = (kHaloCap + 1) * GD, kFwdSF = kTM * kSlotWidth * 4, kFwdAF = kTM * kATS, kFwdArF = (kHaloCap + 1) * 4,
              kFwdVF = 2 * 4 * GD;
struct GatThread {
  int tid, lane, wave_u, grp, q;
};
__device__ __forceinline__ GatThread gat_thread() {
  GatThread t;
  t.tid = threadIdx.x;
  t.lane = t.tid & 63;
  t.wave_u = __builtin_amdgcn_readfirstlane(t.tid >> 6);
  t.grp = t.tid / GG::LPR;
  t.q = t.tid % GG::LPR;
  return t;
}

template <int H>
__device__ __forceinline__ void gat_fwd_consts(const GatFwdK &p, const GatFwdLds &L, const GatThread &t, float (&breg)[4][4], float4 &b4) {
  const int lane = t.lane, wave_u = t.wave_u, grp = t.grp, q = t.q;
  float *ldsAr = L.Ar;
  float4 *Xh4 = reinterpret_cast<float4 *>(L.Xh);
  // B operand of this wave's column tile of W x straight from memory (W is 16 KB, cache resident): lane (i, kq) of column tile
  // ct = wave & 3 needs wt[16 kb + 4 kq + r][ct * 16 + i] -- 16 dwords, in flight from the start; no W^T copy in LDS
  {
    const int ct = wave_u & 3, i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) breg[kb][r] = p.wt[(size_t)(16 * kb + 4 * kq + r) * GD + ct * 16 + i];
  }
  b4 = p.bias ? reinterpret_cast<const float4 *>(p.bias)[q] : f4_zero();
  if (grp == 0) {
    Xh4[kHaloCap * GG::LPR + q] = f4_zero();
    if (q < 4) ldsAr[kHaloCap * 4 + q] = 0.f;
  }
}

// sum over the C / 4 adjacent lanes that hold one head's features (4, 8 or 16 lanes of a 16-lane row group): quad permutes, then
// the half-row / row mirrors; every lane of the head ends with the total
template <int LPH>
__device__ __forceinline__ float head_sum(float v) {
  v += dpp_mov<0xB1>(v);                      // quad_perm [1, 0, 3, 2]
  v += dpp_mov<0x4E>(v);                      // quad_perm [2, 3, 0, 1]
  if (LPH >= 8) v += dpp_mov<0x141>(v);       // row_half_mirror
  if (LPH == 16) v += dpp_mov<0x140>(v);      // row_mirror
  return v;
}

// From the barrier that makes the staged rows visible to the tile's pre-activation row of this thread (bias added).
// Round 3: transform, then aggregate -- W x of the STAGED rows comes off the matrix pipe once per tile ([rows staged][64] x [64][64]:
// <= 48 products per wave on a pipe that idles otherwise), the score halves are 4-float dots with a_l / a_r summed over the C / 4
// lanes of a head, and an entry of the aggregation costs ONE fma4 per lane (its own head's coefficient times its quad of the
// entry's W x row) instead of one per head; no [32][heads * 64] tile, no second product, two barriers fewer.  (Round 2 aggregated
// the INPUT rows per head and multiplied afterwards: 16 fma per entry and lane and sixteen-lane reductions for the scores -- the
// kernel was VALU-bound.)
template <int H>
__device__ __forceinline__ float4 gat_fwd_compute(const GatFwdK &p, const GatFwdLds &L, const GatThread &t, const TileMeta &m, int tile,
                                                  const float (&breg)[4][4], float4 b4) {
  constexpr int C = GD / H;
  constexpr int LPH = C / 4;
  const int lane = t.lane, wave_u = t.wave_u, grp = t.grp, q = t.q;
  float *ldsS = L.S, *ldsWX = L.A, *ldsAr = L.Ar, *ldsAl = L.V;
  float4 *Xh4 = reinterpret_cast<float4 *>(L.Xh);
  const int hq = (4 * q) / C;                                                    // this lane's head
  const float4 al4 = *reinterpret_cast<const float4 *>(p.a + (size_t)hq * 2 * C + (4 * q) % C);
  const float4 ar4 = *reinterpret_cast<const float4 *>(p.a + (size_t)hq * 2 * C + C + (4 * q) % C);
  const int hc = __builtin_amdgcn_readfirstlane(p.tile_info[tile].x);            // rows staged (own rows first)
  NGPDE_GST(1);
  __syncthreads();   // staged rows (DMA) visible
  NGPDE_GST(2);
  // ---- WX[hh][o] = sum_j x_hh[j] W[j][o]: <= 6 row tiles x 4 column tiles, wave w: column tile w & 3, row tiles (w >> 2), + 2, + 4
  {
    const int ct = wave_u & 3, i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int mm = 0; mm < 3; ++mm) {
      const int rt = (wave_u >> 2) + 2 * mm;
      if (rt * 16 < hc) {   // wave-uniform
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const float4 a4 = Xh4[(rt * 16 + i) * GG::LPR + 4 * kb + kq];
          acc = mfma16(a4.x, breg[kb][0], acc);
          acc = mfma16(a4.y, breg[kb][1], acc);
          acc = mfma16(a4.z, breg[kb][2], acc);
          acc = mfma16(a4.w, breg[kb][3], acc);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) ldsWX[(rt * 16 + 4 * kq + reg) * GG::TS + ct * 16 + i] = acc[reg];
      }
    }
    if (grp == 0) *reinterpret_cast<float4 *>(&ldsWX[kHaloCap * GG::TS + 4 * q]) = f4_zero();   // the all-zero row's product
  }
  NGPDE_GST(3);
  __syncthreads();
  NGPDE_GST(4);
  // ---- score halves: ar of every staged row (one lane per head writes it), al of the own row
  float al[4] = {0.f, 0.f, 0.f, 0.f};
  {
#pragma unroll
    for (int r = 0; r < GG::HI; ++r) {
      const int hh = grp + r * GG::GROUPS;
      if (r * GG::GROUPS < hc) {   // wave-uniform (rows of tiles that were not formed are never referenced)
        const float4 wx = *reinterpret_cast<const float4 *>(&ldsWX[hh * GG::TS + 4 * q]);
        const float sr = head_sum<LPH>(dot4(wx, ar4));
        if ((q & (LPH - 1)) == 0) ldsAr[hh * 4 + hq] = sr;
        if (r == 0) {
          const float sl = head_sum<LPH>(dot4(wx, al4));
          if ((q & (LPH - 1)) == 0) ldsAl[grp * 4 + hq] = sl;
        }
      }
    }
  }
  __syncthreads();   // ar of rows staged by other waves (and the own al) visible
  {
    const float4 a4 = *reinterpret_cast<const float4 *>(&ldsAl[grp * 4]);
    al[0] = a4.x; al[1] = a4.y; al[2] = a4.z; al[3] = a4.w;
  }

  // ---- softmax over the row's entries: lane q owns entries q and q + 16, all heads
  const int deg = m.sc.x >= 0 ? m.sc.z : 0;
  float av[2][4];
  {
    float sc[2][4], mx[4];
    bool pos[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const float4 arv = reinterpret_cast<const float4 *>(ldsAr)[m.my[s]];
      const float ar[4] = {arv.x, arv.y, arv.z, arv.w};
      const bool valid = q + 16 * s < deg;
#pragma unroll
      for (int k = 0; k < H; ++k) {
        const float v = al[k] + ar[k];
        pos[s][k] = v > 0.f;
        sc[s][k] = valid ? (v > 0.f ? v : p.slope * v) : -INFINITY;
      }
    }
#pragma unroll
    for (int k = 0; k < H; ++k) mx[k] = row_max16(fmaxf(sc[0][k], sc[1][k]));
    float inv[4];
#pragma unroll
    for (int k = 0; k < H; ++k) {
#pragma unroll
      for (int s = 0; s < 2; ++s) av[s][k] = (q + 16 * s < deg) ? fast_exp(sc[s][k] - mx[k]) : 0.f;
      const float sum = row_sum16(av[0][k] + av[1][k]);
      inv[k] = deg > 0 ? fast_rcp(sum) : 0.f;
    }
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float sg[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < H; ++k) {
        av[s][k] *= inv[k];
        sg[k] = pos[s][k] ? av[s][k] : -av[s][k];        // sign bit = leakyrelu's branch, for the pullback
      }
#pragma unroll
      for (int k = H; k < 4; ++k) av[s][k] = 0.f;
      reinterpret_cast<float4 *>(ldsS)[grp * kSlotWidth + q + 16 * s] = make_float4(av[s][0], av[s][1], av[s][2], av[s][3]);
      if (p.alpha && q + 16 * s < deg) {
        float *dst = p.alpha + (size_t)(m.sc.y + q + 16 * s) * H;
        if (H == 4) *reinterpret_cast<float4 *>(dst) = make_float4(sg[0], sg[1], sg[2], sg[3]);
        else if (H == 2) *reinterpret_cast<float2 *>(dst) = make_float2(sg[0], sg[1]);
        else dst[0] = sg[0];
      }
    }
  }
  // the coefficients are read back by the lanes of the same group (= same wave): no workgroup barrier needed
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  NGPDE_GST(5);
  // ---- aggregation of the entries' W x rows, this lane's head
  float4 acc = f4_zero();
  {
    const int wmax = wave_max_deg(deg);
#pragma unroll
    for (int jw = 0; jw < 8; ++jw) {
      if (jw * 4 < wmax) {   // wave-uniform
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          const float4 wx = *reinterpret_cast<const float4 *>(&ldsWX[slot_byte(m.w, jw, jb) * GG::TS + 4 * q]);
          acc = f4_fma(ldsS[(grp * kSlotWidth + jw * 4 + jb) * 4 + hq], wx, acc);
        }
      }
    }
  }
  NGPDE_GST(9);
  return f4_add(acc, b4);
}

template <int H>
__global__ __launch_bounds__(kThreads, 4) void gat_layer_fwd_kernel(const GatFwdK p) {
  __shared__ __attribute__((aligned(16))) float ldsXh[kFwdXhF];
  __shared__ __attribute__((aligned(16))) float ldsS[kFwdSF];
  __shared__ __attribute__((aligned(16))) float ldsA[kFwdAF];
  __shared__ __attribute__((aligned(16))) float ldsAr[kFwdArF];
  __shared__ __attribute__((aligned(16))) float ldsV[kFwdVF];
  const GatFwdLds L = {ldsXh, ldsS, ldsA, ldsAr, ldsV};
  const GatThread t = gat_thread();
  const int tile = xcd_tile(blockIdx.x, p.n_tiles);
  NGPDE_GST(0);
  TileMeta m;
  tile_meta(p.halo, p.slots, p.sched, p.x, tile, t.grp, t.q, ldsXh, m);
  float breg[4][4];
  float4 b4;
  gat_fwd_consts<H>(p, L, t, breg, b4);
  const float4 z = gat_fwd_compute<H>(p, L, t, m, tile, breg, b4);
  if (m.sc.x >= 0) {
    const size_t idx4 = (size_t)m.sc.x * GG::LPR + t.q;
    if (p.save_z) reinterpret_cast<float4 *>(p.save_z)[idx4] = z;
    reinterpret_cast<float4 *>(p.y)[idx4] = f4_act(p.act, z);
  }
  NGPDE_GST(10);
}

// ---------------------------------------------------------------------------------------------------
// pullback, by target:  dz, d alpha, softmax / leakyrelu pullback -> dscore, dal, db slabs
// ---------------------------------------------------------------------------------------------------
struct GatBwdTK {
  NGPDE_GST_FIELD
  const float *x, *wt, *dy, *yz, *alpha;
  const int4 *sched;
  const int2 *halo;
  const int2 *tile_info;   // [n_tiles] {rows staged, -}
  const uint8_t *slots;
  int n_tiles, act;
  float slope;
  float *dz, *dscore, *dal, *slab_db;
};

// write-through store (see node_persistent.hip: store_sc1 -- the s_nop 4 covers an SGPR base re-materialised by a spill reload right
// in front of the asm, the s_nop 1 the rewrite of the data registers): rows / entries another workgroup of the SAME launch reads
typedef float gat_f4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void gat_store_sc1(float *base, unsigned byte_off, float4 v) {
  gat_f4v t = {v.x, v.y, v.z, v.w};
  asm volatile("s_nop 4\n\tglobal_store_dwordx4 %0, %1, %2 sc1\n\ts_nop 1" ::"v"(byte_off), "v"(t), "s"(base) : "memory");
}
// agent-scope (sc1) loads of four consecutive floats written by another workgroup of the same launch
__device__ __forceinline__ float4 gat_load4_sc1(const float *ptr) {
  float4 v;
  v.x = __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.y = __hip_atomic_load(ptr + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.z = __hip_atomic_load(ptr + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  v.w = __hip_atomic_load(ptr + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return v;
}

struct GatBwdTLds {
  float *Xh;   // [(kHaloCap + 1)][GD] staged input rows
  float *DZ;   // [kTM][TS]  the dz tile
  float *DA;   // [(kHaloCap + 1)][TS] W x of the staged rows, then [kTM][kSlotWidth][4] d alpha per (row, entry, head)
};
constexpr int kBwdWXF = (kHaloCap + 1) * GG::TS;
constexpr int kBwdDZF = kTM * GG::TS, kBwdDAF = kBwdWXF + kTM * kSlotWidth * 4;

// The by-target half of the pullback for one tile, from the thread's dz row (staged input rows in flight or landed) to dscore / dal;
// returns this thread's partial of db (valid where tid % DBP == 0: column tid / DBP).  PAD: dscore goes to the tile's own
// 128-byte-aligned block [tile][32 rows][32 entries][H] with write-through stores instead of the by-target list order -- the
// persistent solver's by-source half reads it in the same launch, and entries of two tiles must not share a cache line.
template <int H, bool PAD>
__device__ __forceinline__ float gat_bwd_target_compute(const GatBwdTK &p, const GatBwdTLds &L, const GatThread &t, const TileMeta &m,
                                                        int tile, float4 dz) {
  constexpr int C = GD / H;
  const int tid = t.tid, lane = t.lane, wave_u = t.wave_u, grp = t.grp, q = t.q;
  float *ldsDZ = L.DZ, *ldsWX = L.DA, *ldsE = L.DA + kBwdWXF;
  float4 *Xh4 = reinterpret_cast<float4 *>(L.Xh);
  // d alpha_{e,k} = (W_k dz_{t,k}) . x_s = dz_{t,k} . (W x_s)_k: the products W x of the STAGED rows come off the matrix pipe once per
  // tile ([rows staged][64] x [64][64], the layer's own W x) and every entry then costs one 4-float dot per lane and a sum over the
  // C / 4 lanes of its head -- instead of a 64-float dot per head reduced over all 16 lanes of the row group (the VALU work that
  // dominated this half).  B operand of this wave's column tile: lane (i, kq) needs wt[16 kb + 4 kq + r][ct*16 + i], in flight
  // from the start (W is 16 KB, cache resident)
  float bw[4][4];
  {
    const int ct = wave_u & 3, i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) bw[kb][r] = p.wt[(size_t)(16 * kb + 4 * kq + r) * GD + ct * 16 + i];
  }
  const int hc = __builtin_amdgcn_readfirstlane(p.tile_info[tile].x);   // rows staged (own rows first)
  const bool ok = m.sc.x >= 0;
  const int deg = ok ? m.sc.z : 0;
  // this lane's two entries of the row: saved coefficients (sign = leakyrelu branch)
  float as[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const bool valid = q + 16 * s < deg;
    const float *src = p.alpha + (size_t)(m.sc.y + (valid ? q + 16 * s : 0)) * H;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    if (valid) {
      if (H == 4) { const float4 t = *reinterpret_cast<const float4 *>(src); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
      else if (H == 2) { const float2 t = *reinterpret_cast<const float2 *>(src); v[0] = t.x; v[1] = t.y; }
      else v[0] = src[0];
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) as[s][k] = v[k];
  }
  if (grp == 0) Xh4[kHaloCap * GG::LPR + q] = f4_zero();
  *reinterpret_cast<float4 *>(&ldsDZ[grp * GG::TS + 4 * q]) = dz;
  __syncthreads();
  NGPDE_GSTP(NGPDE_GSTP_OF(p), 7);
  float db_part;
  {   // db partial: column sums of the dz tile (8 adjacent lanes hold row-partials of one column)
    const int dbc = tid / GG::DBP, dbpart = tid % GG::DBP;
    float s = 0.f;
#pragma unroll
    for (int n = dbpart; n < kTM; n += GG::DBP) s += ldsDZ[n * GG::TS + dbc];
#pragma unroll
    for (int o = 1; o < GG::DBP; o <<= 1) s += __shfl_xor(s, o);
    db_part = s;
  }
  // WX[hh][o] = sum_j x_hh[j] W[j][o] for the staged rows: <= 6 row tiles x 4 column tiles, wave w: column tile w & 3, row tiles
  // (w >> 2), + 2, + 4
  {
    const int ct = wave_u & 3, i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int mm = 0; mm < 3; ++mm) {
      const int rt = (wave_u >> 2) + 2 * mm;
      if (rt * 16 < hc) {   // wave-uniform
        f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
          const float4 a4 = Xh4[(rt * 16 + i) * GG::LPR + 4 * kb + kq];
          acc = mfma16(a4.x, bw[kb][0], acc);
          acc = mfma16(a4.y, bw[kb][1], acc);
          acc = mfma16(a4.z, bw[kb][2], acc);
          acc = mfma16(a4.w, bw[kb][3], acc);
        }
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) ldsWX[(rt * 16 + 4 * kq + reg) * GG::TS + ct * 16 + i] = acc[reg];
      }
    }
    if (grp == 0) *reinterpret_cast<float4 *>(&ldsWX[kHaloCap * GG::TS + 4 * q]) = f4_zero();   // the all-zero row's product
  }
  __syncthreads();
  NGPDE_GSTP(NGPDE_GSTP_OF(p), 8);
  // d alpha of every entry of the row: lane q holds features 4q .. 4q + 3 of its row's dz, i.e. of head 4q / C; partial dot with the
  // entry's W x row, summed over the C / 4 lanes of the head (quad permutes, then half-row / row mirrors); one lane per head
  // writes it to the (row, entry, head) table, the lane that owns the entry reads its heads back
  float da[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  {
    constexpr int LPH = C / 4;   // lanes per head: 4, 8 or 16
    const int hq = (4 * q) / C;
    const float4 dzo = *reinterpret_cast<const float4 *>(&ldsDZ[grp * GG::TS + 4 * q]);
    const int wmax = wave_max_deg(deg);
#pragma unroll
    for (int jw = 0; jw < 8; ++jw) {
      if (jw * 4 < wmax) {   // wave-uniform
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          const int j = jw * 4 + jb;
          const float4 wx = *reinterpret_cast<const float4 *>(&ldsWX[slot_byte(m.w, jw, jb) * GG::TS + 4 * q]);
          float sum = dot4(dzo, wx);
          sum += dpp_mov<0xB1>(sum);                      // quad_perm [1, 0, 3, 2]
          sum += dpp_mov<0x4E>(sum);                      // quad_perm [2, 3, 0, 1]
          if (LPH >= 8) sum += dpp_mov<0x141>(sum);       // row_half_mirror
          if (LPH == 16) sum += dpp_mov<0x140>(sum);      // row_mirror
          if ((q & (LPH - 1)) == 0) ldsE[(grp * kSlotWidth + j) * 4 + hq] = sum;
        }
      }
    }
    // the table is read back by lanes of the same group (= same wave): no workgroup barrier needed
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int sI = 0; sI < 2; ++sI) {
      if (q + 16 * sI < wmax) {   // (entries beyond the wave's longest row were not written)
        const float4 e4 = *reinterpret_cast<const float4 *>(&ldsE[(grp * kSlotWidth + q + 16 * sI) * 4]);
        da[sI][0] = e4.x; da[sI][1] = e4.y; da[sI][2] = e4.z; da[sI][3] = e4.w;
      }
    }
  }
  NGPDE_GSTP(NGPDE_GSTP_OF(p), 9);
  // softmax pullback: dlogit = alpha (d alpha - sum alpha d alpha); leakyrelu' by the saved sign
  float dsc[2][4], dalv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < H; ++k) {
    const float a0 = fabsf(as[0][k]), a1 = fabsf(as[1][k]);
    const float t = row_sum16(fmaf(a0, da[0][k], a1 * da[1][k]));
    dsc[0][k] = a0 * (da[0][k] - t) * (as[0][k] < 0.f ? p.slope : 1.0f);
    dsc[1][k] = a1 * (da[1][k] - t) * (as[1][k] < 0.f ? p.slope : 1.0f);
    if (q >= deg) dsc[0][k] = 0.f;
    if (q + 16 >= deg) dsc[1][k] = 0.f;
    dalv[k] = row_sum16(dsc[0][k] + dsc[1][k]);
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    if (q + 16 * s < deg) {
      const float4 v4 = make_float4(dsc[s][0], H > 1 ? dsc[s][1] : 0.f, H > 2 ? dsc[s][2] : 0.f, H > 2 ? dsc[s][3] : 0.f);
      if constexpr (PAD) {
        gat_store_sc1(p.dscore, (unsigned)((((size_t)tile * kTM + grp) * kSlotWidth + q + 16 * s) * 4 * sizeof(float)), v4);
      } else {
        float *dst = p.dscore + (size_t)(m.sc.y + q + 16 * s) * H;
        if (H == 4) *reinterpret_cast<float4 *>(dst) = v4;
        else if (H == 2) *reinterpret_cast<float2 *>(dst) = make_float2(dsc[s][0], dsc[s][1]);
        else dst[0] = dsc[s][0];
      }
    }
  }
  if (ok && q < H) p.dal[(size_t)m.sc.x * H + q] = sel4(dalv, q);
  return db_part;
}

template <int H>
__global__ __launch_bounds__(kThreads, 4) void gat_layer_bwd_target_kernel(const GatBwdTK p) {
  __shared__ __attribute__((aligned(16))) float ldsXh[kFwdXhF];
  __shared__ __attribute__((aligned(16))) float ldsDZ[kBwdDZF];
  __shared__ __attribute__((aligned(16))) float ldsDA[kBwdDAF];
  const GatBwdTLds L = {ldsXh, ldsDZ, ldsDA};
  const GatThread t = gat_thread();
  const int tile = xcd_tile(blockIdx.x, p.n_tiles);
  TileMeta m;
  tile_meta(p.halo, p.slots, p.sched, p.x, tile, t.grp, t.q, ldsXh, m);
  const bool ok = m.sc.x >= 0;
  const size_t idx4 = (size_t)max(m.sc.x, 0) * GG::LPR + t.q;
  float4 dz = reinterpret_cast<const float4 *>(p.dy)[idx4];
  if (p.yz) dz = f4_mul(dz, f4_dact(p.act, reinterpret_cast<const float4 *>(p.yz)[idx4]));
  if (!ok) dz = f4_zero();
  if (ok && p.dz) reinterpret_cast<float4 *>(p.dz)[idx4] = dz;
  const float db_part = gat_bwd_target_compute<H, false>(p, L, t, m, tile, dz);
  if (t.tid % GG::DBP == 0) p.slab_db[(size_t)tile * GD + t.tid / GG::DBP] = db_part;
}

// ---------------------------------------------------------------------------------------------------
// pullback, by source:  dWx = sum alpha dz[t] + dal a_l + dar a_r;  dx = dWx W^T;  dW, u_l, u_r slabs
// ---------------------------------------------------------------------------------------------------
struct GatBwdSK {
  NGPDE_GST_FIELD
  const float *gz, *x, *wt, *a, *alpha, *dscore, *dal;
  const int4 *sched;
  const int2 *halo;
  const uint8_t *slots;
  const int *xpos;
  const int *xpad;   // by-source list position -> entry of the padded dscore blocks (persistent solver), else null
  int n_tiles, n_edges;
  float *dx, *slab_dw, *slab_u;
};

struct GatBwdSLds {
  float *Xh;    // [(kHaloCap + 1)][GD] staged dz rows; later the dx tile
  float *S;     // [kTM][kSlotWidth][4]
  float *DWX;   // [kTM][TS]
  float *XT;    // [kTM][TS]
  float *Bt;    // [GD][TS]   W, a straight copy
  float *DD;    // [kTM][8]   dal | dar of the tile's rows
};
constexpr int kBwdSF = kTM * kSlotWidth * 4, kBwdTileF = kTM * GG::TS, kBwdBtF = GD * GG::TS, kBwdDDF = kTM * 8;

// dx = dWx x B with B[k = out feature][j = in feature] = wt[j][k]: Bt[j][k] = wt[j][k], a straight copy
__device__ __forceinline__ void gat_load_bt(const float *wt, float *ldsBt, int tid) {
#pragma unroll
  for (int k = 0; k < GG::W4; ++k) {
    const int idx = tid + k * kThreads;
    *reinterpret_cast<float4 *>(&ldsBt[((idx * 4) / GD) * GG::TS + (idx * 4) % GD]) = reinterpret_cast<const float4 *>(wt)[idx];
  }
}

// The by-source half's loads that do not depend on the by-target halves of this pullback (gat_bwd_source_indices: static per tile;
// gat_bwd_source_prefetch: the saved coefficients and the row of x): the persistent solver issues both BEFORE it waits for its
// neighbours' dz / dscore -- behind the wait only the rows and the dscore entries are left (adjoint launch 7.21 -> 7.10 ms at C3's
// size, profiles/r06_o_gat_adjoint_variants.txt; the same file has what did NOT pay: the indices kept in registers for the whole
// launch, W / coefficients / stage adjoints asked for in one batch at the head of the by-target half, the forward's combination
// rows in one batch -- every one of them costs registers the kernels do not have (128, 53 spilled) and came out 1 - 40 % slower).
struct GatBwdSPre {
  int pp[2], xq[2];     // this lane's two outgoing entries: position in the by-target list, entry of the padded dscore blocks (PAD)
  float va[2][4];       // their saved coefficients
  float4 xo;            // the row of x
};
template <bool PAD>
__device__ __forceinline__ void gat_bwd_source_indices(const GatBwdSK &p, const GatThread &t, int4 sc, GatBwdSPre &r) {
  const int q = t.q;
  const int deg = sc.x >= 0 ? sc.z : 0;
  const int elast = p.n_edges - 1;      // (>= 0: gat_layer_fused_supported refuses a graph without edges)
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    // (a lane beyond its row's degree re-reads the row's first entry and drops it: the same address as lane 0, so no further request --
    // reading on into the next rows' entries made the persistent adjoint 25 % slower, its dscore loads go to memory)
    const int ent = max(min(sc.y + (q + 16 * s < deg ? q + 16 * s : 0), elast), 0);
    r.pp[s] = p.xpos[ent];
    r.xq[s] = PAD ? p.xpad[ent] : 0;
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    asm volatile("" : "+v"(r.pp[s]));
    asm volatile("" : "+v"(r.xq[s]));
  }
}
template <int H>
__device__ __forceinline__ void gat_bwd_source_prefetch(const GatBwdSK &p, const GatThread &t, int4 sc, GatBwdSPre &r) {
  r.xo = reinterpret_cast<const float4 *>(p.x)[(size_t)max(sc.x, 0) * GG::LPR + t.q];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int k = 0; k < 4; ++k) r.va[s][k] = 0.f;
    if (H == 4) { const float4 t4 = *reinterpret_cast<const float4 *>(p.alpha + (size_t)r.pp[s] * 4); r.va[s][0] = t4.x; r.va[s][1] = t4.y; r.va[s][2] = t4.z; r.va[s][3] = t4.w; }
    else if (H == 2) { const float2 t2 = *reinterpret_cast<const float2 *>(p.alpha + (size_t)r.pp[s] * 2); r.va[s][0] = t2.x; r.va[s][1] = t2.y; }
    else r.va[s][0] = p.alpha[r.pp[s]];
  }
}

// The by-source half of the pullback for one tile whose dz halo rows are staged (in flight or landed): returns this thread's quad of
// the tile's dx row; dW / u accumulate in the caller's registers.  Ends with the barrier after which the LDS regions may be reused.
template <int H, bool PAD>
__device__ __forceinline__ float4 gat_bwd_source_core(const GatBwdSK &p, const GatBwdSLds &L, const GatThread &t, const TileMeta &m,
                                                      float4 al4, float4 ar4, f32x4 (&dw)[GG::DWT], float &uacc, const GatBwdSPre &pre) {
  constexpr int C = GD / H;
  constexpr int NT = GG::CT * GG::CT;
  const int tid = t.tid, lane = t.lane, wave_u = t.wave_u, grp = t.grp, q = t.q;
  float *ldsXh = L.Xh, *ldsS = L.S, *ldsDWX = L.DWX, *ldsXT = L.XT, *ldsBt = L.Bt, *ldsDD = L.DD;
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
  const int hq = (4 * q) / C;                                                    // this lane's head
  const bool ok = m.sc.x >= 0;
  const int deg = ok ? m.sc.z : 0;
  float4 xo = pre.xo;
  // Every load here and in the two functions above is unconditional, from a clamped position, pinned after its batch and neutralised
  // afterwards.  As `valid ? load : 0` each was an exec-masked branch whose join waits for the load, and the lane's two entries went
  // one after the other: list position, wait, coefficient + dscore, wait, twice -- four round trips behind the row gather instead of two.
  float dalq = p.dal[(size_t)max(m.sc.x, 0) * H + min(q, H - 1)];
  // dscore of this lane's two outgoing entries (the same edges in the by-target list)
  float av[2][4], ds[2][4];
  float va[2][4], vd[2][4];
#pragma unroll
  for (int s = 0; s < 2; ++s) {
#pragma unroll
    for (int k = 0; k < 4; ++k) { va[s][k] = pre.va[s][k]; vd[s][k] = 0.f; }
    if (PAD) {   // the persistent solver: dscore from the target tile's padded block, sc1
      const float4 u = gat_load4_sc1(p.dscore + (size_t)pre.xq[s] * 4);
      vd[s][0] = u.x; vd[s][1] = u.y; vd[s][2] = u.z; vd[s][3] = u.w;
    } else {
      if (H == 4) { const float4 u = *reinterpret_cast<const float4 *>(p.dscore + (size_t)pre.pp[s] * 4); vd[s][0] = u.x; vd[s][1] = u.y; vd[s][2] = u.z; vd[s][3] = u.w; }
      else if (H == 2) { const float2 u = *reinterpret_cast<const float2 *>(p.dscore + (size_t)pre.pp[s] * 2); vd[s][0] = u.x; vd[s][1] = u.y; }
      else vd[s][0] = p.dscore[pre.pp[s]];
    }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int k = 0; k < H; ++k) {
      asm volatile("" : "+v"(va[s][k]));
      asm volatile("" : "+v"(vd[s][k]));
    }
  asm volatile("" : "+v"(dalq));
  asm volatile("" : "+v"(xo.x)); asm volatile("" : "+v"(xo.y)); asm volatile("" : "+v"(xo.z)); asm volatile("" : "+v"(xo.w));
  if (!ok) xo = f4_zero();
  if (!(ok && q < H)) dalq = 0.f;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const bool valid = q + 16 * s < deg;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      av[s][k] = valid ? fabsf(va[s][k]) : 0.f;
      ds[s][k] = valid ? vd[s][k] : 0.f;
    }
    reinterpret_cast<float4 *>(ldsS)[grp * kSlotWidth + q + 16 * s] = make_float4(av[s][0], av[s][1], av[s][2], av[s][3]);
  }
  if (grp == 0) Xh4[kHaloCap * GG::LPR + q] = f4_zero();
  float darv[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int k = 0; k < H; ++k) darv[k] = row_sum16(ds[0][k] + ds[1][k]);
  // dal of the row's heads: lanes q < H hold one each
  float dalv[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) dalv[k] = __shfl(dalq, k, GG::LPR);
  NGPDE_GSTP(NGPDE_GSTP_OF(p), 10);
  __syncthreads();   // staged dz rows visible (coefficients are group-private, same wave)
  NGPDE_GSTP(NGPDE_GSTP_OF(p), 11);
  float4 g = f4_zero();
  {
    const int wmax = wave_max_deg(deg);
#pragma unroll
    for (int jw = 0; jw < 8; ++jw) {
      if (jw * 4 < wmax) {   // wave-uniform
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          const float4 zv = Xh4[slot_byte(m.w, jw, jb) * GG::LPR + q];
          g = f4_fma(ldsS[(grp * kSlotWidth + jw * 4 + jb) * 4 + hq], zv, g);
        }
      }
    }
  }
  g = f4_fma(sel4(dalv, hq), al4, g);
  g = f4_fma(sel4(darv, hq), ar4, g);
  if (!ok) g = f4_zero();
  *reinterpret_cast<float4 *>(&ldsDWX[grp * GG::TS + 4 * q]) = g;
  *reinterpret_cast<float4 *>(&ldsXT[grp * GG::TS + 4 * q]) = xo;
  if (q < 4) ldsDD[grp * 8 + q] = ok ? sel4(dalv, q) : 0.f;
  else if (q < 8) ldsDD[grp * 8 + q] = ok ? sel4(darv, q - 4) : 0.f;
  NGPDE_GSTP(NGPDE_GSTP_OF(p), 12);
  __syncthreads();   // tiles complete, staged rows dead
  NGPDE_GSTP(NGPDE_GSTP_OF(p), 13);
  mfma_rows_times_bt<GD>(ldsDWX, ldsBt, ldsXh, wave_u, lane);
  {   // dWt[i][o] += sum_n x[n][i] dWx[n][o]
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int mm = 0; mm < GG::DWT; ++mm) {
      const int t2 = wave_u + GG::WAVES * mm;
      if (t2 < NT) {
        const int mt = t2 / GG::CT, nt = t2 % GG::CT;
#pragma unroll
        for (int ks = 0; ks < kTM / 4; ++ks)
          dw[mm] = mfma16(ldsXT[(4 * ks + kq) * GG::TS + mt * 16 + i], ldsDWX[(4 * ks + kq) * GG::TS + nt * 16 + i], dw[mm]);
      }
    }
  }
  NGPDE_GSTP(NGPDE_GSTP_OF(p), 14);
  {   // u_which,k[i] += sum_n (dal | dar)[n][k] x[n][i]
    const int which = tid >> 8, hk = (tid >> 6) & 3, i = tid & 63;
#pragma unroll 8
    for (int n = 0; n < kTM; ++n) uacc = fmaf(ldsDD[n * 8 + which * 4 + hk], ldsXT[n * GG::TS + i], uacc);
  }
  __syncthreads();
  const float4 dxv = *reinterpret_cast<const float4 *>(&ldsXh[grp * GG::TS + 4 * q]);
  __syncthreads();   // the next tile's rows land where the dx tile is
  return dxv;
}
template <int H, bool PAD>
__device__ __forceinline__ float4 gat_bwd_source_compute(const GatBwdSK &p, const GatBwdSLds &L, const GatThread &t, const TileMeta &m,
                                                         float4 al4, float4 ar4, f32x4 (&dw)[GG::DWT], float &uacc) {
  GatBwdSPre pre;
  gat_bwd_source_indices<PAD>(p, t, m.sc, pre);
  gat_bwd_source_prefetch<H>(p, t, m.sc, pre);
  return gat_bwd_source_core<H, PAD>(p, L, t, m, al4, ar4, dw, uacc, pre);
}

// this workgroup's share of da from its u accumulators (needs W in L.Bt)
template <int H>
__device__ __forceinline__ void gat_bwd_source_finish(float *slab_u_wg, const GatBwdSLds &L, const GatThread &t, float uacc) {
  constexpr int C = GD / H;
  const int tid = t.tid;
  float *ldsS = L.S, *ldsBt = L.Bt;
  // this workgroup's share of da:  da[(which*C + c) + 2C k] = sum_i W[k*C + c][i] u_which,k[i]   (linear in u: summed over the
  // workgroups by the reduction kernel)
  ldsS[tid] = uacc;
  __syncthreads();
  if (tid < 2 * GD) {
    const int k = tid / (2 * C), r = tid % (2 * C), which = r / C, cc = r % C;
    float sacc = 0.f;
#pragma unroll 8
    for (int i = 0; i < GD; ++i) sacc = fmaf(ldsBt[i * GG::TS + k * C + cc], ldsS[which * 256 + k * 64 + i], sacc);
    slab_u_wg[tid] = sacc;
  }
}

template <int H>
__global__ __launch_bounds__(kThreads, 4) void gat_layer_bwd_source_kernel(const GatBwdSK p) {
  constexpr int C = GD / H;
  __shared__ __attribute__((aligned(16))) float ldsXh[kFwdXhF];
  __shared__ __attribute__((aligned(16))) float ldsS[kBwdSF];
  __shared__ __attribute__((aligned(16))) float ldsDWX[kBwdTileF];
  __shared__ __attribute__((aligned(16))) float ldsXT[kBwdTileF];
  __shared__ __attribute__((aligned(16))) float ldsBt[kBwdBtF];
  __shared__ __attribute__((aligned(16))) float ldsDD[kBwdDDF];
  const GatBwdSLds L = {ldsXh, ldsS, ldsDWX, ldsXT, ldsBt, ldsDD};
  const GatThread t = gat_thread();
  const int q = t.q;
  const int n_wg = (p.n_tiles + kSrcTiles - 1) / kSrcTiles;
  const int wg = xcd_tile(blockIdx.x, n_wg);
  const int hq = (4 * q) / C;
  gat_load_bt(p.wt, ldsBt, t.tid);
  const float4 al4 = *reinterpret_cast<const float4 *>(p.a + (size_t)hq * 2 * C + (4 * q) % C);
  const float4 ar4 = *reinterpret_cast<const float4 *>(p.a + (size_t)hq * 2 * C + C + (4 * q) % C);
  constexpr int NT = GG::CT * GG::CT;
  f32x4 dw[GG::DWT];
#pragma unroll
  for (int mm = 0; mm < GG::DWT; ++mm) dw[mm] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float uacc = 0.f;
  for (int tt = 0; tt < kSrcTiles; ++tt) {
    const int tile = wg * kSrcTiles + tt;
    if (tile >= p.n_tiles) break;   // uniform
    TileMeta m;
    tile_meta(p.halo, p.slots, p.sched, p.gz, tile, t.grp, t.q, ldsXh, m);
    const float4 dxv = gat_bwd_source_compute<H, false>(p, L, t, m, al4, ar4, dw, uacc);
    if (m.sc.x >= 0 && p.dx) reinterpret_cast<float4 *>(p.dx)[(size_t)m.sc.x * GG::LPR + q] = dxv;
  }
  float4 *slab4 = reinterpret_cast<float4 *>(p.slab_dw + (size_t)wg * GD * GD);
#pragma unroll
  for (int mm = 0; mm < GG::DWT; ++mm) {
    const int t2 = t.wave_u + GG::WAVES * mm;
    if (t2 < NT) slab4[t2 * 64 + t.lane] = make_float4(dw[mm][0], dw[mm][1], dw[mm][2], dw[mm][3]);
  }
  gat_bwd_source_finish<H>(p.slab_u + (size_t)wg * 2 * GD, L, t, uacc);
}

// slabs -> dWt (row-major [in][out]), db, da.  Blocks 0 .. 63: 64 elements of dW each; block 64: db; blocks 65, 66: da.
__global__ __launch_bounds__(1024) void gat_layer_reduce_kernel(const float *__restrict__ slab_dw, int n_wg,
                                                                const float *__restrict__ slab_db, int n_tiles,
                                                                const float *__restrict__ slab_da, float *__restrict__ dwt,
                                                                float *__restrict__ db, float *__restrict__ da) {
  __shared__ float part[16][64];
  const int el = threadIdx.x & 63, pid = threadIdx.x >> 6;
  const int b = blockIdx.x;
  const float *slab = b < 64 ? slab_dw : b == 64 ? slab_db : slab_da;
  const int n = b == 64 ? n_tiles : n_wg, len = b < 64 ? GD * GD : b == 64 ? GD : 2 * GD;
  const int e = (b < 64 ? b : b == 64 ? 0 : b - 65) * 64 + el;
  if (b == 64 && db == nullptr) return;
  float s0 = 0.f, s1 = 0.f;
  int k = pid;
  for (; k + 16 < n; k += 32) {
    s0 += slab[(size_t)k * len + e];
    s1 += slab[(size_t)(k + 16) * len + e];
  }
  for (; k < n; k += 16) s0 += slab[(size_t)k * len + e];
  part[pid][el] = s0 + s1;
  __syncthreads();
  if (pid == 0) {
    float v = 0.f;
#pragma unroll
    for (int k2 = 0; k2 < 16; ++k2) v += part[k2][el];
    if (b == 64) {
      db[el] = v;
    } else if (b > 64) {
      da[e] = v;
    } else {   // e = (tt * 64 + lane) * 4 + reg  ->  dWt[(mt*16 + 4*kq + reg) * 64 + nt*16 + i]
      const int reg = e & 3, ln = (e >> 2) & 63, tt = e >> 8, mt = tt / GG::CT, nt = tt % GG::CT;
      dwt[(mt * 16 + 4 * (ln >> 4) + reg) * GD + nt * 16 + (ln & 15)] = v;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The layer as the right-hand side of a fixed-step explicit Runge-Kutta solve, device-resident: ONE persistent launch for the
// forward solve, ONE for the discrete adjoint (BASELINE config 3 "GAT as ODE RHS"; /root/reference/docs/src/tutorials/VMH.md:85-89
// NeuralODE(layer), graph_node.md:44-66).  A workgroup keeps its tile for the whole solve and runs the SAME per-tile code as the
// one-launch layer above (gat_fwd_compute, gat_bwd_target_compute, gat_bwd_source_compute), with the Runge-Kutta combinations
// of its own rows in between -- in the order and with the coefficients of the generic solver (node.py: _rk_forward / _rk_backward
// on ngpde_rk_stage_combine), so u(T) and du0 are bitwise those of the generic path.  Tiles synchronise through per-tile phase
// flags as in node_persistent.hip (wait list = symmetric closure of "my halo references a row of yours" over both directions,
// rows exchanged by write-through stores and sc1 LDS-DMA loads, bounded spins with an abort word):
//   forward, phase (n, i):  wait for the neighbours' phase before; stage the input rows (own + halo) of stage i; the layer; the
//     input of the next stage (or the step update) of the own rows -> the next slot of `xs`; publish.  One hand-off per
//     right-hand side.  With a tape every phase has its own slot of xs (it IS the tape of stage inputs), else two slots ping-pong.
//   adjoint, phase (n, i) in reverse:  K-bar_i of the own rows (lambda, U-bar_j: own rows, in memory, same thread writes and
//     reads); by-target half -> dz rows and dscore blocks (write-through, double-buffered by phase parity); publish; wait for
//     the neighbours' publish of the SAME phase; by-source half -> U-bar_i = dx.  One hand-off per right-hand side.
//     dW / u / db accumulate in registers over the whole adjoint; one slab per tile at the end, reduced by gat_layer_reduce_kernel.
// ---------------------------------------------------------------------------------------------------
constexpr int kGatNbrStride = 64;   // = node_persistent.hip's wait-list stride (node_persistent_setup builds the lists)
struct GatSync {
  const int *nbr;        // [n_tiles][64] wait lists, -1 padded
  unsigned *flags;       // one 128-byte line per tile: the last published phase
  unsigned *abort_word;
};

// every tile of the wait list has published phase >= need (wave 0 polls: one flag per lane, lane 63 the abort word); false: aborted
__device__ __forceinline__ bool gat_wait(const GatSync &s, const GatThread &t, int my_nbr, unsigned need, int *s_ok) {
  if (need == 0) return true;
  if (t.wave_u == 0) {
    const unsigned *addr = (t.lane == 63) ? s.abort_word : (my_nbr >= 0 ? s.flags + 32 * my_nbr : nullptr);
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    bool ok = true;
    for (unsigned it = 1;; ++it) {
      unsigned f = need;
      if (addr) f = __hip_atomic_load(addr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (__any((int)(t.lane == 63 && f != 0))) { ok = false; break; }              // somebody gave up
      if (__all((int)(t.lane == 63 || f >= need))) break;
      if ((it & 1023u) == 0 && __builtin_amdgcn_s_memrealtime() - t0 > 200000000ull) {   // ~2 s of the 100 MHz counter
        if (t.lane == 0) __hip_atomic_store(s.abort_word, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ok = false;
        break;
      }
      __builtin_amdgcn_s_sleep(1);
    }
    if (t.lane == 0) *s_ok = ok ? 1 : 0;
  }
  __syncthreads();
  return *s_ok != 0;
}
// every storing wave drains, the workgroup meets, ONE lane publishes
__device__ __forceinline__ void gat_publish(const GatSync &s, const GatThread &t, int tile, unsigned ph) {
  wait_vmcnt0();
  __syncthreads();
  if (t.tid == 0) __hip_atomic_store(s.flags + 32 * tile, ph, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
#define NGPDE_GAT_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ float4 gat_ld4(const float *base, unsigned byte_off) {
  const gat_f4v v = *reinterpret_cast<NGPDE_GAT_GLOBAL const gat_f4v *>(reinterpret_cast<uintptr_t>(base) + byte_off);
  return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void gat_st4(float *base, unsigned byte_off, float4 v) {
  const gat_f4v w = {v.x, v.y, v.z, v.w};
  *reinterpret_cast<NGPDE_GAT_GLOBAL gat_f4v *>(reinterpret_cast<uintptr_t>(base) + byte_off) = w;
}
__device__ __forceinline__ float4 gat_nan4() {
  const float n = __int_as_float(0x7fc00000);
  return make_float4(n, n, n, n);
}

struct GatNodeFwdK {
  GatFwdK l;            // wt, a, bias, lists by target, n_tiles, act, slope (x, y, alpha, save_z are set per phase / unused)
  GatSync s;
  int n_steps, S, taped;
  const float *u_in;    // [N][64]
  float *u_out;         // [N][64]
  float *xs;            // stage inputs [n_steps * S (taped) or 2][N][64]; slot 0 holds u0 at launch
  float *yz;            // [n_steps * S][N][64]: y (relu) or z, or null (no tape / identity)
  float *alpha;         // [n_steps * S][E][H] or null
  float *kbuf;          // [S][N][64]: stage derivatives of the current step (own rows)
  size_t row_elems, alpha_elems;
  const float *cf;      // [(S + 1)][8]: row i < S: coefficient of k_j in the input of stage i; row S: in the step update
  // batches of identical structures (BATCH kernels): member mb reads u_in + mb * row_elems, owns xs / yz / alpha + mb * the
  // strides below; two members at a time, slot sl = mb & 1 with its own flag words (s.flags + sl * flag_stride) and its own
  // kbuf + sl * 7 * row_elems (rows 0..5: k_j, row 6: u of the current step)
  int n_members;
  size_t flag_stride, xs_stride, yz_stride, alpha_stride;
};

// BATCH: a block-diagonal batch of identical structures (test/runtests.jl:89-102), two members at a time per workgroup: while one
// member's rows and flag travel to the neighbours the workgroup computes the other member's phase (the idea of node_persistent.hip's
// two-slot kernels; here nothing but the accumulators lives in registers across phases, so the slots simply take turns).
template <int H>
__global__ __launch_bounds__(kThreads, 4) void gat_node_fwd_persistent_batch_kernel(const GatNodeFwdK p) {
  __shared__ __attribute__((aligned(16))) float ldsXh[kFwdXhF];
  __shared__ __attribute__((aligned(16))) float ldsS[kFwdSF];
  __shared__ __attribute__((aligned(16))) float ldsA[kFwdAF];
  __shared__ __attribute__((aligned(16))) float ldsAr[kFwdArF];
  __shared__ __attribute__((aligned(16))) float ldsV[kFwdVF];
  __shared__ float ldsC[64];
  __shared__ int s_ok;
  const GatFwdLds L = {ldsXh, ldsS, ldsA, ldsAr, ldsV};
  const GatThread t = gat_thread();
  const int tile = xcd_tile(blockIdx.x, p.l.n_tiles);
  TileMeta m;
  HaloRegs<GD> hr;
  tile_meta_load(p.l.halo, p.l.slots, p.l.sched, tile, t.grp, t.q, hr, m);
  tile_meta_words(hr, m);
  float breg[4][4];
  float4 b4;
  gat_fwd_consts<H>(p.l, L, t, breg, b4);
  if (t.tid < 64) ldsC[t.tid] = p.cf[t.tid];
  if (t.tid == 0) s_ok = 1;
  const int my_nbr = p.s.nbr[(size_t)tile * kGatNbrStride + t.lane];
  const bool ok = m.sc.x >= 0;
  const unsigned own = (unsigned)max(m.sc.x, 0) * (unsigned)(GD * 4) + (unsigned)(t.q * 16);
  const int S = p.S;
  const unsigned P = (unsigned)(p.n_steps * S);
  GatFwdK l = p.l;
  bool dead = false;
  unsigned ph0 = 0;
  __syncthreads();
  for (int mb = 0; mb < p.n_members && !dead; mb += 2, ph0 += P) {
    const int nsl = min(2, p.n_members - mb);
    for (int sl = 0; sl < nsl; ++sl)   // u of the slot's member -> row 6 of the slot's scratch
      if (ok) gat_st4(p.kbuf + (size_t)(sl * 7 + 6) * p.row_elems, own, gat_ld4(p.u_in + (size_t)(mb + sl) * p.row_elems, own));
    for (int n = 0; n < p.n_steps && !dead; ++n) {
      for (int i = 0; i < S && !dead; ++i) {
        const unsigned lp = (unsigned)(n * S + i) + 1, ph = ph0 + lp;
        const size_t e = lp - 1;
        const bool last = lp == P;
        for (int sl = 0; sl < nsl; ++sl) {
          GatSync ys = p.s;
          ys.flags = p.s.flags + (size_t)sl * p.flag_stride;
          float *xs = p.xs + (size_t)(mb + sl) * p.xs_stride, *kb = p.kbuf + (size_t)sl * 7 * p.row_elems;
          const float *X = xs + (p.taped ? e : (e & 1)) * p.row_elems;
          const int row = (i + 1 < S) ? i + 1 : S;
          if (!gat_wait(ys, t, my_nbr, lp > 1 ? ph - 1 : 0u, &s_ok)) { dead = true; break; }
          halo_round2<GD, true, 16>(reinterpret_cast<const float4 *>(X), t.q, t.grp, ldsXh, hr);
          l.alpha = p.alpha ? p.alpha + (size_t)(mb + sl) * p.alpha_stride + e * p.alpha_elems : nullptr;
          const float4 z = gat_fwd_compute<H>(l, L, t, m, tile, breg, b4);
          const float4 y = f4_act(l.act, z);
          if (p.yz && ok) gat_st4(p.yz + (size_t)(mb + sl) * p.yz_stride + e * p.row_elems, own, l.act == NGPDE_ACT_RELU ? y : z);
          float4 v = f4_scale(1.0f, gat_ld4(kb + (size_t)6 * p.row_elems, own));
          for (int j = 0; j < i; ++j) v = f4_fma(ldsC[row * 8 + j], gat_ld4(kb + (size_t)j * p.row_elems, own), v);
          v = f4_fma(ldsC[row * 8 + i], y, v);
          if (ok) {
            if (i + 1 < S) gat_st4(kb + (size_t)i * p.row_elems, own, y);
            else gat_st4(kb + (size_t)6 * p.row_elems, own, v);
            if (last) gat_st4(p.u_out + (size_t)(mb + sl) * p.row_elems, own, v);
            else gat_store_sc1(xs + (p.taped ? e + 1 : ((e + 1) & 1)) * p.row_elems, own, v);
          }
          if (!last) gat_publish(ys, t, tile, ph);
          else __syncthreads();   // (the other slot's DMA must not land in rows this slot's waves still read)
        }
      }
    }
  }
  if (dead && ok)
    for (int mb = 0; mb < p.n_members; ++mb) gat_st4(p.u_out + (size_t)mb * p.row_elems, own, gat_nan4());
}

template <int H>
__global__ __launch_bounds__(kThreads, 4) void gat_node_fwd_persistent_kernel(const GatNodeFwdK p) {
  __shared__ __attribute__((aligned(16))) float ldsXh[kFwdXhF];
  __shared__ __attribute__((aligned(16))) float ldsS[kFwdSF];
  __shared__ __attribute__((aligned(16))) float ldsA[kFwdAF];
  __shared__ __attribute__((aligned(16))) float ldsAr[kFwdArF];
  __shared__ __attribute__((aligned(16))) float ldsV[kFwdVF];
  __shared__ float ldsC[64];
  __shared__ int s_ok;
  const GatFwdLds L = {ldsXh, ldsS, ldsA, ldsAr, ldsV};
  const GatThread t = gat_thread();
  const int tile = xcd_tile(blockIdx.x, p.l.n_tiles);
  TileMeta m;
  HaloRegs<GD> hr;
  tile_meta_load(p.l.halo, p.l.slots, p.l.sched, tile, t.grp, t.q, hr, m);
  tile_meta_words(hr, m);
  float breg[4][4];
  float4 b4;
  gat_fwd_consts<H>(p.l, L, t, breg, b4);
  if (t.tid < 64) ldsC[t.tid] = p.cf[t.tid];
  if (t.tid == 0) s_ok = 1;
  const int my_nbr = p.s.nbr[(size_t)tile * kGatNbrStride + t.lane];
  const bool ok = m.sc.x >= 0;
  const unsigned own = (unsigned)max(m.sc.x, 0) * (unsigned)(GD * 4) + (unsigned)(t.q * 16);
  float4 u = ok ? gat_ld4(p.u_in, own) : f4_zero();
  const int S = p.S;
  GatFwdK l = p.l;
  bool dead = false;
  unsigned ph = 0;
  __syncthreads();
  for (int n = 0; n < p.n_steps && !dead; ++n) {
    for (int i = 0; i < S; ++i) {
      ++ph;
      NGPDE_GSTP(NGPDE_GSTP_OF(l), 0);
      const size_t e = ph - 1;
      const float *X = p.xs + (p.taped ? e : (e & 1)) * p.row_elems;
      // the input of the next stage / the step update, as ngpde_rk_stage_combine forms it: 1 * u, then the k_j in order
      const int row = (i + 1 < S) ? i + 1 : S;
      if (!gat_wait(p.s, t, my_nbr, ph - 1, &s_ok)) { dead = true; break; }
      NGPDE_GSTP(NGPDE_GSTP_OF(l), 13);
      halo_round2<GD, true, 16>(reinterpret_cast<const float4 *>(X), t.q, t.grp, ldsXh, hr);
      l.alpha = p.alpha ? p.alpha + e * p.alpha_elems : nullptr;
      const float4 z = gat_fwd_compute<H>(l, L, t, m, tile, breg, b4);
      const float4 y = f4_act(l.act, z);
      if (p.yz && ok) gat_st4(p.yz + e * p.row_elems, own, l.act == NGPDE_ACT_RELU ? y : z);
      float4 v = f4_scale(1.0f, u);
      for (int j = 0; j < i; ++j) v = f4_fma(ldsC[row * 8 + j], gat_ld4(p.kbuf + (size_t)j * p.row_elems, own), v);
      v = f4_fma(ldsC[row * 8 + i], y, v);
      if (i + 1 < S && ok) gat_st4(p.kbuf + (size_t)i * p.row_elems, own, y);   // (a padding row's thread addresses node 0)
      else u = v;
      const bool last = (n == p.n_steps - 1 && i == S - 1);
      if (ok) {
        if (last) gat_st4(p.u_out, own, v);
        else gat_store_sc1(p.xs + (p.taped ? e + 1 : ((e + 1) & 1)) * p.row_elems, own, v);
      }
      NGPDE_GSTP(NGPDE_GSTP_OF(l), 14);
      if (!last) gat_publish(p.s, t, tile, ph);
      NGPDE_GSTP(NGPDE_GSTP_OF(l), 15);
    }
  }
  if (dead && ok) gat_st4(p.u_out, own, gat_nan4());
}

struct GatNodeBwdK {
  NGPDE_GST_FIELD
  GatBwdTK t;           // wt, lists by target, n_tiles, act, slope, dal (x, alpha, dscore per phase; dy, yz, dz, slab_db unused)
  GatBwdSK s;           // wt, a, dal, lists by source, xpos, xpad, n_tiles, slab_dw, slab_u (gz, x, alpha, dscore per phase; dx unused)
  GatSync y;
  int n_steps, S;
  const float *xs, *yz, *alpha;   // the forward launch's tape
  const float *duT;               // [N][64]
  float *lam;                     // [N][64] out: du0
  float *ubar;                    // [S][N][64]: stage adjoints of the current step (own rows)
  float *dzbuf;                   // [2][N][64]
  float *dscore;                  // [2][n_tiles][32][32][4]
  float *slab_db;                 // [n_tiles][64]
  size_t row_elems, alpha_elems, dscore_elems;
  const float *cb;                // [S][8]: cb[i*8 + i] = dt b_i, cb[i*8 + j] (j > i) = dt a_ji
  // batches (BATCH kernel): member mb owns xs / yz / alpha + mb * stride, duT / lam + mb * row_elems; slot sl = mb & 1 owns
  // flags + sl * flag_stride, ubar + sl * 6 rows, dzbuf + sl * 2 rows arrays, dscore + sl * 2 blocks, dal + sl * dal_stride
  int n_members;
  size_t flag_stride, xs_stride, yz_stride, alpha_stride, dal_stride;
};

// BATCH adjoint: per phase, the by-target halves of both slots (each published as soon as its stores have drained), then the
// by-source halves -- each slot's wait for its neighbours has the other slot's half in front of it.  lambda lives in p.lam (own
// rows, same thread), the weight-gradient accumulators run on over slots and members.
template <int H>
__global__ __launch_bounds__(kThreads, 4) void gat_node_bwd_persistent_batch_kernel(const GatNodeBwdK p) {
  constexpr int C = GD / H;
  __shared__ __attribute__((aligned(16))) float lds[kFwdXhF + kBwdSF + 2 * kBwdTileF + kBwdBtF + kBwdDDF];
  __shared__ float ldsC[64];
  __shared__ int s_ok;
  float *ldsXh = lds;
  const GatBwdTLds LT = {ldsXh, ldsXh + kFwdXhF, ldsXh + kFwdXhF + kBwdDZF};
  float *o = ldsXh + kFwdXhF;
  const GatBwdSLds LS = {ldsXh, o, o + kBwdSF, o + kBwdSF + kBwdTileF, o + kBwdSF + 2 * kBwdTileF, o + kBwdSF + 2 * kBwdTileF + kBwdBtF};
  const GatThread t = gat_thread();
  const int tile = xcd_tile(blockIdx.x, p.t.n_tiles);
  if (t.tid < 64) ldsC[t.tid] = p.cb[t.tid];
  if (t.tid == 0) s_ok = 1;
  const int my_nbr = p.y.nbr[(size_t)tile * kGatNbrStride + t.lane];
  const int hq = (4 * t.q) / C;
  const float4 al4 = *reinterpret_cast<const float4 *>(p.s.a + (size_t)hq * 2 * C + (4 * t.q) % C);
  const float4 ar4 = *reinterpret_cast<const float4 *>(p.s.a + (size_t)hq * 2 * C + C + (4 * t.q) % C);
  constexpr int NT = GG::CT * GG::CT;
  f32x4 dw[GG::DWT];
#pragma unroll
  for (int mm = 0; mm < GG::DWT; ++mm) dw[mm] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float uacc = 0.f, dbacc = 0.f;
  const int node = p.t.sched[(size_t)tile * kTM + t.grp].x;
  const bool ok = node >= 0;
  const unsigned own = (unsigned)max(node, 0) * (unsigned)(GD * 4) + (unsigned)(t.q * 16);
  const int S = p.S;
  const unsigned P = (unsigned)(p.n_steps * S);
  GatBwdTK tk = p.t;
  GatBwdSK sk = p.s;
  GatBwdSPre spre;   // the by-source half's loads from the tape, asked for in front of its wait
  bool dead = false;
  unsigned ph0 = 0;
  __syncthreads();
  for (int mb = 0; mb < p.n_members && !dead; mb += 2, ph0 += P) {
    const int nsl = min(2, p.n_members - mb);
    for (int sl = 0; sl < nsl; ++sl)
      if (ok) gat_st4(p.lam + (size_t)(mb + sl) * p.row_elems, own, gat_ld4(p.duT + (size_t)(mb + sl) * p.row_elems, own));
    unsigned lp = 0;
    for (int n = p.n_steps - 1; n >= 0 && !dead; --n) {
      for (int i = S - 1; i >= 0 && !dead; --i) {
        ++lp;
        const unsigned ph = ph0 + lp;
        const size_t e = (size_t)n * S + i;
        for (int sl = 0; sl < nsl; ++sl) {   // ---- by target
          GatSync ys = p.y;
          ys.flags = p.y.flags + (size_t)sl * p.flag_stride;
          const float *X = p.xs + (size_t)(mb + sl) * p.xs_stride + e * p.row_elems;
          float *lam = p.lam + (size_t)(mb + sl) * p.row_elems, *ub = p.ubar + (size_t)sl * 6 * p.row_elems;
          TileMeta mt;
          tile_meta(p.t.halo, p.t.slots, p.t.sched, X, tile, t.grp, t.q, ldsXh, mt);
          tk.alpha = p.alpha + (size_t)(mb + sl) * p.alpha_stride + e * p.alpha_elems;
          tk.dscore = p.dscore + (size_t)(sl * 2 + (ph & 1)) * p.dscore_elems;
          tk.dal = p.t.dal + (size_t)sl * p.dal_stride;
          const float4 lamv = gat_ld4(lam, own);
          float4 yzv = f4_zero();
          if (p.yz) yzv = gat_ld4(p.yz + (size_t)(mb + sl) * p.yz_stride + e * p.row_elems, own);
          float4 v = f4_scale(ldsC[i * 8 + i], ok ? lamv : f4_zero());
          for (int j = i + 1; j < S; ++j) v = f4_fma(ldsC[i * 8 + j], gat_ld4(ub + (size_t)j * p.row_elems, own), v);
          if (p.yz) v = f4_mul(v, f4_dact(tk.act, yzv));
          if (!ok) v = f4_zero();
          float *dzb = p.dzbuf + (size_t)(sl * 2 + (ph & 1)) * p.row_elems;
          if (ok) gat_store_sc1(dzb, own, v);
          dbacc += gat_bwd_target_compute<H, true>(tk, LT, t, mt, tile, v);
          gat_publish(ys, t, tile, ph);
        }
        gat_load_bt(p.s.wt, LS.Bt, t.tid);   // (once per phase: the by-source half leaves it alone)
        for (int sl = 0; sl < nsl; ++sl) {   // ---- by source
          GatSync ys = p.y;
          ys.flags = p.y.flags + (size_t)sl * p.flag_stride;
          const float *X = p.xs + (size_t)(mb + sl) * p.xs_stride + e * p.row_elems;
          float *lam = p.lam + (size_t)(mb + sl) * p.row_elems, *ub = p.ubar + (size_t)sl * 6 * p.row_elems;
          const float *dzb = p.dzbuf + (size_t)(sl * 2 + (ph & 1)) * p.row_elems;
          TileMeta ms;
          HaloRegs<GD> hrs;
          tile_meta_load(p.s.halo, p.s.slots, p.s.sched, tile, t.grp, t.q, hrs, ms);
          tile_meta_words(hrs, ms);
          sk.x = X;
          sk.alpha = p.alpha + (size_t)(mb + sl) * p.alpha_stride + e * p.alpha_elems;
          sk.dscore = p.dscore + (size_t)(sl * 2 + (ph & 1)) * p.dscore_elems;
          sk.dal = p.s.dal + (size_t)sl * p.dal_stride;
          gat_bwd_source_indices<true>(sk, t, ms.sc, spre);
          gat_bwd_source_prefetch<H>(sk, t, ms.sc, spre);   // the tape's share of the half's loads: under the wait
          if (!gat_wait(ys, t, my_nbr, ph, &s_ok)) { dead = true; break; }
          halo_round2<GD, true, 16>(reinterpret_cast<const float4 *>(dzb), t.q, t.grp, ldsXh, hrs);
          const float4 dxv = gat_bwd_source_core<H, true>(sk, LS, t, ms, al4, ar4, dw, uacc, spre);
          if (i > 0) {
            if (ok) gat_st4(ub + (size_t)i * p.row_elems, own, dxv);
          } else {
            float4 w = f4_scale(1.0f, ok ? gat_ld4(lam, own) : f4_zero());
            w = f4_fma(1.0f, dxv, w);
            for (int j = 1; j < S; ++j) w = f4_fma(1.0f, gat_ld4(ub + (size_t)j * p.row_elems, own), w);
            if (ok) gat_st4(lam, own, w);
          }
        }
      }
    }
  }
  if (ok && dead)
    for (int mb = 0; mb < p.n_members; ++mb) gat_st4(p.lam + (size_t)mb * p.row_elems, own, gat_nan4());
  const float bad = __int_as_float(0x7fc00000);
  float4 *slab4 = reinterpret_cast<float4 *>(p.s.slab_dw + (size_t)tile * GD * GD);
#pragma unroll
  for (int mm = 0; mm < GG::DWT; ++mm) {
    const int t2 = t.wave_u + GG::WAVES * mm;
    if (t2 < NT) slab4[t2 * 64 + t.lane] = dead ? gat_nan4() : make_float4(dw[mm][0], dw[mm][1], dw[mm][2], dw[mm][3]);
  }
  if (t.tid % GG::DBP == 0) p.slab_db[(size_t)tile * GD + t.tid / GG::DBP] = dead ? bad : dbacc;
  __syncthreads();
  gat_load_bt(p.s.wt, LS.Bt, t.tid);
  __syncthreads();
  gat_bwd_source_finish<H>(p.s.slab_u + (size_t)tile * 2 * GD, LS, t, dead ? bad : uacc);
}

template <int H>
__global__ __launch_bounds__(kThreads, 4) void gat_node_bwd_persistent_kernel(const GatNodeBwdK p) {
  constexpr int C = GD / H;
  // one region for both halves: [Xh | DZ | DA] by target, [Xh | S | DWX | XT | Bt | DD] by source (W is copied into Bt before every
  // by-source half, under the wait: Bt overlaps DA)
  __shared__ __attribute__((aligned(16))) float lds[kFwdXhF + kBwdSF + 2 * kBwdTileF + kBwdBtF + kBwdDDF];
  static_assert(kBwdDZF + kBwdDAF <= kBwdSF + 2 * kBwdTileF + kBwdBtF + kBwdDDF, "the by-target tiles fit the by-source layout");
  __shared__ float ldsC[64];
  __shared__ int s_ok;
  float *ldsXh = lds;
  const GatBwdTLds LT = {ldsXh, ldsXh + kFwdXhF, ldsXh + kFwdXhF + kBwdDZF};
  float *o = ldsXh + kFwdXhF;
  const GatBwdSLds LS = {ldsXh, o, o + kBwdSF, o + kBwdSF + kBwdTileF, o + kBwdSF + 2 * kBwdTileF, o + kBwdSF + 2 * kBwdTileF + kBwdBtF};
  const GatThread t = gat_thread();
  const int tile = xcd_tile(blockIdx.x, p.t.n_tiles);
  if (t.tid < 64) ldsC[t.tid] = p.cb[t.tid];
  if (t.tid == 0) s_ok = 1;
  const int my_nbr = p.y.nbr[(size_t)tile * kGatNbrStride + t.lane];
  const int hq = (4 * t.q) / C;
  const float4 al4 = *reinterpret_cast<const float4 *>(p.s.a + (size_t)hq * 2 * C + (4 * t.q) % C);
  const float4 ar4 = *reinterpret_cast<const float4 *>(p.s.a + (size_t)hq * 2 * C + C + (4 * t.q) % C);
  constexpr int NT = GG::CT * GG::CT;
  f32x4 dw[GG::DWT];
#pragma unroll
  for (int mm = 0; mm < GG::DWT; ++mm) dw[mm] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float uacc = 0.f, dbacc = 0.f;
  const int node = p.t.sched[(size_t)tile * kTM + t.grp].x;   // (the same node in both directions' schedules: checked by the host)
  const bool ok = node >= 0;
  const unsigned own = (unsigned)max(node, 0) * (unsigned)(GD * 4) + (unsigned)(t.q * 16);
  float4 lam = ok ? gat_ld4(p.duT, own) : f4_zero();
  const int S = p.S;
  GatBwdTK tk = p.t;
  GatBwdSK sk = p.s;
  GatBwdSPre spre;   // the by-source half's loads from the tape, asked for in front of its wait
  bool dead = false;
  unsigned ph = 0;
  __syncthreads();
  for (int n = p.n_steps - 1; n >= 0 && !dead; --n) {
    for (int i = S - 1; i >= 0; --i) {
      ++ph;
      NGPDE_GSTP(NGPDE_GSTP_OF(p), 0);
      const size_t e = (size_t)n * S + i;
      const float *X = p.xs + e * p.row_elems;
      // ---- by target.  K-bar_i = (dt b_i) lambda + sum_{j > i} (dt a_ji) U-bar_j, in ngpde_rk_stage_combine's order
      TileMeta mt;
      tile_meta(p.t.halo, p.t.slots, p.t.sched, X, tile, t.grp, t.q, ldsXh, mt);
      tk.alpha = p.alpha + e * p.alpha_elems;
      tk.dscore = p.dscore + (size_t)(ph & 1) * p.dscore_elems;
      float4 yzv = f4_zero();
      if (p.yz) yzv = gat_ld4(p.yz + e * p.row_elems, own);
      float4 v = f4_scale(ldsC[i * 8 + i], lam);
      for (int j = i + 1; j < S; ++j) v = f4_fma(ldsC[i * 8 + j], gat_ld4(p.ubar + (size_t)j * p.row_elems, own), v);
      if (p.yz) v = f4_mul(v, f4_dact(tk.act, yzv));
      if (!ok) v = f4_zero();
      float *dzb = p.dzbuf + (size_t)(ph & 1) * p.row_elems;
      if (ok) gat_store_sc1(dzb, own, v);
      NGPDE_GSTP(NGPDE_GSTP_OF(p), 1);
      dbacc += gat_bwd_target_compute<H, true>(tk, LT, t, mt, tile, v);
      NGPDE_GSTP(NGPDE_GSTP_OF(p), 2);
      gat_publish(p.y, t, tile, ph);
      NGPDE_GSTP(NGPDE_GSTP_OF(p), 3);
      // ---- by source
      TileMeta ms;
      HaloRegs<GD> hrs;
      tile_meta_load(p.s.halo, p.s.slots, p.s.sched, tile, t.grp, t.q, hrs, ms);
      tile_meta_words(hrs, ms);
      gat_load_bt(p.s.wt, LS.Bt, t.tid);
      sk.x = X;
      sk.alpha = tk.alpha;
      sk.dscore = tk.dscore;
      gat_bwd_source_indices<true>(sk, t, ms.sc, spre);
      gat_bwd_source_prefetch<H>(sk, t, ms.sc, spre);   // the tape's share of the half's loads: under the wait
      NGPDE_GSTP(NGPDE_GSTP_OF(p), 4);
      if (!gat_wait(p.y, t, my_nbr, ph, &s_ok)) { dead = true; break; }
      NGPDE_GSTP(NGPDE_GSTP_OF(p), 5);
      halo_round2<GD, true, 16>(reinterpret_cast<const float4 *>(dzb), t.q, t.grp, ldsXh, hrs);
      const float4 dxv = gat_bwd_source_core<H, true>(sk, LS, t, ms, al4, ar4, dw, uacc, spre);
      NGPDE_GSTP(NGPDE_GSTP_OF(p), 6);
      if (i > 0) {
        if (ok) gat_st4(p.ubar + (size_t)i * p.row_elems, own, dxv);
      } else {   // lambda of the step before: 1 * lambda + sum_j 1 * U-bar_j, j ascending
        float4 w = f4_scale(1.0f, lam);
        w = f4_fma(1.0f, dxv, w);
        for (int j = 1; j < S; ++j) w = f4_fma(1.0f, gat_ld4(p.ubar + (size_t)j * p.row_elems, own), w);
        lam = ok ? w : f4_zero();
      }
    }
  }
  if (ok) gat_st4(p.lam, own, dead ? gat_nan4() : lam);
  const float bad = __int_as_float(0x7fc00000);
  float4 *slab4 = reinterpret_cast<float4 *>(p.s.slab_dw + (size_t)tile * GD * GD);
#pragma unroll
  for (int mm = 0; mm < GG::DWT; ++mm) {
    const int t2 = t.wave_u + GG::WAVES * mm;
    if (t2 < NT) slab4[t2 * 64 + t.lane] = dead ? gat_nan4() : make_float4(dw[mm][0], dw[mm][1], dw[mm][2], dw[mm][3]);
  }
  if (t.tid % GG::DBP == 0) p.slab_db[(size_t)tile * GD + t.tid / GG::DBP] = dead ? bad : dbacc;
  __syncthreads();
  gat_load_bt(p.s.wt, LS.Bt, t.tid);
  __syncthreads();
  gat_bwd_source_finish<H>(p.s.slab_u + (size_t)tile * 2 * GD, LS, t, dead ? bad : uacc);
}

inline bool no_fused_gat_layer_env() {   // read on every call: tests flip it inside one process
  const char *e = std::getenv("NGPDE_NO_FUSED_GAT_LAYER");
  return e && e[0] == '1';
}

inline int src_wgs(const ngpde_graph *g) { return (fused_num_blocks(g->n_nodes) + kSrcTiles - 1) / kSrcTiles; }

struct GatWs {   // carving of the caller's pullback workspace (256-byte aligned pieces)
  size_t dz, dscore, dal, slab_db, slab_dw, slab_u, total;
};
inline GatWs gat_ws(const ngpde_graph *g, int heads) {
  auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
  GatWs w;
  size_t o = 0;
  w.dz = o; o += up((size_t)g->n_nodes * GD * 4);
  w.dscore = o; o += up((size_t)std::max<int64_t>(g->n_edges, 1) * heads * 4);
  w.dal = o; o += up((size_t)g->n_nodes * heads * 4);
  w.slab_db = o; o += up((size_t)fused_num_blocks(g->n_nodes) * GD * 4);
  w.slab_dw = o; o += up((size_t)src_wgs(g) * GD * GD * 4);
  w.slab_u = o; o += up((size_t)src_wgs(g) * 2 * GD * 4);
  w.total = o;
  return w;
}

}  // namespace

#ifdef NGPDE_STAMPS
extern "C" int32_t ngpde_debug_set_gat_stamps(unsigned long long *dev_buf) {   // [n_tiles][16] or NULL
  g_gat_stamps = dev_buf;
  return NGPDE_OK;
}
#endif

bool gat_layer_fused_supported(const ngpde_graph *g, int din, int heads, int c) {
  return g && g->has_norm && g->n_edges > 0 && g->by_t.halo_ok && g->by_s.halo_ok && din == GD && heads * c == GD &&
         (heads == 1 || heads == 2 || heads == 4) && (uint64_t)g->n_nodes * GD * 4 < (1ull << 32) && !no_fused_gat_layer_env();
}

size_t gat_layer_workspace_bytes(const ngpde_graph *g, int heads) { return g ? gat_ws(g, heads).total : 0; }

int32_t launch_gat_layer_fwd(const ngpde_graph *g, int heads, float slope, int act, const float *x, const float *wt, const float *a,
                             const float *bias, float *y, float *alpha, float *save_z, hipStream_t stream) {
  if (g->n_nodes == 0) return NGPDE_OK;
  GatFwdK k;
  k.x = x; k.wt = wt; k.a = a; k.bias = bias; k.sched = g->by_t.sched; k.halo = g->by_t.halo; k.slots = g->by_t.slots;
  k.tile_info = g->by_t.tile_info;
  k.n_tiles = fused_num_blocks(g->n_nodes); k.act = act; k.slope = slope; k.y = y; k.alpha = alpha; k.save_z = save_z;
  NGPDE_GST_SET(k)
  const dim3 grid(k.n_tiles), block(kThreads);
  switch (heads) {
    case 1: hipLaunchKernelGGL(gat_layer_fwd_kernel<1>, grid, block, 0, stream, k); break;
    case 2: hipLaunchKernelGGL(gat_layer_fwd_kernel<2>, grid, block, 0, stream, k); break;
    default: hipLaunchKernelGGL(gat_layer_fwd_kernel<4>, grid, block, 0, stream, k); break;
  }
  NGPDE_LAUNCH_CHECK("gat_layer_fwd_kernel");
  return NGPDE_OK;
}

int32_t launch_gat_layer_bwd(const ngpde_graph *g, int heads, float slope, int act, const float *x, const float *wt, const float *a,
                             const float *yz, const float *alpha, const float *dy, float *dx, float *dwt, float *da, float *db,
                             void *workspace, size_t workspace_bytes, hipStream_t stream) {
  const GatWs w = gat_ws(g, heads);
  NGPDE_REQUIRE(workspace && workspace_bytes >= w.total, NGPDE_ERR_WORKSPACE, "ngpde_gat_layer_backward: workspace too small (%zu < %zu bytes)",
                workspace_bytes, w.total);
  char *ws = static_cast<char *>(workspace);
  float *dz = reinterpret_cast<float *>(ws + w.dz), *dscore = reinterpret_cast<float *>(ws + w.dscore);
  float *dal = reinterpret_cast<float *>(ws + w.dal), *slab_db = reinterpret_cast<float *>(ws + w.slab_db);
  float *slab_dw = reinterpret_cast<float *>(ws + w.slab_dw), *slab_u = reinterpret_cast<float *>(ws + w.slab_u);
  const int n_tiles = fused_num_blocks(g->n_nodes), n_wg = src_wgs(g);
  const bool ident = act == NGPDE_ACT_IDENTITY;
  if (g->n_nodes > 0) {
    GatBwdTK t;
    t.x = x; t.wt = wt; t.dy = dy; t.yz = ident ? nullptr : yz; t.alpha = alpha; t.sched = g->by_t.sched; t.halo = g->by_t.halo;
    t.tile_info = g->by_t.tile_info;
    t.slots = g->by_t.slots; t.n_tiles = n_tiles; t.act = act; t.slope = slope; t.dz = ident ? nullptr : dz; t.dscore = dscore;
    t.dal = dal; t.slab_db = slab_db;
    NGPDE_GST_SET(t)
    GatBwdSK s;
    s.gz = ident ? dy : dz; s.x = x; s.wt = wt; s.a = a; s.alpha = alpha; s.dscore = dscore; s.dal = dal; s.sched = g->by_s.sched;
    s.halo = g->by_s.halo; s.slots = g->by_s.slots; s.xpos = g->by_s.xpos; s.n_tiles = n_tiles; s.n_edges = (int)g->n_edges; s.dx = dx; s.slab_dw = slab_dw;
    s.slab_u = slab_u; s.xpad = nullptr;
    NGPDE_GST_SET(s)
    const dim3 block(kThreads);
    switch (heads) {
      case 1:
        hipLaunchKernelGGL(gat_layer_bwd_target_kernel<1>, dim3(n_tiles), block, 0, stream, t);
        hipLaunchKernelGGL(gat_layer_bwd_source_kernel<1>, dim3(n_wg), block, 0, stream, s);
        break;
      case 2:
        hipLaunchKernelGGL(gat_layer_bwd_target_kernel<2>, dim3(n_tiles), block, 0, stream, t);
        hipLaunchKernelGGL(gat_layer_bwd_source_kernel<2>, dim3(n_wg), block, 0, stream, s);
        break;
      default:
        hipLaunchKernelGGL(gat_layer_bwd_target_kernel<4>, dim3(n_tiles), block, 0, stream, t);
        hipLaunchKernelGGL(gat_layer_bwd_source_kernel<4>, dim3(n_wg), block, 0, stream, s);
        break;
    }
    NGPDE_LAUNCH_CHECK("gat_layer_bwd kernels");
  }
  hipLaunchKernelGGL(gat_layer_reduce_kernel, dim3(67), dim3(1024), 0, stream, slab_dw, g->n_nodes > 0 ? n_wg : 0, slab_db,
                     g->n_nodes > 0 ? n_tiles : 0, slab_u, dwt, db, da);
  NGPDE_LAUNCH_CHECK("gat_layer_reduce_kernel");
  return NGPDE_OK;
}


// ---- the persistent solver's host side -------------------------------------------------------------------------------------
namespace {
__global__ void gat_pad_of_p_kernel(const int4 *__restrict__ sched_t, int n_sched, int *__restrict__ pad_of_p) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos >= n_sched) return;
  const int4 sc = sched_t[pos];
  if (sc.x < 0) return;
  for (int j = 0; j < sc.z && j < kSlotWidth; ++j) pad_of_p[sc.y + j] = pos * kSlotWidth + j;
}
__global__ void gat_xpad_kernel(const int *__restrict__ xpos_s, const int *__restrict__ pad_of_p, int64_t n_edges, int *__restrict__ xpad) {
  const int64_t qq = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (qq < n_edges) xpad[qq] = pad_of_p[xpos_s[qq]];
}
__global__ void gat_sched_same_kernel(const int4 *__restrict__ a, const int4 *__restrict__ b, int n, unsigned *bad) {
  const int pos = blockIdx.x * blockDim.x + threadIdx.x;
  if (pos < n && a[pos].x != b[pos].x) *bad = 1u;
}
__global__ void gat_set_word_kernel(unsigned *w, unsigned v) {
  if (threadIdx.x == 0) *w = v;
}
__global__ void gat_latch_fault_kernel(const unsigned *abort_word, unsigned *fault) {
  if (threadIdx.x == 0 && *abort_word != 0) *fault = 1u;
}
inline GatSync gat_sync(const NodePersist &ps) {
  GatSync y;
  y.nbr = ps.nbr; y.flags = ps.sync; y.abort_word = ps.sync + (size_t)ps.n_tiles * 64;   // node_persistent_setup's layout
  return y;
}
}  // namespace

size_t gat_node_dscore_elems(const ngpde_graph *g) { return (size_t)g->n_sched * kSlotWidth * 4; }

bool gat_node_persistent_supported(const ngpde_graph *g, int heads, int c) {
  if (node_persistent_disabled_env() || !gat_layer_fused_supported(g, GD, heads, c)) return false;
  int dev = 0, cus = 0, occ = 1 << 30;
  if (hipGetDevice(&dev) != hipSuccess) return false;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) return false;
  auto take = [&](auto kernel) {   // every workgroup spins for its neighbours: all of them must be resident
    int o = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&o, kernel, kThreads, 0) != hipSuccess) o = 0;
    occ = std::min(occ, o);
  };
  switch (heads) {
    case 1: take(gat_node_fwd_persistent_kernel<1>); take(gat_node_bwd_persistent_kernel<1>);
            take(gat_node_fwd_persistent_batch_kernel<1>); take(gat_node_bwd_persistent_batch_kernel<1>); break;
    case 2: take(gat_node_fwd_persistent_kernel<2>); take(gat_node_bwd_persistent_kernel<2>);
            take(gat_node_fwd_persistent_batch_kernel<2>); take(gat_node_bwd_persistent_batch_kernel<2>); break;
    default: take(gat_node_fwd_persistent_kernel<4>); take(gat_node_bwd_persistent_kernel<4>);
             take(gat_node_fwd_persistent_batch_kernel<4>); take(gat_node_bwd_persistent_batch_kernel<4>); break;
  }
  const int nt = g->n_sched / kTileRows;
  if (nt < 1 || nt > cus * occ) return false;
  // both directions' schedules must name the same node at every position (the solver keeps per-thread state across the halves):
  // compared on the device ONCE per handle (a launch on the NULL stream and a blocking copy -- not something to repeat per solve)
  std::lock_guard<std::mutex> lock(g->lazy_mu);
  if (g->sched_same >= 0) return g->sched_same == 1;
  unsigned *bad = nullptr, h = 1;
  if (hipMalloc((void **)&bad, sizeof(unsigned)) != hipSuccess) return false;
  bool same = hipMemset(bad, 0, sizeof(unsigned)) == hipSuccess;
  if (same) {
    hipLaunchKernelGGL(gat_sched_same_kernel, dim3((g->n_sched + 255) / 256), dim3(256), 0, 0, g->by_t.sched, g->by_s.sched, g->n_sched, bad);
    same = hipMemcpy(&h, bad, sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess && h == 0;
  }
  (void)hipFree(bad);
  g->sched_same = same ? 1 : 0;
  return same;
}

int32_t launch_gat_node_xpad(const ngpde_graph *g, int *pad_of_p, int *xpad, hipStream_t stream) {
  if (g->n_edges == 0) return NGPDE_OK;
  hipLaunchKernelGGL(gat_pad_of_p_kernel, dim3((g->n_sched + 255) / 256), dim3(256), 0, stream, g->by_t.sched, g->n_sched, pad_of_p);
  hipLaunchKernelGGL(gat_xpad_kernel, dim3((unsigned)((g->n_edges + 255) / 256)), dim3(256), 0, stream, g->by_s.xpos, pad_of_p, g->n_edges, xpad);
  NGPDE_LAUNCH_CHECK("gat_xpad_kernel");
  return NGPDE_OK;
}

int32_t launch_gat_node_fwd(const GatNodeFwd &a, hipStream_t stream) {
  const ngpde_graph *g = a.g;
  const NodePersist &ps = *a.ps;
  int32_t st;
  PersistentTurn turn;
  if ((st = turn.enter(stream))) return st;
  if ((st = launch_zero(ps.sync, ps.sync_bytes, stream))) return st;
  GatNodeFwdK k;
  k.l.x = nullptr; k.l.wt = a.wt; k.l.a = a.a; k.l.bias = a.bias; k.l.sched = g->by_t.sched; k.l.halo = g->by_t.halo;
  k.l.tile_info = g->by_t.tile_info;
  k.l.slots = g->by_t.slots; k.l.n_tiles = ps.n_tiles; k.l.act = a.act; k.l.slope = a.slope; k.l.y = nullptr; k.l.alpha = nullptr;
  k.l.save_z = nullptr;
  NGPDE_GST_SET(k.l)
  k.s = gat_sync(ps);
  {
    const char *fa = std::getenv("NGPDE_DEBUG_FORCE_ABORT");
    if (fa && fa[0] == '1') hipLaunchKernelGGL(gat_set_word_kernel, dim3(1), dim3(64), 0, stream, k.s.abort_word, 1u);
  }
  k.n_steps = a.n_steps; k.S = a.S; k.taped = a.taped ? 1 : 0; k.u_in = a.u_in; k.u_out = a.u_out; k.xs = a.xs; k.yz = a.yz;
  k.alpha = a.alpha; k.kbuf = a.kbuf; k.row_elems = (size_t)g->n_nodes * GD; k.alpha_elems = (size_t)std::max<int64_t>(g->n_edges, 1) * a.heads;
  k.cf = a.cf;
  k.n_members = a.n_members; k.flag_stride = (size_t)ps.n_tiles * 32; k.xs_stride = a.xs_stride; k.yz_stride = a.yz_stride;
  k.alpha_stride = a.alpha_stride;
  const dim3 grid(ps.n_tiles), block(kThreads);
#define NGPDE_GN_LAUNCH(KERNEL)                                                                                   \
  if (a.ev_start) hipExtLaunchKernelGGL(KERNEL, grid, block, 0, stream, a.ev_start, a.ev_stop, 0, k);              \
  else hipLaunchKernelGGL(KERNEL, grid, block, 0, stream, k);
  if (a.n_members > 1) {
    switch (a.heads) {
      case 1: NGPDE_GN_LAUNCH(gat_node_fwd_persistent_batch_kernel<1>) break;
      case 2: NGPDE_GN_LAUNCH(gat_node_fwd_persistent_batch_kernel<2>) break;
      default: NGPDE_GN_LAUNCH(gat_node_fwd_persistent_batch_kernel<4>) break;
    }
  } else {
    switch (a.heads) {
      case 1: NGPDE_GN_LAUNCH(gat_node_fwd_persistent_kernel<1>) break;
      case 2: NGPDE_GN_LAUNCH(gat_node_fwd_persistent_kernel<2>) break;
      default: NGPDE_GN_LAUNCH(gat_node_fwd_persistent_kernel<4>) break;
    }
  }
  NGPDE_LAUNCH_CHECK("gat_node_fwd_persistent_kernel");
  hipLaunchKernelGGL(gat_latch_fault_kernel, dim3(1), dim3(64), 0, stream, k.s.abort_word, ps.fault);
  NGPDE_LAUNCH_CHECK("gat_latch_fault_kernel");
  return turn.leave();
}

int32_t launch_gat_node_bwd(const GatNodeBwd &a, hipStream_t stream) {
  const ngpde_graph *g = a.g;
  const NodePersist &ps = *a.ps;
  int32_t st;
  PersistentTurn turn;
  if ((st = turn.enter(stream))) return st;
  if ((st = launch_zero(ps.sync, ps.sync_bytes, stream))) return st;
  GatNodeBwdK k;
  const bool ident = a.act == NGPDE_ACT_IDENTITY;
  k.t.x = nullptr; k.t.wt = a.wt; k.t.dy = nullptr; k.t.yz = nullptr; k.t.alpha = nullptr; k.t.sched = g->by_t.sched; k.t.halo = g->by_t.halo;
  k.t.tile_info = g->by_t.tile_info;
  k.t.slots = g->by_t.slots; k.t.n_tiles = ps.n_tiles; k.t.act = a.act; k.t.slope = a.slope; k.t.dz = nullptr; k.t.dscore = nullptr;
  k.t.dal = a.dal; k.t.slab_db = nullptr;
  k.s.gz = nullptr; k.s.x = nullptr; k.s.wt = a.wt; k.s.a = a.a; k.s.alpha = nullptr; k.s.dscore = nullptr; k.s.dal = a.dal;
  k.s.sched = g->by_s.sched; k.s.halo = g->by_s.halo; k.s.slots = g->by_s.slots; k.s.xpos = g->by_s.xpos; k.s.xpad = a.xpad; k.s.n_edges = (int)g->n_edges;
  k.s.n_tiles = ps.n_tiles; k.s.dx = nullptr; k.s.slab_dw = a.slab_dw; k.s.slab_u = a.slab_u;
  k.y = gat_sync(ps);
  NGPDE_GST_SET(k)
  NGPDE_GST_SET(k.t)
  NGPDE_GST_SET(k.s)
  {
    const char *fa = std::getenv("NGPDE_DEBUG_FORCE_ABORT");
    if (fa && fa[0] == '1') hipLaunchKernelGGL(gat_set_word_kernel, dim3(1), dim3(64), 0, stream, k.y.abort_word, 1u);
  }
  k.n_steps = a.n_steps; k.S = a.S; k.xs = a.xs; k.yz = ident ? nullptr : a.yz; k.alpha = a.alpha; k.duT = a.duT; k.lam = a.lam;
  k.ubar = a.ubar; k.dzbuf = a.dzbuf; k.dscore = a.dscore; k.slab_db = a.slab_db; k.row_elems = (size_t)g->n_nodes * GD;
  k.alpha_elems = (size_t)std::max<int64_t>(g->n_edges, 1) * a.heads; k.dscore_elems = gat_node_dscore_elems(g); k.cb = a.cb;
  k.n_members = a.n_members; k.flag_stride = (size_t)ps.n_tiles * 32; k.xs_stride = a.xs_stride; k.yz_stride = a.yz_stride;
  k.alpha_stride = a.alpha_stride; k.dal_stride = (size_t)g->n_nodes * a.heads;
  NGPDE_REQUIRE(ident || a.yz, NGPDE_ERR_INVALID_ARGUMENT, "persistent GAT adjoint: the saved y / z rows are missing");
  const dim3 grid(ps.n_tiles), block(kThreads);
  if (a.n_members > 1) {
    switch (a.heads) {
      case 1: NGPDE_GN_LAUNCH(gat_node_bwd_persistent_batch_kernel<1>) break;
      case 2: NGPDE_GN_LAUNCH(gat_node_bwd_persistent_batch_kernel<2>) break;
      default: NGPDE_GN_LAUNCH(gat_node_bwd_persistent_batch_kernel<4>) break;
    }
  } else {
    switch (a.heads) {
      case 1: NGPDE_GN_LAUNCH(gat_node_bwd_persistent_kernel<1>) break;
      case 2: NGPDE_GN_LAUNCH(gat_node_bwd_persistent_kernel<2>) break;
      default: NGPDE_GN_LAUNCH(gat_node_bwd_persistent_kernel<4>) break;
    }
  }
#undef NGPDE_GN_LAUNCH
  NGPDE_LAUNCH_CHECK("gat_node_bwd_persistent_kernel");
  hipLaunchKernelGGL(gat_latch_fault_kernel, dim3(1), dim3(64), 0, stream, k.y.abort_word, ps.fault);
  hipLaunchKernelGGL(gat_layer_reduce_kernel, dim3(67), dim3(1024), 0, stream, a.slab_dw, ps.n_tiles, a.slab_db, ps.n_tiles, a.slab_u, a.dwt,
                     a.db, a.da);
  NGPDE_LAUNCH_CHECK("gat_layer_reduce_kernel");
  return turn.leave();
}

}  // namespace ngpde
