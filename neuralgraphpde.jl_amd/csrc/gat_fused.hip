// gat_fused.hip -- a whole GAT-style layer (64 => heads x c = 64, heads in {1, 2, 4}) as ONE forward launch and
// two pullback launches (+ one reduction) on the tile / halo machinery of the fused GCN kernels.
//   y_i = act( ||_k sum_{e: t_e = i} alpha_{e,k} W_k x_{s_e} + b ),   alpha = softmax_e leakyrelu(a_l,k . W_k x_i + a_r,k . W_k x_s)
// [GraphNeuralNetworks.jl GATConv on softmax_edge_neighbors, the primitive the reference re-exports at
//  /root/reference/src/NeuralGraphPDE.jl:7; BASELINE config 3: GATConv 4 heads x 16 on the 16k-node graph.]
//
// The layer is evaluated in a reassociated form so that nothing per node is written but the output:
//   * scores:  a_l,k . W_k x_i = v_l,k . x_i with v_l,k = W_k^T a_l,k (64 floats per head, rebuilt by every workgroup from W and
//     a: 2 KB, 16 products per element), so the logits come straight from the staged INPUT rows;
//   * messages:  sum_e alpha_{e,k} W_k x_s = W_k (sum_e alpha_{e,k} x_s): the tile aggregates its staged input rows once per
//     head (the LDS row read is shared by the heads) into a [32][heads * 64] tile and ONE fp32-MFMA product per head block
//     gives the output tile -- the same 32 x 64 x 64 product as a GCN layer, no W x array in memory.
// Saved for the pullback: alpha ([E][heads], p order) with the sign bit carrying leakyrelu's branch (alpha >= 0).
// Pullback:  (1) by target: dz = dy . act', d alpha_{e,k} = (W_k dz_i,k) . x_s with the 32 x (heads * 64) product on MFMA,
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
__device__ __forceinline__ void tile_meta(const int2 *halo, const uint8_t *slots, const int4 *sched, const float *rows, int tile,
                                          int grp, int q, float *ldsXh, TileMeta &m) {
  HaloRegs<GD> hr;
  halo_round1<GD>(halo, reinterpret_cast<const uint4 *>(slots), nullptr, tile, grp, true, hr);
  const size_t pos = (size_t)tile * kTM + grp;
  m.sc = sched[pos];
  m.my[0] = slots[pos * kSlotWidth + q];
  m.my[1] = slots[pos * kSlotWidth + 16 + q];
  halo_round2<GD, true>(reinterpret_cast<const float4 *>(rows), q, grp, ldsXh, hr);
  m.w[0] = hr.sl[0][0].x; m.w[1] = hr.sl[0][0].y; m.w[2] = hr.sl[0][0].z; m.w[3] = hr.sl[0][0].w;
  m.w[4] = hr.sl[0][1].x; m.w[5] = hr.sl[0][1].y; m.w[6] = hr.sl[0][1].z; m.w[7] = hr.sl[0][1].w;
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
#else
#define NGPDE_GST(k)
#define NGPDE_GST_FIELD
#define NGPDE_GST_SET(kk)
#endif

struct GatFwdK {
  NGPDE_GST_FIELD
  const float *x, *wt, *a, *bias;
  const int4 *sched;
  const int2 *halo;
  const uint8_t *slots;
  int n_tiles, act;
  float slope;
  float *y, *alpha, *save_z;
};

template <int H>
__global__ __launch_bounds__(kThreads, 4) void gat_layer_fwd_kernel(const GatFwdK p) {
  constexpr int C = GD / H;
  // the halo region first: an LDS-DMA destination is a 16-bit offset
  __shared__ __attribute__((aligned(16))) float ldsXh[(kHaloCap + 1) * GD];     // staged input rows
  __shared__ __attribute__((aligned(16))) float ldsS[kTM * kSlotWidth * 4];      // alpha per (row, entry, head); later the output tile
  __shared__ __attribute__((aligned(16))) float ldsA[kTM * kATS];                // per-head aggregates
  __shared__ __attribute__((aligned(16))) float ldsAr[(kHaloCap + 1) * 4];
  __shared__ __attribute__((aligned(16))) float ldsV[2 * 4 * GD];                // v_l, v_r per head
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = tid / GG::LPR, q = tid % GG::LPR;
  const int tile = xcd_tile(blockIdx.x, p.n_tiles);
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);

  NGPDE_GST(0);
  TileMeta m;
  tile_meta(p.halo, p.slots, p.sched, p.x, tile, grp, q, ldsXh, m);
  // v_which,k[i] = sum_c a[which*C + c][k] W[k*C + c][i]: the workgroup reads W once, coalesced (thread: input feature i =
  // tid >> 3, eight consecutive output columns), and the C / 8 adjacent lanes of one (i, head) add their partial dots
  {
    const int i = tid >> 3, part = tid & 7;
    const int hk = (part * 8) / C, c0 = (part * 8) % C;
    const float4 *w4 = reinterpret_cast<const float4 *>(p.wt + (size_t)i * GD + part * 8);
    const float4 *l4 = reinterpret_cast<const float4 *>(p.a + (size_t)hk * 2 * C + c0);
    const float4 *r4 = reinterpret_cast<const float4 *>(p.a + (size_t)hk * 2 * C + C + c0);
    const float4 w0 = w4[0], w1 = w4[1];
    float pl = dot4(w0, l4[0]) + dot4(w1, l4[1]), pr = dot4(w0, r4[0]) + dot4(w1, r4[1]);
#pragma unroll
    for (int o = 1; o < C / 8; o <<= 1) {
      pl += __shfl_xor(pl, o);
      pr += __shfl_xor(pr, o);
    }
    if (part % (C / 8) == 0) {
      ldsV[hk * GD + i] = pl;
      ldsV[4 * GD + hk * GD + i] = pr;
    }
  }
  // B operand of this wave's output tile straight from memory (W is 16 KB, cache resident): lane (i, kq) of column tile ct
  // needs wt[16 kb + 4 kq + r][ct * 16 + i] -- 16 dwords, in flight from the start; no W^T copy in LDS
  float breg[4][4];
  {
    const int ct = wave_u >> 1, i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int r = 0; r < 4; ++r) breg[kb][r] = p.wt[(size_t)(16 * kb + 4 * kq + r) * GD + ct * 16 + i];
  }
  const float4 b4 = p.bias ? reinterpret_cast<const float4 *>(p.bias)[q] : f4_zero();
  if (grp == 0) {
    Xh4[kHaloCap * GG::LPR + q] = f4_zero();
    if (q < 4) ldsAr[kHaloCap * 4 + q] = 0.f;
  }
  NGPDE_GST(1);
  __syncthreads();   // staged rows (DMA) and v vectors visible
  NGPDE_GST(2);

  // ---- score halves from the staged rows: ar of every staged row, al of the own row (= slot `grp`)
  float al[4] = {0.f, 0.f, 0.f, 0.f};
  {
    float4 vr[H], vl[H];
#pragma unroll
    for (int k = 0; k < H; ++k) {
      vl[k] = reinterpret_cast<const float4 *>(ldsV)[k * 16 + q];
      vr[k] = reinterpret_cast<const float4 *>(ldsV)[(4 + k) * 16 + q];
    }
#pragma unroll
    for (int r = 0; r < GG::HI; ++r) {
      const int hh = grp + r * GG::GROUPS;
      const float4 xv = Xh4[hh * GG::LPR + q];
      float s[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int k = 0; k < H; ++k) s[k] = row_sum16(dot4(xv, vr[k]));
      if (q < 4) ldsAr[hh * 4 + q] = sel4(s, q);
    }
    const float4 xo = Xh4[grp * GG::LPR + q];
#pragma unroll
    for (int k = 0; k < H; ++k) al[k] = row_sum16(dot4(xo, vl[k]));
  }
  NGPDE_GST(3);
  __syncthreads();
  NGPDE_GST(4);

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
  // ---- per-head aggregates of the staged rows (one LDS row read serves all heads)
  {
    const int wmax = wave_max_deg(deg);
    float4 acc[H];
#pragma unroll
    for (int k = 0; k < H; ++k) acc[k] = f4_zero();
#pragma unroll
    for (int jw = 0; jw < 8; ++jw) {
      if (jw * 4 < wmax) {   // wave-uniform
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          const float4 xv = Xh4[slot_byte(m.w, jw, jb) * GG::LPR + q];
          const float4 cf = reinterpret_cast<const float4 *>(ldsS)[grp * kSlotWidth + jw * 4 + jb];
          const float cfa[4] = {cf.x, cf.y, cf.z, cf.w};
#pragma unroll
          for (int k = 0; k < H; ++k) acc[k] = f4_fma(cfa[k], xv, acc[k]);
        }
      }
    }
#pragma unroll
    for (int k = 0; k < H; ++k) *reinterpret_cast<float4 *>(&ldsA[grp * kATS + k * GD + 4 * q]) = acc[k];
  }
  NGPDE_GST(6);
  __syncthreads();   // aggregates complete; the coefficients are dead: their region takes the output tile
  NGPDE_GST(7);
  float *ldsZ = ldsS;
  {   // out[:, ct*16..] = A_head(ct) x W[:, ct*16..]: 2 row tiles x 4 column tiles = one tile per wave
    const int rt = wave_u & 1, ct = wave_u >> 1;
    const int i = lane & 15, kq = lane >> 4;
    const int head = (ct * 16) / C;
    const float *pa = ldsA + (rt * 16 + i) * kATS + head * GD + 4 * kq;
    f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const float4 av4 = *reinterpret_cast<const float4 *>(pa + kb * 16);
      acc = mfma16(av4.x, breg[kb][0], acc);
      acc = mfma16(av4.y, breg[kb][1], acc);
      acc = mfma16(av4.z, breg[kb][2], acc);
      acc = mfma16(av4.w, breg[kb][3], acc);
    }
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) ldsZ[(rt * 16 + 4 * kq + reg) * GG::TS + ct * 16 + i] = acc[reg];
  }
  NGPDE_GST(8);
  __syncthreads();
  NGPDE_GST(9);
  if (m.sc.x >= 0) {
    const size_t idx4 = (size_t)m.sc.x * GG::LPR + q;
    const float4 z = f4_add(*reinterpret_cast<const float4 *>(&ldsZ[grp * GG::TS + 4 * q]), b4);
    if (p.save_z) reinterpret_cast<float4 *>(p.save_z)[idx4] = z;
    reinterpret_cast<float4 *>(p.y)[idx4] = f4_act(p.act, z);
  }
  NGPDE_GST(10);
}

// ---------------------------------------------------------------------------------------------------
// pullback, by target:  dz, d alpha, softmax / leakyrelu pullback -> dscore, dal, db slabs
// ---------------------------------------------------------------------------------------------------
struct GatBwdTK {
  const float *x, *wt, *dy, *yz, *alpha;
  const int4 *sched;
  const int2 *halo;
  const uint8_t *slots;
  int n_tiles, act;
  float slope;
  float *dz, *dscore, *dal, *slab_db;
};

template <int H>
__global__ __launch_bounds__(kThreads, 4) void gat_layer_bwd_target_kernel(const GatBwdTK p) {
  constexpr int C = GD / H;
  __shared__ __attribute__((aligned(16))) float ldsXh[(kHaloCap + 1) * GD];
  // the dz tile feeds the product (its B operand, blocks of W, comes straight from memory); the result is the per-head
  // [32][H*64] tile
  __shared__ __attribute__((aligned(16))) float ldsDZ[kTM * GG::TS];
  __shared__ __attribute__((aligned(16))) float ldsDA[kTM * kATS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = tid / GG::LPR, q = tid % GG::LPR;
  const int tile = xcd_tile(blockIdx.x, p.n_tiles);
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);

  TileMeta m;
  tile_meta(p.halo, p.slots, p.sched, p.x, tile, grp, q, ldsXh, m);
  // B operand of this wave's H output tiles: lane (i, kq) of tile (head, jt) needs wt[jt*16 + i][head*C + 16 kb + 4 kq + r],
  // r = 0..3 contiguous: C / 16 float4 per tile, in flight from the start (W is 16 KB, cache resident)
  float4 breg[H][C / 16];
  {
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int t = 0; t < H; ++t) {
      const int ctile = (wave_u + GG::WAVES * t) >> 1, head = ctile >> 2, jt = ctile & 3;
#pragma unroll
      for (int kb = 0; kb < C / 16; ++kb)
        breg[t][kb] = *reinterpret_cast<const float4 *>(p.wt + (size_t)(jt * 16 + i) * GD + head * C + 16 * kb + 4 * kq);
    }
  }
  const bool ok = m.sc.x >= 0;
  const int deg = ok ? m.sc.z : 0;
  const size_t idx4 = (size_t)max(m.sc.x, 0) * GG::LPR + q;
  float4 dz = reinterpret_cast<const float4 *>(p.dy)[idx4];
  if (p.yz) dz = f4_mul(dz, f4_dact(p.act, reinterpret_cast<const float4 *>(p.yz)[idx4]));
  if (!ok) dz = f4_zero();
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
  if (ok && p.dz) reinterpret_cast<float4 *>(p.dz)[idx4] = dz;
  *reinterpret_cast<float4 *>(&ldsDZ[grp * GG::TS + 4 * q]) = dz;
  __syncthreads();
  {   // db partial: column sums of the dz tile (8 adjacent lanes hold row-partials of one column)
    const int dbc = tid / GG::DBP, dbpart = tid % GG::DBP;
    float s = 0.f;
#pragma unroll
    for (int n = dbpart; n < kTM; n += GG::DBP) s += ldsDZ[n * GG::TS + dbc];
#pragma unroll
    for (int o = 1; o < GG::DBP; o <<= 1) s += __shfl_xor(s, o);
    if (dbpart == 0) p.slab_db[(size_t)tile * GD + dbc] = s;
  }
  // dA[i][k*64 + j] = sum_c dz[i][k*C + c] W[j][k*C + c]: 2 row tiles x (4 H) column tiles, H tiles per wave
  {
    const int i = lane & 15, kq = lane >> 4;
#pragma unroll
    for (int t = 0; t < H; ++t) {
      const int id = wave_u + GG::WAVES * t, rt = id & 1, ctile = id >> 1, head = ctile >> 2;
      const float *pa = ldsDZ + (rt * 16 + i) * GG::TS + head * C + 4 * kq;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < C / 16; ++kb) {
        const float4 a4 = *reinterpret_cast<const float4 *>(pa + kb * 16);
        acc = mfma16(a4.x, breg[t][kb].x, acc);
        acc = mfma16(a4.y, breg[t][kb].y, acc);
        acc = mfma16(a4.z, breg[t][kb].z, acc);
        acc = mfma16(a4.w, breg[t][kb].w, acc);
      }
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) ldsDA[(rt * 16 + 4 * kq + reg) * kATS + ctile * 16 + i] = acc[reg];
    }
  }
  __syncthreads();
  // d alpha of every entry of the row: <dA_k[i], x_s> per head, reduced over the group's 16 lanes; the lane that owns the
  // entry keeps it
  float4 dar[H];
#pragma unroll
  for (int k = 0; k < H; ++k) dar[k] = *reinterpret_cast<const float4 *>(&ldsDA[grp * kATS + k * GD + 4 * q]);
  float da[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  {
    const int wmax = wave_max_deg(deg);
#pragma unroll
    for (int jw = 0; jw < 8; ++jw) {
      if (jw * 4 < wmax) {   // wave-uniform
#pragma unroll
        for (int jb = 0; jb < 4; ++jb) {
          const int j = jw * 4 + jb;
          const float4 xv = Xh4[slot_byte(m.w, jw, jb) * GG::LPR + q];
#pragma unroll
          for (int k = 0; k < H; ++k) {
            const float s = row_sum16(dot4(dar[k], xv));
            da[j >> 4][k] = (q == (j & 15)) ? s : da[j >> 4][k];
          }
        }
      }
    }
  }
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
      float *dst = p.dscore + (size_t)(m.sc.y + q + 16 * s) * H;
      if (H == 4) *reinterpret_cast<float4 *>(dst) = make_float4(dsc[s][0], dsc[s][1], dsc[s][2], dsc[s][3]);
      else if (H == 2) *reinterpret_cast<float2 *>(dst) = make_float2(dsc[s][0], dsc[s][1]);
      else dst[0] = dsc[s][0];
    }
  }
  if (ok && q < H) p.dal[(size_t)m.sc.x * H + q] = sel4(dalv, q);
}

// ---------------------------------------------------------------------------------------------------
// pullback, by source:  dWx = sum alpha dz[t] + dal a_l + dar a_r;  dx = dWx W^T;  dW, u_l, u_r slabs
// ---------------------------------------------------------------------------------------------------
struct GatBwdSK {
  const float *gz, *x, *wt, *a, *alpha, *dscore, *dal;
  const int4 *sched;
  const int2 *halo;
  const uint8_t *slots;
  const int *xpos;
  int n_tiles;
  float *dx, *slab_dw, *slab_u;
};

template <int H>
__global__ __launch_bounds__(kThreads, 4) void gat_layer_bwd_source_kernel(const GatBwdSK p) {
  constexpr int C = GD / H;
  __shared__ __attribute__((aligned(16))) float ldsXh[(kHaloCap + 1) * GD];     // staged dz rows; later the dx tile
  __shared__ __attribute__((aligned(16))) float ldsS[kTM * kSlotWidth * 4];
  __shared__ __attribute__((aligned(16))) float ldsDWX[kTM * GG::TS];
  __shared__ __attribute__((aligned(16))) float ldsXT[kTM * GG::TS];
  __shared__ __attribute__((aligned(16))) float ldsBt[GD * GG::TS];
  __shared__ __attribute__((aligned(16))) float ldsDD[kTM * 8];                  // dal | dar of the tile's rows
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave_u = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = tid / GG::LPR, q = tid % GG::LPR;
  float4 *Xh4 = reinterpret_cast<float4 *>(ldsXh);
  const int n_wg = (p.n_tiles + kSrcTiles - 1) / kSrcTiles;
  const int wg = xcd_tile(blockIdx.x, n_wg);
  const int hq = (4 * q) / C;                                                    // this lane's head
  // dx = dWx x B with B[k = out feature][j = in feature] = wt[j][k]: Bt[j][k] = wt[j][k], a straight copy, resident for all tiles
#pragma unroll
  for (int k = 0; k < GG::W4; ++k) {
    const int idx = tid + k * kThreads;
    *reinterpret_cast<float4 *>(&ldsBt[((idx * 4) / GD) * GG::TS + (idx * 4) % GD]) = reinterpret_cast<const float4 *>(p.wt)[idx];
  }
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
    tile_meta(p.halo, p.slots, p.sched, p.gz, tile, grp, q, ldsXh, m);
    const bool ok = m.sc.x >= 0;
    const int deg = ok ? m.sc.z : 0;
    const size_t idx4 = (size_t)max(m.sc.x, 0) * GG::LPR + q;
    float4 xo = reinterpret_cast<const float4 *>(p.x)[idx4];
    if (!ok) xo = f4_zero();
    float dalq = (ok && q < H) ? p.dal[(size_t)m.sc.x * H + q] : 0.f;
    // this lane's two outgoing entries: coefficient and dscore of the same edge in the by-target list
    float av[2][4], ds[2][4];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const bool valid = q + 16 * s < deg;
      const int pp = valid ? p.xpos[m.sc.y + q + 16 * s] : 0;
      float va[4] = {0.f, 0.f, 0.f, 0.f}, vd[4] = {0.f, 0.f, 0.f, 0.f};
      if (valid) {
        if (H == 4) {
          const float4 t = *reinterpret_cast<const float4 *>(p.alpha + (size_t)pp * 4), u = *reinterpret_cast<const float4 *>(p.dscore + (size_t)pp * 4);
          va[0] = t.x; va[1] = t.y; va[2] = t.z; va[3] = t.w; vd[0] = u.x; vd[1] = u.y; vd[2] = u.z; vd[3] = u.w;
        } else if (H == 2) {
          const float2 t = *reinterpret_cast<const float2 *>(p.alpha + (size_t)pp * 2), u = *reinterpret_cast<const float2 *>(p.dscore + (size_t)pp * 2);
          va[0] = t.x; va[1] = t.y; vd[0] = u.x; vd[1] = u.y;
        } else {
          va[0] = p.alpha[pp]; vd[0] = p.dscore[pp];
        }
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) { av[s][k] = fabsf(va[k]); ds[s][k] = vd[k]; }
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
    __syncthreads();   // staged dz rows visible (coefficients are group-private, same wave)
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
    __syncthreads();   // tiles complete, staged rows dead
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
    {   // u_which,k[i] += sum_n (dal | dar)[n][k] x[n][i]
      const int which = tid >> 8, hk = (tid >> 6) & 3, i = tid & 63;
#pragma unroll 8
      for (int n = 0; n < kTM; ++n) uacc = fmaf(ldsDD[n * 8 + which * 4 + hk], ldsXT[n * GG::TS + i], uacc);
    }
    __syncthreads();
    if (ok && p.dx) reinterpret_cast<float4 *>(p.dx)[idx4] = *reinterpret_cast<const float4 *>(&ldsXh[grp * GG::TS + 4 * q]);
    __syncthreads();   // the next tile's rows land where the dx tile is
  }
  float4 *slab4 = reinterpret_cast<float4 *>(p.slab_dw + (size_t)wg * GD * GD);
#pragma unroll
  for (int mm = 0; mm < GG::DWT; ++mm) {
    const int t2 = wave_u + GG::WAVES * mm;
    if (t2 < NT) slab4[t2 * 64 + lane] = make_float4(dw[mm][0], dw[mm][1], dw[mm][2], dw[mm][3]);
  }
  // this workgroup's share of da:  da[(which*C + c) + 2C k] = sum_i W[k*C + c][i] u_which,k[i]   (linear in u: summed over the
  // workgroups by the reduction kernel)
  ldsS[tid] = uacc;
  __syncthreads();
  if (tid < 2 * GD) {
    const int k = tid / (2 * C), r = tid % (2 * C), which = r / C, cc = r % C;
    float sacc = 0.f;
#pragma unroll 8
    for (int i = 0; i < GD; ++i) sacc = fmaf(ldsBt[i * GG::TS + k * C + cc], ldsS[which * 256 + k * 64 + i], sacc);
    p.slab_u[(size_t)wg * 2 * GD + tid] = sacc;
  }
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
  return g && g->has_norm && g->by_t.halo_ok && g->by_s.halo_ok && din == GD && heads * c == GD &&
         (heads == 1 || heads == 2 || heads == 4) && (uint64_t)g->n_nodes * GD * 4 < (1ull << 32) && !no_fused_gat_layer_env();
}

size_t gat_layer_workspace_bytes(const ngpde_graph *g, int heads) { return g ? gat_ws(g, heads).total : 0; }

int32_t launch_gat_layer_fwd(const ngpde_graph *g, int heads, float slope, int act, const float *x, const float *wt, const float *a,
                             const float *bias, float *y, float *alpha, float *save_z, hipStream_t stream) {
  if (g->n_nodes == 0) return NGPDE_OK;
  GatFwdK k;
  k.x = x; k.wt = wt; k.a = a; k.bias = bias; k.sched = g->by_t.sched; k.halo = g->by_t.halo; k.slots = g->by_t.slots;
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
    t.slots = g->by_t.slots; t.n_tiles = n_tiles; t.act = act; t.slope = slope; t.dz = ident ? nullptr : dz; t.dscore = dscore;
    t.dal = dal; t.slab_db = slab_db;
    GatBwdSK s;
    s.gz = ident ? dy : dz; s.x = x; s.wt = wt; s.a = a; s.alpha = alpha; s.dscore = dscore; s.dal = dal; s.sched = g->by_s.sched;
    s.halo = g->by_s.halo; s.slots = g->by_s.slots; s.xpos = g->by_s.xpos; s.n_tiles = n_tiles; s.dx = dx; s.slab_dw = slab_dw;
    s.slab_u = slab_u;
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

}  // namespace ngpde
