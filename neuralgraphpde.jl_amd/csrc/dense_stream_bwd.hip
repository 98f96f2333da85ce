// dense_stream_bwd.hip -- the whole pullback of a node-level Dense with 64 outputs in ONE streaming launch
//   dz = dy . act'(z),   dX_b = dz W_b^T  (per 64-wide input block b),   dW = [X1 | X2 | narrow]^T dz,   db = colsum(dz)
// for the shapes the edge-function layers put at node level (/root/reference/src/layers.jl:409-418 after the first-layer
// split: [h | d | theta] => 64, [h | d] => 64, [h | m | theta] => 64, 64 => 64): one or two 64-wide blocks plus up to four
// narrow features (coordinates, per-graph theta) that carry no gradient.  The composed path (dense_dz + weight pullback +
// input pullback, dense_mfma.hip) reads dz three times and X once in three launches of 64-row K steps: 0.30 ms per layer at
// BASELINE config 4's shard (524 288 rows) against 0.10 ms for the 402 MB this launch moves.
//
// Persistent workgroups of 4 waves walk 64-row tiles.  Per tile: dy (and z) come through registers -- dz is formed there, feeds
// the rank-1 accumulations of the narrow features and the bias, and is written to LDS --, the 64 x 64 tiles of X_b go memory ->
// LDS by LDS-DMA a tile ahead (NMAIN + 1 rotating buffers); then every wave runs 64 MFMAs per block of X_b^T dz (its 16 input
// features x all 64 outputs, contraction over the tile's rows, accumulators live in registers for the whole launch, all blocks from
// one read of dz) and 64 MFMAs per block of dz W_b^T (all 64 rows x ITS 16 input features, whose rows of W_b it keeps in registers for
// the launch: no weight copy in LDS), whose result leaves through an image buffer as full 256-byte rows.  All tiles use one XOR
// swizzle of the 16-byte chunk index, slot = chunk ^ R(row), R = the row's low two bit pairs swapped, which makes BOTH access
// patterns bank-conflict free: 16-byte row reads (16 rows x one chunk) and the 4-byte transposed reads of the weight product
// (rows 4s .. 4s + 3 x 16 consecutive columns).  These launches are bound by instruction ISSUE on the SIMD (a wave's MFMAs, VALU
// and waits are one in-order stream and two waves share a SIMD), not by memory: every LDS address of the product loops is a
// register set up once per launch plus an immediate, and the reads of a product group are issued in the middle of the group before
// (PairBwdAddr, stream_bwd_dw / stream_bwd_dx / pair_bwd_products; DESIGN 5.4, tools/stamps_pair_bwd.py).
// Per-workgroup dW / db slabs are summed by dense_weight_reduce_kernel in a fixed order: no atomics, reproducible.
#include <algorithm>
#include <cstdlib>

#include <type_traits>

#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

constexpr int kBT = 256, kTR = 64, kD = 64, kPS = kD + 4, kMaxNarrow = 4;

struct DenseBwdK {
  int64_t n;
  int n_tiles, n_narrow, din, act;
  const float *x[2];       // the 64-wide blocks [n][64]
  float *dx[2];            // their gradients, or NULL
  int main_off[2];         // first feature of each block in the virtual vcat (= row of W / dW)
  const float *nx[kMaxNarrow];   // narrow feature f lives at nx[f][(row / ndiv[f]) * nwidth[f]]
  int nwidth[kMaxNarrow], ndiv[kMaxNarrow], nfeat[kMaxNarrow];
  const float *wt, *z, *dy;
  float *partial;          // [gridDim.x][din + 1][64]
  int dephase;
};

template <int S, int N, class F>
__device__ __forceinline__ void static_for(F &&f) {
  if constexpr (S < N) {
    f(std::integral_constant<int, S>{});
    static_for<S + 1, N>(f);
  }
}

__device__ __forceinline__ int swz(int row) { return ((row & 3) << 2) | ((row >> 2) & 3); }   // R(row & 15)
__device__ __forceinline__ int sw_addr(int row, int col) { return row * kD + 4 * ((col >> 2) ^ swz(row)) + (col & 3); }

// X tile of the rows row0 .. row0 + 63 -> swizzled LDS image: four DMA instructions per wave, four rows each
__device__ __forceinline__ void dma_x_tile(const float *x, uint32_t row0, uint32_t n, float *img, int wave, int lane) {
  const int rl = lane >> 4, s16 = lane & 15;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (wave * 4 + j) * 4 + rl;
    const uint32_t gr = min(row0 + r, n - 1);
    const float *g = x + (gr * kD + 4 * (s16 ^ swz(r)));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)(reinterpret_cast<float4 *>(img) + ((wave * 4 + j) * 4) * 16 + lane),
                                     16, 0, 0);
  }
}

// Word offsets of a lane's LDS reads in the products of the two kernels below, per lane (i = lane % 16, kq = lane / 16), computed ONCE
// per launch.  With the swizzle of sw_addr the address of (row 4 s + kq, column 16 c + i), s = 4 m + j, is
//     1024 m + [256 j + 64 kq + 4 ((i / 4) ^ j) + i % 4] + 16 (c ^ kq)
// and that of the 16-byte chunk 4 kh + kq of row 16 rt + i is  1024 rt + [64 i + 16 (kh ^ (i % 4)) + 4 (kq ^ (i / 4))]:  a register per
// (j, c) resp. kh and an immediate per m resp. rt.  Written as sw_addr(4 s + kq, ..) in the loop the compiler either keeps ~ 100 addresses
// in registers across the tile loop or (behind an opaque zero, as both kernels did until round 5) recomputes them per slice: ~ 80 VALU
// instructions per 10 MFMAs, and VALU issue does not overlap a wave's own MFMAs -- the products phase of a tile took 20 k cycles for
// 9.2 k cycles of MFMA (tools/stamps_pair_bwd.py).
struct PairBwdAddr {
  int a[4][4];   // [j][ob]: dy columns 16 ob + i of row slice j (mod 4)
  int aw[4];     // [j]: column 16 wave + i (the X image's and the narrow products' column block)
  int xk[4];     // [kh]: chunk 4 kh + kq of row i
  int nb;        // narrow block: row kq, column i
};
__device__ __forceinline__ PairBwdAddr pair_bwd_addr(int wave, int lane) {
  const int i = lane & 15, kq = lane >> 4;
  PairBwdAddr q;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int b = 256 * j + 64 * kq + 4 * ((i >> 2) ^ j) + (i & 3);
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) q.a[j][ob] = b + 16 * (ob ^ kq);
    q.aw[j] = b + 16 * (wave ^ kq);
  }
#pragma unroll
  for (int kh = 0; kh < 4; ++kh) q.xk[kh] = 64 * i + 16 * (kh ^ (i & 3)) + 4 * (kq ^ (i >> 2));
  q.nb = 16 * kq + i;
  return q;
}

// dW pass of one tile of dense_stream64_bwd_kernel: accW[b] += X_b^T dz for this wave's 16 input features of every block, with the NEXT
// tile's first image on its way by LDS-DMA (issued here: __restrict__ pointers, so that the reads carry no-alias information against the
// DMA's destination -- see pair_bwd_products).  Per row slice NMAIN + 4 LDS words and 4 NMAIN products; the words of slice s + 1 are asked
// for in the middle of slice s's products.
template <int NMAIN>
__device__ __forceinline__ void stream_bwd_dw(const float *__restrict__ x0img, const float *__restrict__ x1img, float *__restrict__ dma_dst,
                                              const float *__restrict__ dz, bool has_next, const float *xnext, uint32_t next_row0, uint32_t n,
                                              int wave, int lane, const PairBwdAddr &q, f32x4 (&accW)[NMAIN][4]) {
  if (has_next) dma_x_tile(xnext, next_row0, n, dma_dst, wave, lane);
  struct WSlice { float av[NMAIN], d[4]; };
  const int kq = lane >> 4;
  int oo[4];   // 16 ((ob ^ kq) - (0 ^ kq)): column block ob relative to block 0
#pragma unroll
  for (int ob = 0; ob < 4; ++ob) oo[ob] = 16 * ((ob ^ kq) - kq);
  asm volatile("" : "+v"(oo[0]), "+v"(oo[1]), "+v"(oo[2]), "+v"(oo[3]));
  auto wload = [&](auto sc, WSlice &f) {
    constexpr int s = decltype(sc)::value, m = s >> 2, j = s & 3;
    f.av[0] = x0img[q.aw[j] + 1024 * m];
    if constexpr (NMAIN == 2) f.av[1] = x1img[q.aw[j] + 1024 * m];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) f.d[ob] = dz[q.a[j][0] + oo[ob] + 1024 * m];   // (one add per read instead of 16 address registers)
  };
  auto wmul = [&](const WSlice &f, int half) {
#pragma unroll
    for (int b = 0; b < NMAIN; ++b) {
      accW[b][2 * half] = mfma16(f.av[b], f.d[2 * half], accW[b][2 * half]);
      accW[b][2 * half + 1] = mfma16(f.av[b], f.d[2 * half + 1], accW[b][2 * half + 1]);
    }
  };
  WSlice fs[2];
  wload(std::integral_constant<int, 0>{}, fs[0]);
  static_for<0, 16>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    __builtin_amdgcn_sched_barrier(0);
    wmul(fs[s & 1], 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (s + 1 < 16) wload(std::integral_constant<int, s + 1>{}, fs[(s + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    wmul(fs[s & 1], 1);
  });
}

// dX_b[:, 16 w .. + 15] = dz W_b[16 w .. + 15, :]^T for all 64 rows of the tile: the wave's 16 rows of W_b in registers (wf), dz rows from
// LDS, 8 product groups of (one row tile, half the contraction) with the next group's two 16-byte reads asked for in the middle of a
// group.  The next tile's SECOND image (NMAIN = 2) leaves for its buffer from here.
__device__ __forceinline__ void stream_bwd_dx(const float *__restrict__ dz, float *__restrict__ dma_dst, bool dma, const float *xnext,
                                              uint32_t next_row0, uint32_t n, int wave, int lane, const PairBwdAddr &q, const f32x4 (&wf)[4],
                                              f32x4 (&accX)[4]) {
  if (dma) dma_x_tile(xnext, next_row0, n, dma_dst, wave, lane);
  auto xload = [&](auto gc, float4 (&a)[2]) {   // group g: row tile g / 2, chunks 4 kh + kq of kh = 2 (g % 2), 2 (g % 2) + 1
    constexpr int g = decltype(gc)::value, rt = g >> 1, k0 = 2 * (g & 1);
    a[0] = *reinterpret_cast<const float4 *>(&dz[q.xk[k0] + 1024 * rt]);
    a[1] = *reinterpret_cast<const float4 *>(&dz[q.xk[k0 + 1] + 1024 * rt]);
  };
  auto xmul = [&](const float4 &a, int rt, int kh) {
    accX[rt] = mfma16(a.x, wf[kh][0], accX[rt]);
    accX[rt] = mfma16(a.y, wf[kh][1], accX[rt]);
    accX[rt] = mfma16(a.z, wf[kh][2], accX[rt]);
    accX[rt] = mfma16(a.w, wf[kh][3], accX[rt]);
  };
  float4 xa[2][2];
  xload(std::integral_constant<int, 0>{}, xa[0]);
  static_for<0, 8>([&](auto gc) {
    constexpr int g = decltype(gc)::value, rt = g >> 1, k0 = 2 * (g & 1);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (k0 == 0) accX[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    xmul(xa[g & 1][0], rt, k0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (g + 1 < 8) xload(std::integral_constant<int, g + 1>{}, xa[(g + 1) & 1]);
    __builtin_amdgcn_sched_barrier(0);
    xmul(xa[g & 1][1], rt, k0 + 1);
  });
}

// LDS: NMAIN + 1 buffers [64][68] -- the swizzled [64][64] DMA images of this tile's X_b and of one image of the next tile, later dX_b on
// its way out -- then dz [64][64] swizzled.  (The images first: the destination of an LDS-DMA has to lie in the first 64 KB.)  50 / 67 KB:
// two workgroups per CU.  A tile:
//   dy . act'(z) of the thread's rows (registers, loaded a tile ago) -> dz in LDS, bias / narrow-feature sums on the VALU;
//   the next tile's dy / z / narrow values and X_0 image leave; the dW pass over this tile's images (all blocks from one read of dz);
//   NMAIN = 2: the next tile's X_1 image leaves for the buffer X_0 just vacated;
//   per block: dX_b products (weights in registers), staged through LDS, stored as whole rows -- the loads of the next tile are collected
//   before the LAST block's stores are issued (vmcnt counts stores: collected at the top of the next tile they would cost a store latency).
// The buffers rotate: (X_0, X_1, free) -> (free, X_0, X_1).
template <int NMAIN>
__global__ __launch_bounds__(kBT, 2) void dense_stream64_bwd_kernel(const DenseBwdK p) {
  constexpr int NB = NMAIN + 1;
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *ldsDz = lds + NB * kTR * kPS;   // [64][64], swizzled
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int rg = tid >> 4, qc = tid & 15;   // staging role: rows rg + 16 p, columns 4 qc .. 4 qc + 3

  f32x4 wf[NMAIN][4];   // W_b[main_off_b + 16 wave + i][16 kh + 4 kq .. + 3]
#pragma unroll
  for (int b = 0; b < NMAIN; ++b)
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      const float4 w4 = *reinterpret_cast<const float4 *>(p.wt + (size_t)(p.main_off[b] + 16 * wave + i) * kD + 16 * kh + 4 * kq);
      wf[b][kh] = (f32x4){w4.x, w4.y, w4.z, w4.w};
    }
  const PairBwdAddr addr = pair_bwd_addr(wave, lane);

  f32x4 accW[NMAIN][4];
#pragma unroll
  for (int b = 0; b < NMAIN; ++b)
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) accW[b][ob] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float4 nacc[kMaxNarrow], bacc = f4_zero();
#pragma unroll
  for (int f = 0; f < kMaxNarrow; ++f) nacc[f] = f4_zero();

  float4 dyr[4], zr[4] = {f4_zero(), f4_zero(), f4_zero(), f4_zero()};
  float nxr[4][kMaxNarrow];
  uint32_t nmagic[kMaxNarrow];   // (uniform: one division per feature and launch)
#pragma unroll
  for (int f = 0; f < kMaxNarrow; ++f) nmagic[f] = div_magic((uint32_t)p.ndiv[f]);
  auto fetch = [&](int tile) {   // dy, z and the narrow features of the thread's four rows (32-bit offsets from uniform bases)
    const uint32_t row0 = (uint32_t)tile * kTR;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const uint32_t r = row0 + rg + 16 * pp;
      const uint32_t rc = min(r, (uint32_t)(p.n - 1));
      const uint32_t e = rc * kD + 4 * qc;
      dyr[pp] = *reinterpret_cast<const float4 *>(p.dy + e);
      if (r >= (uint32_t)p.n) dyr[pp] = f4_zero();   // rows past the end contribute nothing
      if (p.z) zr[pp] = *reinterpret_cast<const float4 *>(p.z + e);
#pragma unroll
      for (int f = 0; f < kMaxNarrow; ++f) nxr[pp][f] = p.nx[f][fast_div(rc, (uint32_t)p.ndiv[f], nmagic[f]) * (uint32_t)p.nwidth[f]];   // (unused slots alias X)
    }
  };

  dephase_second_half(p.dephase);
  int tile = blockIdx.x;
  int b0 = 0, b1 = 1, bf = NMAIN;   // buffers of X_0, X_1 (NMAIN = 2) and the free one
  if (tile < p.n_tiles) {
    fetch(tile);
    dma_x_tile(p.x[0], (uint32_t)tile * kTR, (uint32_t)p.n, lds + b0 * kTR * kPS, wave, lane);
    if (NMAIN == 2) dma_x_tile(p.x[1], (uint32_t)tile * kTR, (uint32_t)p.n, lds + b1 * kTR * kPS, wave, lane);
  }
  wait_vmcnt0();
  for (; tile < p.n_tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * kTR;
    float *img0 = lds + b0 * kTR * kPS, *img1 = lds + (NMAIN == 2 ? b1 : b0) * kTR * kPS, *imgf = lds + bf * kTR * kPS;
    __syncthreads();   // the previous tile's dz and outgoing dX are consumed
    // dz of the thread's rows -> LDS; bias and narrow-feature accumulations (rank-1 updates on the VALU)
    if (p.z) {   // dz = dy . act'(z): one uniform activation switch for the 16 values
      f4n_dact<4>(p.act, zr);
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) dyr[pp] = f4_mul(dyr[pp], zr[pp]);
    }
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const float4 dz = dyr[pp];
      const int r = rg + 16 * pp;
      *reinterpret_cast<float4 *>(&ldsDz[r * kD + 4 * (qc ^ swz(r))]) = dz;
      bacc = f4_add(bacc, dz);
#pragma unroll
      for (int f = 0; f < kMaxNarrow; ++f) nacc[f] = f4_fma(nxr[pp][f], dz, nacc[f]);
    }
    __syncthreads();
    const int tnext = tile + gridDim.x;
    const bool has_next = tnext < p.n_tiles;
    if (has_next) fetch(tnext);   // in flight during the products
    stream_bwd_dw<NMAIN>(img0, img1, imgf, ldsDz, has_next, p.x[0], (uint32_t)tnext * kTR, (uint32_t)p.n, wave, lane, addr, accW);
    __syncthreads();   // every wave is done with the X images
    // the staging buffer of the outgoing dX: X_0's image (NMAIN = 1), X_1's (NMAIN = 2: X_0's takes the next tile's X_1)
    float *out = NMAIN == 2 ? img1 : img0;
#pragma unroll
    for (int b = 0; b < NMAIN; ++b) {
      const bool last = b == NMAIN - 1;
      if (p.dx[b] == nullptr) {   // (workgroup-uniform)
        if (NMAIN == 2 && b == 0 && has_next) dma_x_tile(p.x[1], (uint32_t)tnext * kTR, (uint32_t)p.n, img0, wave, lane);
        if (last) wait_vmcnt0();
        continue;
      }
      f32x4 accX[4];
      stream_bwd_dx(ldsDz, img0, NMAIN == 2 && b == 0 && has_next, p.x[NMAIN - 1], (uint32_t)tnext * kTR, (uint32_t)p.n, wave, lane, addr, wf[b], accX);
      if (last) wait_vmcnt0();   // the next tile's loads and images, before this block's stores are issued
      if (b > 0) __syncthreads();   // the previous block's rows have left the staging buffer
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) out[(16 * rt + 4 * kq + reg) * kPS + 16 * wave + i] = accX[rt][reg];
      __syncthreads();
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const int r = rg + 16 * pp;
        if (row0 + r < p.n)
          *reinterpret_cast<float4 *>(p.dx[b] + (row0 + r) * kD + 4 * qc) = *reinterpret_cast<const float4 *>(&out[r * kPS + 4 * qc]);
      }
    }
    if (NMAIN == 2) { const int t0 = b0; b0 = bf; bf = b1; b1 = t0; }   // (X_0, X_1, free) -> (free, X_0, X_1)
    else { const int t0 = b0; b0 = bf; bf = t0; }
  }

  // ---- this workgroup's slab: the MFMA accumulators directly, narrow features and bias after a fixed-order sum over the
  // 16 row groups
  float *slab = p.partial + (size_t)blockIdx.x * (p.din + 1) * kD;
#pragma unroll
  for (int b = 0; b < NMAIN; ++b)
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) slab[(size_t)(p.main_off[b] + 16 * wave + 4 * kq + reg) * kD + 16 * ob + i] = accW[b][ob][reg];
  __syncthreads();
  float *red = lds;   // [16 row groups][5][64]
#pragma unroll
  for (int f = 0; f < kMaxNarrow; ++f) *reinterpret_cast<float4 *>(&red[(rg * 5 + f) * kD + 4 * qc]) = nacc[f];
  *reinterpret_cast<float4 *>(&red[(rg * 5 + 4) * kD + 4 * qc]) = bacc;
  __syncthreads();
  for (int idx = tid; idx < 5 * kD; idx += kBT) {
    const int f = idx / kD, o = idx % kD;
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) s += red[(g * 5 + f) * kD + o];
    if (f == 4) slab[(size_t)p.din * kD + o] = s;
    else if (f < p.n_narrow) slab[(size_t)p.nfeat[f] * kD + o] = s;
  }
}

// ---- the pullbacks of TWO Dense layers that share their 64-wide leading block (ngpde_dense_pair_forward's pair: the target and
// source halves P, Q of a message MLP's first layer; no activation of their own -- it is applied per edge) in one launch:
//   dX = dy_a Wa^T + dy_b Wb^T [+ addend],   dWa = [X | narrow_a]^T dy_a,   dWb = [X | narrow_b]^T dy_b,   db_a, db_b
// X is read once and its gradient written once, already summed (the composed path: two launches, two gradient arrays, an add).
// Same tiles, swizzle and slabs as dense_stream64_bwd_kernel; the wave that owns input features 16 w .. + 15 for the weight
// products also owns those COLUMNS of dX, with its 16 rows of Wa and Wb in registers (no weight copy in LDS).
struct DensePairBwdK {
  int64_t n;
  int n_tiles;
  const float *x, *dx_add;
  float *dx;
  int din[2], main_off[2], n_narrow[2];
  const float *wt[2], *dy[2];
  const float *nx[2][kMaxNarrow];
  int nwidth[2][kMaxNarrow], ndiv[2][kMaxNarrow], nfeat[2][kMaxNarrow];
  float *partial[2];   // [gridDim.x][din_s + 1][64]
  int dephase;
#ifdef NGPDE_STAMPS
  unsigned long long *stamps;   // diagnostic build only (tools/stamps_pair_bwd.py): [gridDim.x][16], the workgroup's 4th tile
#endif
};
#ifdef NGPDE_STAMPS
unsigned long long *g_pair_bwd_stamps = nullptr;
#define PBWD_STAMP(k) do { if (threadIdx.x == 0 && p.stamps && it == 3) p.stamps[(size_t)blockIdx.x * 16 + (k)] = clock64(); } while (0)
#else
#define PBWD_STAMP(k)
#endif

// The products of one tile of dense_pair64_bwd_kernel with the NEXT tile's X image on its way by LDS-DMA.  The pointers are
// __restrict__ so that the DMA and the LDS reads carry no-alias information: without it the compiler puts s_waitcnt vmcnt(0) in
// front of the first LDS read behind a global_load_lds (the read might alias the DMA's destination), i.e. waits for the
// prefetch at once (dense_mfma.hip, products_beside_dma).
__device__ __forceinline__ void pair_bwd_products(const float *__restrict__ ximg, float *__restrict__ xnext, const float *__restrict__ dz0,
                                                  const float *__restrict__ dz1, const float *__restrict__ nblk, bool has_next,
                                                  const float *x, uint32_t next_row0, uint32_t n, int wave, int lane, const PairBwdAddr &q,
                                                  const f32x4 (&wf)[2][4], f32x4 (&accW)[2][4], f32x4 (&accN)[2], f32x4 (&accX)[4],
                                                  unsigned long long *st = nullptr) {   // (st: diagnostic build)
  if (has_next) dma_x_tile(x, next_row0, n, xnext, wave, lane);
  if (st) st[12] = clock64();
  // ---- dW_s += X^T dy_s: this wave's 16 input features x 64 outputs, both sides from one read of X; the narrow block
  // (features + ones) x this wave's 16 outputs.  The 13 LDS words of row slice s + 1 are asked for in the MIDDLE of the 10 products
  // of slice s (two register sets, scheduling fences): written as "read, then multiply" the compiler emits read -> s_waitcnt
  // lgkmcnt(0) -> two MFMAs; it never waits with a count, so reads placed in FRONT of a product group are waited for at once, while
  // behind five products they have 160 cycles to land.
  struct WSlice { float av, d0[4], d1[4], n0, n1, z0, z1; };
  auto wload = [&](auto sc, WSlice &f) {
    constexpr int s = decltype(sc)::value, m = s >> 2, j = s & 3;
    f.av = ximg[q.aw[j] + 1024 * m];
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) {
      f.d0[ob] = dz0[q.a[j][ob] + 1024 * m];
      f.d1[ob] = dz1[q.a[j][ob] + 1024 * m];
    }
    f.n0 = nblk[q.nb + 64 * s];
    f.n1 = nblk[q.nb + 64 * s + kTR * 16];
    f.z0 = dz0[q.aw[j] + 1024 * m];
    f.z1 = dz1[q.aw[j] + 1024 * m];
  };
  auto xload = [&](auto gc, float4 &a0, float4 &a1) {   // product group g of dX: row tile g / 4, 16-deep slice g % 4 of the contraction
    constexpr int g = decltype(gc)::value, rt = g >> 2, kh = g & 3;
    a0 = *reinterpret_cast<const float4 *>(&dz0[q.xk[kh] + 1024 * rt]);
    a1 = *reinterpret_cast<const float4 *>(&dz1[q.xk[kh] + 1024 * rt]);
  };
  auto wmul = [&](const WSlice &f, int half) {
    if (half == 0) {
      accW[0][0] = mfma16(f.av, f.d0[0], accW[0][0]);
      accW[1][0] = mfma16(f.av, f.d1[0], accW[1][0]);
      accW[0][1] = mfma16(f.av, f.d0[1], accW[0][1]);
      accW[1][1] = mfma16(f.av, f.d1[1], accW[1][1]);
      accN[0] = mfma16(f.n0, f.z0, accN[0]);
    } else {
      accW[0][2] = mfma16(f.av, f.d0[2], accW[0][2]);
      accW[1][2] = mfma16(f.av, f.d1[2], accW[1][2]);
      accW[0][3] = mfma16(f.av, f.d0[3], accW[0][3]);
      accW[1][3] = mfma16(f.av, f.d1[3], accW[1][3]);
      accN[1] = mfma16(f.n1, f.z1, accN[1]);
    }
  };
  auto xmul = [&](const float4 &a, int sd, int rt, int kh) {
    accX[rt] = mfma16(a.x, wf[sd][kh][0], accX[rt]);
    accX[rt] = mfma16(a.y, wf[sd][kh][1], accX[rt]);
    accX[rt] = mfma16(a.z, wf[sd][kh][2], accX[rt]);
    accX[rt] = mfma16(a.w, wf[sd][kh][3], accX[rt]);
  };
  WSlice fs[2];
  float4 xa[2][2];
  wload(std::integral_constant<int, 0>{}, fs[0]);
  static_for<0, 16>([&](auto sc) {
    constexpr int s = decltype(sc)::value;
    __builtin_amdgcn_sched_barrier(0);
    wmul(fs[s & 1], 0);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (s + 1 < 16) wload(std::integral_constant<int, s + 1>{}, fs[(s + 1) & 1]);
    else xload(std::integral_constant<int, 0>{}, xa[0][0], xa[0][1]);   // the first group of dX behind the last slice
    __builtin_amdgcn_sched_barrier(0);
    wmul(fs[s & 1], 1);
  });
  if (st) st[13] = clock64();
  // ---- dX[:, 16 w .. + 15] = dy_a Wa^T + dy_b Wb^T: all 64 rows, this wave's 16 columns, weights from registers; same pipeline
  static_for<0, 16>([&](auto gc) {
    constexpr int g = decltype(gc)::value, rt = g >> 2, kh = g & 3;
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (kh == 0) accX[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    xmul(xa[g & 1][0], 0, rt, kh);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (g + 1 < 16) xload(std::integral_constant<int, g + 1>{}, xa[(g + 1) & 1][0], xa[(g + 1) & 1][1]);
    __builtin_amdgcn_sched_barrier(0);
    xmul(xa[g & 1][1], 1, rt, kh);
  });
}

__global__ __launch_bounds__(kBT, 2) void dense_pair64_bwd_kernel(const DensePairBwdK p) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float *ldsDz0 = lds, *ldsDz1 = lds + kTR * kD;   // [64][64] each, swizzled
  float *ldsX0 = ldsDz1 + kTR * kD;                // two [64][64] swizzled DMA images of X (this tile's, the next one's on its
  float *ldsX1 = ldsX0 + kTR * kPS;                // way); this tile's becomes [64][68] dX on its way out
  float *ldsN = ldsX1 + kTR * kPS;                 // [2][64][16]: per side the narrow features (columns 0..3), a column of ones
                                                   // (4: the bias gradient), zeros -- a 16-wide "input block" for the matrix pipe
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i = lane & 15, kq = lane >> 4;
  const int rg = tid >> 4, qc = tid & 15;

  f32x4 wf[2][4];   // W_s[main_off_s + 16 wave + i][16 kh + 4 kq .. + 3]
#pragma unroll
  for (int sd = 0; sd < 2; ++sd)
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      const float4 w4 = *reinterpret_cast<const float4 *>(p.wt[sd] + (size_t)(p.main_off[sd] + 16 * wave + i) * kD + 16 * kh + 4 * kq);
      wf[sd][kh] = (f32x4){w4.x, w4.y, w4.z, w4.w};
    }
  for (int idx = tid; idx < 2 * kTR * 16; idx += kBT) ldsN[idx] = ((idx & 15) == 4) ? 1.0f : 0.f;
  f32x4 accW[2][4], accN[2];
#pragma unroll
  for (int sd = 0; sd < 2; ++sd) {
    accN[sd] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int ob = 0; ob < 4; ++ob) accW[sd][ob] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  const PairBwdAddr addr = pair_bwd_addr(wave, lane);
  // prefetch roles: dy rows (rg + 16 pp, columns 4 qc ..) of both sides; narrow values: thread -> (side, row, feature pair)
  const int nsd = tid >> 7, nrow = (tid >> 1) & 63, nf = 2 * (tid & 1);
  float4 dyr[2][4];
  float nv0 = 0.f, nv1 = 0.f;
  // where this thread's two narrow features live: resolved ONCE (left inside fetch the compiler re-reads the six words from the
  // kernel-argument segment per tile, behind the dy loads in the queue, and waits for all of them: a memory round trip per tile)
  unsigned long long nb0u = (unsigned long long)(nsd ? (nf ? p.nx[1][2] : p.nx[1][0]) : (nf ? p.nx[0][2] : p.nx[0][0]));
  unsigned long long nb1u = (unsigned long long)(nsd ? (nf ? p.nx[1][3] : p.nx[1][1]) : (nf ? p.nx[0][3] : p.nx[0][1]));
  uint32_t nw0 = nsd ? (nf ? p.nwidth[1][2] : p.nwidth[1][0]) : (nf ? p.nwidth[0][2] : p.nwidth[0][0]);
  uint32_t nw1 = nsd ? (nf ? p.nwidth[1][3] : p.nwidth[1][1]) : (nf ? p.nwidth[0][3] : p.nwidth[0][1]);
  uint32_t nd0 = nsd ? (nf ? p.ndiv[1][2] : p.ndiv[1][0]) : (nf ? p.ndiv[0][2] : p.ndiv[0][0]);
  uint32_t nd1 = nsd ? (nf ? p.ndiv[1][3] : p.ndiv[1][1]) : (nf ? p.ndiv[0][3] : p.ndiv[0][1]);
  uint32_t nm0 = div_magic(nd0), nm1 = div_magic(nd1);
  asm volatile("" : "+v"(nb0u), "+v"(nb1u), "+v"(nw0), "+v"(nw1), "+v"(nd0), "+v"(nd1), "+v"(nm0), "+v"(nm1));   // (held in registers from here on)
  typedef const __attribute__((address_space(1))) float *gptr_t;   // (a pointer that went through the pin is "generic": say global again)
  const gptr_t nb0 = (gptr_t)nb0u, nb1 = (gptr_t)nb1u;
  auto fetch = [&](int tile) {
    const uint32_t row0 = (uint32_t)tile * kTR;
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const uint32_t r = row0 + rg + 16 * pp;
      const uint32_t e = min(r, (uint32_t)(p.n - 1)) * kD + 4 * qc;
#pragma unroll
      for (int sd = 0; sd < 2; ++sd) {
        dyr[sd][pp] = *reinterpret_cast<const float4 *>(p.dy[sd] + e);
        if (r >= (uint32_t)p.n) dyr[sd][pp] = f4_zero();   // rows past the end contribute nothing
      }
    }
    const uint32_t rc = min(row0 + nrow, (uint32_t)(p.n - 1));
    nv0 = nb0[fast_div(rc, nd0, nm0) * nw0];   // (unused slots alias X: their dW rows are never written)
    nv1 = nb1[fast_div(rc, nd1, nm1) * nw1];
  };

  // Per tile: the tile's dy rows and narrow values (registers, loaded a tile ago) -> LDS; then, with the next tile's loads
  // and X image in flight, the products; the loads are collected BEFORE this tile's dX stores are issued (vmcnt counts stores
  // too: collected at the top of the next tile they would cost a store latency per tile).
  dephase_second_half(p.dephase);
  int tile = blockIdx.x, it = 0;
  if (tile < p.n_tiles) {
    fetch(tile);
    dma_x_tile(p.x, (uint32_t)tile * kTR, (uint32_t)p.n, ldsX0, wave, lane);
  }
  wait_vmcnt0();
  for (; tile < p.n_tiles; tile += gridDim.x, ++it) {
    const int64_t row0 = (int64_t)tile * kTR;
    float *cur = (it & 1) ? ldsX1 : ldsX0, *nxt = (it & 1) ? ldsX0 : ldsX1;
    PBWD_STAMP(0);
    __syncthreads();   // the previous tile's dz tiles and outgoing dX are consumed
    PBWD_STAMP(1);
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int r = rg + 16 * pp;
      *reinterpret_cast<float4 *>(&ldsDz0[r * kD + 4 * (qc ^ swz(r))]) = dyr[0][pp];
      *reinterpret_cast<float4 *>(&ldsDz1[r * kD + 4 * (qc ^ swz(r))]) = dyr[1][pp];
    }
    *reinterpret_cast<float2 *>(&ldsN[(nsd * kTR + nrow) * 16 + nf]) = make_float2(nv0, nv1);
    __syncthreads();
    PBWD_STAMP(2);
    const int tnext = tile + gridDim.x;
    if (tnext < p.n_tiles) fetch(tnext);
    PBWD_STAMP(3);
    f32x4 accX[4];
    pair_bwd_products(cur, nxt, ldsDz0, ldsDz1, ldsN, tnext < p.n_tiles, p.x, (uint32_t)tnext * kTR, (uint32_t)p.n, wave, lane, addr, wf,
                      accW, accN, accX
#ifdef NGPDE_STAMPS
                      , (threadIdx.x == 0 && p.stamps && it == 3) ? p.stamps + (size_t)blockIdx.x * 16 : nullptr
#endif
                      );
    PBWD_STAMP(4);
    wait_vmcnt0();   // (tracked by the compiler: behind an asm wait it would drain vmcnt again at the next barrier, with the addend
                     // loads below in flight)
    // the addend's rows: in flight while dX is staged.  Inline loads: written as C++ loads the compiler sinks each of them
    // into the guarded store below -- load, wait (for the previous store with it), add, store, four times in a row
    PBWD_STAMP(5);
    f32x4 addv[4];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) addv[pp] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (p.dx_add) {
#pragma unroll
      for (int pp = 0; pp < 4; ++pp) {
        const float *src = p.dx_add + min(row0 + rg + 16 * pp, p.n - 1) * kD + 4 * qc;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(addv[pp]) : "v"(src));
      }
    }
    __syncthreads();   // every wave is done with the X image
    PBWD_STAMP(6);
#pragma unroll
    for (int rt = 0; rt < 4; ++rt)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) cur[(16 * rt + 4 * kq + reg) * kPS + 16 * wave + i] = accX[rt][reg];
    __syncthreads();
    PBWD_STAMP(7);
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(addv[0]), "+v"(addv[1]), "+v"(addv[2]), "+v"(addv[3]) : : "memory");
    PBWD_STAMP(8);
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int r = rg + 16 * pp;
      if (row0 + r < p.n) {
        const float4 v = *reinterpret_cast<const float4 *>(&cur[r * kPS + 4 * qc]);
        *reinterpret_cast<float4 *>(p.dx + (row0 + r) * kD + 4 * qc) = make_float4(v.x + addv[pp][0], v.y + addv[pp][1], v.z + addv[pp][2], v.w + addv[pp][3]);
      }
    }
    PBWD_STAMP(9);
#ifdef NGPDE_STAMPS
    if (threadIdx.x == 0 && p.stamps && it == 3) {   // which CU / XCD the workgroup runs on: who shares a CU with whom
      unsigned hw, xcc;
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
      asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
      p.stamps[(size_t)blockIdx.x * 16 + 10] = hw;
      p.stamps[(size_t)blockIdx.x * 16 + 11] = xcc;
    }
#endif
  }

  // ---- slabs of both sides: the accumulators directly (row 4 of the narrow block = the bias gradient)
#pragma unroll
  for (int sd = 0; sd < 2; ++sd) {
    float *slab = p.partial[sd] + (size_t)blockIdx.x * (p.din[sd] + 1) * kD;
#pragma unroll
    for (int ob = 0; ob < 4; ++ob)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) slab[(size_t)(p.main_off[sd] + 16 * wave + 4 * kq + reg) * kD + 16 * ob + i] = accW[sd][ob][reg];
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int f = 4 * kq + reg;
      if (f < p.n_narrow[sd]) slab[(size_t)p.nfeat[sd][f < kMaxNarrow ? f : 0] * kD + 16 * wave + i] = accN[sd][reg];
      else if (f == 4) slab[(size_t)p.din[sd] * kD + 16 * wave + i] = accN[sd][reg];
    }
  }
}

int dephase_cycles() {
  static const int v = [] { const char *e = std::getenv("NGPDE_DENSE_DEPHASE"); return e ? std::atoi(e) : 0; }();
  return v;
}
bool env_on(const char *name) {
  const char *e = std::getenv(name);
  return e && e[0] == '1';
}

}  // namespace

// Applies when: 64 outputs, one or two blocks of exactly 64 features (row_div 1, 16-byte aligned), every other feature
// narrow (<= 4 in total) and without a gradient request, enough rows for a resident wave of tiles, and the slabs fit the dz
// area of the caller's workspace.  NGPDE_DENSE_NO_STREAM_BWD=1 keeps the composed path (read per call).
int dense_stream_bwd_grid(int64_t n, const SegTable &t, int din, int dout, float *const *dseg) {
  if (env_on("NGPDE_DENSE_NO_STREAM_BWD") || dout != kD || n < 32768 || n > (1 << 24)) return 0;   // (32-bit byte offsets)
  int n_main = 0, n_narrow = 0;
  for (int b = 0; b < t.n; ++b) {
    const bool grad = dseg && dseg[b] && t.row_div[b] == 1;
    if (t.width[b] == kD && t.row_div[b] == 1 && (reinterpret_cast<uintptr_t>(t.ptr[b]) & 15) == 0 &&
        (!grad || (reinterpret_cast<uintptr_t>(dseg[b]) & 15) == 0)) {
      ++n_main;
    } else {
      if (grad) return 0;
      n_narrow += t.width[b];
    }
  }
  if (n_main < 1 || n_main > 2 || n_narrow > kMaxNarrow) return 0;
  const int n_tiles = (int)((n + kTR - 1) / kTR);
  int grid = std::min(n_tiles, 256 * 2);
  grid = (int)std::min<int64_t>(grid, n / (din + 1));   // slabs live in the [n][64] dz area of the workspace
  return grid >= 256 ? grid : 0;
}

int32_t launch_dense_stream_bwd(int64_t n, const SegTable &t, int din, int act, const float *wt, const float *z, const float *dy,
                                float *const *dseg, float *dwt, float *dbias, float *slabs, int grid, hipStream_t stream) {
  DenseBwdK k{};
  k.n = n; k.n_tiles = (int)((n + kTR - 1) / kTR); k.din = din; k.act = act;
  k.wt = wt; k.z = (act == NGPDE_ACT_IDENTITY) ? nullptr : z; k.dy = dy; k.partial = slabs; k.dephase = dephase_cycles();
  int n_main = 0;
  for (int b = 0; b < t.n; ++b) {
    const bool grad = dseg && dseg[b] && t.row_div[b] == 1;
    if (t.width[b] == kD && t.row_div[b] == 1 && (reinterpret_cast<uintptr_t>(t.ptr[b]) & 15) == 0 &&
        (!grad || (reinterpret_cast<uintptr_t>(dseg[b]) & 15) == 0)) {
      k.x[n_main] = t.ptr[b]; k.dx[n_main] = grad ? dseg[b] : nullptr; k.main_off[n_main] = t.offset[b];
      ++n_main;
    } else {
      for (int c = 0; c < t.width[b]; ++c) {
        k.nx[k.n_narrow] = t.ptr[b] + c; k.nwidth[k.n_narrow] = t.width[b]; k.ndiv[k.n_narrow] = t.row_div[b];
        k.nfeat[k.n_narrow] = t.offset[b] + c;
        ++k.n_narrow;
      }
    }
  }
  for (int f = k.n_narrow; f < kMaxNarrow; ++f) { k.nx[f] = k.x[0]; k.nwidth[f] = kD; k.ndiv[f] = 1; k.nfeat[f] = 0; }
  const size_t lds = ((size_t)kTR * kD + (size_t)(n_main + 1) * kTR * kPS) * sizeof(float);   // 50 / 67 KB
  auto launch = [&](auto kernel) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(kBT), lds, stream, k);
    return hipSuccess;
  };
  const hipError_t le = n_main == 1 ? launch(dense_stream64_bwd_kernel<1>) : launch(dense_stream64_bwd_kernel<2>);
  if (le != hipSuccess) return fail(NGPDE_ERR_HIP, "dense_stream64_bwd_kernel: LDS request of %zu bytes refused: %s", lds, hipGetErrorString(le));
  NGPDE_LAUNCH_CHECK("dense_stream64_bwd_kernel");
  return launch_dense_weight_reduce(grid, din, kD, slabs, dwt, dbias, stream);
}

// ---- the pair pullback: both sides one 64-wide leading block at the SAME address + narrow blocks, 64 outputs, no activation
static bool pair_side_ok(const SegTable &t, int din) {
  if (t.n < 1 || t.width[0] != kD || t.row_div[0] != 1 || (reinterpret_cast<uintptr_t>(t.ptr[0]) & 15)) return false;
  return din - kD <= kMaxNarrow;
}
int dense_pair_bwd_grid(int64_t n, const SegTable &ta, int dina, const SegTable &tb, int dinb) {
  if (env_on("NGPDE_DENSE_NO_STREAM_BWD") || n < 32768 || n > (1 << 24)) return 0;
  if (!pair_side_ok(ta, dina) || !pair_side_ok(tb, dinb) || ta.ptr[0] != tb.ptr[0]) return 0;
  return std::min((int)((n + kTR - 1) / kTR), 256 * 2);
}
size_t dense_pair_bwd_workspace(int grid, int dina, int dinb) { return (size_t)grid * (dina + dinb + 2) * kD * sizeof(float) + 512; }

int32_t launch_dense_pair_bwd(int64_t n, const SegTable &ta, int dina, const float *wta, const float *dya, float *dwta, float *dba,
                              const SegTable &tb, int dinb, const float *wtb, const float *dyb, float *dwtb, float *dbb, float *dx,
                              const float *dx_add, void *workspace, int grid, hipStream_t stream) {
  DensePairBwdK k{};
  k.n = n; k.n_tiles = (int)((n + kTR - 1) / kTR); k.x = ta.ptr[0]; k.dx = dx; k.dx_add = dx_add; k.dephase = dephase_cycles();
#ifdef NGPDE_STAMPS
  k.stamps = g_pair_bwd_stamps;
#endif
  const SegTable *ts[2] = {&ta, &tb};
  const int dins[2] = {dina, dinb};
  const float *wts[2] = {wta, wtb}, *dys[2] = {dya, dyb};
  float *slabs = (float *)workspace;
  for (int sd = 0; sd < 2; ++sd) {
    const SegTable &t = *ts[sd];
    k.din[sd] = dins[sd]; k.main_off[sd] = t.offset[0]; k.wt[sd] = wts[sd]; k.dy[sd] = dys[sd];
    k.partial[sd] = slabs;
    slabs += (size_t)grid * (dins[sd] + 1) * kD;
    int nn = 0;
    for (int b = 1; b < t.n; ++b)
      for (int c = 0; c < t.width[b]; ++c) {
        k.nx[sd][nn] = t.ptr[b] + c; k.nwidth[sd][nn] = t.width[b]; k.ndiv[sd][nn] = t.row_div[b]; k.nfeat[sd][nn] = t.offset[b] + c;
        ++nn;
      }
    k.n_narrow[sd] = nn;
    for (int f = nn; f < kMaxNarrow; ++f) { k.nx[sd][f] = k.x; k.nwidth[sd][f] = kD; k.ndiv[sd][f] = 1; k.nfeat[sd][f] = 0; }
  }
  const size_t lds = ((size_t)2 * kTR * kD + (size_t)2 * kTR * kPS + (size_t)2 * kTR * 16) * sizeof(float);   // 74.8 KB: two per CU
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dense_pair64_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return fail(NGPDE_ERR_HIP, "dense_pair64_bwd_kernel: LDS request of %zu bytes refused: %s", lds, hipGetErrorString(e));
  hipLaunchKernelGGL(dense_pair64_bwd_kernel, dim3(grid), dim3(kBT), lds, stream, k);
  NGPDE_LAUNCH_CHECK("dense_pair64_bwd_kernel");
  int32_t st;
  if ((st = launch_dense_weight_reduce(grid, dina, kD, k.partial[0], dwta, dba, stream))) return st;
  return launch_dense_weight_reduce(grid, dinb, kD, k.partial[1], dwtb, dbb, stream);
}

}  // namespace ngpde

#ifdef NGPDE_STAMPS
extern "C" int32_t ngpde_debug_set_pair_bwd_stamps(unsigned long long *buf) {
  ngpde::g_pair_bwd_stamps = buf;
  return 0;
}
#endif
