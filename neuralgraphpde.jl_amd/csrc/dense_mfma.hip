// dense_mfma.hip -- fp32 MFMA kernels of the concat-free Dense (Lux.Dense on a virtual vcat of blocks) used
// by the edge-function layers of /root/reference/src/layers.jl (:106, :316, :328, :409, :418, :523) over N node
// columns or E edge columns:  y = act([X1 | X2 | ...] Wt + b), its input pullback and its weight pullback.
// 64 x 64 output tile per workgroup (4 waves x (16 rows x 64 cols)), K in chunks of 16 staged through LDS with
// the B operand stored transposed, so every operand fetch is one ds_read_b128 feeding four
// v_mfma_f32_16x16x4_f32 k-steps (exact fp32); the next chunk's global loads are in flight during the MFMAs.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

constexpr int BM = 64, BN = 64, BK = 16, LS = BK + 4;   // LDS row stride 20 floats: 16-byte aligned b128 rows

// rows are < 2^31 (host-checked), so the per-graph row division is a 32-bit one and only taken when a block asks for it
__device__ __forceinline__ int64_t seg_row(int64_t row, int row_div) {
  return row_div == 1 ? row : (int64_t)((uint32_t)row / (uint32_t)row_div);
}

// the block of the segmented input that holds feature k, as a select chain over the (static) table entries: the table
// stays in SGPRs / kernel arguments (a loop with early exit made the compiler spill it to scratch)
struct SegRef {
  const float *ptr;
  int width, row_div, offset, end, vec;
};
__device__ __forceinline__ SegRef seg_find(const SegTable &s, int k) {
  SegRef r{s.ptr[0], s.width[0], s.row_div[0], s.offset[0], s.offset[1], s.vec[0]};
  if (s.n > 1 && k >= s.offset[1]) r = SegRef{s.ptr[1], s.width[1], s.row_div[1], s.offset[1], s.offset[2], s.vec[1]};
  if (s.n > 2 && k >= s.offset[2]) r = SegRef{s.ptr[2], s.width[2], s.row_div[2], s.offset[2], s.offset[3], s.vec[2]};
  if (s.n > 3 && k >= s.offset[3]) r = SegRef{s.ptr[3], s.width[3], s.row_div[3], s.offset[3], s.offset[4], s.vec[3]};
  return r;
}

// caller guarantees k < din (= s.offset[4])
__device__ __forceinline__ float seg_load(const SegTable &s, int64_t row, int k) {
  const SegRef r = seg_find(s, k);
  return r.ptr[seg_row(row, r.row_div) * r.width + (k - r.offset)];
}

// acc[ct] += A[16 rows of this wave][BK] x B[BK][16 ct]   from LDS (A row-major [BM][LS], Bt [BN][LS])
__device__ __forceinline__ void mfma_chunk(const float *ldsA, const float *ldsBt, int wave, int lane, f32x4 (&acc)[4]) {
  const int i = lane & 15, kq = lane >> 4;
  const float4 a4 = *reinterpret_cast<const float4 *>(&ldsA[(wave * 16 + i) * LS + 4 * kq]);
  float4 b4[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) b4[ct] = *reinterpret_cast<const float4 *>(&ldsBt[(ct * 16 + i) * LS + 4 * kq]);
  const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const float bv[4] = {b4[ct].x, b4[ct].y, b4[ct].z, b4[ct].w};
      acc[ct] = mfma16(av[r], bv[r], acc[ct]);
    }
  }
}

// ---- forward: y[n][o] = act(sum_k X[n][k] wt[k][o] + b[o]) -----------------------------------------------------
// Split-K (part_stride > 0): workgroup z contracts the features [z * kper, (z + 1) * kper) only and writes its raw partial
// products to y + z * part_stride (no bias, no activation); the caller sums the partials (the GCN path does it inside the
// aggregation that follows, src/layers.jl:220-223).  For tall-K / few-row shapes such as GCNConv(1433 => 16) on a 2.7k-node
// graph: 43 row tiles alone would leave 5/6 of the chip idle behind 90 dependent K steps each.
// (the body of one 64 x 64 output tile, shared by the single-problem kernel and the multi-problem one below)
__device__ __forceinline__ void dense_mfma_fwd_tile(int64_t n, const SegTable &segs, int din_all, int dout, int act,
                                                    const float *__restrict__ wt, const float *__restrict__ bias, float *__restrict__ y,
                                                    float *__restrict__ save_z, int kper, size_t part_stride, int bx, int by, int bz,
                                                    float *ldsA, float *ldsBt) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)bx * BM;
  const int col0 = by * BN;
  const int kbeg = bz * kper, din = min(din_all, kbeg + kper);   // this workgroup's feature range [kbeg, din)
  if (part_stride) { y += (size_t)bz * part_stride; bias = nullptr; save_z = nullptr; act = NGPDE_ACT_IDENTITY; }
  // staging roles: A element (row = tid / 16 + 16 p, k = tid % 16), B element (k = tid / 64 + 4 p, col = tid % 64)
  const int ar = tid >> 4, ak = tid & 15, bk = tid >> 6, bc = tid & 63;
  float areg[4], breg[4];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t r = row0 + ar + 16 * p;
      areg[p] = (r < n && k0 + ak < din) ? seg_load(segs, r, k0 + ak) : 0.f;
      const int k = k0 + bk + 4 * p;
      breg[p] = (k < din && col0 + bc < dout) ? wt[(size_t)k * dout + col0 + bc] : 0.f;
    }
  };
  f32x4 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  fetch(kbeg);
  for (int k0 = kbeg; k0 < din; k0 += BK) {
    __syncthreads();   // previous chunk fully consumed
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      ldsA[(ar + 16 * p) * LS + ak] = areg[p];
      ldsBt[bc * LS + bk + 4 * p] = breg[p];
    }
    __syncthreads();
    if (k0 + BK < din) fetch(k0 + BK);   // in flight during the MFMAs
    mfma_chunk(ldsA, ldsBt, wave, lane, acc);
  }
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int o = col0 + ct * 16 + i;
    const float b = (bias && o < dout) ? bias[o] : 0.f;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t r = row0 + wave * 16 + 4 * kq + reg;
      if (r < n && o < dout) {
        const float z = acc[ct][reg] + b;
        if (save_z) save_z[r * dout + o] = z;
        y[r * dout + o] = act_apply(act, z);
      }
    }
  }
}

__global__ __launch_bounds__(256) void dense_mfma_fwd_kernel(int64_t n, SegTable segs, int din_all, int dout, int act,
                                                             const float *__restrict__ wt, const float *__restrict__ bias,
                                                             float *__restrict__ y, float *__restrict__ save_z, int kper,
                                                             size_t part_stride) {
  __shared__ __attribute__((aligned(16))) float ldsA[BM * LS], ldsBt[BN * LS];
  dense_mfma_fwd_tile(n, segs, din_all, dout, act, wt, bias, y, save_z, kper, part_stride, blockIdx.x, blockIdx.y, blockIdx.z, ldsA, ldsBt);
}

// Several INDEPENDENT small Dense layers in one launch (<= 4 problems; a 1-D grid over the 64 x 64 output tiles of all of them).
// GNOConv's node-level terms P, Q, B2 h and W h (/root/reference/src/layers.jl:523,536: 4 096 rows each) are four launches of a
// few dozen workgroups behind 2-8 dependent K chunks -- 8-16 us apiece, one after the other; together they are one such latency.
struct MultiFwdK {
  int count;
  int tile0[5];       // first tile of problem q in the grid; tile0[count] = the grid size
  int col_tiles[4];
  struct Prob {
    int64_t n;
    SegTable segs;
    int din, dout, act;
    const float *wt, *bias;
    float *y, *save_z;
  } p[4];
};
__global__ __launch_bounds__(256) void dense_mfma_multi_fwd_kernel(const MultiFwdK mk) {
  __shared__ __attribute__((aligned(16))) float ldsA[BM * LS], ldsBt[BN * LS];
  const int b = blockIdx.x;
  int q = 0;
  while (q + 1 < mk.count && b >= mk.tile0[q + 1]) ++q;   // block-uniform
  const int local = b - mk.tile0[q];
  const MultiFwdK::Prob &pr = mk.p[q];
  dense_mfma_fwd_tile(pr.n, pr.segs, pr.din, pr.dout, pr.act, pr.wt, pr.bias, pr.y, pr.save_z, pr.din, (size_t)0, local / mk.col_tiles[q],
                      local % mk.col_tiles[q], 0, ldsA, ldsBt);
}

// ---- input pullback: dX[n][k] = sum_o dz[n][o] wt[k][o], written into the blocks that ask for it -------------------
// Split over the OUTPUT features (the contraction of this product) when `part` is given: workgroup z contracts
// [z * oper, (z + 1) * oper); z = 0 writes into the gradient block itself, z >= 1 into part + (z - 1) * part_stride, summed
// afterwards (single-block inputs only).  For the pullback of a few-row x very-wide Dense such as GNOConv's T = W2 (x) h
// (4096 rows, 8192 outputs): 128 workgroups behind 512 dependent K steps otherwise.
__global__ __launch_bounds__(256) void dense_mfma_bwd_input_kernel(int64_t n, SegGrad segs, int din, int dout_all,
                                                                   const float *__restrict__ dz,
                                                                   const float *__restrict__ wt, int oper, float *part,
                                                                   size_t part_stride) {
  __shared__ __attribute__((aligned(16))) float ldsA[BM * LS], ldsBt[BN * LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int col0 = blockIdx.y * BN;          // columns of dX = input features k
  const int obeg = blockIdx.z * oper, dout = min(dout_all, obeg + oper);   // this workgroup's output-feature range [obeg, dout)
  if (blockIdx.z > 0) segs.ptr[0] = part + (size_t)(blockIdx.z - 1) * part_stride;
  const int ar = tid >> 4, ak = tid & 15;    // A = dz: (row, o)
  const int bcol = tid >> 2, bo4 = (tid & 3) * 4;   // Bt[col = k][o]: thread loads 4 consecutive o of one k row
  float areg[4];
  float4 breg;
  auto fetch = [&](int o0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t r = row0 + ar + 16 * p;
      areg[p] = (r < n && o0 + ak < dout) ? dz[r * dout_all + o0 + ak] : 0.f;
    }
    const int k = col0 + bcol;
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = (k < din && o0 + bo4 + j < dout) ? wt[(size_t)k * dout_all + o0 + bo4 + j] : 0.f;
    breg = make_float4(t[0], t[1], t[2], t[3]);
  };
  f32x4 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  fetch(obeg);
  for (int o0 = obeg; o0 < dout; o0 += BK) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) ldsA[(ar + 16 * p) * LS + ak] = areg[p];
    *reinterpret_cast<float4 *>(&ldsBt[bcol * LS + bo4]) = breg;
    __syncthreads();
    if (o0 + BK < dout) fetch(o0 + BK);
    mfma_chunk(ldsA, ldsBt, wave, lane, acc);
  }
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int k = col0 + ct * 16 + i;
    if (k >= din) continue;
    int sg = -1;
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (q < segs.n && k >= segs.offset[q] && k < segs.offset[q + 1]) sg = q;
    if (sg < 0 || !segs.ptr[sg]) continue;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int64_t r = row0 + wave * 16 + 4 * kq + reg;
      if (r < n) segs.ptr[sg][r * segs.width[sg] + (k - segs.offset[sg])] = acc[ct][reg];
    }
  }
}

// ---- wide-tile variants (128 rows x 64 columns per workgroup, 32-deep K steps, b128 global loads where a block of the
// segmented input allows it, outputs transposed through LDS into full 256-byte row stores).  Used whenever the problem
// has enough 128-row tiles to fill the chip; the 64-row kernels above serve the small cases.
constexpr int BM2 = 128, BK2 = 32, LS2 = BK2 + 4, OS2 = BN + 4;
constexpr int kWideLds = (BM2 * OS2 > (BM2 + BN) * LS2) ? BM2 * OS2 : (BM2 + BN) * LS2;   // floats

// 4 consecutive features k..k+3 of row `row` of the segmented input (k % 4 == 0)
__device__ __forceinline__ float4 seg_load4(const SegTable &s, int64_t row, int k, int din) {
  const SegRef r = seg_find(s, k);
  if (r.vec && k + 4 <= r.end)
    return *reinterpret_cast<const float4 *>(r.ptr + seg_row(row, r.row_div) * r.width + (k - r.offset));
  float t[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) t[j] = (k + j < din) ? seg_load(s, row, k + j) : 0.f;
  return make_float4(t[0], t[1], t[2], t[3]);
}

// acc[rt][ct] += A[32 rows of this wave][BK2] x B[BK2][64]  from LDS (A row-major [BM2][LS2], Bt [BN][LS2])
__device__ __forceinline__ void mfma_step_wide(const float *ldsA, const float *ldsBt, int wave, int lane, f32x4 (&acc)[2][4]) {
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int kh = 0; kh < BK2 / 16; ++kh) {
    float4 a4[2], b4[4];
#pragma unroll
    for (int rt = 0; rt < 2; ++rt) a4[rt] = *reinterpret_cast<const float4 *>(&ldsA[(wave * 32 + rt * 16 + i) * LS2 + 16 * kh + 4 * kq]);
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) b4[ct] = *reinterpret_cast<const float4 *>(&ldsBt[(ct * 16 + i) * LS2 + 16 * kh + 4 * kq]);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
#pragma unroll
      for (int rt = 0; rt < 2; ++rt) {
        const float av[4] = {a4[rt].x, a4[rt].y, a4[rt].z, a4[rt].w};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const float bv[4] = {b4[ct].x, b4[ct].y, b4[ct].z, b4[ct].w};
          acc[rt][ct] = mfma16(av[r], bv[r], acc[rt][ct]);
        }
      }
    }
  }
}

// accumulators -> LDS tile out[128][OS2] (every wave owns its 32 rows)
__device__ __forceinline__ void acc_to_lds(float *out, int wave, int lane, const f32x4 (&acc)[2][4]) {
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) out[(wave * 32 + rt * 16 + 4 * kq + reg) * OS2 + ct * 16 + i] = acc[rt][ct][reg];
}

// din_main <= din: the leading features contracted on the matrix pipe; the (few, narrow) trailing ones -- node coordinates, the
// per-graph parameters theta of MPPDEConv (src/layers.jl:409-418) -- enter as a rank-(din - din_main) update in the epilogue, so
// that a 64 + 2 + 2 wide input costs two 32-deep K steps, not three with a ragged, scalar-loaded last one.
__global__ __launch_bounds__(256, 4) void dense_wide_fwd_kernel(int64_t n, SegTable segs, int din_all, int dout, int act,
                                                             const float *__restrict__ wt, const float *__restrict__ bias,
                                                             float *__restrict__ y, float *__restrict__ save_z, int din) {
  __shared__ __attribute__((aligned(16))) float lds[kWideLds];
  float *ldsA = lds, *ldsBt = lds + BM2 * LS2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * BM2;
  const int col0 = blockIdx.y * BN;
  // staging roles: A float4 (row = tid / 8 + 32 p, k = 4 (tid % 8)); B: column tid % 64, k = 8 (tid / 64) .. + 7
  const int ar = tid >> 3, ak = 4 * (tid & 7), bc = tid & 63, bk = 8 * (tid >> 6);
  float4 areg[4];
  float breg[8];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t r = row0 + ar + 32 * p;
      areg[p] = (r < n && k0 + ak < din) ? seg_load4(segs, r, k0 + ak, din) : f4_zero();
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + bk + j;
      breg[j] = (k < din && col0 + bc < dout) ? wt[(size_t)k * dout + col0 + bc] : 0.f;
    }
  };
  f32x4 acc[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  fetch(0);
  for (int k0 = 0; k0 < din; k0 += BK2) {
    __syncthreads();   // previous step fully consumed
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<float4 *>(&ldsA[(ar + 32 * p) * LS2 + ak]) = areg[p];
    *reinterpret_cast<float4 *>(&ldsBt[bc * LS2 + bk]) = make_float4(breg[0], breg[1], breg[2], breg[3]);
    *reinterpret_cast<float4 *>(&ldsBt[bc * LS2 + bk + 4]) = make_float4(breg[4], breg[5], breg[6], breg[7]);
    __syncthreads();
    if (k0 + BK2 < din) fetch(k0 + BK2);   // in flight during the MFMAs
    mfma_step_wide(ldsA, ldsBt, wave, lane, acc);
  }
  __syncthreads();
  acc_to_lds(lds, wave, lane, acc);
  __syncthreads();
  // epilogue: thread (row = tid / 16 + 16 p, columns 4 (tid % 16) .. + 3): bias, activation, 16-byte stores
  const int oc = 4 * (tid & 15);
  const bool vec = (dout % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(save_z)) & 15) == 0;
  float b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = (bias && col0 + oc + j < dout) ? bias[col0 + oc + j] : 0.f;
  float4 zz[8], aa[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const float4 v = *reinterpret_cast<const float4 *>(&lds[((tid >> 4) + 16 * p) * OS2 + oc]);
    zz[p] = make_float4(v.x + b[0], v.y + b[1], v.z + b[2], v.w + b[3]);
  }
  for (int k = din; k < din_all; ++k) {   // the narrow trailing features (uniform trip count, usually 0, 2 or 4)
    float w[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) w[j] = (col0 + oc + j < dout) ? wt[(size_t)k * dout + col0 + oc + j] : 0.f;
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int64_t r = row0 + (tid >> 4) + 16 * p;
      const float xv = r < n ? seg_load(segs, r, k) : 0.f;
      zz[p] = make_float4(fmaf(xv, w[0], zz[p].x), fmaf(xv, w[1], zz[p].y), fmaf(xv, w[2], zz[p].z), fmaf(xv, w[3], zz[p].w));
    }
  }
#pragma unroll
  for (int p = 0; p < 8; ++p) aa[p] = zz[p];
  f4n_act<8>(act, aa);   // one uniform activation switch for the thread's 32 values
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int64_t r = row0 + (tid >> 4) + 16 * p;
    if (r >= n || col0 + oc >= dout) continue;
    if (vec) {   // dout % 4 == 0: the four columns exist and the address is 16-byte aligned
      if (save_z) nt_store4(save_z + r * dout + col0 + oc, zz[p]);
      nt_store4(y + r * dout + col0 + oc, aa[p]);
    } else {
      const float z[4] = {zz[p].x, zz[p].y, zz[p].z, zz[p].w}, a4[4] = {aa[p].x, aa[p].y, aa[p].z, aa[p].w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (col0 + oc + j < dout) {
          if (save_z) save_z[r * dout + col0 + oc + j] = z[j];
          y[r * dout + col0 + oc + j] = a4[j];
        }
    }
  }
}

// ---- 128 x 128 output tiles for wide outputs (GNOConv's node-level T = W2 (x) h: 4096 x 128 => 8192, 8.6 GFLOP at config 5) ------
// Four waves as 2 x 2, each 64 x 64 outputs = 16 accumulator tiles: 64 MFMAs per 16-deep K chunk from 4 + 4 operand reads (the
// 128 x 64 tiles above: 64 MFMAs from 12 reads and half the outputs per weight element fetched), K chunks double-buffered in LDS
// (one barrier per chunk), the next chunk's global loads in flight during the products; every wave takes its output out
// through a private 16 x 64 LDS patch, row tile by row tile, as full 256-byte row segments.  Single 16-byte-loadable input
// block, din % 16 == 0, dout % 4 == 0.
constexpr int BG = 128, BKG = 16, LSG = BKG + 4;
__global__ __launch_bounds__(256, 3) void dense_gemm128_fwd_kernel(int64_t n, const float *__restrict__ x, int din, int dout, int act,
                                                                const float *__restrict__ wt, const float *__restrict__ bias,
                                                                float *__restrict__ y, float *__restrict__ save_z) {
  constexpr int BS = BG + 4, kBuf = BG * LSG + BKG * BS;   // per buffer: A [128][LSG], B [16][BS] row-major (k, col)
  __shared__ __attribute__((aligned(16))) float lds[2 * kBuf];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, kq = lane >> 4;
  const int64_t row0 = (int64_t)blockIdx.x * BG;
  const int col0 = blockIdx.y * BG;
  // staging roles: A float4 (row = tid / 4 + 64 p, k = 4 (tid % 4)); B float4 (k = tid / 32 + 8 p, columns 4 (tid % 32) .. + 3)
  const int ar = tid >> 2, ak = 4 * (tid & 3), bk = tid >> 5, bc = 4 * (tid & 31);
  // (named registers: float4 arrays captured by the lambdas end up in scratch memory, with a wait in front of every store)
  float4 areg0, areg1, breg0, breg1;
  const float *xa0 = x + min(row0 + ar, n - 1) * din + ak, *xa1 = x + min(row0 + ar + 64, n - 1) * din + ak;   // rows past the end
  const float *wb0 = wt + (size_t)bk * dout + min(col0 + bc, dout - 4), *wb1 = wb0 + (size_t)8 * dout;        // read the last one
  auto fetch = [&](int k0) {
    areg0 = *reinterpret_cast<const float4 *>(xa0 + k0);
    areg1 = *reinterpret_cast<const float4 *>(xa1 + k0);
    breg0 = *reinterpret_cast<const float4 *>(wb0 + (size_t)k0 * dout);
    breg1 = *reinterpret_cast<const float4 *>(wb1 + (size_t)k0 * dout);
  };
  // B stays row-major in LDS (16-byte stores, no transposition: scalar stores into a transposed tile would hit 4 of the 64
  // banks); the operand reads are four 4-byte reads per fragment, conflict-free with the row stride 132 (rows 4 kq + r land 16
  // banks apart, the 16 columns of a lane group side by side)
  auto stage = [&](int buf) {
    float *A = lds + buf * kBuf, *B = A + BG * LSG;
    *reinterpret_cast<float4 *>(&A[ar * LSG + ak]) = areg0;
    *reinterpret_cast<float4 *>(&A[(ar + 64) * LSG + ak]) = areg1;
    *reinterpret_cast<float4 *>(&B[bk * BS + bc]) = breg0;
    *reinterpret_cast<float4 *>(&B[(bk + 8) * BS + bc]) = breg1;
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  fetch(0);
  stage(0);
  __syncthreads();
  const int nk = din / BKG;
  for (int kc = 0; kc < nk; ++kc) {
    if (kc + 1 < nk) fetch((kc + 1) * BKG);   // in flight during the MFMAs
    const float *A = lds + (kc & 1) * kBuf, *B = A + BG * LSG;
    float4 a4[4];
    float bv[4][4];
#pragma unroll
    for (int rt = 0; rt < 4; ++rt) a4[rt] = *reinterpret_cast<const float4 *>(&A[(64 * wr + 16 * rt + i) * LSG + 4 * kq]);
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) bv[r][ct] = B[(4 * kq + r) * BS + 64 * wc + 16 * ct + i];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const float av = r == 0 ? a4[rt].x : r == 1 ? a4[rt].y : r == 2 ? a4[rt].z : a4[rt].w;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = mfma16(av, bv[r][ct], acc[rt][ct]);
      }
    if (kc + 1 < nk) stage((kc + 1) & 1);   // the other buffer: last read in chunk kc - 1, before the barrier below
    __syncthreads();
  }
  // epilogue: this wave's 64 x 64 outputs, 16 rows at a time through its own LDS patch
  float *patch = lds + wave * (16 * OS2);
  const int pc = 4 * (lane & 15);
  float b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = (bias && col0 + 64 * wc + pc + j < dout) ? bias[col0 + 64 * wc + pc + j] : 0.f;
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) patch[(4 * kq + reg) * OS2 + 16 * ct + i] = acc[rt][ct][reg];
    float4 zz[4], aa[4];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const float4 v = *reinterpret_cast<const float4 *>(&patch[((lane >> 4) + 4 * pp) * OS2 + pc]);
      zz[pp] = make_float4(v.x + b[0], v.y + b[1], v.z + b[2], v.w + b[3]);
      aa[pp] = zz[pp];
    }
    f4n_act<4>(act, aa);
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int64_t r = row0 + 64 * wr + 16 * rt + (lane >> 4) + 4 * pp;
      const int c = col0 + 64 * wc + pc;
      if (r < n && c < dout) {
        if (save_z) nt_store4(save_z + r * dout + c, zz[pp]);
        nt_store4(y + r * dout + c, aa[pp]);
      }
    }
  }
}

// ---- the same 128 x 128 tiles for the two pullbacks of such a layer, with the contraction split over workgroups -----------------
//   input pullback   dX [n][din]   = dz [n][dout] x Wt^T          (contraction over dout: both operands contraction-contiguous)
//   weight pullback  dWt[din][dout] = X^T [din][n] x dz [n][dout]  (contraction over the rows: both operands row-major in k)
// C [M][N] (+)= A x B over k in [z kper, (z + 1) kper), z = blockIdx.z: split 0 writes c0, split z > 0 writes cpart + (z - 1)
// part_stride (the callers' add_partials / dense_weight_reduce passes sum the slabs in a fixed order).  A_MN: A is stored
// [k][lda] with m contiguous (else [m][lda], k contiguous); B_K: B is stored [n][ldb] with k contiguous (else [k][ldb]).  A
// contraction-contiguous operand is staged as [128][16 + 4] and read as one 16-byte fragment per 16 x 4 block; the other kind
// stays row-major [16][128 + 4] and is read 4 bytes at a time (dense_gemm128_fwd_kernel's two forms).  K, kper multiples of 16;
// M, N multiples of 4.
// one 128 x 128 tile of C = A x B over the contraction range [kbeg, kend) (the body of the two kernels below)
template <bool A_MN, bool B_K>
__device__ __forceinline__ void gemm128_tile(int M, int N, int kbeg, int kend, const float *__restrict__ A, int lda, const float *__restrict__ B,
                                             int ldb, float *__restrict__ C, int ldc) {
  constexpr int BS = BG + 4, kOp = BG * LSG, kBuf = 2 * kOp;   // per buffer: two operand tiles of at most 128 x 20 floats
  __shared__ __attribute__((aligned(16))) float lds[2 * kBuf];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wr = wave >> 1, wc = wave & 1;
  const int i = lane & 15, kq = lane >> 4;
  const int m0 = blockIdx.x * BG, n0 = blockIdx.y * BG;
  // staging roles: contraction-contiguous operand: float4 (row = tid / 4 + 64 p, k = 4 (tid % 4)); row-major-in-k operand: float4
  // (k = tid / 32 + 8 p, column 4 (tid % 32) .. + 3); rows / columns past the end re-read the last ones (never stored)
  const int kr = tid >> 2, kk = 4 * (tid & 3), rk = tid >> 5, rc = 4 * (tid & 31);
  const float *pa0, *pa1, *pb0, *pb1;
  if (A_MN) { pa0 = A + (size_t)rk * lda + min(m0 + rc, M - 4); pa1 = pa0 + (size_t)8 * lda; }
  else { pa0 = A + (size_t)min(m0 + kr, M - 1) * lda + kk; pa1 = A + (size_t)min(m0 + kr + 64, M - 1) * lda + kk; }
  if (B_K) { pb0 = B + (size_t)min(n0 + kr, N - 1) * ldb + kk; pb1 = B + (size_t)min(n0 + kr + 64, N - 1) * ldb + kk; }
  else { pb0 = B + (size_t)rk * ldb + min(n0 + rc, N - 4); pb1 = pb0 + (size_t)8 * ldb; }
  float4 areg0, areg1, breg0, breg1;
  auto fetch = [&](int k0) {
    areg0 = *reinterpret_cast<const float4 *>(A_MN ? pa0 + (size_t)k0 * lda : pa0 + k0);
    areg1 = *reinterpret_cast<const float4 *>(A_MN ? pa1 + (size_t)k0 * lda : pa1 + k0);
    breg0 = *reinterpret_cast<const float4 *>(B_K ? pb0 + k0 : pb0 + (size_t)k0 * ldb);
    breg1 = *reinterpret_cast<const float4 *>(B_K ? pb1 + k0 : pb1 + (size_t)k0 * ldb);
  };
  auto stage = [&](int buf) {
    float *At = lds + buf * kBuf, *Bt = At + kOp;
    if (A_MN) {
      *reinterpret_cast<float4 *>(&At[rk * BS + rc]) = areg0;
      *reinterpret_cast<float4 *>(&At[(rk + 8) * BS + rc]) = areg1;
    } else {
      *reinterpret_cast<float4 *>(&At[kr * LSG + kk]) = areg0;
      *reinterpret_cast<float4 *>(&At[(kr + 64) * LSG + kk]) = areg1;
    }
    if (B_K) {
      *reinterpret_cast<float4 *>(&Bt[kr * LSG + kk]) = breg0;
      *reinterpret_cast<float4 *>(&Bt[(kr + 64) * LSG + kk]) = breg1;
    } else {
      *reinterpret_cast<float4 *>(&Bt[rk * BS + rc]) = breg0;
      *reinterpret_cast<float4 *>(&Bt[(rk + 8) * BS + rc]) = breg1;
    }
  };
  f32x4 acc[4][4];
#pragma unroll
  for (int rt = 0; rt < 4; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int nk = kend > kbeg ? (kend - kbeg) / BKG : 0;   // (an empty split writes zeros: the reduce passes read every slab)
  if (nk > 0) {
    fetch(kbeg);
    stage(0);
  }
  __syncthreads();
  for (int kc = 0; kc < nk; ++kc) {
    if (kc + 1 < nk) fetch(kbeg + (kc + 1) * BKG);   // in flight during the MFMAs
    const float *At = lds + (kc & 1) * kBuf, *Bt = At + kOp;
    float av[4][4], bv[4][4];   // [r][tile]
    if (A_MN) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int rt = 0; rt < 4; ++rt) av[r][rt] = At[(4 * kq + r) * BS + 64 * wr + 16 * rt + i];
    } else {
#pragma unroll
      for (int rt = 0; rt < 4; ++rt) {
        const float4 a4 = *reinterpret_cast<const float4 *>(&At[(64 * wr + 16 * rt + i) * LSG + 4 * kq]);
        av[0][rt] = a4.x; av[1][rt] = a4.y; av[2][rt] = a4.z; av[3][rt] = a4.w;
      }
    }
    if (B_K) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float4 b4 = *reinterpret_cast<const float4 *>(&Bt[(64 * wc + 16 * ct + i) * LSG + 4 * kq]);
        bv[0][ct] = b4.x; bv[1][ct] = b4.y; bv[2][ct] = b4.z; bv[3][ct] = b4.w;
      }
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) bv[r][ct] = Bt[(4 * kq + r) * BS + 64 * wc + 16 * ct + i];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int rt = 0; rt < 4; ++rt)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = mfma16(av[r][rt], bv[r][ct], acc[rt][ct]);
    if (kc + 1 < nk) stage((kc + 1) & 1);
    __syncthreads();
  }
  // this wave's 64 x 64 outputs, 16 rows at a time through its own LDS patch, as full 256-byte row segments
  float *patch = lds + wave * (16 * OS2);
  const int pc = 4 * (lane & 15);
#pragma unroll
  for (int rt = 0; rt < 4; ++rt) {
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) patch[(4 * kq + reg) * OS2 + 16 * ct + i] = acc[rt][ct][reg];
#pragma unroll
    for (int pp = 0; pp < 4; ++pp) {
      const int r = m0 + 64 * wr + 16 * rt + (lane >> 4) + 4 * pp, c = n0 + 64 * wc + pc;
      if (r < M && c < N) *reinterpret_cast<float4 *>(C + (size_t)r * ldc + c) = *reinterpret_cast<const float4 *>(&patch[((lane >> 4) + 4 * pp) * OS2 + pc]);
    }
  }
}

template <bool A_MN, bool B_K>
__global__ __launch_bounds__(256, 3) void dense_gemm128_split_kernel(int M, int N, int K, int kper, const float *__restrict__ A, int lda,
                                                                     const float *__restrict__ B, int ldb, float *__restrict__ c0,
                                                                     float *__restrict__ cpart, size_t part_stride, int ldc) {
  const int kbeg = blockIdx.z * kper, kend = min(K, kbeg + kper);
  float *C = blockIdx.z == 0 ? c0 : cpart + (size_t)(blockIdx.z - 1) * part_stride;
  gemm128_tile<A_MN, B_K>(M, N, kbeg, kend, A, lda, B, ldb, C, ldc);
}

// the row-major x row-major form with up to two further SHORT products of the same output shape in the same launch: blockIdx.z <
// nsplit takes range z of the main contraction, blockIdx.z = nsplit + e the whole of product e; every z writes its own slab
// (gno_gform.hip: G W2' split 16 ways, plus hsum B2 and h W -- three launches' worth of workgroups in one)
struct GemmSide {
  int n;
  const float *A[2], *B[2];
  int lda[2], ldb[2], K[2];
};
__global__ __launch_bounds__(256, 3) void dense_gemm128_split_nn_kernel(int M, int N, int K, int kper, int nsplit, const float *__restrict__ A, int lda,
                                                                        const float *__restrict__ B, int ldb, float *__restrict__ slabs,
                                                                        size_t slab_stride, int ldc, const GemmSide side) {
  const int z = blockIdx.z;
  float *C = slabs + (size_t)z * slab_stride;
  if (z < nsplit) {
    gemm128_tile<false, false>(M, N, z * kper, min(K, (z + 1) * kper), A, lda, B, ldb, C, ldc);
  } else {
    const int e = z - nsplit;
    gemm128_tile<false, false>(M, N, 0, side.K[e], side.A[e], side.lda[e], side.B[e], side.ldb[e], C, ldc);
  }
}

// ---- streaming form of the wide forward for a 64-deep contraction and <= 64 outputs (the node-level Dense of the edge-function
// layers: h => 64, [h | d | theta] => 64).  Persistent workgroups (one resident wave of them, three per CU) walk the 128-row
// tiles; W^T is staged once per workgroup; the WHOLE 128 x 64 input tile goes memory -> LDS by LDS-DMA in one burst (32 KB in
// flight per workgroup, no register staging), then eight waves of 16 rows run 64 MFMAs each without another global load.  The
// DMA writes four 256-byte rows linearly per instruction, which as a plain row-major image would put the 16 rows of an MFMA
// operand read on the same banks; so lane s of a row fetches chunk s ^ (row & 15) of that row -- still one coalesced 256-byte row
// per 16 lanes -- and the operand read of chunk k4 of row r looks at slot k4 ^ (r & 15): conflict-free.  Trailing narrow
// features enter in the epilogue as in dense_wide_fwd_kernel.
// Measured at 524 288 x 64 => 64 (tools/bench_dense.py; the matrix-pipe floor is 27 us, a copy of the same bytes 37 us):
// 128-row tiles through registers 82 us -> four waves per SIMD 76 -> whole tile by LDS-DMA, one workgroup per tile 75-79 ->
// persistent 68.  By the SQ counters (tools/pmc_dense.sh) the waves are 56 % issue-stalled and 32 % parked with the matrix
// pipe 38 % busy; staggering the resident workgroups by thirds of a tile period changed nothing, and a barrier-free form in
// which every wave pipelines its own 16 rows and stores from the accumulator registers (64-byte segments) was slower (98 us).
constexpr int kStreamThreads = 512;
constexpr int kStreamLds = BM2 * OS2 + BN * (64 + 4);   // [input tile, later the output tile (128 x 68)] [W^T, resident]
__global__ __launch_bounds__(kStreamThreads, 6) void dense_stream64_fwd_kernel(int64_t n, SegTable segs, int din_all, int dout, int act,
                                                                               const float *__restrict__ wt, const float *__restrict__ bias,
                                                                               float *__restrict__ y, float *__restrict__ save_z, int n_tiles) {
  constexpr int KM = 64, BS = KM + 4;
  __shared__ __attribute__((aligned(16))) float lds[kStreamLds];
  float *ldsA = lds, *ldsBt = lds + BM2 * OS2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  {   // W -> Bt[col][k], transposed through registers (coalesced dword loads down the columns), once per workgroup
    const int bc = tid & 63, bk = 8 * (tid >> 6);
    float w[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) w[j] = (bc < dout) ? wt[(size_t)(bk + j) * dout + bc] : 0.f;
    *reinterpret_cast<float4 *>(&ldsBt[bc * BS + bk]) = make_float4(w[0], w[1], w[2], w[3]);
    *reinterpret_cast<float4 *>(&ldsBt[bc * BS + bk + 4]) = make_float4(w[4], w[5], w[6], w[7]);
  }
  const int i = lane & 15, kq = lane >> 4;
  const int oc = 4 * (tid & 15);
  const bool vec = (dout % 4 == 0) && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(save_z)) & 15) == 0;
  float b[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = (bias && oc + j < dout) ? bias[oc + j] : 0.f;
  for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * BM2;
    __syncthreads();   // the previous tile's output rows have left the LDS tile that the next rows overwrite (first pass: W^T written)
    {   // input tile: four DMA instructions per wave, four rows each
      const int rl = lane >> 4, s16 = lane & 15;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int r = (wave * 4 + j) * 4 + rl;
        const int64_t gr = min(row0 + r, n - 1);                  // rows past the end read the last row (never stored)
        const int k = 4 * (s16 ^ (r & 15));
        const SegRef sr = seg_find(segs, k);
        const float *g = sr.ptr + seg_row(gr, sr.row_div) * sr.width + (k - sr.offset);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                         (__attribute__((address_space(3))) void *)(reinterpret_cast<float4 *>(ldsA) + ((wave * 4 + j) * 4) * 16 + lane),
                                         16, 0, 0);
      }
    }
    wait_vmcnt0();
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
    {
      const int r = wave * 16 + i;
#pragma unroll
      for (int kh = 0; kh < KM / 16; ++kh) {
        const float4 a4 = reinterpret_cast<const float4 *>(ldsA)[r * 16 + ((4 * kh + kq) ^ (r & 15))];
        float4 b4[4];
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) b4[ct] = *reinterpret_cast<const float4 *>(&ldsBt[(ct * 16 + i) * BS + 16 * kh + 4 * kq]);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
#pragma unroll
          for (int ct = 0; ct < 4; ++ct) {
            const float bv[4] = {b4[ct].x, b4[ct].y, b4[ct].z, b4[ct].w};
            acc[ct] = mfma16(av[rr], bv[rr], acc[ct]);
          }
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) lds[(wave * 16 + 4 * kq + reg) * OS2 + ct * 16 + i] = acc[ct][reg];
    __syncthreads();
    // epilogue: thread (row = tid / 16 + 32 p, columns 4 (tid % 16) .. + 3)
    float4 zz[4], aa[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const float4 v = *reinterpret_cast<const float4 *>(&lds[((tid >> 4) + 32 * p) * OS2 + oc]);
      zz[p] = make_float4(v.x + b[0], v.y + b[1], v.z + b[2], v.w + b[3]);
    }
    for (int k = KM; k < din_all; ++k) {   // the narrow trailing features
      float w[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) w[j] = (oc + j < dout) ? wt[(size_t)k * dout + oc + j] : 0.f;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const int64_t r = row0 + (tid >> 4) + 32 * p;
        const float xv = r < n ? seg_load(segs, r, k) : 0.f;
        zz[p] = make_float4(fmaf(xv, w[0], zz[p].x), fmaf(xv, w[1], zz[p].y), fmaf(xv, w[2], zz[p].z), fmaf(xv, w[3], zz[p].w));
      }
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) aa[p] = zz[p];
    f4n_act<4>(act, aa);
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t r = row0 + (tid >> 4) + 32 * p;
      if (r >= n || oc >= dout) continue;
      if (vec) {
        if (save_z) nt_store4(save_z + r * dout + oc, zz[p]);
        nt_store4(y + r * dout + oc, aa[p]);
      } else {
        const float z[4] = {zz[p].x, zz[p].y, zz[p].z, zz[p].w}, a4[4] = {aa[p].x, aa[p].y, aa[p].z, aa[p].w};
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (oc + j < dout) {
            if (save_z) save_z[r * dout + oc + j] = z[j];
            y[r * dout + oc + j] = a4[j];
          }
      }
    }
  }
}

// ---- two Dense layers in one streaming launch, weights in registers ------------------------------------------------------------
// dense_pair_fwd_kernel: y_a = act_a([X | narrow_a] Wa + ba), y_b = act_b([X | narrow_b] Wb + bb) from ONE pass over the 64-wide
// block X -- the two node-level first-layer terms P, Q of the edge-function layers (/root/reference/src/layers.jl:409-410 split
// at node level: P from [h_i; d_i; theta], Q from [h_j; -d_j]).  dense_chain_fwd_kernel: y = act2(act1([X0 | X1 | narrow] W1 +
// b1) W2 + b2) with the 64-wide intermediate kept in LDS -- the node update psi of MPPDEConv / gamma of VMHConv (:418, :328).
// Both walk 128-row tiles with persistent workgroups of 8 waves as dense_stream64_fwd_kernel does (input tiles by LDS-DMA into
// XOR-swizzled images, narrow trailing features and bias in the epilogue), but wave w owns 16 OUTPUT COLUMNS (w % 4) of 64 rows
// (w / 4) and keeps its W^T fragments -- 16 registers per 64-deep input block -- for the whole launch: no weight copy in LDS, so
// two workgroups fit a CU beside 35-67 KB of tiles, and the only LDS reads of the products are the 16 A fragments per tile.
struct StreamOut {
  SegTable segs;   // the Dense's whole virtual vcat (leading blocks: the 64-wide ones; trailing: narrow features)
  int din_all, dout, act;
  const float *wt, *bias;
  float *y, *save_z;
};

__device__ __forceinline__ void load_wfrag(const float *__restrict__ wt, int dout, int k0, int col, int kq, f32x4 (&breg)[4]) {
#pragma unroll
  for (int kh = 0; kh < 4; ++kh)
#pragma unroll
    for (int r = 0; r < 4; ++r) breg[kh][r] = (col < dout) ? wt[(size_t)(k0 + 16 * kh + 4 * kq + r) * dout + col] : 0.f;
}

// TR x 64 tile memory -> LDS image (slot = chunk ^ (row & 15)), TR / 32 DMA instructions per wave
template <int TR>
__device__ __forceinline__ void dma_tile(const float *__restrict__ x, int64_t row0, int64_t n, float *img, int wave, int lane) {
  const int rl = lane >> 4, s16 = lane & 15;
#pragma unroll
  for (int j = 0; j < TR / 32; ++j) {
    const int r = (wave * (TR / 32) + j) * 4 + rl;
    const int64_t gr = min(row0 + r, n - 1);   // rows past the end read the last row (never stored)
    const float *g = x + gr * 64 + 4 * (s16 ^ (r & 15));
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)(reinterpret_cast<float4 *>(img) + ((wave * (TR / 32) + j) * 4) * 16 + lane),
                                     16, 0, 0);
  }
}

// acc[rt] += rows ((TR / 2) half + 16 rt ..) of a swizzled image (SWZ) or of a row-major tile of stride OS2 x the wave's fragments
template <int TR, bool SWZ>
__device__ __forceinline__ void mfma_rows_regs(const float *t, int half, int lane, const f32x4 (&breg)[4], f32x4 (&acc)[TR / 32]) {
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int rt = 0; rt < TR / 32; ++rt) {
    const int r = (TR / 2) * half + 16 * rt + i;
#pragma unroll
    for (int kh = 0; kh < 4; ++kh) {
      const float4 a4 = SWZ ? reinterpret_cast<const float4 *>(t)[r * 16 + ((4 * kh + kq) ^ (r & 15))]
                            : *reinterpret_cast<const float4 *>(&t[r * OS2 + 16 * kh + 4 * kq]);
      acc[rt] = mfma16(a4.x, breg[kh][0], acc[rt]);
      acc[rt] = mfma16(a4.y, breg[kh][1], acc[rt]);
      acc[rt] = mfma16(a4.z, breg[kh][2], acc[rt]);
      acc[rt] = mfma16(a4.w, breg[kh][3], acc[rt]);
    }
  }
}
// Products of a tile with the NEXT tile's image(s) on their way by LDS-DMA.  The compiler puts s_waitcnt vmcnt(0) in front of
// every LDS access it cannot prove disjoint from the destination of a global_load_lds in flight -- with both buffers carved
// from one dynamic LDS array that is every access: the products would wait for the very prefetch they are meant to cover (and
// for the previous tile's stores with it).  Issued inside a function whose image and destination pointers are __restrict__,
// the DMA and the reads carry scoped no-alias information and the wait is not emitted; the caller collects the DMA itself
// (s_waitcnt vmcnt(0), then a barrier) before anyone reads the destination.
__device__ __forceinline__ unsigned lds_byte_addr(const void *ptr) { return (unsigned)(uintptr_t)ptr; }   // low half of a generic LDS pointer
template <int TR, int NIMG, int NW>
__device__ __forceinline__ void products_beside_dma(const float *__restrict__ cur, float *__restrict__ nxt, int img_stride, bool has_next,
                                                    const float *x0, const float *x1, int64_t next_row0, int64_t n, int wave, int lane,
                                                    int half, const f32x4 (&w0)[4], const f32x4 (&w1)[4], f32x4 (&acc0)[TR / 32],
                                                    f32x4 (&acc1)[TR / 32]) {
  if (has_next) {
    dma_tile<TR>(x0, next_row0, n, nxt, wave, lane);
    if (NIMG == 2) dma_tile<TR>(x1, next_row0, n, nxt + img_stride, wave, lane);
  }
  if (NIMG == 1) {   // one image, NW weight sets -> NW outputs
    mfma_rows_regs<TR, true>(cur, half, lane, w0, acc0);
    if (NW == 2) mfma_rows_regs<TR, true>(cur, half, lane, w1, acc1);
  } else {           // two images, one weight set each -> one output
    mfma_rows_regs<TR, true>(cur, half, lane, w0, acc0);
    mfma_rows_regs<TR, true>(cur + img_stride, half, lane, w1, acc0);
  }
}
template <int TR>
__device__ __forceinline__ void stage_cols(float *tile, int half, int ct, int lane, const f32x4 (&acc)[TR / 32]) {
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int rt = 0; rt < TR / 32; ++rt)
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) tile[((TR / 2) * half + 16 * rt + 4 * kq + reg) * OS2 + 16 * ct + i] = acc[rt][reg];
}

// Narrow features (<= kNarrow per Dense) travel through two small LDS tables: wn[f][64] = their weight rows (once per
// workgroup), xn[row][kNarrowAll] = their values on the tile's rows (one or two loads per thread at the top of the tile, in
// flight under the products) -- the epilogue then has no global load of its own.
constexpr int kNarrow = 4, kNarrowAll = 8;
__device__ __forceinline__ void narrow_weights(const StreamOut &o, int kmain, float *wn, int tid) {
  for (int idx = tid; idx < kNarrow * 64; idx += kStreamThreads) {
    const int f = idx >> 6, c = idx & 63;
    wn[idx] = (kmain + f < o.din_all && c < o.dout) ? o.wt[(size_t)(kmain + f) * o.dout + c] : 0.f;
  }
}
// A thread's narrow feature value on a row: where the feature lives is resolved once (narrow_ref), then one load per tile.
// (Resolved per tile in the two arms of a branch -- features of Dense a in some lanes, of Dense b in others -- the two loads
// target the same register, and the compiler waits for the first to land before it issues the second: a whole memory latency at
// the top of every tile.)
struct NarrowRef {
  const float *ptr;   // column of the feature in its block (NULL: no such feature)
  int width, row_div;
  uint32_t magic;     // div_magic(row_div)
};
__device__ __forceinline__ NarrowRef narrow_ref(const StreamOut &o, int kmain, int f) {
  if (kmain + f >= o.din_all) return NarrowRef{nullptr, 0, 1, 0xFFFFFFFFu};
  const SegRef r = seg_find(o.segs, kmain + f);
  return NarrowRef{r.ptr + (kmain + f - r.offset), r.width, r.row_div, div_magic((uint32_t)r.row_div)};
}
__device__ __forceinline__ float narrow_fetch(const NarrowRef &q, int64_t r, int64_t n) {
  return (q.ptr && r < n) ? q.ptr[(int64_t)fast_div((uint32_t)r, (uint32_t)q.row_div, q.magic) * q.width] : 0.f;
}

// epilogue of one Dense on a staged TR x 64 tile: thread (row = tid / 16 + 32 p, columns 4 (tid % 16) .. + 3) adds bias and the
// narrow features (tables xn / wn), saves z, applies the activation, stores y; WRITEBACK: the activations replace the tile
// (input of a second layer)
template <int TR, bool WRITEBACK>
__device__ __forceinline__ void stream_epilogue(const StreamOut &o, int n_narrow, const float *xn, const float *wn, float *tile,
                                                int64_t row0, int64_t n, int tid, const float (&b)[4]) {
  const int oc = 4 * (tid & 15);
  const bool vec = (o.dout % 4 == 0) && ((reinterpret_cast<uintptr_t>(o.y) | reinterpret_cast<uintptr_t>(o.save_z)) & 15) == 0;
  float4 zz[TR / 32], aa[TR / 32];
#pragma unroll
  for (int p = 0; p < TR / 32; ++p) {
    const float4 v = *reinterpret_cast<const float4 *>(&tile[((tid >> 4) + 32 * p) * OS2 + oc]);
    zz[p] = make_float4(v.x + b[0], v.y + b[1], v.z + b[2], v.w + b[3]);
  }
  for (int f = 0; f < n_narrow; ++f) {
    const float4 w = *reinterpret_cast<const float4 *>(&wn[f * 64 + oc]);
#pragma unroll
    for (int p = 0; p < TR / 32; ++p) {
      const float xv = xn[((tid >> 4) + 32 * p) * kNarrowAll + f];
      zz[p] = make_float4(fmaf(xv, w.x, zz[p].x), fmaf(xv, w.y, zz[p].y), fmaf(xv, w.z, zz[p].z), fmaf(xv, w.w, zz[p].w));
    }
  }
#pragma unroll
  for (int p = 0; p < TR / 32; ++p) aa[p] = zz[p];
  f4n_act<TR / 32>(o.act, aa);
#pragma unroll
  for (int p = 0; p < TR / 32; ++p) {
    if (WRITEBACK) *reinterpret_cast<float4 *>(&tile[((tid >> 4) + 32 * p) * OS2 + oc]) = aa[p];
    const int64_t r = row0 + (tid >> 4) + 32 * p;
    if (r >= n || oc >= o.dout) continue;
    if (vec) {
      if (o.save_z) nt_store4(o.save_z + r * o.dout + oc, zz[p]);
      if (o.y) nt_store4(o.y + r * o.dout + oc, aa[p]);
    } else {
      const float z[4] = {zz[p].x, zz[p].y, zz[p].z, zz[p].w}, a4[4] = {aa[p].x, aa[p].y, aa[p].z, aa[p].w};
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (oc + j < o.dout) {
          if (o.save_z) o.save_z[r * o.dout + oc + j] = z[j];
          if (o.y) o.y[r * o.dout + oc + j] = a4[j];
        }
    }
  }
}

// Both kernels double-buffer the input images: the DMA of tile t + 1 is issued at the top of tile t into the buffer tile t - 1
// left behind, and collected at the top of tile t + 1 -- a whole tile of arithmetic later.
constexpr int kPairTR = 64;
#ifdef NGPDE_STAMPS
extern unsigned long long *g_pair_stamps;
unsigned long long *g_pair_stamps = nullptr;   // diagnostic build only (tools/stamps_pair.py): [n_blocks][16], the workgroup's 4th tile
#define PAIR_STAMP(k) do { if (threadIdx.x == 0 && stamps && it == 3) stamps[(size_t)blockIdx.x * 16 + (k)] = clock64(); } while (0)
#else
#define PAIR_STAMP(k)
#endif
__global__ __launch_bounds__(kStreamThreads, 4) void dense_pair_fwd_kernel(int64_t n, int n_tiles, const float *__restrict__ x,
                                                                           const StreamOut a, const StreamOut b, int dephase
#ifdef NGPDE_STAMPS
                                                                           , unsigned long long *stamps
#endif
                                                                           ) {
  constexpr int TR = kPairTR;
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  // dyn: two buffers [TR][OS2], each the X image of a tile, then each output on its way out
  float *wn = dyn + 2 * TR * OS2;                  // [2][kNarrow][64]
  float *xn = wn + 2 * kNarrow * 64;               // [TR][kNarrowAll]: features 0..3 of a, 4..7 of b
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3, half = wave >> 2;
  f32x4 wa[4], wb[4];
  load_wfrag(a.wt, a.dout, 0, 16 * ct + (lane & 15), lane >> 4, wa);
  load_wfrag(b.wt, b.dout, 0, 16 * ct + (lane & 15), lane >> 4, wb);
  narrow_weights(a, 64, wn, tid);
  narrow_weights(b, 64, wn + kNarrow * 64, tid);
  const int na = a.din_all - 64, nb = b.din_all - 64;
  const int oc = 4 * (tid & 15);
  float ba[4], bb[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    ba[j] = (a.bias && oc + j < a.dout) ? a.bias[oc + j] : 0.f;
    bb[j] = (b.bias && oc + j < b.dout) ? b.bias[oc + j] : 0.f;
  }
  const int nr = tid >> 3, nf = tid & 7;   // (64 rows x 8 narrow features, 0..3 of a and 4..7 of b: one value per thread)
  NarrowRef nq = narrow_ref(a, 64, nf & 3);
  {
    const NarrowRef nqb = narrow_ref(b, 64, nf & 3);
    if (nf >= 4) nq = nqb;
  }
  dephase_second_half(dephase);
  const int G = (int)gridDim.x;
  int t = blockIdx.x, it = 0;
  if (t < n_tiles) dma_tile<TR>(x, (int64_t)t * TR, n, dyn, wave, lane);
  for (; t < n_tiles; t += G, ++it) {
    const int64_t row0 = (int64_t)t * TR;
    float *cur = dyn + (it & 1) * (TR * OS2), *nxt = dyn + ((it + 1) & 1) * (TR * OS2);
    PAIR_STAMP(0);
    if (it == 0) wait_vmcnt0();
    __syncthreads();   // this tile's image has landed (collected below, a tile ago); the previous tile's outputs have left LDS
    PAIR_STAMP(1);
    // narrow features of this tile's rows, then the next tile's image
    const float xv = narrow_fetch(nq, row0 + nr, n);
    f32x4 acca[TR / 32], accb[TR / 32];
#pragma unroll
    for (int rt = 0; rt < TR / 32; ++rt) acca[rt] = accb[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    products_beside_dma<TR, 1, 2>(cur, nxt, 0, t + G < n_tiles, x, nullptr, (int64_t)(t + G) * TR, n, wave, lane, half, wa, wb, acca, accb);
    PAIR_STAMP(2);
    // collect the next tile's image HERE, behind the products and before this tile's stores are issued: a wait at the top of the
    // next tile would also wait for those stores (vmcnt counts them) -- a full store latency per tile
    wait_vmcnt0();
    PAIR_STAMP(3);
    xn[nr * kNarrowAll + nf] = xv;
    __syncthreads();
    PAIR_STAMP(4);
    stage_cols<TR>(cur, half, ct, lane, acca);
    __syncthreads();
    PAIR_STAMP(5);
    stream_epilogue<TR, false>(a, na, xn, wn, cur, row0, n, tid, ba);
    PAIR_STAMP(6);
    __syncthreads();
    stage_cols<TR>(cur, half, ct, lane, accb);
    __syncthreads();
    PAIR_STAMP(7);
    stream_epilogue<TR, false>(b, nb, xn + 4, wn + kNarrow * 64, cur, row0, n, tid, bb);
    PAIR_STAMP(8);
  }
}

// l1.y (the activations a1) and l1.save_z may be NULL (inference); l1.dout == 64.  64-row tiles (128-row ones spill at the 128
// registers two workgroups per CU leave a wave).
template <int NIN>
__global__ __launch_bounds__(kStreamThreads, 4) void dense_chain_fwd_kernel(int64_t n, int n_tiles, const float *__restrict__ x0,
                                                                            const float *__restrict__ x1, const StreamOut l1,
                                                                            const StreamOut l2, int dephase) {
  constexpr int TR = 64;
  constexpr int kBuf = TR * OS2 + (NIN == 2 ? TR * 64 : 0);   // [TR][OS2]: X0 image, then z1 / a1, then y; [TR][64]: X1 image
  extern __shared__ __attribute__((aligned(16))) float dyn[];
  float *wn = dyn + 2 * kBuf;                      // [kNarrow][64]
  float *xn = wn + kNarrow * 64;                   // [TR][kNarrowAll]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave & 3, half = wave >> 2;
  f32x4 w1[NIN][4], w2[4];
#pragma unroll
  for (int bI = 0; bI < NIN; ++bI) load_wfrag(l1.wt, 64, 64 * bI, 16 * ct + (lane & 15), lane >> 4, w1[bI]);
  load_wfrag(l2.wt, l2.dout, 0, 16 * ct + (lane & 15), lane >> 4, w2);
  narrow_weights(l1, 64 * NIN, wn, tid);
  const int nn = l1.din_all - 64 * NIN;
  const int oc = 4 * (tid & 15);
  float *bl = xn + TR * kNarrowAll;                // [2][64]: the two bias vectors (kept out of the registers)
  if (tid < 128) {
    const int c = tid & 63;
    bl[tid] = tid < 64 ? (l1.bias ? l1.bias[c] : 0.f) : ((l2.bias && c < l2.dout) ? l2.bias[c] : 0.f);
  }
  auto dma_in = [&](int tile, float *bufp) {
    dma_tile<TR>(x0, (int64_t)tile * TR, n, bufp, wave, lane);
    if (NIN == 2) dma_tile<TR>(x1, (int64_t)tile * TR, n, bufp + TR * OS2, wave, lane);
  };
  // narrow features of a tile's rows: thread -> (row tid / 8, feature tid % 8 < 4), resolved once
  const int nr = tid >> 3, nf = tid & 7;
  const NarrowRef nq = nf < kNarrow ? narrow_ref(l1, 64 * NIN, nf) : NarrowRef{nullptr, 0, 1, 0xFFFFFFFFu};
  dephase_second_half(dephase);
  const int G = (int)gridDim.x;
  int t = blockIdx.x, it = 0;
  if (t < n_tiles) dma_in(t, dyn);
  for (; t < n_tiles; t += G, ++it) {
    const int64_t row0 = (int64_t)t * TR;
    float *cur = dyn + (it & 1) * kBuf, *nxt = dyn + ((it + 1) & 1) * kBuf;
    if (it == 0) wait_vmcnt0();
    __syncthreads();
    const float xv = narrow_fetch(nq, row0 + nr, n);
    f32x4 acc[TR / 32];
#pragma unroll
    for (int rt = 0; rt < TR / 32; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    products_beside_dma<TR, NIN, 1>(cur, nxt, TR * OS2, t + G < n_tiles, x0, x1, (int64_t)(t + G) * TR, n, wave, lane, half,
                                    w1[0], w1[NIN - 1], acc, acc);
    wait_vmcnt0();   // the next tile's images, before this tile's stores (see dense_pair_fwd_kernel)
    if (nf < kNarrow) xn[nr * kNarrowAll + nf] = xv;
    __syncthreads();
    stage_cols<TR>(cur, half, ct, lane, acc);
    __syncthreads();
    {
      const float4 bv = *reinterpret_cast<const float4 *>(&bl[oc]);
      const float b1[4] = {bv.x, bv.y, bv.z, bv.w};
      stream_epilogue<TR, true>(l1, nn, xn, wn, cur, row0, n, tid, b1);   // z1 -> a1 in place (+ saves)
    }
    __syncthreads();
#pragma unroll
    for (int rt = 0; rt < TR / 32; ++rt) acc[rt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    mfma_rows_regs<TR, false>(cur, half, lane, w2, acc);
    __syncthreads();
    stage_cols<TR>(cur, half, ct, lane, acc);
    __syncthreads();
    {
      const float4 bv = *reinterpret_cast<const float4 *>(&bl[64 + oc]);
      const float b2[4] = {bv.x, bv.y, bv.z, bv.w};
      stream_epilogue<TR, false>(l2, 0, xn, wn, cur, row0, n, tid, b2);
    }
  }
}

// (oper, part, part_stride: the same split over the output features as in dense_mfma_bwd_input_kernel; oper = dout_all: none)
__global__ __launch_bounds__(256) void dense_wide_bwd_input_kernel(int64_t n, SegGrad segs, int din, int dout_all,
                                                                   const float *__restrict__ dz,
                                                                   const float *__restrict__ wt, int oper, float *part,
                                                                   size_t part_stride) {
  __shared__ __attribute__((aligned(16))) float lds[kWideLds];
  float *ldsA = lds, *ldsBt = lds + BM2 * LS2;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * BM2;
  const int col0 = blockIdx.y * BN;          // columns of dX = input features k
  const int obeg = blockIdx.z * oper, dout = min(dout_all, obeg + oper);   // this workgroup's output-feature range [obeg, dout)
  if (blockIdx.z > 0) segs.ptr[0] = part + (size_t)(blockIdx.z - 1) * part_stride;
  const int ar = tid >> 3, ak = 4 * (tid & 7);       // A = dz: float4 (row, o)
  const int bcol = tid >> 2, bo = 8 * (tid & 3);     // Bt[col = k][o]: 8 consecutive o of one weight row
  const bool avec = (dout_all % 4 == 0) && ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(wt)) & 15) == 0;
  float4 areg[4], breg[2];
  auto fetch = [&](int o0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t r = row0 + ar + 32 * p;
      const int o = o0 + ak;
      if (r < n && avec && o + 4 <= dout) {
        areg[p] = *reinterpret_cast<const float4 *>(dz + r * dout_all + o);
      } else {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (r < n && o + j < dout) ? dz[r * dout_all + o + j] : 0.f;
        areg[p] = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
    const int k = col0 + bcol;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int o = o0 + bo + 4 * h;
      if (k < din && avec && o + 4 <= dout) {
        breg[h] = *reinterpret_cast<const float4 *>(wt + (size_t)k * dout_all + o);
      } else {
        float t[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) t[j] = (k < din && o + j < dout) ? wt[(size_t)k * dout_all + o + j] : 0.f;
        breg[h] = make_float4(t[0], t[1], t[2], t[3]);
      }
    }
  };
  f32x4 acc[2][4];
#pragma unroll
  for (int rt = 0; rt < 2; ++rt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[rt][ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  fetch(obeg);
  for (int o0 = obeg; o0 < dout; o0 += BK2) {
    __syncthreads();
#pragma unroll
    for (int p = 0; p < 4; ++p) *reinterpret_cast<float4 *>(&ldsA[(ar + 32 * p) * LS2 + ak]) = areg[p];
    *reinterpret_cast<float4 *>(&ldsBt[bcol * LS2 + bo]) = breg[0];
    *reinterpret_cast<float4 *>(&ldsBt[bcol * LS2 + bo + 4]) = breg[1];
    __syncthreads();
    if (o0 + BK2 < dout) fetch(o0 + BK2);
    mfma_step_wide(ldsA, ldsBt, wave, lane, acc);
  }
  __syncthreads();
  acc_to_lds(lds, wave, lane, acc);
  __syncthreads();
  // epilogue: thread (row = tid / 16 + 16 p, features col0 + 4 (tid % 16) .. + 3) -> the block that owns each feature
  const int oc = 4 * (tid & 15);
  const int k = col0 + oc;
  if (k >= din) return;
  int sg = -1;
#pragma unroll
  for (int q = 0; q < 4; ++q)
    if (q < segs.n && k >= segs.offset[q] && k < segs.offset[q + 1]) sg = q;
  const bool whole = sg >= 0 && k + 4 <= segs.offset[sg + 1] && (segs.offset[sg] % 4 == 0) && (segs.width[sg] % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(segs.ptr[sg]) & 15) == 0);
#pragma unroll
  for (int p = 0; p < 8; ++p) {
    const int rl = (tid >> 4) + 16 * p;
    const int64_t r = row0 + rl;
    if (r >= n) continue;
    const float4 v = *reinterpret_cast<const float4 *>(&lds[rl * OS2 + oc]);
    if (whole) {
      if (segs.ptr[sg]) *reinterpret_cast<float4 *>(segs.ptr[sg] + r * segs.width[sg] + (k - segs.offset[sg])) = v;
    } else {
      const float t[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int kj = k + j;
        if (kj >= din) continue;
        int s2 = -1;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          if (q < segs.n && kj >= segs.offset[q] && kj < segs.offset[q + 1]) s2 = q;
        if (s2 >= 0 && segs.ptr[s2]) segs.ptr[s2][r * segs.width[s2] + (kj - segs.offset[s2])] = t[j];
      }
    }
  }
}

// ---- weight pullback: partial[chunk][k][o] = sum_{rows of chunk} X[row][k] dz[row][o]; row k == din of `partial`
// holds the bias gradient sum_rows dz[row][o], accumulated by the blockIdx.x == 0 tiles from the staged dz chunk
__global__ __launch_bounds__(256) void dense_mfma_bwd_weight_kernel(int64_t n, SegTable segs, int din, int dout,
                                                                    const float *__restrict__ dz, int64_t rows_per_chunk,
                                                                    float *__restrict__ partial) {
  __shared__ __attribute__((aligned(16))) float ldsA[BM * LS], ldsBt[BN * LS];   // At[k][nn], Bt[o][nn]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int k0 = blockIdx.x * BM;            // rows of dWt (input features, + bias row)
  const int col0 = blockIdx.y * BN;          // cols of dWt (outputs)
  const int64_t r0 = (int64_t)blockIdx.z * rows_per_chunk, r1 = min(n, r0 + rows_per_chunk);
  // staging: thread (sn = tid / 64, c = tid % 64) carries rows 4 sn .. 4 sn + 3 of the 16-row step for column c of the
  // X block and of the dz block -> ONE b128 LDS store per operand into the transposed tiles
  const int sn = tid >> 6, scol = tid & 63;
  // the X column of this thread never changes: resolve its block of the segmented input once
  const float *abase = nullptr;
  int awidth = 0, adiv = 1;
  {
    const int k = k0 + scol;
#pragma unroll
    for (int i = 3; i >= 0; --i)
      if (i < segs.n && k < segs.offset[i + 1] && k >= segs.offset[i] && k < din) {
        abase = segs.ptr[i] + (k - segs.offset[i]);
        awidth = segs.width[i];
        adiv = segs.row_div[i];
      }
  }
  const float *bbase = (col0 + scol < dout) ? dz + col0 + scol : nullptr;
  float areg[2][4], breg[2][4];
  auto fetch = [&](int64_t rr0, float (&ar)[4], float (&br)[4]) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t r = rr0 + 4 * sn + p;
      ar[p] = (r < r1 && abase) ? abase[seg_row(r, adiv) * awidth] : 0.f;
      br[p] = (r < r1 && bbase) ? bbase[r * dout] : 0.f;
    }
  };
  f32x4 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float bsum = 0.f;
  auto step = [&](float (&ar)[4], float (&br)[4], int64_t next) {
    __syncthreads();
    *reinterpret_cast<float4 *>(&ldsA[scol * LS + 4 * sn]) = make_float4(ar[0], ar[1], ar[2], ar[3]);
    *reinterpret_cast<float4 *>(&ldsBt[scol * LS + 4 * sn]) = make_float4(br[0], br[1], br[2], br[3]);
    __syncthreads();
    fetch(next, ar, br);                 // two steps ahead: in flight across this step's and the next step's MFMAs
    if (blockIdx.x == 0 && tid < BN) {   // bias row: column sums of the staged dz chunk (one wave, 4 b128 LDS reads)
#pragma unroll
      for (int nn = 0; nn < BK; nn += 4) {
        const float4 v = *reinterpret_cast<const float4 *>(&ldsBt[tid * LS + nn]);
        bsum += (v.x + v.y) + (v.z + v.w);
      }
    }
    mfma_chunk(ldsA, ldsBt, wave, lane, acc);
  };
  fetch(r0, areg[0], breg[0]);
  fetch(r0 + BK, areg[1], breg[1]);
  for (int64_t rr = r0; rr < r1; rr += 2 * BK) {   // rows past r1 stage zeros: harmless
    step(areg[0], breg[0], rr + 2 * BK);
    if (rr + BK < r1) step(areg[1], breg[1], rr + 3 * BK);
  }
  const int i = lane & 15, kq = lane >> 4;
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int o = col0 + ct * 16 + i;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int k = k0 + wave * 16 + 4 * kq + reg;
      if (k < din && o < dout) partial[((size_t)blockIdx.z * (din + 1) + k) * dout + o] = acc[ct][reg];
    }
  }
  if (blockIdx.x == 0 && tid < BN && col0 + tid < dout)
    partial[((size_t)blockIdx.z * (din + 1) + din) * dout + col0 + tid] = bsum;
}

// dwt / db = sum over the chunk slabs, in a fixed order: 64 elements x 4 chunk lanes per workgroup
__global__ __launch_bounds__(256) void dense_weight_reduce_kernel(int nchunk, int din, int dout, const float *__restrict__ partial,
                                                                  float *__restrict__ dwt, float *__restrict__ db) {
  __shared__ float part[4][64];
  const int e = threadIdx.x & 63, cl = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + e;
  const int total = (din + 1) * dout;
  // eight independent partial sums per thread: the loop is a chain of L2 round trips (a slab element is touched once), so the
  // loads in flight per round, not the adds, set its length (282 slabs: 9 rounds instead of 18)
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
  if (idx < total) {
    int c = cl;
    for (; c + 28 < nchunk; c += 32) {
      const float v0 = partial[(size_t)c * total + idx], v1 = partial[(size_t)(c + 4) * total + idx];
      const float v2 = partial[(size_t)(c + 8) * total + idx], v3 = partial[(size_t)(c + 12) * total + idx];
      const float v4 = partial[(size_t)(c + 16) * total + idx], v5 = partial[(size_t)(c + 20) * total + idx];
      const float v6 = partial[(size_t)(c + 24) * total + idx], v7 = partial[(size_t)(c + 28) * total + idx];
      s0 += v0; s1 += v1; s2 += v2; s3 += v3; s4 += v4; s5 += v5; s6 += v6; s7 += v7;
    }
    for (; c + 12 < nchunk; c += 16) {
      s0 += partial[(size_t)c * total + idx];
      s1 += partial[(size_t)(c + 4) * total + idx];
      s2 += partial[(size_t)(c + 8) * total + idx];
      s3 += partial[(size_t)(c + 12) * total + idx];
    }
    for (; c < nchunk; c += 4) s0 += partial[(size_t)c * total + idx];
  }
  part[cl][e] = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
  __syncthreads();
  if (cl == 0 && idx < total) {
    const float s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    if (idx < din * dout) dwt[idx] = s;
    else if (db) db[idx - din * dout] = s;
  }
}

}  // namespace

// 128-row tiles once they alone give every CU two workgroups
static bool use_wide_tiles(int64_t n, int cols) {
  static const bool off = getenv("NGPDE_DENSE_NARROW") != nullptr;
  return !off && ((n + BM2 - 1) / BM2) * ((cols + BN - 1) / BN) >= 512;
}

int32_t launch_dense_seg_fwd(int64_t n, const SegTable &segs, int din, int dout, int act, const float *wt,
                             const float *bias, float *y, float *save_z, hipStream_t stream) {
  if (n == 0 || dout == 0) return NGPDE_OK;
  if (const int sgrid = dense_small_fwd_grid(n, din, dout))   // 17 .. 64 inputs, at most 64 outputs, latency-bound row counts: one contraction pass
    return launch_dense_small_fwd(n, segs, din, dout, act, wt, bias, y, save_z, sgrid, stream);
  {   // wide outputs from one 16-byte-loadable block: 128 x 128 tiles
    static const bool no_gemm = getenv("NGPDE_DENSE_NO_GEMM128") != nullptr;
    const int64_t tiles = ((n + BG - 1) / BG) * ((dout + BG - 1) / BG);
    if (!no_gemm && segs.n == 1 && segs.vec[0] && segs.row_div[0] == 1 && din % BKG == 0 && dout % 4 == 0 && dout >= BG && tiles >= 512 &&
        ((reinterpret_cast<uintptr_t>(wt) | reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(save_z)) & 15) == 0) {
      hipLaunchKernelGGL(dense_gemm128_fwd_kernel, dim3((unsigned)((n + BG - 1) / BG), (dout + BG - 1) / BG), dim3(256), 0, stream, n,
                         segs.ptr[0], din, dout, act, wt, bias, y, save_z);
      NGPDE_LAUNCH_CHECK("dense_gemm128_fwd_kernel");
      return NGPDE_OK;
    }
  }
  if (use_wide_tiles(n, dout)) {
    // trailing narrow blocks (<= 8 features in total behind a multiple of 32) leave the K loop: see dense_wide_fwd_kernel
    int din_main = din;
    for (int i = segs.n - 1; i >= 1 && din - segs.offset[i] <= 8; --i)
      if (segs.offset[i] % BK2 == 0) din_main = segs.offset[i];
    // a 64-deep contraction over 16-byte-loadable blocks: the streaming form (whole input tile by LDS-DMA in one burst)
    static const bool no_stream = getenv("NGPDE_DENSE_NO_STREAM") != nullptr;
    bool stream_ok = !no_stream && din_main == 64 && dout <= 64;
    for (int i = 0; i < segs.n && segs.offset[i] < din_main; ++i) stream_ok = stream_ok && segs.vec[i] && segs.offset[i + 1] <= din_main;
    if (stream_ok) {
      static int cus = 0, per_cu = 0;
      if (cus == 0) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
        (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dense_stream64_fwd_kernel, kStreamThreads, 0);
        cus = std::max(cus, 1); per_cu = std::max(per_cu, 1);
      }
      const int n_tiles = (int)((n + BM2 - 1) / BM2);
      hipLaunchKernelGGL(dense_stream64_fwd_kernel, dim3((unsigned)std::min(n_tiles, cus * per_cu)), dim3(kStreamThreads), 0, stream,
                         n, segs, din, dout, act, wt, bias, y, save_z, n_tiles);
      NGPDE_LAUNCH_CHECK("dense_stream64_fwd_kernel");
      return NGPDE_OK;
    }
    hipLaunchKernelGGL(dense_wide_fwd_kernel, dim3((unsigned)((n + BM2 - 1) / BM2), (dout + BN - 1) / BN), dim3(256), 0, stream,
                       n, segs, din, dout, act, wt, bias, y, save_z, din_main);
    NGPDE_LAUNCH_CHECK("dense_wide_fwd_kernel");
    return NGPDE_OK;
  }
  hipLaunchKernelGGL(dense_mfma_fwd_kernel, dim3((unsigned)((n + BM - 1) / BM), (dout + BN - 1) / BN), dim3(256), 0, stream,
                     n, segs, din, dout, act, wt, bias, y, save_z, din, (size_t)0);
  NGPDE_LAUNCH_CHECK("dense_mfma_fwd_kernel");
  return NGPDE_OK;
}

// count <= 4 independent problems in ONE launch of the 64 x 64-tile kernel (see dense_mfma_multi_fwd_kernel)
int32_t launch_dense_multi_fwd(int count, const int64_t *n, const SegTable *segs, const int *din, const int *dout, const int *act,
                               const float *const *wt, const float *const *bias, float *const *y, float *const *save_z, hipStream_t stream) {
  MultiFwdK mk;
  mk.count = 0;
  int tiles = 0;
  for (int q = 0; q < count; ++q) {
    if (n[q] == 0 || dout[q] == 0) continue;
    const int k = mk.count++;
    mk.tile0[k] = tiles;
    mk.col_tiles[k] = (dout[q] + BN - 1) / BN;
    tiles += (int)((n[q] + BM - 1) / BM) * mk.col_tiles[k];
    mk.p[k].n = n[q]; mk.p[k].segs = segs[q]; mk.p[k].din = din[q]; mk.p[k].dout = dout[q]; mk.p[k].act = act[q];
    mk.p[k].wt = wt[q]; mk.p[k].bias = bias[q]; mk.p[k].y = y[q]; mk.p[k].save_z = save_z[q];
  }
  if (mk.count == 0) return NGPDE_OK;
  mk.tile0[mk.count] = tiles;
  hipLaunchKernelGGL(dense_mfma_multi_fwd_kernel, dim3((unsigned)tiles), dim3(256), 0, stream, mk);
  NGPDE_LAUNCH_CHECK("dense_mfma_multi_fwd_kernel");
  return NGPDE_OK;
}

// number of K splits worth taking for X[n][din] x W[din][dout]: only when the row tiles alone leave most of the chip idle and
// the contraction is long; every split gets a multiple of BK features
int dense_fwd_splits(int64_t n, int din, int dout) {
  const int64_t tiles = ((n + BM - 1) / BM) * ((dout + BN - 1) / BN);
  if (tiles >= 256 || din < 256) return 1;
  const int64_t want = (512 + tiles - 1) / tiles;
  return (int)std::max<int64_t>(1, std::min<int64_t>(want, din / 64));
}

// partial[z][n][dout] = X[:, range z] x W[range z, :], z < nsplit (nsplit from dense_fwd_splits)
int32_t launch_dense_seg_fwd_splitk(int64_t n, const SegTable &segs, int din, int dout, const float *wt, float *partial, int nsplit,
                                    hipStream_t stream) {
  if (n == 0 || dout == 0) return NGPDE_OK;
  const int kper = ((din + nsplit - 1) / nsplit + BK - 1) / BK * BK;
  hipLaunchKernelGGL(dense_mfma_fwd_kernel, dim3((unsigned)((n + BM - 1) / BM), (dout + BN - 1) / BN, (din + kper - 1) / kper),
                     dim3(256), 0, stream, n, segs, din, dout, NGPDE_ACT_IDENTITY, wt, nullptr, partial, nullptr, kper,
                     (size_t)n * dout);
  NGPDE_LAUNCH_CHECK("dense_mfma_fwd_kernel (split-K)");
  return NGPDE_OK;
}
int dense_fwd_split_count(int din, int nsplit) {   // partial slabs launch_dense_seg_fwd_splitk actually writes
  const int kper = ((din + nsplit - 1) / nsplit + BK - 1) / BK * BK;
  return (din + kper - 1) / kper;
}

// C [M][N] = A [M][K] x B [K][N] (both row-major) on the 128 x 128 tiles, the contraction split into nsplit ranges of whole 16-chunks:
// range z writes slab z of `slabs` ([nsplit + n_side][M][ldc], slab_stride floats apart); up to two further short products
// side_a[e] [M][side_k[e]] x side_b[e] [side_k[e]][N] (row-major, leading dimensions side_k[e] and N) write slabs nsplit + e in the
// same launch; the caller sums the slabs in order.  For a node-level product with a long contraction and few row tiles
// (gno_gform.hip: 4096 x 8192 x 128).  K and side_k multiples of 16, N % 4 == 0.
int32_t launch_gemm128_split_nn(int M, int N, int K, int nsplit, const float *A, int lda, const float *B, int ldb, float *slabs,
                                size_t slab_stride, int ldc, int n_side, const float *const *side_a, const float *const *side_b, const int *side_k,
                                hipStream_t stream) {
  if (M == 0 || N == 0) return NGPDE_OK;
  const int kper = ((K + nsplit - 1) / nsplit + BKG - 1) / BKG * BKG;
  NGPDE_REQUIRE(K % BKG == 0 && N % 4 == 0 && (size_t)kper * nsplit >= (size_t)K && n_side >= 0 && n_side <= 2, NGPDE_ERR_DIMENSION_MISMATCH,
                "gemm128 split: K = %d must be a multiple of %d and N = %d of 4", K, BKG, N);
  GemmSide side{};
  side.n = n_side;
  for (int e = 0; e < n_side; ++e) {
    NGPDE_REQUIRE(side_k[e] % BKG == 0 && side_a[e] && side_b[e], NGPDE_ERR_DIMENSION_MISMATCH, "gemm128 split: side product %d: K = %d", e, side_k[e]);
    side.A[e] = side_a[e]; side.B[e] = side_b[e]; side.lda[e] = side_k[e]; side.ldb[e] = N; side.K[e] = side_k[e];
  }
  hipLaunchKernelGGL(dense_gemm128_split_nn_kernel, dim3((M + BG - 1) / BG, (N + BG - 1) / BG, nsplit + n_side), dim3(256), 0, stream, M, N, K, kper,
                     nsplit, A, lda, B, ldb, slabs, slab_stride, ldc, side);
  NGPDE_LAUNCH_CHECK("dense_gemm128_split_nn_kernel");
  return NGPDE_OK;
}

int32_t launch_dense_seg_bwd_input(int64_t n, const SegGrad &segs, int din, int dout, const float *dz, const float *wt,
                                   hipStream_t stream) {
  if (n == 0 || din == 0) return NGPDE_OK;
  if (use_wide_tiles(n, din)) {
    hipLaunchKernelGGL(dense_wide_bwd_input_kernel, dim3((unsigned)((n + BM2 - 1) / BM2), (din + BN - 1) / BN), dim3(256), 0,
                       stream, n, segs, din, dout, dz, wt, dout, (float *)nullptr, (size_t)0);
    NGPDE_LAUNCH_CHECK("dense_wide_bwd_input_kernel");
    return NGPDE_OK;
  }
  hipLaunchKernelGGL(dense_mfma_bwd_input_kernel, dim3((unsigned)((n + BM - 1) / BM), (din + BN - 1) / BN), dim3(256), 0,
                     stream, n, segs, din, dout, dz, wt, dout, (float *)nullptr, (size_t)0);
  NGPDE_LAUNCH_CHECK("dense_mfma_bwd_input_kernel");
  return NGPDE_OK;
}

namespace {
// out[i] += parts[i] + parts[stride + i] + ... (nparts slabs, in slab order)
__global__ void add_partials_kernel(int64_t count, int nparts, size_t stride, const float *__restrict__ parts, float *__restrict__ out) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x) {
    float s0 = out[i], s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int z = 0;
    for (; z + 3 <= nparts; z += 3) {
      s1 += parts[(size_t)z * stride + i];
      s2 += parts[(size_t)(z + 1) * stride + i];
      s3 += parts[(size_t)(z + 2) * stride + i];
    }
    for (; z < nparts; ++z) s1 += parts[(size_t)z * stride + i];
    out[i] = (s0 + s1) + (s2 + s3);
  }
}
}  // namespace

// splits of the input pullback over the output features (see dense_mfma_bwd_input_kernel): 1 = none
int dense_bwd_input_splits(int64_t n, int din, int dout) {
  const int64_t tiles = ((n + BM2 - 1) / BM2) * ((din + BN - 1) / BN);
  if (tiles >= 256 || dout < 512) return 1;
  // enough workgroups for a few rounds of the chip, each still with a dozen or more K steps (the partial slabs cost a pass too)
  return (int)std::max<int64_t>(1, std::min<int64_t>((2048 + tiles - 1) / tiles, dout / 256));
}
size_t dense_bwd_input_split_bytes(int64_t n, int din, int dout) {
  const int ns = dense_bwd_input_splits(n, din, dout);
  return ns > 1 ? (size_t)(ns - 1) * (size_t)n * din * sizeof(float) : 0;
}
// single-block input pullback, split over the output features; `part` holds dense_bwd_input_split_bytes
int32_t launch_dense_bwd_input_splitk(int64_t n, float *dx, int din, int dout, const float *dz, const float *wt, float *part,
                                      hipStream_t stream) {
  if (n == 0 || din == 0) return NGPDE_OK;
  const int ns = dense_bwd_input_splits(n, din, dout);
  const int oper = ((dout + ns - 1) / ns + BK2 - 1) / BK2 * BK2;
  const int nz = (dout + oper - 1) / oper;
  {   // 128 x 128 tiles (GNOConv's T = W2 (x) h: 4096 x 128 <= 8192)
    static const bool no_gemm = getenv("NGPDE_DENSE_NO_GEMM128") != nullptr;
    if (!no_gemm && nz > 1 && din >= BG && din % 4 == 0 && dout % BKG == 0 && oper % BKG == 0 && n < (1 << 30) &&
        ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(wt) | reinterpret_cast<uintptr_t>(dx) | reinterpret_cast<uintptr_t>(part)) & 15) == 0) {
      hipLaunchKernelGGL((dense_gemm128_split_kernel<false, true>), dim3((unsigned)((n + BG - 1) / BG), (din + BG - 1) / BG, nz), dim3(256), 0,
                         stream, (int)n, din, dout, oper, dz, dout, wt, dout, dx, part, (size_t)n * din, din);
      NGPDE_LAUNCH_CHECK("dense_gemm128_split_kernel (input pullback)");
      const int64_t count = n * din;
      hipLaunchKernelGGL(add_partials_kernel, dim3((unsigned)std::min<int64_t>((count + 255) / 256, 4096)), dim3(256), 0, stream, count,
                         nz - 1, (size_t)n * din, part, dx);
      NGPDE_LAUNCH_CHECK("add_partials_kernel");
      return NGPDE_OK;
    }
  }
  SegGrad segs;
  segs.n = 1; segs.ptr[0] = dx; segs.width[0] = din;
  for (int i = 1; i <= 4; ++i) segs.offset[i] = din;
  // the 128-row tile kernel: 16-byte loads of dz and W rows (the 64-row kernel reads dz in 64-byte pieces)
  hipLaunchKernelGGL(dense_wide_bwd_input_kernel, dim3((unsigned)((n + BM2 - 1) / BM2), (din + BN - 1) / BN, nz), dim3(256), 0, stream,
                     n, segs, din, dout, dz, wt, oper, part, (size_t)n * din);
  NGPDE_LAUNCH_CHECK("dense_mfma_bwd_input_kernel (split)");
  if (nz > 1) {
    const int64_t count = n * din;
    hipLaunchKernelGGL(add_partials_kernel, dim3((unsigned)std::min<int64_t>((count + 255) / 256, 4096)), dim3(256), 0, stream, count,
                       nz - 1, (size_t)n * din, part, dx);
    NGPDE_LAUNCH_CHECK("add_partials_kernel");
  }
  return NGPDE_OK;
}

// row chunks of the weight pullback: enough (tile x chunk) workgroups to cover the chip several times over (the chunk
// loop is a dependent global-load chain, ~16 rows per trip), at least 64 rows per chunk, at most 1024 partial slabs
int32_t launch_dense_weight_reduce(int nchunk, int din, int dout, const float *partial, float *dwt, float *db, hipStream_t stream) {
  const int total = (din + 1) * dout;
  hipLaunchKernelGGL(dense_weight_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, stream, nchunk, din, dout, partial, dwt, db);
  NGPDE_LAUNCH_CHECK("dense_weight_reduce_kernel");
  return NGPDE_OK;
}

int dense_weight_chunks(int64_t n, int din, int dout) {
  const int64_t tiles = (int64_t)std::max(1, (din + BM - 1) / BM) * std::max(1, (dout + BN - 1) / BN);
  const int64_t want = (4096 + tiles - 1) / tiles;
  return (int)std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(1024, want), (n + 63) / 64));
}

int32_t launch_dense_seg_bwd_weight(int64_t n, const SegTable &segs, int din, int dout, const float *dz, float *dwt,
                                    float *db, float *partial, hipStream_t stream) {
  if (dout == 0) return NGPDE_OK;
  const int nchunk = dense_weight_chunks(n, din, dout);
  const int64_t rpc = std::max<int64_t>(BK, (((n + nchunk - 1) / nchunk) + BK - 1) / BK * BK);
  {   // 128 x 128 tiles for a wide layer without bias gradient (GNOConv's T): slabs in the layout dense_weight_reduce_kernel sums
    static const bool no_gemm = getenv("NGPDE_DENSE_NO_GEMM128") != nullptr;
    if (!no_gemm && db == nullptr && nchunk > 1 && segs.n == 1 && segs.vec[0] && segs.row_div[0] == 1 && din >= BG && din % 4 == 0 &&
        dout >= BG && dout % 4 == 0 && n % BKG == 0 && n < (1 << 30) &&
        ((reinterpret_cast<uintptr_t>(dz) | reinterpret_cast<uintptr_t>(partial)) & 15) == 0) {
      const size_t slab = (size_t)(din + 1) * dout;
      hipLaunchKernelGGL((dense_gemm128_split_kernel<true, false>), dim3((din + BG - 1) / BG, (dout + BG - 1) / BG, nchunk), dim3(256), 0, stream,
                         din, dout, (int)n, (int)rpc, segs.ptr[0], din, dz, dout, partial, partial + slab, slab, dout);
      NGPDE_LAUNCH_CHECK("dense_gemm128_split_kernel (weight pullback)");
      return launch_dense_weight_reduce(nchunk, din, dout, partial, dwt, db, stream);
    }
  }
  hipLaunchKernelGGL(dense_mfma_bwd_weight_kernel, dim3(std::max(1, (din + BM - 1) / BM), (dout + BN - 1) / BN, nchunk), dim3(256), 0,
                     stream, n, segs, din, dout, dz, rpc, partial);
  NGPDE_LAUNCH_CHECK("dense_mfma_bwd_weight_kernel");
  const int total = (din + 1) * dout;
  hipLaunchKernelGGL(dense_weight_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, stream, nchunk, din, dout, partial,
                     dwt, db);
  NGPDE_LAUNCH_CHECK("dense_weight_reduce_kernel");
  return NGPDE_OK;
}


// ---- launches of the two-layer streaming forwards ----------------------------------------------------------------------------------
static int stream_grid(const void *kernel, size_t dyn_lds, int n_tiles) {
  int dev = 0, cus = 0, per_cu = 0;
  (void)hipGetDevice(&dev);
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, kStreamThreads, dyn_lds);
  return std::min(n_tiles, std::max(cus, 1) * std::max(per_cu, 1));
}
// leading blocks of exactly 64 features that LDS-DMA can fetch (row_div 1, 16-byte aligned); the rest must be narrow
static int stream_main_blocks(const SegTable &t, int din) {
  int nm = 0;
  while (nm < t.n && t.width[nm] == 64 && t.row_div[nm] == 1 && t.vec[nm]) ++nm;
  return (nm >= 1 && din - 64 * nm <= kNarrow) ? nm : 0;
}
static bool stream2_enabled(int64_t n) {
  const char *e = getenv("NGPDE_DENSE_NO_STREAM2");   // read per call: the tests switch it at run time
  return !(e && e[0] == '1') && (n + BM2 - 1) / BM2 >= 512;
}

static int dephase_cycles() {
  static const int v = [] { const char *e = std::getenv("NGPDE_DENSE_DEPHASE"); return e ? std::atoi(e) : 0; }();
  return v;
}

bool dense_pair_fwd_applicable(int64_t n, const SegTable &ta, int dina, int douta, const SegTable &tb, int dinb, int doutb) {
  return stream2_enabled(n) && douta <= 64 && doutb <= 64 && stream_main_blocks(ta, dina) == 1 && stream_main_blocks(tb, dinb) == 1 &&
         ta.ptr[0] == tb.ptr[0];
}
int32_t launch_dense_pair_fwd(int64_t n, const SegTable &ta, int dina, int douta, int acta, const float *wta, const float *ba, float *ya,
                              float *za, const SegTable &tb, int dinb, int doutb, int actb, const float *wtb, const float *bb, float *yb,
                              float *zb, hipStream_t stream) {
  StreamOut a{ta, dina, douta, acta, wta, ba, ya, za}, b{tb, dinb, doutb, actb, wtb, bb, yb, zb};
  const int n_tiles = (int)((n + kPairTR - 1) / kPairTR);
  const size_t lds = ((size_t)2 * kPairTR * OS2 + 2 * kNarrow * 64 + (size_t)kPairTR * kNarrowAll) * sizeof(float);
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(dense_pair_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (e != hipSuccess) return fail(NGPDE_ERR_HIP, "dense_pair_fwd_kernel: LDS request of %zu bytes refused: %s", lds, hipGetErrorString(e));
  static int cap = 0;   // resident workgroups of this kernel on this device (one occupancy query per process)
  if (!cap) cap = stream_grid(reinterpret_cast<const void *>(dense_pair_fwd_kernel), lds, 1 << 30);
  const int grid = std::min(n_tiles, cap);
#ifdef NGPDE_STAMPS
  hipLaunchKernelGGL(dense_pair_fwd_kernel, dim3((unsigned)grid), dim3(kStreamThreads), lds, stream, n, n_tiles, ta.ptr[0], a, b, dephase_cycles(), g_pair_stamps);
#else
  hipLaunchKernelGGL(dense_pair_fwd_kernel, dim3((unsigned)grid), dim3(kStreamThreads), lds, stream, n, n_tiles, ta.ptr[0], a, b, dephase_cycles());
#endif
  NGPDE_LAUNCH_CHECK("dense_pair_fwd_kernel");
  return NGPDE_OK;
}

bool dense_chain_fwd_applicable(int64_t n, const SegTable &t1, int din1, int dmid, int dout2) {
  const int nm = stream_main_blocks(t1, din1);
  return stream2_enabled(n) && dmid == 64 && dout2 <= 64 && (nm == 1 || nm == 2);
}
int32_t launch_dense_chain_fwd(int64_t n, const SegTable &t1, int din1, int act1, const float *wt1, const float *b1, float *a1, float *z1,
                               int dout2, int act2, const float *wt2, const float *b2, float *y, float *z2, hipStream_t stream) {
  SegTable t2;
  t2.n = 1; t2.width[0] = 64; t2.offset[1] = t2.offset[2] = t2.offset[3] = t2.offset[4] = 64;
  StreamOut l1{t1, din1, 64, act1, wt1, b1, a1, z1}, l2{t2, 64, dout2, act2, wt2, b2, y, z2};
  const int nm = stream_main_blocks(t1, din1);
  const int tr = 64;
  const int n_tiles = (int)((n + tr - 1) / tr);
  const size_t lds = ((size_t)2 * (tr * OS2 + (nm == 2 ? tr * 64 : 0)) + kNarrow * 64 + (size_t)tr * kNarrowAll + 128) * sizeof(float);
  auto launch = [&](auto kernel) -> hipError_t {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    static int cap[3] = {0, 0, 0};
    if (!cap[nm]) cap[nm] = stream_grid(reinterpret_cast<const void *>(kernel), lds, 1 << 30);
    const int grid = std::min(n_tiles, cap[nm]);
    hipLaunchKernelGGL(kernel, dim3((unsigned)grid), dim3(kStreamThreads), lds, stream, n, n_tiles, t1.ptr[0], nm == 2 ? t1.ptr[1] : nullptr,
                       l1, l2, dephase_cycles());
    return hipSuccess;
  };
  const hipError_t le = nm == 2 ? launch(dense_chain_fwd_kernel<2>) : launch(dense_chain_fwd_kernel<1>);
  if (le != hipSuccess) return fail(NGPDE_ERR_HIP, "dense_chain_fwd_kernel: LDS request of %zu bytes refused: %s", lds, hipGetErrorString(le));
  NGPDE_LAUNCH_CHECK("dense_chain_fwd_kernel");
  return NGPDE_OK;
}

}  // namespace ngpde

#ifdef NGPDE_STAMPS
extern "C" int32_t ngpde_debug_set_pair_stamps(unsigned long long *buf) {
  ngpde::g_pair_stamps = buf;
  return 0;
}
#endif
