// dense_mfma.hip -- fp32 MFMA kernels of the concat-free Dense (Lux.Dense on a virtual vcat of blocks) used
// by the edge-function layers of /root/reference/src/layers.jl (:106, :316, :328, :409, :418, :523) over N node
// columns or E edge columns:  y = act([X1 | X2 | ...] Wt + b), its input pullback and its weight pullback.
// 64 x 64 output tile per workgroup (4 waves x (16 rows x 64 cols)), K in chunks of 16 staged through LDS with
// the B operand stored transposed, so every operand fetch is one ds_read_b128 feeding four
// v_mfma_f32_16x16x4_f32 k-steps (exact fp32); the next chunk's global loads are in flight during the MFMAs.
#include <algorithm>

#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

#define NGPDE_LAUNCH_CHECK(name)                                                         \
  do {                                                                                   \
    hipError_t _e = hipGetLastError();                                                   \
    if (_e != hipSuccess) return fail(NGPDE_ERR_HIP, "%s launch failed: %s", name, hipGetErrorString(_e)); \
  } while (0)

constexpr int BM = 64, BN = 64, BK = 16, LS = BK + 4;   // LDS row stride 20 floats: 16-byte aligned b128 rows

// rows are < 2^31 (host-checked), so the per-graph row division is a 32-bit one and only taken when a block asks for it
__device__ __forceinline__ int64_t seg_row(int64_t row, int row_div) {
  return row_div == 1 ? row : (int64_t)((uint32_t)row / (uint32_t)row_div);
}

__device__ __forceinline__ float seg_load(const SegTable &s, int64_t row, int k) {
#pragma unroll
  for (int i = 0; i < 4; ++i)
    if (i < s.n && k < s.offset[i + 1]) return s.ptr[i][seg_row(row, s.row_div[i]) * s.width[i] + (k - s.offset[i])];
  return 0.f;
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
__global__ __launch_bounds__(256) void dense_mfma_fwd_kernel(int64_t n, SegTable segs, int din, int dout, int act,
                                                             const float *__restrict__ wt, const float *__restrict__ bias,
                                                             float *__restrict__ y, float *__restrict__ save_z) {
  __shared__ __attribute__((aligned(16))) float ldsA[BM * LS], ldsBt[BN * LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int col0 = blockIdx.y * BN;
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
  fetch(0);
  for (int k0 = 0; k0 < din; k0 += BK) {
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

// ---- input pullback: dX[n][k] = sum_o dz[n][o] wt[k][o], written into the blocks that ask for it -------------------
__global__ __launch_bounds__(256) void dense_mfma_bwd_input_kernel(int64_t n, SegGrad segs, int din, int dout,
                                                                   const float *__restrict__ dz,
                                                                   const float *__restrict__ wt) {
  __shared__ __attribute__((aligned(16))) float ldsA[BM * LS], ldsBt[BN * LS];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int64_t row0 = (int64_t)blockIdx.x * BM;
  const int col0 = blockIdx.y * BN;          // columns of dX = input features k
  const int ar = tid >> 4, ak = tid & 15;    // A = dz: (row, o)
  const int bcol = tid >> 2, bo4 = (tid & 3) * 4;   // Bt[col = k][o]: thread loads 4 consecutive o of one k row
  float areg[4];
  float4 breg;
  auto fetch = [&](int o0) {
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const int64_t r = row0 + ar + 16 * p;
      areg[p] = (r < n && o0 + ak < dout) ? dz[r * dout + o0 + ak] : 0.f;
    }
    const int k = col0 + bcol;
    float t[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) t[j] = (k < din && o0 + bo4 + j < dout) ? wt[(size_t)k * dout + o0 + bo4 + j] : 0.f;
    breg = make_float4(t[0], t[1], t[2], t[3]);
  };
  f32x4 acc[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
  fetch(0);
  for (int o0 = 0; o0 < dout; o0 += BK) {
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
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (idx < total) {
    int c = cl;
    for (; c + 12 < nchunk; c += 16) {
      s0 += partial[(size_t)c * total + idx];
      s1 += partial[(size_t)(c + 4) * total + idx];
      s2 += partial[(size_t)(c + 8) * total + idx];
      s3 += partial[(size_t)(c + 12) * total + idx];
    }
    for (; c < nchunk; c += 4) s0 += partial[(size_t)c * total + idx];
  }
  part[cl][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (cl == 0 && idx < total) {
    const float s = (part[0][e] + part[1][e]) + (part[2][e] + part[3][e]);
    if (idx < din * dout) dwt[idx] = s;
    else if (db) db[idx - din * dout] = s;
  }
}

}  // namespace

int32_t launch_dense_seg_fwd(int64_t n, const SegTable &segs, int din, int dout, int act, const float *wt,
                             const float *bias, float *y, float *save_z, hipStream_t stream) {
  if (n == 0 || dout == 0) return NGPDE_OK;
  hipLaunchKernelGGL(dense_mfma_fwd_kernel, dim3((unsigned)((n + BM - 1) / BM), (dout + BN - 1) / BN), dim3(256), 0, stream,
                     n, segs, din, dout, act, wt, bias, y, save_z);
  NGPDE_LAUNCH_CHECK("dense_mfma_fwd_kernel");
  return NGPDE_OK;
}

int32_t launch_dense_seg_bwd_input(int64_t n, const SegGrad &segs, int din, int dout, const float *dz, const float *wt,
                                   hipStream_t stream) {
  if (n == 0 || din == 0) return NGPDE_OK;
  hipLaunchKernelGGL(dense_mfma_bwd_input_kernel, dim3((unsigned)((n + BM - 1) / BM), (din + BN - 1) / BN), dim3(256), 0,
                     stream, n, segs, din, dout, dz, wt);
  NGPDE_LAUNCH_CHECK("dense_mfma_bwd_input_kernel");
  return NGPDE_OK;
}

// row chunks of the weight pullback: enough (tile x chunk) workgroups to cover the chip several times over (the chunk
// loop is a dependent global-load chain, ~16 rows per trip), at least 64 rows per chunk, at most 1024 partial slabs
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
  hipLaunchKernelGGL(dense_mfma_bwd_weight_kernel, dim3(std::max(1, (din + BM - 1) / BM), (dout + BN - 1) / BN, nchunk), dim3(256), 0,
                     stream, n, segs, din, dout, dz, rpc, partial);
  NGPDE_LAUNCH_CHECK("dense_mfma_bwd_weight_kernel");
  const int total = (din + 1) * dout;
  hipLaunchKernelGGL(dense_weight_reduce_kernel, dim3((total + 63) / 64), dim3(256), 0, stream, nchunk, din, dout, partial,
                     dwt, db);
  NGPDE_LAUNCH_CHECK("dense_weight_reduce_kernel");
  return NGPDE_OK;
}

}  // namespace ngpde
