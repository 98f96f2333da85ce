// dense_small_bwd.hip -- the whole pullback of a Dense layer with at most 64 inputs and at most 64 outputs in ONE launch
//   dz = dy . act'(z),   dX = dz W^T  (into the blocks of the virtual vcat that carry a gradient),   dW = X^T dz,   db = colsum(dz)
// for the message / update MLPs of the edge-function layers at the widths the reference's tutorials use (Lux Dense inside
// ExplicitEdgeConv / VMHConv / MPPDEConv, /root/reference/src/layers.jl:103-111, :313-326, :402-418; VMH.md:75-79 has
// 4 => 60 => 60 => 60 => 40 and 41 => 60 => 60 => 60 => 1) at any row count.  The composed path (dense_dz + weight pullback +
// slab reduction + input pullback: four launches of 4-9 us each at 3 000 - 18 000 rows, where every launch is one latency-bound
// wave of workgroups) was 63 % of a right-hand side + pullback of the VMH tutorial's NeuralODE; this is one launch + the reduction.
//
// A workgroup of four waves walks 64-row tiles.  Per tile: X (through the segment table), dy and z come in as 16 dwords per
// thread each (a thread keeps ONE column: its block of the vcat is resolved once), dz is formed in registers and both tiles go
// to LDS, zero-padded to 64 x 64.  Wave w then runs 64 MFMAs of dz W^T for rows 16 w .. 16 w + 15 (W zero-padded in LDS for
// the whole launch) and stores the result from its accumulators, and 64 MFMAs of X^T dz for input features 16 w .. 16 w + 15
// (contraction over the tile's 64 rows; accumulators live in registers for the whole launch).  Per-workgroup dW / db slabs in the
// layout of dense_mfma_bwd_weight_kernel, summed by dense_weight_reduce_kernel in a fixed order: no atomics, reproducible.
#include <algorithm>
#include <cstdlib>

#include "common.h"
#include "device_utils.h"

namespace ngpde {

namespace {

constexpr int kT = 256, kR = 64, kW = 64;
constexpr int kSZ = kW + 4;    // dz and W tiles: 16-byte rows, conflict-free row reads (16 rows x one 16-byte chunk)
constexpr int kSX = kW + 16;   // X tile: read only transposed (rows 4 s + kq, 16 consecutive columns): stride 80 spreads kq over the banks

struct SmallBwdK {
  int64_t n;
  int n_tiles, din, dout, act;
  SegTable segs;
  SegGrad grads;   // ptr[b] == NULL: block b carries no gradient
  const float *wt, *z, *dy;
  float *partial;   // [gridDim.x][din + 1][dout]
};

__device__ __forceinline__ int64_t srow(int64_t row, int row_div) {
  return row_div == 1 ? row : (int64_t)((uint32_t)row / (uint32_t)row_div);
}

// dz of a thread's 16 values with ONE uniform switch (a switch per element is a scalar branch chain per element)
template <int ACT>
__device__ __forceinline__ void dz16(float (&dv)[16], const float (&zv)[16]) {
#pragma unroll
  for (int s = 0; s < 16; ++s) dv[s] *= dact_c<ACT>(zv[s]);
}

__global__ __launch_bounds__(kT) void dense_small_bwd_kernel(const SmallBwdK p) {
  __shared__ __attribute__((aligned(16))) float ldsW[kW * kSZ], ldsDZ[kR * kSZ], ldsX[kR * kSX];
  __shared__ float ldsDb[4][kW];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kq = lane >> 4;
  const int din = p.din, dout = p.dout;
  // W for the input pullback: B[k = o][j = in] = wt[in][o]  ->  Bt[j = in][k = o] = wt[in][o], a straight (zero-padded) copy
  // (all sixteen loads of a thread first, then the LDS writes: a load -> write loop would take sixteen L2 round trips in turn)
  {
    float wv[kW * kW / kT];
#pragma unroll
    for (int k = 0; k < kW * kW / kT; ++k) {
      const int in = (tid >> 6) + 4 * k, o = tid & 63;
      wv[k] = (in < din && o < dout) ? p.wt[(size_t)in * dout + o] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < kW * kW / kT; ++k) ldsW[((tid >> 6) + 4 * k) * kSZ + (tid & 63)] = wv[k];
  }
  // this thread's column of the tiles (rows rq + 4 s): the block of the vcat that holds it, resolved once
  const int c = tid & 63, rq = tid >> 6;
  const float *xbase = nullptr;
  int xwidth = 0, xdiv = 1;
#pragma unroll
  for (int b = 3; b >= 0; --b)
    if (b < p.segs.n && c < p.segs.offset[b + 1] && c >= p.segs.offset[b] && c < din) {
      xbase = p.segs.ptr[b] + (c - p.segs.offset[b]);
      xwidth = p.segs.width[b];
      xdiv = p.segs.row_div[b];
    }
  // ... and the gradient block of the columns this lane stores after the input product (column 16 ct + i16)
  float *gbase[4];
  int gwidth[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const int col = 16 * ct + i16;
    gbase[ct] = nullptr;
    gwidth[ct] = 0;
#pragma unroll
    for (int b = 3; b >= 0; --b)
      if (b < p.grads.n && col < p.grads.offset[b + 1] && col >= p.grads.offset[b] && col < din && p.grads.ptr[b]) {
        gbase[ct] = p.grads.ptr[b] + (col - p.grads.offset[b]);
        gwidth[ct] = p.grads.width[b];
      }
  }
  const bool dcol = c < dout;
  f32x4 accW[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) accW[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dbacc = 0.f;

  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * kR;
    float xv[16], dv[16], zv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int64_t r = row0 + rq + 4 * s;
      const bool ok = r < p.n;
      xv[s] = (ok && xbase) ? xbase[srow(r, xdiv) * xwidth] : 0.f;
      dv[s] = (ok && dcol) ? p.dy[r * dout + c] : 0.f;
      zv[s] = (ok && dcol && p.z) ? p.z[r * dout + c] : 0.f;
    }
    if (p.z) { NGPDE_ACT_DISPATCH(p.act, dz16, dv, zv) }   // (rows / columns beyond the problem: dv = 0)
    __syncthreads();   // the previous tile's products are done with the LDS tiles (and W is in LDS)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float dz = dv[s];
      dbacc += dz;
      ldsDZ[(rq + 4 * s) * kSZ + c] = dz;
      ldsX[(rq + 4 * s) * kSX + c] = xv[s];
    }
    __syncthreads();
    // ---- dX rows 16 wave .. + 15 = dz x W^T: contraction over the outputs
    {
      f32x4 acc[4];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        const float4 a4 = *reinterpret_cast<const float4 *>(&ldsDZ[(wave * 16 + i16) * kSZ + 16 * kb + 4 * kq]);
        const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const float4 b4 = *reinterpret_cast<const float4 *>(&ldsW[(ct * 16 + i16) * kSZ + 16 * kb + 4 * kq]);
          const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) acc[ct] = mfma16(av[r], bv[r], acc[ct]);
        }
      }
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        if (gbase[ct]) {
#pragma unroll
          for (int reg = 0; reg < 4; ++reg) {
            const int64_t r = row0 + wave * 16 + 4 * kq + reg;
            if (r < p.n) gbase[ct][r * gwidth[ct]] = acc[ct][reg];
          }
        }
      }
    }
    // ---- dW rows (input features) 16 wave .. + 15 += X^T dz: contraction over the tile's rows
#pragma unroll
    for (int ks = 0; ks < kR / 4; ++ks) {
      const float a = ldsX[(4 * ks + kq) * kSX + wave * 16 + i16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) accW[nt] = mfma16(a, ldsDZ[(4 * ks + kq) * kSZ + nt * 16 + i16], accW[nt]);
    }
  }
  // ---- slab of this workgroup
  float *slab = p.partial + (size_t)blockIdx.x * (din + 1) * dout;
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) {
    const int o = nt * 16 + i16;
#pragma unroll
    for (int reg = 0; reg < 4; ++reg) {
      const int k = wave * 16 + 4 * kq + reg;
      if (k < din && o < dout) slab[(size_t)k * dout + o] = accW[nt][reg];
    }
  }
  ldsDb[rq][c] = dbacc;
  __syncthreads();
  if (tid < kW && tid < dout) slab[(size_t)din * dout + tid] = (ldsDb[0][tid] + ldsDb[1][tid]) + (ldsDb[2][tid] + ldsDb[3][tid]);
}

// ---- forward of the same shapes: y = act(X W + b) with the whole contraction in one pass -----------------------------------------
// (the general kernel walks the inputs in 16-feature steps with two barriers each: four to five dependent rounds at din = 60)
struct SmallFwdK {
  int64_t n;
  int n_tiles, din, dout, act;
  SegTable segs;
  const float *wt, *bias;
  float *y, *save_z;
};

template <int ACT>
__device__ __forceinline__ void act16(float (&v)[16]) {
#pragma unroll
  for (int s = 0; s < 16; ++s) v[s] = act_c<ACT>(v[s]);
}

__global__ __launch_bounds__(kT) void dense_small_fwd_kernel(const SmallFwdK p) {
  __shared__ __attribute__((aligned(16))) float ldsW[kW * kSZ], ldsX[kR * kSZ];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int i16 = lane & 15, kq = lane >> 4;
  const int din = p.din, dout = p.dout;
  const int c = tid & 63, rq = tid >> 6;
  // B[k = in][j = o] = wt[in][o]  ->  Bt[j = o][k = in]: a transposing (zero-padded) copy, loads first
  {
    float wv[kW * kW / kT];
#pragma unroll
    for (int k = 0; k < kW * kW / kT; ++k) {
      const int in = rq + 4 * k;
      wv[k] = (in < din && c < dout) ? p.wt[(size_t)in * dout + c] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < kW * kW / kT; ++k) ldsW[c * kSZ + rq + 4 * k] = wv[k];
  }
  const float *xbase = nullptr;
  int xwidth = 0, xdiv = 1;
#pragma unroll
  for (int b = 3; b >= 0; --b)
    if (b < p.segs.n && c < p.segs.offset[b + 1] && c >= p.segs.offset[b] && c < din) {
      xbase = p.segs.ptr[b] + (c - p.segs.offset[b]);
      xwidth = p.segs.width[b];
      xdiv = p.segs.row_div[b];
    }
  float bo[4];   // bias of the columns this lane holds after the product (column 16 ct + i16)
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) bo[ct] = (p.bias && 16 * ct + i16 < dout) ? p.bias[16 * ct + i16] : 0.f;

  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * kR;
    float xv[16];
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int64_t r = row0 + rq + 4 * s;
      xv[s] = (r < p.n && xbase) ? xbase[srow(r, xdiv) * xwidth] : 0.f;
    }
    __syncthreads();   // the previous tile's product is done with the X tile (and W is in LDS)
#pragma unroll
    for (int s = 0; s < 16; ++s) ldsX[(rq + 4 * s) * kSZ + c] = xv[s];
    __syncthreads();
    f32x4 acc[4];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) acc[ct] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      const float4 a4 = *reinterpret_cast<const float4 *>(&ldsX[(wave * 16 + i16) * kSZ + 16 * kb + 4 * kq]);
      const float av[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float4 b4 = *reinterpret_cast<const float4 *>(&ldsW[(ct * 16 + i16) * kSZ + 16 * kb + 4 * kq]);
        const float bv[4] = {b4.x, b4.y, b4.z, b4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[ct] = mfma16(av[r], bv[r], acc[ct]);
      }
    }
    float zv[16];
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) zv[4 * ct + reg] = acc[ct][reg] + bo[ct];
    if (p.save_z) {
#pragma unroll
      for (int ct = 0; ct < 4; ++ct)
#pragma unroll
        for (int reg = 0; reg < 4; ++reg) {
          const int64_t r = row0 + wave * 16 + 4 * kq + reg;
          const int o = 16 * ct + i16;
          if (r < p.n && o < dout) p.save_z[r * dout + o] = zv[4 * ct + reg];
        }
    }
    NGPDE_ACT_DISPATCH(p.act, act16, zv)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int64_t r = row0 + wave * 16 + 4 * kq + reg;
        const int o = 16 * ct + i16;
        if (r < p.n && o < dout) p.y[r * dout + o] = zv[4 * ct + reg];
      }
  }
}

}  // namespace

// the forward of the same shapes in one contraction pass (0: not this form's shape)
int dense_small_fwd_grid(int64_t n, int din, int dout) {
  static const bool off = std::getenv("NGPDE_DENSE_NO_SMALL_FWD") != nullptr;
  if (off || din < 17 || din > kW || dout < 1 || dout > kW || n < 1 || n > 65536) return 0;   // (din <= 16 is ONE step of the general kernel)
  return (int)std::min<int64_t>((n + kR - 1) / kR, 1024);
}

int32_t launch_dense_small_fwd(int64_t n, const SegTable &t, int din, int dout, int act, const float *wt, const float *bias, float *y,
                               float *save_z, int grid, hipStream_t stream) {
  SmallFwdK k{};
  k.n = n; k.n_tiles = (int)((n + kR - 1) / kR); k.din = din; k.dout = dout; k.act = act;
  k.segs = t; k.wt = wt; k.bias = bias; k.y = y; k.save_z = save_z;
  hipLaunchKernelGGL(dense_small_fwd_kernel, dim3(grid), dim3(kT), 0, stream, k);
  NGPDE_LAUNCH_CHECK("dense_small_fwd_kernel");
  return NGPDE_OK;
}

// grid of the one-launch form, or 0 when the shape is not its own: widths up to 64, few enough rows that the composed path's
// launches are latency-bound (beyond that its 16-byte / LDS-DMA loads win), and enough rows for the slabs to fit the [n][dout] dz
// area of the composed path's workspace (which this form does not use otherwise)
int dense_small_bwd_grid(int64_t n, int din, int dout) {
  static const bool off = std::getenv("NGPDE_DENSE_NO_SMALL_BWD") != nullptr;
  if (off || din < 1 || din > kW || dout < 1 || dout > kW || n < 1 || n > 65536) return 0;
  const int64_t n_tiles = (n + kR - 1) / kR;
  const int64_t grid = std::min<int64_t>(std::min<int64_t>(n_tiles, 1024), n / (din + 1));
  return grid >= 1 ? (int)grid : 0;
}

int32_t launch_dense_small_bwd(int64_t n, const SegTable &t, int din, int dout, int act, const float *wt, const float *z, const float *dy,
                               float *const *dseg, float *dwt, float *dbias, float *slabs, int grid, hipStream_t stream) {
  SmallBwdK k{};
  k.n = n; k.n_tiles = (int)((n + kR - 1) / kR); k.din = din; k.dout = dout; k.act = act;
  k.segs = t;
  k.grads.n = t.n;
  for (int b = 0; b < t.n; ++b) {
    k.grads.ptr[b] = (dseg && t.row_div[b] == 1) ? dseg[b] : nullptr;   // per-graph blocks carry no gradient (@ignore_derivatives, :397, :418)
    k.grads.width[b] = t.width[b];
  }
  for (int b = 0; b <= 4; ++b) k.grads.offset[b] = t.offset[b];
  k.wt = wt; k.z = (act == NGPDE_ACT_IDENTITY) ? nullptr : z; k.dy = dy; k.partial = slabs;
  hipLaunchKernelGGL(dense_small_bwd_kernel, dim3(grid), dim3(kT), 0, stream, k);
  NGPDE_LAUNCH_CHECK("dense_small_bwd_kernel");
  return launch_dense_weight_reduce(grid, din, dout, slabs, dwt, dbias, stream);
}

}  // namespace ngpde
