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

// the loaded value is needed whatever a later select does with it: keeps the compiler from sinking the load under the select's
// condition (an exec-masked branch and a full wait per load)
#define NGPDE_KEEP(x) asm volatile("" : "+v"(x))

constexpr int kT = 256, kR = 64, kW = 64;
constexpr int kSZ = kW + 4;    // dz and W tiles: 16-byte rows, conflict-free row reads (16 rows x one 16-byte chunk)
constexpr int kSX = kW + 16;   // X tile: read only transposed (rows 4 s + kq, 16 consecutive columns): stride 80 spreads kq over the banks

// diagnostic build only (tools/stamps_small_dense.py): wall-clock stamps (100 MHz) of thread 0 of every workgroup, [n_blocks][8]
#ifdef NGPDE_STAMPS
unsigned long long *g_small_stamps = nullptr;
#define NGPDE_SST_FIELD unsigned long long *stamps;
#define NGPDE_SST(p, k) do { if (threadIdx.x == 0 && (p).stamps) (p).stamps[(size_t)blockIdx.x * 8 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define NGPDE_SST_SET(kk) kk.stamps = g_small_stamps;
#else
#define NGPDE_SST_FIELD
#define NGPDE_SST(p, k)
#define NGPDE_SST_SET(kk)
#endif

struct SmallBwdK {
  NGPDE_SST_FIELD
  int64_t n;
  int n_tiles, din, dout, act;
  SegTable segs;
  SegGrad grads;   // ptr[b] == NULL: block b carries no gradient
  const float *wt, *z, *dy;
  float *partial;   // [gridDim.x][din + 1][dout]
};

// the block of the virtual vcat that holds column `col`, with every table field read unconditionally (selects, no branches): the
// table lives in kernel-argument memory, and a field read under a branch is a dependent scalar load of its own -- twenty of them in a
// row, each waited for, were most of the 5 us these kernels spent before their first product
struct ColRef {
  const float *base;   // element (row 0, col) of the block, or NULL when col is beyond the layer's inputs
  int width, div;
};
// (uniform values pinned to scalar registers: left to itself the compiler fetched the table per lane with vector loads -- a dependent
// round trip in front of the tile loads)
__device__ __forceinline__ int sgpr(int v) { return __builtin_amdgcn_readfirstlane(v); }
template <class T>
__device__ __forceinline__ T *sgpr_ptr(T *p) {
  const uintptr_t u = reinterpret_cast<uintptr_t>(p);
  const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu)), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(u >> 32));
  return reinterpret_cast<T *>(((uintptr_t)hi << 32) | lo);
}
template <class Table, class Ptr>
__device__ __forceinline__ ColRef table_column(const Table &s, const int (&row_div)[4], int col, int din) {
  const int n = sgpr(s.n), o0 = sgpr(s.offset[0]), o1 = sgpr(s.offset[1]), o2 = sgpr(s.offset[2]), o3 = sgpr(s.offset[3]);
  const int w0 = sgpr(s.width[0]), w1 = sgpr(s.width[1]), w2 = sgpr(s.width[2]), w3 = sgpr(s.width[3]);
  const int d0 = sgpr(row_div[0]), d1 = sgpr(row_div[1]), d2 = sgpr(row_div[2]), d3 = sgpr(row_div[3]);
  Ptr p0 = sgpr_ptr(s.ptr[0]), p1 = sgpr_ptr(s.ptr[1]), p2 = sgpr_ptr(s.ptr[2]), p3 = sgpr_ptr(s.ptr[3]);
  const bool b1 = n > 1 && col >= o1, b2 = n > 2 && col >= o2, b3 = n > 3 && col >= o3;
  Ptr ptr = b3 ? p3 : (b2 ? p2 : (b1 ? p1 : p0));
  const int off = b3 ? o3 : (b2 ? o2 : (b1 ? o1 : o0));
  ColRef r;
  r.width = b3 ? w3 : (b2 ? w2 : (b1 ? w1 : w0));
  r.div = b3 ? d3 : (b2 ? d2 : (b1 ? d1 : d0));
  r.base = (col < din && ptr) ? ptr + (col - off) : nullptr;
  return r;
}
__device__ __forceinline__ ColRef seg_column(const SegTable &s, int col, int din) {
  return table_column<SegTable, const float *>(s, s.row_div, col, din);
}
__device__ __forceinline__ ColRef grad_column(const SegGrad &s, int col, int din) {
  const int ones[4] = {1, 1, 1, 1};
  return table_column<SegGrad, float *>(s, ones, col, din);
}

// block row = row / row_div without a branch or an integer division (rows < 2^17 here): float estimate, corrected by one either way.
// (row_div differs between the lanes of a wave -- each lane has its own column, hence its own block -- so a `row_div == 1 ?`
// shortcut would be a divergent branch per load.)
__device__ __forceinline__ int srow32(int row, int row_div, float inv) {
  int q = (int)((float)row * inv);
  q -= (q * row_div > row) ? 1 : 0;
  q += ((q + 1) * row_div <= row) ? 1 : 0;
  return q;
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
  NGPDE_SST(p, 0);
  // W for the input pullback: B[k = o][j = in] = wt[in][o]  ->  Bt[j = in][k = o] = wt[in][o], a straight (zero-padded) copy.  Its loads
  // go out here, the first tile's loads right behind them, and only then the LDS writes: one round trip for both (a W copy completed
  // first cost a second one: 4-5 us before the first product instead of 2.5)
  // (every load of these kernels is unconditional, from a clamped address, and neutralised afterwards: a `cond ? load : 0` compiles
  // to an exec-masked branch per load -- 64 of them were most of the 5 us in front of the first product)
  float wv[kW * kW / kT];
#pragma unroll
  for (int k = 0; k < kW * kW / kT; ++k) {
    const int in = (tid >> 6) + 4 * k, o = tid & 63;
    wv[k] = p.wt[min(in, din - 1) * dout + min(o, dout - 1)];
  }
  // this thread's column of the tiles (rows rq + 4 s): the block of the vcat that holds it, resolved once
  const int c = tid & 63, rq = tid >> 6;
  const ColRef xc = seg_column(p.segs, c, din);
  const float *xbase = xc.base;
  const int xwidth = xc.width, xdiv = xc.div;
  // ... and the gradient block of the columns this lane stores after the input product (column 16 ct + i16)
  float *gbase[4];
  int gwidth[4];
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) {
    const ColRef gc = grad_column(p.grads, 16 * ct + i16, din);
    gbase[ct] = const_cast<float *>(gc.base);
    gwidth[ct] = gc.width;
  }
  const bool dcol = c < dout;
  f32x4 accW[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) accW[nt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  float dbacc = 0.f;

  float xv[16], dv[16], zv[16];
  // 32-bit offsets advanced by additions (n <= 65 536 rows of at most 64 floats): sixteen 64-bit multiplies per array and thread were
  // a microsecond of quarter-rate instructions in front of the loads
  const float *xb = xbase ? xbase : p.wt;            // (a column without a block reads something valid and is zeroed)
  const int xw = xbase ? xwidth : 0;
  const float xinv = 1.0f / (float)xdiv;
  const int cc = min(c, dout - 1);
  auto load_tile = [&](int tile) {
    const int r0 = tile * kR + rq, nlast = (int)p.n - 1;
    const int od_max = nlast * dout + cc;
    int od = r0 * dout + cc;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      xv[s] = xb[srow32(min(r0 + 4 * s, nlast), xdiv, xinv) * xw];
      dv[s] = p.dy[min(od, od_max)];
      od += 4 * dout;
    }
    if (p.z) {   // uniform
      od = r0 * dout + cc;
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        zv[s] = p.z[min(od, od_max)];
        od += 4 * dout;
      }
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      NGPDE_KEEP(xv[s]); NGPDE_KEEP(dv[s]);
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const bool ok = r0 + 4 * s <= nlast;
      xv[s] = (ok && xbase) ? xv[s] : 0.f;
      dv[s] = (ok && dcol) ? dv[s] : 0.f;
    }
  };
  load_tile(blockIdx.x);
#pragma unroll
  for (int k = 0; k < kW * kW / kT; ++k) NGPDE_KEEP(wv[k]);
#pragma unroll
  for (int k = 0; k < kW * kW / kT; ++k)
    ldsW[((tid >> 6) + 4 * k) * kSZ + (tid & 63)] = ((tid >> 6) + 4 * k < din && (tid & 63) < dout) ? wv[k] : 0.f;
  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * kR;
    if (tile != (int)blockIdx.x) load_tile(tile);
    if (p.z) { NGPDE_ACT_DISPATCH(p.act, dz16, dv, zv) }   // (rows / columns beyond the problem: dv = 0)
    NGPDE_SST(p, 1);   // loads landed, dz formed
    __syncthreads();   // the previous tile's products are done with the LDS tiles (and W is in LDS)
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float dz = dv[s];
      dbacc += dz;
      ldsDZ[(rq + 4 * s) * kSZ + c] = dz;
      ldsX[(rq + 4 * s) * kSX + c] = xv[s];
    }
    __syncthreads();
    NGPDE_SST(p, 2);   // tiles in LDS
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
            const int r = (int)row0 + wave * 16 + 4 * kq + reg;
            if (r < (int)p.n) gbase[ct][r * gwidth[ct]] = acc[ct][reg];
          }
        }
      }
    }
    NGPDE_SST(p, 3);   // dX product + stores issued
    // ---- dW rows (input features) 16 wave .. + 15 += X^T dz: contraction over the tile's rows
#pragma unroll
    for (int ks = 0; ks < kR / 4; ++ks) {
      const float a = ldsX[(4 * ks + kq) * kSX + wave * 16 + i16];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) accW[nt] = mfma16(a, ldsDZ[(4 * ks + kq) * kSZ + nt * 16 + i16], accW[nt]);
    }
  }
  NGPDE_SST(p, 4);   // dW product
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
  NGPDE_SST(p, 5);   // slab stores issued
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
  // B[k = in][j = o] = wt[in][o]  ->  Bt[j = o][k = in]: a transposing (zero-padded) copy; its loads and the first tile's in one batch
  float wv[kW * kW / kT];
#pragma unroll
  for (int k = 0; k < kW * kW / kT; ++k) {
    const int in = rq + 4 * k;
    wv[k] = p.wt[min(in, din - 1) * dout + min(c, dout - 1)];
  }
  const ColRef xc = seg_column(p.segs, c, din);
  const float *xbase = xc.base;
  const int xwidth = xc.width, xdiv = xc.div;
  float bo[4];   // bias of the columns this lane holds after the product (column 16 ct + i16)
#pragma unroll
  for (int ct = 0; ct < 4; ++ct) bo[ct] = (p.bias && 16 * ct + i16 < dout) ? p.bias[16 * ct + i16] : 0.f;

  float xv[16];
  const float *xb = xbase ? xbase : p.wt;
  const int xw = xbase ? xwidth : 0;
  const float xinv = 1.0f / (float)xdiv;
  auto load_tile = [&](int tile) {
    const int r0 = tile * kR + rq, nlast = (int)p.n - 1;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      xv[s] = xb[srow32(min(r0 + 4 * s, nlast), xdiv, xinv) * xw];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) NGPDE_KEEP(xv[s]);
#pragma unroll
    for (int s = 0; s < 16; ++s) xv[s] = (r0 + 4 * s <= nlast && xbase) ? xv[s] : 0.f;
  };
  load_tile(blockIdx.x);
#pragma unroll
  for (int k = 0; k < kW * kW / kT; ++k) NGPDE_KEEP(wv[k]);
#pragma unroll
  for (int k = 0; k < kW * kW / kT; ++k) ldsW[c * kSZ + rq + 4 * k] = (rq + 4 * k < din && c < dout) ? wv[k] : 0.f;
  for (int tile = blockIdx.x; tile < p.n_tiles; tile += gridDim.x) {
    const int64_t row0 = (int64_t)tile * kR;
    if (tile != (int)blockIdx.x) load_tile(tile);
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
          const int r = (int)row0 + wave * 16 + 4 * kq + reg;
          const int o = 16 * ct + i16;
          if (r < (int)p.n && o < dout) p.save_z[r * dout + o] = zv[4 * ct + reg];
        }
    }
    NGPDE_ACT_DISPATCH(p.act, act16, zv)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int reg = 0; reg < 4; ++reg) {
        const int r = (int)row0 + wave * 16 + 4 * kq + reg;
        const int o = 16 * ct + i16;
        if (r < (int)p.n && o < dout) p.y[r * dout + o] = zv[4 * ct + reg];
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
  NGPDE_SST_SET(k)
  hipLaunchKernelGGL(dense_small_bwd_kernel, dim3(grid), dim3(kT), 0, stream, k);
  NGPDE_LAUNCH_CHECK("dense_small_bwd_kernel");
  return launch_dense_weight_reduce(grid, din, dout, slabs, dwt, dbias, stream);
}

}  // namespace ngpde

#ifdef NGPDE_STAMPS
extern "C" int32_t ngpde_debug_set_small_dense_stamps(unsigned long long *dev_buf) {   // [n_blocks][8] or NULL
  ngpde::g_small_stamps = dev_buf;
  return NGPDE_OK;
}
#endif
