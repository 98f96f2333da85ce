// neighbors.hip -- neighbour search on the device: the COO lists GNNGraphs.radius_graph / knn_graph produce on the host
// with NearestNeighbors.jl trees ([UPSTREAM] GraphNeuralNetworks.jl, re-exported at /root/reference/src/NeuralGraphPDE.jl:4),
// and a space-filling-curve node order that serves as the locality schedule of a never-seen point-cloud graph without a
// host traversal.
//
// Semantics (the tests compare bit for bit with a brute-force float32 restatement):
//   d2(i, j) = sum over the coordinates, in order, of (p_i - p_j)^2, every operation rounded to float (no fused
//   multiply-add: the library is compiled with -ffp-contract=off); radius graph: j is a neighbour of i iff
//   d2 <= r*r (float product) and, with graph_id, both belong to the same graph; k-NN: the k smallest (d2, j) pairs.
//   dir = :in (default): neighbours are SOURCES, i the target; the list is ordered by i, neighbours ascending by index
//   (radius) or by (d2, index) (k-NN).  The reference's tree searches return an implementation-defined order of the
//   same edge set.
// Method: uniform cell grid over the bounding box (cell >= 1.01 r, so the 3^dim block around a point's cell holds its
// neighbours), points sorted by (graph, cell) with a stable radix sort, one thread per point walking contiguous cell
// ranges; counts -> scan -> fill -> per-row sort.  k-NN walks rings of cells until the k-th distance is proven.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace ngpde {
namespace {

constexpr int kB = 256;
inline unsigned blocks_for(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + kB - 1) / kB); }

struct Scratch {   // frees on scope exit
  std::vector<void *> ptrs;
  ~Scratch() {
    for (void *p : ptrs) (void)hipFree(p);
  }
  template <class T>
  int32_t get(T **p, size_t count) {
    *p = nullptr;
    NGPDE_HIP_CHECK(hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
    ptrs.push_back(*p);
    return NGPDE_OK;
  }
};

int bits_for(int64_t n) {
  int b = 1;
  while (((int64_t)1 << b) < n) ++b;
  return b;
}

struct Grid {
  float lo[3];
  float inv;        // 1 / cell edge
  float cell;       // cell edge
  int nc[3];        // cells per dimension
  int cells;        // per graph
};

__device__ __forceinline__ unsigned ordered_bits(float v) {
  const unsigned u = __float_as_uint(v);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
inline float from_ordered_bits(unsigned u) {
  const unsigned b = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
  float f;
  std::memcpy(&f, &b, 4);
  return f;
}

// box[0..2] = min, box[3..5] = max (ordered-bit encoding); also checks graph ids
__global__ void bbox_kernel(int64_t n, int dim, const float *__restrict__ pts, const int32_t *__restrict__ gid, int id_base,
                            int n_graphs, unsigned *__restrict__ box, int *__restrict__ bad_id) {
  unsigned mn[3] = {0xffffffffu, 0xffffffffu, 0xffffffffu}, mx[3] = {0u, 0u, 0u};
  bool bad = false;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
    for (int d = 0; d < dim; ++d) {
      const unsigned u = ordered_bits(pts[i * dim + d]);
      mn[d] = min(mn[d], u);
      mx[d] = max(mx[d], u);
    }
    if (gid) {
      const int g = gid[i] - id_base;
      bad |= (g < 0 || g >= n_graphs);
    }
  }
  for (int d = 0; d < dim; ++d) {
    for (int off = 32; off > 0; off >>= 1) {
      mn[d] = min(mn[d], (unsigned)__shfl_xor((int)mn[d], off));
      mx[d] = max(mx[d], (unsigned)__shfl_xor((int)mx[d], off));
    }
    if ((threadIdx.x & 63) == 0) {
      atomicMin(&box[d], mn[d]);
      atomicMax(&box[3 + d], mx[d]);
    }
  }
  if (bad) atomicOr(bad_id, 1);
}

template <int DIM>
__device__ __forceinline__ void cell_of(const Grid &g, const float *p, int *c) {
#pragma unroll
  for (int d = 0; d < DIM; ++d) c[d] = min(g.nc[d] - 1, max(0, (int)((p[d] - g.lo[d]) * g.inv)));
}
template <int DIM>
__device__ __forceinline__ int linear_cell(const Grid &g, const int *c) {
  int l = c[0];
#pragma unroll
  for (int d = 1; d < DIM; ++d) l = l * g.nc[d] + c[d];
  return l;
}

template <int DIM>
__global__ void cell_key_kernel(int64_t n, Grid g, const float *__restrict__ pts, const int32_t *__restrict__ gid, int id_base,
                                unsigned *__restrict__ key, int32_t *__restrict__ iota) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float p[DIM];
  int c[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) p[d] = pts[i * DIM + d];
  cell_of<DIM>(g, p, c);
  const int graph = gid ? gid[i] - id_base : 0;
  key[i] = (unsigned)graph * (unsigned)g.cells + (unsigned)linear_cell<DIM>(g, c);
  iota[i] = (int32_t)i;
}

template <int DIM>
__global__ void gather_points_kernel(int64_t n, const float *__restrict__ pts, const int32_t *__restrict__ idx,
                                     float *__restrict__ out) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  const int64_t i = idx[p];
#pragma unroll
  for (int d = 0; d < DIM; ++d) out[p * DIM + d] = pts[i * DIM + d];
}

// start[c] = first sorted position whose key is >= c   (c = 0 .. total)
__global__ void cell_start_kernel(int64_t total, int64_t n, const unsigned *__restrict__ key_sorted, int32_t *__restrict__ start) {
  const int64_t c = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (c > total) return;
  int64_t lo = 0, hi = n;
  while (lo < hi) {
    const int64_t mid = (lo + hi) >> 1;
    if ((int64_t)key_sorted[mid] < c) lo = mid + 1;
    else hi = mid;
  }
  start[c] = (int32_t)lo;
}

template <int DIM>
__device__ __forceinline__ float dist2(const float *a, const float *b) {
  float d2 = 0.f;
#pragma unroll
  for (int d = 0; d < DIM; ++d) {
    const float df = a[d] - b[d];
    d2 = d2 + df * df;     // not contracted: -ffp-contract=off
  }
  return d2;
}

// Visits every sorted position q in the 3^DIM block of cells around point p's cell (the cells along the last
// coordinate are contiguous in key order: one range per row of the block).
template <int DIM, class F>
__device__ __forceinline__ void for_block(const Grid &g, int graph, const int *c, const int32_t *__restrict__ start, F &&f) {
  const int base = graph * g.cells;
  const int l0 = max(c[DIM - 1] - 1, 0), l1 = min(c[DIM - 1] + 1, g.nc[DIM - 1] - 1);
  if constexpr (DIM == 1) {
    for (int q = start[base + l0], e = start[base + l1 + 1]; q < e; ++q) f(q);
  } else if constexpr (DIM == 2) {
    for (int a = max(c[0] - 1, 0); a <= min(c[0] + 1, g.nc[0] - 1); ++a) {
      const int row = base + a * g.nc[1];
      for (int q = start[row + l0], e = start[row + l1 + 1]; q < e; ++q) f(q);
    }
  } else {
    for (int a = max(c[0] - 1, 0); a <= min(c[0] + 1, g.nc[0] - 1); ++a)
      for (int b = max(c[1] - 1, 0); b <= min(c[1] + 1, g.nc[1] - 1); ++b) {
        const int row = base + (a * g.nc[1] + b) * g.nc[2];
        for (int q = start[row + l0], e = start[row + l1 + 1]; q < e; ++q) f(q);
      }
  }
}

// FILL = false: deg[i] = number of neighbours; FILL = true: writes them (unsorted) at rowptr[i]
template <int DIM, bool FILL>
__global__ void radius_kernel(int64_t n, Grid g, float r2, int self_loops, int out_base, const float *__restrict__ spts,
                              const int32_t *__restrict__ sidx, const unsigned *__restrict__ key_sorted,
                              const int32_t *__restrict__ start, int32_t *__restrict__ deg, unsigned long long *__restrict__ total,
                              const int32_t *__restrict__ rowptr, int32_t *__restrict__ nbr, int32_t *__restrict__ self) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int cnt = 0;
  if (p < n) {
    float a[DIM];
    int c[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) a[d] = spts[p * DIM + d];
    cell_of<DIM>(g, a, c);
    const int graph = (int)(key_sorted[p] / (unsigned)g.cells);
    const int32_t i = sidx[p];
    int32_t w = FILL ? rowptr[i] : 0;
    for_block<DIM>(g, graph, c, start, [&](int q) {
      float b[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d) b[d] = spts[(int64_t)q * DIM + d];
      const bool hit = dist2<DIM>(a, b) <= r2 && (self_loops || q != (int)p);
      if (hit) {
        if constexpr (FILL) {
          nbr[w] = sidx[q] + out_base;
          self[w] = i + out_base;
          ++w;
        }
        ++cnt;
      }
    });
    if constexpr (!FILL) deg[i] = cnt;
  }
  if constexpr (!FILL) {
    unsigned long long s = (unsigned long long)cnt;
    for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
    if ((threadIdx.x & 63) == 0 && s) atomicAdd(total, s);
  }
}

// k nearest (d2, index) pairs per point; the running list lives in LDS, one column per thread:
// list entry m of thread l at [m * 64 + l]
constexpr int kKnnThreads = 64;
template <int DIM>
__global__ void knn_kernel(int64_t n, Grid g, int k, int self_loops, int out_base, int dir_out, const float *__restrict__ spts,
                           const int32_t *__restrict__ sidx, const unsigned *__restrict__ key_sorted,
                           const int32_t *__restrict__ start, int32_t *__restrict__ s_out, int32_t *__restrict__ t_out,
                           int *__restrict__ short_rows) {
  extern __shared__ float knn_lds[];
  float *ld = knn_lds + threadIdx.x;                        // distances
  int *li = (int *)(knn_lds + (size_t)k * kKnnThreads) + threadIdx.x;   // original indices
  const int64_t p = (int64_t)blockIdx.x * kKnnThreads + threadIdx.x;
  if (p >= n) return;
  float a[DIM];
  int c[DIM];
#pragma unroll
  for (int d = 0; d < DIM; ++d) a[d] = spts[p * DIM + d];
  cell_of<DIM>(g, a, c);
  const int graph = (int)(key_sorted[p] / (unsigned)g.cells);
  const int base = graph * g.cells;
  const int32_t i = sidx[p];
  int have = 0;
  int max_ring = 0;
#pragma unroll
  for (int d = 0; d < DIM; ++d) max_ring = max(max_ring, max(c[d], g.nc[d] - 1 - c[d]));
  auto offer = [&](int q) {
    if (!self_loops && q == (int)p) return;
    float b[DIM];
#pragma unroll
    for (int d = 0; d < DIM; ++d) b[d] = spts[(int64_t)q * DIM + d];
    const float d2 = dist2<DIM>(a, b);
    const int j = sidx[q];
    if (have == k) {
      const float wd = ld[(k - 1) * kKnnThreads];
      const int wj = li[(k - 1) * kKnnThreads];
      if (!(d2 < wd || (d2 == wd && j < wj))) return;
    }
    int m = (have < k) ? have : k - 1;                       // slot that opens up
    while (m > 0) {
      const float pd = ld[(m - 1) * kKnnThreads];
      const int pj = li[(m - 1) * kKnnThreads];
      if (pd < d2 || (pd == d2 && pj < j)) break;
      ld[m * kKnnThreads] = pd;
      li[m * kKnnThreads] = pj;
      --m;
    }
    ld[m * kKnnThreads] = d2;
    li[m * kKnnThreads] = j;
    if (have < k) ++have;
  };
  for (int ring = 0; ring <= max_ring; ++ring) {
    // cells at Chebyshev distance exactly `ring` from the home cell
    if constexpr (DIM == 1) {
      for (int s = -1; s <= 1; s += 2) {
        const int x = c[0] + s * ring;
        if (x < 0 || x >= g.nc[0] || (ring == 0 && s > 0)) continue;
        for (int q = start[base + x], e = start[base + x + 1]; q < e; ++q) offer(q);
      }
    } else if constexpr (DIM == 2) {
      for (int dx = -ring; dx <= ring; ++dx) {
        const int x = c[0] + dx;
        if (x < 0 || x >= g.nc[0]) continue;
        const int step = (abs(dx) == ring || ring == 0) ? 1 : 2 * ring;
        for (int dy = -ring; dy <= ring; dy += step) {
          const int y = c[1] + dy;
          if (y < 0 || y >= g.nc[1]) continue;
          const int cell = base + x * g.nc[1] + y;
          for (int q = start[cell], e = start[cell + 1]; q < e; ++q) offer(q);
        }
      }
    } else {
      for (int dx = -ring; dx <= ring; ++dx) {
        const int x = c[0] + dx;
        if (x < 0 || x >= g.nc[0]) continue;
        for (int dy = -ring; dy <= ring; ++dy) {
          const int y = c[1] + dy;
          if (y < 0 || y >= g.nc[1]) continue;
          const bool shell = abs(dx) == ring || abs(dy) == ring;
          const int step = (shell || ring == 0) ? 1 : 2 * ring;
          for (int dz = -ring; dz <= ring; dz += step) {
            const int z = c[2] + dz;
            if (z < 0 || z >= g.nc[2]) continue;
            const int cell = base + (x * g.nc[1] + y) * g.nc[2] + z;
            for (int q = start[cell], e = start[cell + 1]; q < e; ++q) offer(q);
          }
        }
      }
    }
    // every unvisited point is at least ring * cell away (the point lies inside its home cell)
    if (have == k) {
      const float reach = ((float)ring - 0.02f) * g.cell;
      if (ring > 0 && ld[(k - 1) * kKnnThreads] < reach * reach) break;
    }
  }
  if (have < k) atomicOr(short_rows, 1);
  int32_t *nb = dir_out ? t_out : s_out, *me = dir_out ? s_out : t_out;
  for (int m = 0; m < k; ++m) {
    nb[(int64_t)i * k + m] = (m < have ? li[m * kKnnThreads] : i) + out_base;
    me[(int64_t)i * k + m] = i + out_base;
  }
}

// ---- space-filling-curve keys ---------------------------------------------------------------------------------------
struct Quant {
  float lo[3];
  float scale[3];
  int bits;
};

__device__ __forceinline__ unsigned hilbert2(unsigned x, unsigned y, int bits) {   // position along the Hilbert curve
  unsigned d = 0;
  const unsigned side = 1u << bits;
  for (unsigned s = side >> 1; s > 0; s >>= 1) {
    const unsigned rx = (x & s) ? 1u : 0u, ry = (y & s) ? 1u : 0u;
    d += s * s * ((3u * rx) ^ ry);
    if (ry == 0) {
      if (rx == 1) {
        x = side - 1 - x;
        y = side - 1 - y;
      }
      const unsigned tmp = x;
      x = y;
      y = tmp;
    }
  }
  return d;
}

__device__ __forceinline__ unsigned spread3(unsigned v) {   // 10 bits -> every third bit
  v &= 0x3ffu;
  v = (v | (v << 16)) & 0x030000ffu;
  v = (v | (v << 8)) & 0x0300f00fu;
  v = (v | (v << 4)) & 0x030c30c3u;
  v = (v | (v << 2)) & 0x09249249u;
  return v;
}

template <int DIM>
__global__ void curve_key_kernel(int64_t n, Quant qz, const float *__restrict__ pts, const int32_t *__restrict__ gid, int id_base,
                                 unsigned long long *__restrict__ key, int32_t *__restrict__ iota) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned q[DIM];
  const int top = (int)((1u << qz.bits) - 1u);
#pragma unroll
  for (int d = 0; d < DIM; ++d) q[d] = (unsigned)min(top, max(0, (int)((pts[i * DIM + d] - qz.lo[d]) * qz.scale[d])));
  unsigned code;
  if constexpr (DIM == 1) code = q[0];
  else if constexpr (DIM == 2) code = hilbert2(q[0], q[1], qz.bits);
  else code = (spread3(q[0]) << 2) | (spread3(q[1]) << 1) | spread3(q[2]);
  const unsigned long long graph = gid ? (unsigned long long)(gid[i] - id_base) : 0ull;
  key[i] = (graph << 32) | code;
  iota[i] = (int32_t)i;
}

int32_t bounding_box(int64_t n, int dim, const float *pts, const int32_t *gid, int id_base, int n_graphs, hipStream_t stream,
                     Scratch &sc, float *lo, float *hi) {
  unsigned *box = nullptr;
  int *bad = nullptr;
  int32_t st;
  if ((st = sc.get(&box, 6)) || (st = sc.get(&bad, 1))) return st;
  const unsigned init[6] = {0xffffffffu, 0xffffffffu, 0xffffffffu, 0u, 0u, 0u};
  NGPDE_HIP_CHECK(hipMemcpyAsync(box, init, sizeof(init), hipMemcpyHostToDevice, stream));
  NGPDE_HIP_CHECK(hipMemsetAsync(bad, 0, sizeof(int), stream));
  hipLaunchKernelGGL(bbox_kernel, dim3(std::min<unsigned>(blocks_for(n), 1024u)), dim3(kB), 0, stream, n, dim, pts, gid, id_base,
                     n_graphs, box, bad);
  NGPDE_LAUNCH_CHECK("bbox_kernel");
  unsigned h[6];
  int h_bad = 0;
  NGPDE_HIP_CHECK(hipMemcpyAsync(h, box, sizeof(h), hipMemcpyDeviceToHost, stream));
  NGPDE_HIP_CHECK(hipMemcpyAsync(&h_bad, bad, sizeof(int), hipMemcpyDeviceToHost, stream));
  NGPDE_HIP_CHECK(hipStreamSynchronize(stream));
  NGPDE_REQUIRE(!h_bad, NGPDE_ERR_INVALID_ARGUMENT, "graph_indicator holds an id outside %d:%d", id_base, id_base + n_graphs - 1);
  for (int d = 0; d < 3; ++d) {
    lo[d] = d < dim ? from_ordered_bits(h[d]) : 0.f;
    hi[d] = d < dim ? from_ordered_bits(h[3 + d]) : 0.f;
    NGPDE_REQUIRE(std::isfinite(lo[d]) && std::isfinite(hi[d]), NGPDE_ERR_INVALID_ARGUMENT,
                  "points hold a non-finite coordinate");
  }
  return NGPDE_OK;
}

// cell edge >= want, at most ~budget cells in total over all graphs, at most kMaxCellsPerDim per dimension: a cell
// coordinate (p - lo) * inv then carries an absolute rounding error <= 8192 * 2^-23 = 1e-3 cells, inside the 1 % by which
// the cell edge exceeds the radius (and inside the margin of the k-NN ring bound)
constexpr int kMaxCellsPerDim = 8192;
Grid make_grid(int dim, const float *lo, const float *hi, float want, int64_t budget, int n_graphs) {
  Grid g{};
  float cell = want;
  float ext_max = 0.f;
  for (int d = 0; d < dim; ++d) ext_max = std::max(ext_max, hi[d] - lo[d]);
  if (!std::isfinite(cell)) cell = std::max(ext_max, 1.f) * 2.f;              // everything in one cell
  if (!(cell > 0.f)) cell = ext_max > 0.f ? ext_max / (float)kMaxCellsPerDim : 1.f;
  cell = std::max(cell, ext_max / (float)kMaxCellsPerDim);
  for (;;) {
    int64_t cells = 1;
    for (int d = 0; d < 3; ++d) {
      g.nc[d] = 1;
      if (d < dim) g.nc[d] = (int)std::min<double>((double)kMaxCellsPerDim, std::floor((double)(hi[d] - lo[d]) / cell) + 1.0);
      cells *= g.nc[d];
    }
    if (cells * n_graphs <= budget || cells == 1) {
      g.cells = (int)cells;
      break;
    }
    cell *= 1.26f;
  }
  for (int d = 0; d < 3; ++d) g.lo[d] = lo[d];
  g.cell = cell;
  g.inv = 1.0f / cell;
  return g;
}

// total cells over all graphs: ~4 per point, and (graph, cell) keys must fit 32 bits
inline int64_t cell_budget(int64_t n) { return std::min<int64_t>(std::max<int64_t>(4 * n, 1 << 16), (int64_t)1 << 30); }

struct Sorted {
  unsigned *key = nullptr;    // sorted (graph, cell) keys
  int32_t *idx = nullptr;     // original index per sorted position
  float *pts = nullptr;       // points in sorted order
  int32_t *start = nullptr;   // [total cells + 1]
};

template <int DIM>
int32_t sort_into_cells(int64_t n, const Grid &g, int n_graphs, const float *pts, const int32_t *gid, int id_base,
                        hipStream_t stream, Scratch &sc, Sorted &out) {
  const int64_t total = (int64_t)g.cells * n_graphs;
  unsigned *key = nullptr;
  int32_t *iota = nullptr;
  int32_t st;
  if ((st = sc.get(&key, (size_t)n)) || (st = sc.get(&iota, (size_t)n)) || (st = sc.get(&out.key, (size_t)n)) ||
      (st = sc.get(&out.idx, (size_t)n)) || (st = sc.get(&out.pts, (size_t)n * DIM)) || (st = sc.get(&out.start, (size_t)total + 1)))
    return st;
  hipLaunchKernelGGL(cell_key_kernel<DIM>, dim3(blocks_for(n)), dim3(kB), 0, stream, n, g, pts, gid, id_base, key, iota);
  NGPDE_LAUNCH_CHECK("cell_key_kernel");
  size_t sb = 0;
  const unsigned end_bit = (unsigned)bits_for(std::max<int64_t>(total, 2));
  NGPDE_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, sb, key, out.key, iota, out.idx, (size_t)n, 0u, end_bit, stream));
  void *tmp = nullptr;
  if ((st = sc.get((char **)&tmp, sb))) return st;
  NGPDE_HIP_CHECK(rocprim::radix_sort_pairs(tmp, sb, key, out.key, iota, out.idx, (size_t)n, 0u, end_bit, stream));
  hipLaunchKernelGGL(gather_points_kernel<DIM>, dim3(blocks_for(n)), dim3(kB), 0, stream, n, pts, out.idx, out.pts);
  NGPDE_LAUNCH_CHECK("gather_points_kernel");
  hipLaunchKernelGGL(cell_start_kernel, dim3(blocks_for(total + 1)), dim3(kB), 0, stream, total, n, out.key, out.start);
  NGPDE_LAUNCH_CHECK("cell_start_kernel");
  return NGPDE_OK;
}

template <int DIM>
int32_t radius_graph_impl(int64_t n, const float *pts, float r, const int32_t *gid, int n_graphs, int id_base, int self_loops,
                          int dir_out, int out_base, int64_t capacity, int32_t *s, int32_t *t, int64_t *n_edges,
                          hipStream_t stream) {
  Scratch sc;
  float lo[3], hi[3];
  int32_t st;
  if ((st = bounding_box(n, DIM, pts, gid, id_base, n_graphs, stream, sc, lo, hi))) return st;
  const Grid g = make_grid(DIM, lo, hi, r * 1.01f, cell_budget(n), n_graphs);
  Sorted so;
  if ((st = sort_into_cells<DIM>(n, g, n_graphs, pts, gid, id_base, stream, sc, so))) return st;
  int32_t *deg = nullptr, *rowptr = nullptr;
  unsigned long long *total = nullptr;
  if ((st = sc.get(&deg, (size_t)n + 1)) || (st = sc.get(&rowptr, (size_t)n + 1)) || (st = sc.get(&total, 1))) return st;
  NGPDE_HIP_CHECK(hipMemsetAsync(total, 0, sizeof(unsigned long long), stream));
  NGPDE_HIP_CHECK(hipMemsetAsync(deg, 0, ((size_t)n + 1) * sizeof(int32_t), stream));
  const float r2 = r * r;
  hipLaunchKernelGGL((radius_kernel<DIM, false>), dim3(blocks_for(n)), dim3(kB), 0, stream, n, g, r2, self_loops, out_base, so.pts,
                     so.idx, so.key, so.start, deg, total, (const int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr);
  NGPDE_LAUNCH_CHECK("radius_kernel(count)");
  unsigned long long h_total = 0;
  NGPDE_HIP_CHECK(hipMemcpyAsync(&h_total, total, sizeof(h_total), hipMemcpyDeviceToHost, stream));
  NGPDE_HIP_CHECK(hipStreamSynchronize(stream));
  *n_edges = (int64_t)h_total;
  if (!s && !t) return NGPDE_OK;   // count only
  NGPDE_REQUIRE(h_total <= 0x7fffffffull, NGPDE_ERR_UNSUPPORTED, "radius graph has %llu edges: more than int32 positions", h_total);
  NGPDE_REQUIRE((int64_t)h_total <= capacity, NGPDE_ERR_INVALID_ARGUMENT,
                "radius graph has %llu edges, the output arrays hold %lld", h_total, (long long)capacity);
  if (h_total == 0) return NGPDE_OK;
  size_t tb = 0;
  NGPDE_HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, deg, rowptr, 0, (size_t)n + 1, rocprim::plus<int32_t>(), stream));
  void *tmp = nullptr;
  if ((st = sc.get((char **)&tmp, tb))) return st;
  NGPDE_HIP_CHECK(rocprim::exclusive_scan(tmp, tb, deg, rowptr, 0, (size_t)n + 1, rocprim::plus<int32_t>(), stream));
  int32_t *nbr = nullptr;
  if ((st = sc.get(&nbr, (size_t)h_total))) return st;
  int32_t *nb_out = dir_out ? t : s, *me_out = dir_out ? s : t;
  hipLaunchKernelGGL((radius_kernel<DIM, true>), dim3(blocks_for(n)), dim3(kB), 0, stream, n, g, r2, self_loops, out_base, so.pts,
                     so.idx, so.key, so.start, (int32_t *)nullptr, (unsigned long long *)nullptr, (const int32_t *)rowptr, nbr, me_out);
  NGPDE_LAUNCH_CHECK("radius_kernel(fill)");
  size_t sb = 0;
  const unsigned end_bit = (unsigned)bits_for(std::max<int64_t>(n + out_base + 1, 2));
  NGPDE_HIP_CHECK(rocprim::segmented_radix_sort_keys(nullptr, sb, nbr, nb_out, (unsigned)h_total, (unsigned)n, rowptr, rowptr + 1, 0u,
                                                     end_bit, stream));
  void *tmp2 = nullptr;
  if ((st = sc.get((char **)&tmp2, sb))) return st;
  NGPDE_HIP_CHECK(rocprim::segmented_radix_sort_keys(tmp2, sb, nbr, nb_out, (unsigned)h_total, (unsigned)n, rowptr, rowptr + 1, 0u,
                                                     end_bit, stream));
  NGPDE_HIP_CHECK(hipStreamSynchronize(stream));
  return NGPDE_OK;
}

template <int DIM>
int32_t knn_graph_impl(int64_t n, const float *pts, int k, const int32_t *gid, int n_graphs, int id_base, int self_loops, int dir_out,
                       int out_base, int32_t *s, int32_t *t, hipStream_t stream) {
  Scratch sc;
  float lo[3], hi[3];
  int32_t st;
  if ((st = bounding_box(n, DIM, pts, gid, id_base, n_graphs, stream, sc, lo, hi))) return st;
  // about k/2 points per cell on average: the first ring usually settles a point
  double vol = 1.0;
  for (int d = 0; d < DIM; ++d) vol *= std::max((double)hi[d] - (double)lo[d], 1e-30);
  const double per_graph = std::max(1.0, (double)n / n_graphs);
  const float want = (float)std::pow(vol * std::max(1.0, 0.5 * k) / per_graph, 1.0 / DIM);
  const Grid g = make_grid(DIM, lo, hi, want, cell_budget(n), n_graphs);
  Sorted so;
  if ((st = sort_into_cells<DIM>(n, g, n_graphs, pts, gid, id_base, stream, sc, so))) return st;
  int *short_rows = nullptr;
  if ((st = sc.get(&short_rows, 1))) return st;
  NGPDE_HIP_CHECK(hipMemsetAsync(short_rows, 0, sizeof(int), stream));
  const size_t lds = (size_t)k * kKnnThreads * 8;
  NGPDE_HIP_CHECK(hipFuncSetAttribute((const void *)knn_kernel<DIM>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  hipLaunchKernelGGL(knn_kernel<DIM>, dim3((unsigned)((n + kKnnThreads - 1) / kKnnThreads)), dim3(kKnnThreads), lds, stream, n, g, k,
                     self_loops, out_base, dir_out, so.pts, so.idx, so.key, so.start, s, t, short_rows);
  NGPDE_LAUNCH_CHECK("knn_kernel");
  int h_short = 0;
  NGPDE_HIP_CHECK(hipMemcpyAsync(&h_short, short_rows, sizeof(int), hipMemcpyDeviceToHost, stream));
  NGPDE_HIP_CHECK(hipStreamSynchronize(stream));
  NGPDE_REQUIRE(!h_short, NGPDE_ERR_INVALID_ARGUMENT, "knn_graph: a graph has fewer than k%s points", self_loops ? "" : " + 1");
  return NGPDE_OK;
}

template <int DIM>
int32_t spatial_order_impl(int64_t n, const float *pts, const int32_t *gid, int n_graphs, int id_base, int32_t *order,
                           hipStream_t stream) {
  Scratch sc;
  float lo[3], hi[3];
  int32_t st;
  if ((st = bounding_box(n, DIM, pts, gid, id_base, n_graphs, stream, sc, lo, hi))) return st;
  Quant qz{};
  qz.bits = DIM == 1 ? 30 : (DIM == 2 ? 16 : 10);
  for (int d = 0; d < DIM; ++d) {
    const float ext = hi[d] - lo[d];
    qz.lo[d] = lo[d];
    qz.scale[d] = ext > 0.f ? (float)(1u << qz.bits) / ext : 0.f;
  }
  unsigned long long *key = nullptr, *key_sorted = nullptr;
  int32_t *iota = nullptr;
  if ((st = sc.get(&key, (size_t)n)) || (st = sc.get(&key_sorted, (size_t)n)) || (st = sc.get(&iota, (size_t)n))) return st;
  hipLaunchKernelGGL(curve_key_kernel<DIM>, dim3(blocks_for(n)), dim3(kB), 0, stream, n, qz, pts, gid, id_base, key, iota);
  NGPDE_LAUNCH_CHECK("curve_key_kernel");
  size_t sb = 0;
  const unsigned end_bit = 32u + (unsigned)bits_for(std::max<int64_t>(n_graphs, 2));
  NGPDE_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, sb, key, key_sorted, iota, order, (size_t)n, 0u, end_bit, stream));
  void *tmp = nullptr;
  if ((st = sc.get((char **)&tmp, sb))) return st;
  NGPDE_HIP_CHECK(rocprim::radix_sort_pairs(tmp, sb, key, key_sorted, iota, order, (size_t)n, 0u, end_bit, stream));
  NGPDE_HIP_CHECK(hipStreamSynchronize(stream));
  return NGPDE_OK;
}

int32_t check_common(int64_t n, int32_t dim, const float *pts, const int32_t *gid, int32_t n_graphs) {
  NGPDE_REQUIRE(n >= 0 && n <= 0x7fffffffLL, NGPDE_ERR_INVALID_ARGUMENT, "number of points %lld outside 0:2^31-1", (long long)n);
  NGPDE_REQUIRE(dim >= 1 && dim <= 3, NGPDE_ERR_UNSUPPORTED, "neighbour search supports 1, 2 or 3 coordinates, got %d", dim);
  NGPDE_REQUIRE(n == 0 || pts != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "points is NULL");
  NGPDE_REQUIRE(n_graphs >= 1, NGPDE_ERR_INVALID_ARGUMENT, "n_graphs must be >= 1");
  NGPDE_REQUIRE(gid != nullptr || n_graphs == 1, NGPDE_ERR_INVALID_ARGUMENT, "n_graphs > 1 needs a graph_indicator");
  return NGPDE_OK;
}

}  // namespace
}  // namespace ngpde

using namespace ngpde;

extern "C" {

int32_t ngpde_radius_graph(int64_t n, int32_t dim, const float *points, float r, const int32_t *graph_id, int32_t n_graphs,
                           int32_t id_base, int32_t self_loops, int32_t dir_out, int32_t index_base, int64_t capacity, int32_t *s,
                           int32_t *t, int64_t *n_edges, ngpde_stream_t stream) {
  NGPDE_RANGE();
  int32_t st = check_common(n, dim, points, graph_id, n_graphs);
  if (st) return st;
  NGPDE_REQUIRE(n_edges != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "n_edges is NULL");
  NGPDE_REQUIRE(r >= 0.f, NGPDE_ERR_INVALID_ARGUMENT, "radius must be >= 0 (got %g)", (double)r);
  NGPDE_REQUIRE((s == nullptr) == (t == nullptr), NGPDE_ERR_INVALID_ARGUMENT, "s and t must both be given or both be NULL");
  *n_edges = 0;
  if (n == 0) return NGPDE_OK;
  hipStream_t hs = (hipStream_t)stream;
  switch (dim) {
    case 1: return radius_graph_impl<1>(n, points, r, graph_id, n_graphs, id_base, self_loops, dir_out, index_base, capacity, s, t, n_edges, hs);
    case 2: return radius_graph_impl<2>(n, points, r, graph_id, n_graphs, id_base, self_loops, dir_out, index_base, capacity, s, t, n_edges, hs);
    default: return radius_graph_impl<3>(n, points, r, graph_id, n_graphs, id_base, self_loops, dir_out, index_base, capacity, s, t, n_edges, hs);
  }
}

int32_t ngpde_knn_graph(int64_t n, int32_t dim, const float *points, int32_t k, const int32_t *graph_id, int32_t n_graphs,
                        int32_t id_base, int32_t self_loops, int32_t dir_out, int32_t index_base, int32_t *s, int32_t *t,
                        ngpde_stream_t stream) {
  NGPDE_RANGE();
  int32_t st = check_common(n, dim, points, graph_id, n_graphs);
  if (st) return st;
  NGPDE_REQUIRE(k >= 0 && k <= NGPDE_KNN_MAX_K, NGPDE_ERR_UNSUPPORTED, "knn_graph supports 0 <= k <= %d, got %d", NGPDE_KNN_MAX_K, k);
  NGPDE_REQUIRE(n * (int64_t)k <= 0x7fffffffLL, NGPDE_ERR_UNSUPPORTED, "knn graph has more than int32 positions");
  if (n == 0 || k == 0) return NGPDE_OK;
  NGPDE_REQUIRE(s != nullptr && t != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "s or t is NULL");
  NGPDE_REQUIRE(n >= (int64_t)k + (self_loops ? 0 : 1), NGPDE_ERR_INVALID_ARGUMENT, "knn_graph: a graph has fewer than k%s points",
                self_loops ? "" : " + 1");
  hipStream_t hs = (hipStream_t)stream;
  switch (dim) {
    case 1: return knn_graph_impl<1>(n, points, k, graph_id, n_graphs, id_base, self_loops, dir_out, index_base, s, t, hs);
    case 2: return knn_graph_impl<2>(n, points, k, graph_id, n_graphs, id_base, self_loops, dir_out, index_base, s, t, hs);
    default: return knn_graph_impl<3>(n, points, k, graph_id, n_graphs, id_base, self_loops, dir_out, index_base, s, t, hs);
  }
}

int32_t ngpde_spatial_order(int64_t n, int32_t dim, const float *points, const int32_t *graph_id, int32_t n_graphs, int32_t id_base,
                            int32_t *order, ngpde_stream_t stream) {
  NGPDE_RANGE();
  int32_t st = check_common(n, dim, points, graph_id, n_graphs);
  if (st) return st;
  if (n == 0) return NGPDE_OK;
  NGPDE_REQUIRE(order != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "order is NULL");
  hipStream_t hs = (hipStream_t)stream;
  switch (dim) {
    case 1: return spatial_order_impl<1>(n, points, graph_id, n_graphs, id_base, order, hs);
    case 2: return spatial_order_impl<2>(n, points, graph_id, n_graphs, id_base, order, hs);
    default: return spatial_order_impl<3>(n, points, graph_id, n_graphs, id_base, order, hs);
  }
}

}  // extern "C"
