// graph_device.hip -- derived-graph handle built ON THE DEVICE from a COO list that already lives in HBM
// (training loops swap the graph every minibatch, /root/reference/docs/src/tutorials/
// VMH.md:132-134 `st = updategraph(st, g)` after `g |> gpu`; the host builder in graph.hip costs 3.6 ms for the C2
// graph and 104 ms for a 64-trajectory C4 batch, more than the layer it feeds).
//
// Same arrays, bit for bit, as the host builder (tests compare every one):
//   CSR by target / by source   stable LSD radix sort of (key, COO position) pairs (rocPRIM) => COO order inside a row
//   xpos                        inverse-permutation kernels
//   c, ent, sched, ell          one thread per node / entry / schedule row
//   halo lists + slot bytes     one workgroup per 32-row tile: first-occurrence ranks of the tile's distinct columns
// The locality order (BFS-grown clusters, graph.hip: locality_order) is the one sequential step: it is taken from the
// caller when supplied (a batch of graphs reuses the cached order of each member, offset by its first node; so does a
// graph seen in an earlier minibatch), and computed on the host from the downloaded CSR otherwise.
#include <algorithm>
#include <cstring>
#include <vector>

#include <rocprim/rocprim.hpp>

#include "common.h"

namespace ngpde {

std::vector<int32_t> locality_order_host(int64_t n, const std::vector<int32_t> &rp_in, const std::vector<int32_t> &col_in,
                                         const std::vector<int32_t> &rp_out, const std::vector<int32_t> &col_out, int tile);

namespace {

constexpr int kB = 256;
inline unsigned blocks_for(int64_t n) { return (unsigned)std::max<int64_t>(1, (n + kB - 1) / kB); }

template <class I>
__global__ void convert_kernel(int64_t m, int64_t n, int base, const I *__restrict__ s, const I *__restrict__ t,
                               int32_t *__restrict__ s32, int32_t *__restrict__ t32, int32_t *__restrict__ iota,
                               unsigned long long *__restrict__ first_bad) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= m) return;
  const int64_t a = (int64_t)s[e] - base, b = (int64_t)t[e] - base;
  if (a < 0 || a >= n || b < 0 || b >= n) {
    atomicMin(first_bad, (unsigned long long)e);
    s32[e] = 0;
    t32[e] = 0;
  } else {
    s32[e] = (int32_t)a;
    t32[e] = (int32_t)b;
  }
  iota[e] = (int32_t)e;
}

__global__ void count_kernel(int64_t m, const int32_t *__restrict__ key, int32_t *__restrict__ deg) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e < m) atomicAdd(&deg[key[e]], 1);
}

__global__ void gather_col_kernel(int64_t m, const int32_t *__restrict__ eid, const int32_t *__restrict__ other,
                                  int32_t *__restrict__ col, int32_t *__restrict__ pos_of_edge) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  const int32_t e = eid[p];
  col[p] = other[e];
  pos_of_edge[e] = (int32_t)p;
}

__global__ void xpos_kernel(int64_t m, const int32_t *__restrict__ eid, const int32_t *__restrict__ pos_other,
                            int32_t *__restrict__ xpos) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p < m) xpos[p] = pos_other[eid[p]];
}

__global__ void max_degree_kernel(int64_t n, const int32_t *__restrict__ rowptr, int32_t *__restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int d = (i < n) ? rowptr[i + 1] - rowptr[i] : 0;
  for (int off = 32; off > 0; off >>= 1) d = max(d, __shfl_xor(d, off));
  if ((threadIdx.x & 63) == 0 && d > 0) atomicMax(out, d);
}

// d = degree(g; dir = :in[, edge_weight]) with the self loop's weight 1 (src/layers.jl:210-224), summed in float in COO
// order inside the row as scatter(+) does; c = 1 / sqrt(d)
__global__ void norm_kernel(int64_t n, int self_loops, int weighted, const int32_t *__restrict__ rowptr_t,
                            const int32_t *__restrict__ eid_t, const float *__restrict__ w, float *__restrict__ c) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float acc = 0.f;
  for (int32_t p = rowptr_t[i]; p < rowptr_t[i + 1]; ++p) acc += weighted ? w[eid_t[p]] : 1.0f;
  const float d = acc + (self_loops ? 1.0f : 0.0f);
  c[i] = 1.0f / sqrtf(d);
}

__global__ void ent_kernel(int64_t m, const int32_t *__restrict__ col, const int32_t *__restrict__ eid,
                           const float *__restrict__ w, const float *__restrict__ c, int2 *__restrict__ ent) {
  const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= m) return;
  const int32_t v = col[p];
  const float coef = (w ? w[eid[p]] : 1.0f) * c[v];
  ent[p] = make_int2(v, __float_as_int(coef));
}

__global__ void sched_kernel(int64_t n, int64_t n_sched, const int32_t *__restrict__ order, const int32_t *__restrict__ rowptr,
                             const float *__restrict__ c, const int2 *__restrict__ ent, int4 *__restrict__ sched,
                             int2 *__restrict__ ell) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_sched) return;
  int4 e = make_int4(-1, 0, 0, 0);
  int rs = 0, deg = 0;
  if (k < n) {
    const int32_t v = order[k];
    rs = rowptr[v];
    deg = rowptr[v + 1] - rs;
    e = make_int4(v, rs, deg, __float_as_int(c[v]));
  }
  sched[k] = e;
  for (int j = 0; j < kEllWidth; ++j) ell[k * kEllWidth + j] = (j < deg) ? ent[rs + j] : make_int2(0, 0);
}

// One workgroup per tile.  Slot numbering of the host builder: the tile's own rows take slots 0..31 (padding rows
// included), then every other distinct column in order of first appearance while scanning rows 0..31, entries in
// row order.  info.x = slot count if the tile fits (all degrees <= kSlotWidth, count <= kHaloCap), else 0.
__global__ __launch_bounds__(256) void halo_kernel(int64_t n, const int32_t *__restrict__ order, const int32_t *__restrict__ rowptr,
                                                   const int32_t *__restrict__ col, const int32_t *__restrict__ eid,
                                                   const float *__restrict__ c, const float *__restrict__ w,
                                                   int2 *__restrict__ halo, int2 *__restrict__ info,
                                                   uint8_t *__restrict__ slots, float *__restrict__ slot_w) {
  constexpr int kMax = kTileRows * kSlotWidth;
  __shared__ int own[kTileRows], rs[kTileRows], off[kTileRows + 1];
  __shared__ int L[kMax], slot[kMax], firstflag[kMax], scan[kMax];
  __shared__ int too_wide;
  const int64_t tl = blockIdx.x;
  const int tid = threadIdx.x;
  if (tid == 0) too_wide = 0;
  __syncthreads();
  if (tid < kTileRows) {
    const int64_t pos = tl * kTileRows + tid;
    int v = -1, r = 0, d = 0;
    if (pos < n) {
      v = order[pos];
      r = rowptr[v];
      d = rowptr[v + 1] - r;
      if (d > kSlotWidth) atomicOr(&too_wide, 1);
    }
    own[tid] = v;
    rs[tid] = r;
    off[tid + 1] = min(d, kSlotWidth);
  }
  // halo / slots defaults (as the host builder's assign())
  for (int s = tid; s < kHaloCap; s += 256) halo[tl * kHaloCap + s] = make_int2(0, 0);
  for (int s = tid; s < kTileRows * kSlotWidth; s += 256) {
    slots[tl * kTileRows * kSlotWidth + s] = (uint8_t)kHaloCap;
    if (slot_w) slot_w[tl * kTileRows * kSlotWidth + s] = 0.f;
  }
  __syncthreads();
  if (tid == 0) {
    off[0] = 0;
    for (int k = 0; k < kTileRows; ++k) off[k + 1] += off[k];
  }
  __syncthreads();
  const int total = off[kTileRows];
  // own rows -> halo slots 0..31
  if (tid < kTileRows && own[tid] >= 0) halo[tl * kHaloCap + tid] = make_int2(own[tid], __float_as_int(c[own[tid]]));
  if (too_wide) {   // the host builder stops at the first over-wide row: mark the tile as not fitting
    if (tid == 0) info[tl] = make_int2(0, 0);
    return;
  }
  // the tile's entries in scan order
  for (int k = 0; k < kTileRows; ++k)
    for (int j = tid; j < off[k + 1] - off[k]; j += 256) L[off[k] + j] = col[rs[k] + j];
  __syncthreads();
  // slot of own columns; first-occurrence flags of the others
  for (int i = tid; i < total; i += 256) {
    const int v = L[i];
    int sl = -1;
    for (int k = 0; k < kTileRows; ++k)
      if (own[k] == v) { sl = k; break; }
    int first = 0;
    if (sl < 0) {
      first = 1;
      for (int j = 0; j < i; ++j)
        if (L[j] == v) { first = 0; break; }
    }
    slot[i] = sl;
    firstflag[i] = first;
  }
  __syncthreads();
  // inclusive prefix sum of the flags (<= 1024 entries): Hillis-Steele in LDS
  for (int i = tid; i < total; i += 256) scan[i] = firstflag[i];
  __syncthreads();
  for (int d = 1; d < total; d <<= 1) {
    int v[4];
    int cnt = 0;
    for (int i = tid; i < total; i += 256) v[cnt++] = scan[i] + (i >= d ? scan[i - d] : 0);
    __syncthreads();
    cnt = 0;
    for (int i = tid; i < total; i += 256) scan[i] = v[cnt++];
    __syncthreads();
  }
  const int distinct = total > 0 ? scan[total - 1] : 0;
  const int count = kTileRows + distinct;
  for (int i = tid; i < total; i += 256)
    if (firstflag[i]) {
      const int sl = kTileRows + scan[i] - 1;
      slot[i] = sl;
      if (sl < kHaloCap) halo[tl * kHaloCap + sl] = make_int2(L[i], __float_as_int(c[L[i]]));
    }
  __syncthreads();
  for (int i = tid; i < total; i += 256)
    if (slot[i] < 0) {
      const int v = L[i];
      for (int j = 0; j < i; ++j)
        if (L[j] == v) { slot[i] = slot[j]; break; }   // L[j] is the first occurrence: its slot is final
    }
  __syncthreads();
  const bool fits = count <= kHaloCap;
  if (fits) {
    for (int k = 0; k < kTileRows; ++k)
      for (int j = tid; j < off[k + 1] - off[k]; j += 256) {
        const int64_t pos = tl * kTileRows + k;
        slots[pos * kSlotWidth + j] = (uint8_t)slot[off[k] + j];
        if (slot_w) slot_w[pos * kSlotWidth + j] = w[eid[rs[k] + j]];
      }
  }
  if (tid == 0) info[tl] = make_int2(fits ? count : 0, 0);
}

// bad[0] |= a tile does not fit; bad[2] = max halo count
__global__ void all_fit_kernel(int64_t n_tiles, const int2 *__restrict__ info, int32_t *__restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_tiles) return;
  if (info[i].x == 0) atomicOr(bad, 1);
  else atomicMax(bad + 2, info[i].x);
}

// a caller-supplied node order must be a permutation of 0..n-1: the schedule kernels index with it
__global__ void order_check_kernel(int64_t n, const int32_t *__restrict__ order, int32_t *__restrict__ seen, int32_t *__restrict__ bad) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int32_t o = order[i];
  if (o < 0 || o >= n) atomicOr(bad, 1);
  else if (atomicAdd(&seen[o], 1) != 0) atomicOr(bad, 1);
}

template <class T>
int32_t dalloc(T **p, size_t count) {
  *p = nullptr;
  NGPDE_HIP_CHECK(hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
  return NGPDE_OK;
}

struct Scratch {   // frees on scope exit
  std::vector<void *> ptrs;
  ~Scratch() {
    for (void *p : ptrs) (void)hipFree(p);
  }
  template <class T>
  int32_t get(T **p, size_t count) {
    int32_t st = dalloc(p, count);
    if (!st) ptrs.push_back(*p);
    return st;
  }
};

int bits_for(int64_t n) {
  int b = 1;
  while (((int64_t)1 << b) < n) ++b;
  return b;
}

// CSR of one direction: stable sort of the COO positions by `key`
int32_t build_csr_device(int64_t n, int64_t m, const int32_t *key, const int32_t *other, const int32_t *iota, Csr &out,
                         int32_t *pos_of_edge, Scratch &sc, hipStream_t stream) {
  int32_t st;
  if ((st = dalloc(&out.rowptr, (size_t)n + 1)) || (st = dalloc(&out.col, (size_t)m)) || (st = dalloc(&out.eid, (size_t)m)))
    return st;
  int32_t *deg = nullptr, *keys_sorted = nullptr;
  if ((st = sc.get(&deg, (size_t)n + 1)) || (st = sc.get(&keys_sorted, (size_t)m))) return st;
  NGPDE_HIP_CHECK(hipMemsetAsync(deg, 0, ((size_t)n + 1) * sizeof(int32_t), stream));
  if (m > 0) {
    hipLaunchKernelGGL(count_kernel, dim3(blocks_for(m)), dim3(kB), 0, stream, m, key, deg);
    NGPDE_LAUNCH_CHECK("count_kernel");
  }
  size_t tb = 0;
  NGPDE_HIP_CHECK(rocprim::exclusive_scan(nullptr, tb, deg, out.rowptr, 0, (size_t)n + 1, rocprim::plus<int32_t>(), stream));
  void *tmp = nullptr;
  if ((st = sc.get((char **)&tmp, tb))) return st;
  NGPDE_HIP_CHECK(rocprim::exclusive_scan(tmp, tb, deg, out.rowptr, 0, (size_t)n + 1, rocprim::plus<int32_t>(), stream));
  if (m > 0) {
    size_t sb = 0;
    const unsigned end_bit = (unsigned)bits_for(std::max<int64_t>(n, 2));
    NGPDE_HIP_CHECK(rocprim::radix_sort_pairs(nullptr, sb, key, keys_sorted, iota, out.eid, (size_t)m, 0u, end_bit, stream));
    void *tmp2 = nullptr;
    if ((st = sc.get((char **)&tmp2, sb))) return st;
    NGPDE_HIP_CHECK(rocprim::radix_sort_pairs(tmp2, sb, key, keys_sorted, iota, out.eid, (size_t)m, 0u, end_bit, stream));
    hipLaunchKernelGGL(gather_col_kernel, dim3(blocks_for(m)), dim3(kB), 0, stream, m, out.eid, other, out.col, pos_of_edge);
    NGPDE_LAUNCH_CHECK("gather_col_kernel");
  }
  return NGPDE_OK;
}

template <class T>
int32_t download(std::vector<T> &dst, const T *src, size_t count, hipStream_t stream) {
  dst.resize(count);
  if (count) NGPDE_HIP_CHECK(hipMemcpyAsync(dst.data(), src, count * sizeof(T), hipMemcpyDeviceToHost, stream));
  return NGPDE_OK;
}

void free_norm(ngpde_graph *g) {
  for (Csr *c2 : {&g->by_t, &g->by_s}) {
    for (void **p : {(void **)&c2->halo, (void **)&c2->tile_info, (void **)&c2->slots, (void **)&c2->slot_w, (void **)&c2->ell,
                     (void **)&c2->ent, (void **)&c2->sched})
      if (*p) { (void)hipFree(*p); *p = nullptr; }
    c2->halo_ok = false;
  }
  if (g->c) { (void)hipFree(g->c); g->c = nullptr; }
  if (g->w_coo) { (void)hipFree(g->w_coo); g->w_coo = nullptr; }
  g->has_norm = false;
}

}  // namespace

// normalisation, schedule, fixed-width block and halo lists of a handle whose CSR lists and order are on the device
int32_t set_gcn_norm_device(ngpde_graph *g, int add_self_loops, const float *w_dev, int weighted_degree, hipStream_t stream) {
  const int64_t n = g->n_nodes, m = g->n_edges;
  free_norm(g);
  int32_t st;
  if ((st = dalloc(&g->c, (size_t)n))) return st;
  if (n > 0) {
    hipLaunchKernelGGL(norm_kernel, dim3(blocks_for(n)), dim3(kB), 0, stream, n, add_self_loops, weighted_degree, g->by_t.rowptr,
                       g->by_t.eid, w_dev, g->c);
    NGPDE_LAUNCH_CHECK("norm_kernel");
  }
  const int64_t n_tiles = g->n_sched / kTileRows;
  int32_t *bad = nullptr;
  Scratch sc;
  if ((st = sc.get(&bad, 8))) return st;   // per direction: {not-fitting flag, -, max halo, -}
  NGPDE_HIP_CHECK(hipMemsetAsync(bad, 0, 8 * sizeof(int32_t), stream));
  int dir = 0;
  for (Csr *c2 : {&g->by_t, &g->by_s}) {
    if ((st = dalloc(&c2->ent, (size_t)m)) || (st = dalloc(&c2->sched, (size_t)g->n_sched)) ||
        (st = dalloc(&c2->ell, (size_t)g->n_sched * kEllWidth)) || (st = dalloc(&c2->halo, (size_t)n_tiles * kHaloCap)) ||
        (st = dalloc(&c2->tile_info, (size_t)n_tiles)) || (st = dalloc(&c2->slots, (size_t)g->n_sched * kSlotWidth)))
      return st;
    if (w_dev && (st = dalloc(&c2->slot_w, (size_t)g->n_sched * kSlotWidth))) return st;
    if (m > 0) {
      hipLaunchKernelGGL(ent_kernel, dim3(blocks_for(m)), dim3(kB), 0, stream, m, c2->col, c2->eid, w_dev, g->c, c2->ent);
      NGPDE_LAUNCH_CHECK("ent_kernel");
    }
    if (g->n_sched > 0) {
      hipLaunchKernelGGL(sched_kernel, dim3(blocks_for(g->n_sched)), dim3(kB), 0, stream, n, (int64_t)g->n_sched, g->order,
                         c2->rowptr, g->c, c2->ent, c2->sched, c2->ell);
      NGPDE_LAUNCH_CHECK("sched_kernel");
      hipLaunchKernelGGL(halo_kernel, dim3((unsigned)n_tiles), dim3(256), 0, stream, n, g->order, c2->rowptr, c2->col, c2->eid,
                         g->c, w_dev, c2->halo, c2->tile_info, c2->slots, c2->slot_w);
      NGPDE_LAUNCH_CHECK("halo_kernel");
      hipLaunchKernelGGL(all_fit_kernel, dim3(blocks_for(n_tiles)), dim3(kB), 0, stream, n_tiles, c2->tile_info, bad + 4 * dir);
      NGPDE_LAUNCH_CHECK("all_fit_kernel");
    }
    ++dir;
  }
  int32_t h_bad[8] = {1, 0, 0, 0, 1, 0, 0, 0};
  NGPDE_HIP_CHECK(hipMemcpyAsync(h_bad, bad, sizeof(h_bad), hipMemcpyDeviceToHost, stream));
  NGPDE_HIP_CHECK(hipStreamSynchronize(stream));
  g->by_t.halo_ok = n_tiles > 0 && h_bad[0] == 0;
  g->by_s.halo_ok = n_tiles > 0 && h_bad[4] == 0;
  if (w_dev && m > 0 && !(g->by_t.halo_ok && g->by_s.halo_ok)) {   // (kept for the persistent solver's hub geometry: graph.hip)
    if ((st = dalloc(&g->w_coo, (size_t)m))) return st;
    NGPDE_HIP_CHECK(hipMemcpyAsync(g->w_coo, w_dev, (size_t)m * sizeof(float), hipMemcpyDeviceToDevice, stream));
    NGPDE_HIP_CHECK(hipStreamSynchronize(stream));
  }
  g->by_t.max_halo = h_bad[2];
  g->by_s.max_halo = h_bad[6];
  g->self_loops = add_self_loops ? 1 : 0;
  g->has_norm = true;
  return NGPDE_OK;
}

template <class I>
int32_t graph_create_device(int64_t n_nodes, int64_t n_edges, const I *s, const I *t, int index_base, int32_t n_graphs,
                            const int32_t *order_dev, hipStream_t stream, ngpde_graph **out) {
  ngpde_graph *g = new (std::nothrow) ngpde_graph();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "out of host memory");
  g->n_nodes = n_nodes;
  g->n_edges = n_edges;
  g->n_graphs = n_graphs;
  g->device_built = true;
  g->n_sched = (int32_t)(((n_nodes + kTileRows - 1) / kTileRows) * kTileRows);
  auto bail = [&](int32_t st) {
    ngpde_graph_destroy(g);
    return st;
  };
  Scratch sc;
  int32_t st;
  int32_t *s32 = nullptr, *t32 = nullptr, *iota = nullptr, *pos_t = nullptr, *pos_s = nullptr, *maxdeg = nullptr;
  unsigned long long *first_bad = nullptr;
  if ((st = sc.get(&s32, (size_t)n_edges)) || (st = sc.get(&t32, (size_t)n_edges)) || (st = sc.get(&iota, (size_t)n_edges)) ||
      (st = sc.get(&pos_t, (size_t)n_edges)) || (st = sc.get(&pos_s, (size_t)n_edges)) || (st = sc.get(&maxdeg, 2)) ||
      (st = sc.get(&first_bad, 1)))
    return bail(st);
  if (hipMemsetAsync(first_bad, 0xff, sizeof(unsigned long long), stream) != hipSuccess ||
      hipMemsetAsync(maxdeg, 0, 2 * sizeof(int32_t), stream) != hipSuccess)
    return bail(fail(NGPDE_ERR_HIP, "hipMemsetAsync failed"));
  if (n_edges > 0) {
    hipLaunchKernelGGL(convert_kernel<I>, dim3(blocks_for(n_edges)), dim3(kB), 0, stream, n_edges, n_nodes, index_base, s, t, s32,
                       t32, iota, first_bad);
    if (hipGetLastError() != hipSuccess) return bail(fail(NGPDE_ERR_HIP, "convert_kernel launch failed"));
  }
  if ((st = build_csr_device(n_nodes, n_edges, t32, s32, iota, g->by_t, pos_t, sc, stream)) ||
      (st = build_csr_device(n_nodes, n_edges, s32, t32, iota, g->by_s, pos_s, sc, stream)))
    return bail(st);
  if ((st = dalloc(&g->by_t.xpos, (size_t)n_edges)) || (st = dalloc(&g->by_s.xpos, (size_t)n_edges))) return bail(st);
  if (n_edges > 0) {
    hipLaunchKernelGGL(xpos_kernel, dim3(blocks_for(n_edges)), dim3(kB), 0, stream, n_edges, g->by_t.eid, pos_s, g->by_t.xpos);
    hipLaunchKernelGGL(xpos_kernel, dim3(blocks_for(n_edges)), dim3(kB), 0, stream, n_edges, g->by_s.eid, pos_t, g->by_s.xpos);
  }
  if (n_nodes > 0) {
    hipLaunchKernelGGL(max_degree_kernel, dim3(blocks_for(n_nodes)), dim3(kB), 0, stream, n_nodes, g->by_t.rowptr, maxdeg);
    hipLaunchKernelGGL(max_degree_kernel, dim3(blocks_for(n_nodes)), dim3(kB), 0, stream, n_nodes, g->by_s.rowptr, maxdeg + 1);
  }
  if (hipGetLastError() != hipSuccess) return bail(fail(NGPDE_ERR_HIP, "graph construction kernel launch failed"));
  unsigned long long h_bad = 0;
  int32_t h_max[2] = {0, 0};
  if (hipMemcpyAsync(&h_bad, first_bad, sizeof(h_bad), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipMemcpyAsync(h_max, maxdeg, sizeof(h_max), hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)
    return bail(fail(NGPDE_ERR_HIP, "graph construction failed: %s", hipGetErrorString(hipGetLastError())));
  if (h_bad != ~0ull)
    return bail(fail(NGPDE_ERR_DIMENSION_MISMATCH, "DimensionMismatch: edge %lld references a node outside 1:%lld",
                     (long long)h_bad + index_base, (long long)n_nodes));
  g->max_in_degree = h_max[0];
  g->max_out_degree = h_max[1];
  if ((st = dalloc(&g->order, (size_t)n_nodes))) return bail(st);
  if (order_dev) {
    int32_t *seen = nullptr;
    if ((st = sc.get(&seen, (size_t)n_nodes + 1))) return bail(st);
    if (hipMemsetAsync(seen, 0, ((size_t)n_nodes + 1) * sizeof(int32_t), stream) != hipSuccess)
      return bail(fail(NGPDE_ERR_HIP, "hipMemsetAsync failed"));
    int32_t h_seen = 0;
    if (n_nodes) {
      hipLaunchKernelGGL(order_check_kernel, dim3(blocks_for(n_nodes)), dim3(kB), 0, stream, n_nodes, order_dev, seen, seen + n_nodes);
      if (hipGetLastError() != hipSuccess || hipMemcpyAsync(&h_seen, seen + n_nodes, sizeof(int32_t), hipMemcpyDeviceToHost, stream) != hipSuccess ||
          hipStreamSynchronize(stream) != hipSuccess)
        return bail(fail(NGPDE_ERR_HIP, "checking the node order failed"));
    }
    if (h_seen) return bail(fail(NGPDE_ERR_INVALID_ARGUMENT, "order is not a permutation of the %lld nodes", (long long)n_nodes));
    if (n_nodes && hipMemcpyAsync(g->order, order_dev, (size_t)n_nodes * sizeof(int32_t), hipMemcpyDeviceToDevice, stream) != hipSuccess)
      return bail(fail(NGPDE_ERR_HIP, "copying the node order failed"));
  } else {   // the one sequential step: BFS-grown clusters on the host, from the downloaded lists
    std::vector<int32_t> rp_t, col_t, rp_s, col_s;
    if ((st = download(rp_t, g->by_t.rowptr, (size_t)n_nodes + 1, stream)) || (st = download(col_t, g->by_t.col, (size_t)n_edges, stream)) ||
        (st = download(rp_s, g->by_s.rowptr, (size_t)n_nodes + 1, stream)) || (st = download(col_s, g->by_s.col, (size_t)n_edges, stream)))
      return bail(st);
    if (hipStreamSynchronize(stream) != hipSuccess) return bail(fail(NGPDE_ERR_HIP, "download of the CSR lists failed"));
    g->h_order = locality_order_host(n_nodes, rp_t, col_t, rp_s, col_s, kTileRows);
    if (n_nodes && hipMemcpyAsync(g->order, g->h_order.data(), (size_t)n_nodes * sizeof(int32_t), hipMemcpyHostToDevice, stream) != hipSuccess)
      return bail(fail(NGPDE_ERR_HIP, "upload of the node order failed"));
    if (hipStreamSynchronize(stream) != hipSuccess) return bail(fail(NGPDE_ERR_HIP, "upload of the node order failed"));
  }
  *out = g;
  return NGPDE_OK;
}

template int32_t graph_create_device<int32_t>(int64_t, int64_t, const int32_t *, const int32_t *, int, int32_t, const int32_t *,
                                              hipStream_t, ngpde_graph **);
template int32_t graph_create_device<int64_t>(int64_t, int64_t, const int64_t *, const int64_t *, int, int32_t, const int32_t *,
                                              hipStream_t, ngpde_graph **);

}  // namespace ngpde
