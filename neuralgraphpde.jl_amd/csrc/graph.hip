// graph.hip -- derived-graph handle: COO (as held by a Julia GNNGraph) -> CSR by target + CSR by
// source, built once per graph instead of on every layer call (the reference re-runs add_self_loops,
// degree and the COO walk per call: /root/reference/src/layers.jl:211,224,228-232).
#include <algorithm>
#include <cmath>
#include <deque>
#include <cstring>
#include <new>

#include "common.h"
#include <dlfcn.h>

namespace ngpde {
const RoctxApi &roctx_api() {
  static const RoctxApi api = [] {
    RoctxApi a;
    // Ranges are for profiling runs: on when the marker library is ALREADY in the process (rocprofv3 preloads it: RTLD_NOLOAD finds
    // it without loading anything) or when NGPDE_ROCTX=1 asks for it; a production process never pulls a profiler library in, and
    // never with global symbol visibility.  NGPDE_NO_ROCTX=1 turns them off even under a profiler.
    const char *off = std::getenv("NGPDE_NO_ROCTX");
    if (off && off[0] == '1') return a;
    void *h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_LOCAL | RTLD_NOLOAD);
    const char *on = std::getenv("NGPDE_ROCTX");
    if (!h && on && on[0] == '1') {
      h = dlopen("librocprofiler-sdk-roctx.so", RTLD_NOW | RTLD_LOCAL);
      if (!h) h = dlopen("librocprofiler-sdk-roctx.so.1", RTLD_NOW | RTLD_LOCAL);
    }
    if (!h) return a;
    a.push = reinterpret_cast<int (*)(const char *)>(dlsym(h, "roctxRangePushA"));
    a.pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
    if (!a.push || !a.pop) a.push = nullptr, a.pop = nullptr;
    return a;
  }();
  return api;
}
}  // namespace ngpde

namespace ngpde {

std::string &last_error() {
  thread_local std::string msg;
  return msg;
}

int32_t fail(int32_t code, const char *fmt, ...) {
  char buf[1024];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  last_error() = buf;
  return code;
}

namespace {

// stable counting sort of the edge list by `key`; `other` becomes the column of each entry
void build_csr(int64_t n, int64_t m, const std::vector<int32_t> &key, const std::vector<int32_t> &other,
               Csr &out) {
  out.h_rowptr.assign(n + 1, 0);
  for (int64_t e = 0; e < m; ++e) out.h_rowptr[key[e] + 1]++;
  for (int64_t i = 0; i < n; ++i) out.h_rowptr[i + 1] += out.h_rowptr[i];
  out.h_col.resize(m);
  out.h_eid.resize(m);
  std::vector<int32_t> cursor(out.h_rowptr.begin(), out.h_rowptr.end() - 1);
  for (int64_t e = 0; e < m; ++e) {
    int32_t p = cursor[key[e]]++;
    out.h_col[p] = other[e];
    out.h_eid[p] = (int32_t)e;
  }
}

}  // namespace

// BFS-grown clusters of `tile` nodes, emitted in a breadth-first sweep (see ngpde_graph::h_order)
std::vector<int32_t> locality_order_host(int64_t n, const std::vector<int32_t> &rp_in, const std::vector<int32_t> &col_in,
                                         const std::vector<int32_t> &rp_out, const std::vector<int32_t> &col_out, int tile) {
  std::vector<int32_t> order;
  order.reserve((size_t)n);
  std::vector<char> taken((size_t)n, 0);
  std::deque<int32_t> frontier;
  int64_t next_free = 0;
  std::vector<int32_t> local;
  auto neighbours = [&](int32_t v, auto &&f) {
    for (int32_t p = rp_in[v]; p < rp_in[v + 1]; ++p) f(col_in[p]);
    for (int32_t p = rp_out[v]; p < rp_out[v + 1]; ++p) f(col_out[p]);
  };
  while ((int64_t)order.size() < n) {
    local.clear();
    size_t head = 0;
    while ((int)local.size() < tile && (int64_t)(order.size() + local.size()) < n) {
      if (head == local.size()) {  // need a (new) seed: oldest frontier node, else the next untouched index
        int32_t seed = -1;
        while (!frontier.empty()) {
          int32_t v = frontier.front();
          frontier.pop_front();
          if (!taken[v]) { seed = v; break; }
        }
        if (seed < 0) {
          while (taken[next_free]) ++next_free;
          seed = (int32_t)next_free;
        }
        taken[seed] = 1;
        local.push_back(seed);
      }
      const int32_t v = local[head++];
      neighbours(v, [&](int32_t w) {
        if (taken[w]) return;
        if ((int)local.size() < tile) { taken[w] = 1; local.push_back(w); }
        else frontier.push_back(w);
      });
    }
    for (size_t k = head; k < local.size(); ++k)
      neighbours(local[k], [&](int32_t w) { if (!taken[w]) frontier.push_back(w); });
    order.insert(order.end(), local.begin(), local.end());
  }
  return order;
}

namespace {

template <class T>
int32_t upload(T **dst, const T *src, size_t count) {
  *dst = nullptr;
  size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  NGPDE_HIP_CHECK(hipMalloc((void **)dst, bytes));
  if (count) NGPDE_HIP_CHECK(hipMemcpy(*dst, src, count * sizeof(T), hipMemcpyHostToDevice));
  return NGPDE_OK;
}

int32_t upload_csr(Csr &c, int64_t n, int64_t m) {
  int32_t st;
  if ((st = upload(&c.rowptr, c.h_rowptr.data(), (size_t)n + 1))) return st;
  if ((st = upload(&c.col, c.h_col.data(), (size_t)m))) return st;
  if ((st = upload(&c.eid, c.h_eid.data(), (size_t)m))) return st;
  return NGPDE_OK;
}

void free_csr(Csr &c) {
  if (c.rowptr) (void)hipFree(c.rowptr);
  if (c.col) (void)hipFree(c.col);
  if (c.eid) (void)hipFree(c.eid);
  if (c.ent) (void)hipFree(c.ent);
  if (c.sched) (void)hipFree(c.sched);
  if (c.ell) (void)hipFree(c.ell);
  if (c.xpos) (void)hipFree(c.xpos);
  if (c.halo) (void)hipFree(c.halo);
  if (c.tile_info) (void)hipFree(c.tile_info);
  if (c.slots) (void)hipFree(c.slots);
  if (c.slot_w) (void)hipFree(c.slot_w);
  c = Csr();
}

}  // namespace
}  // namespace ngpde

namespace ngpde {

int32_t graph_halo_inverse(const ngpde_graph *g, int stride, const HaloInverse **out) {
  std::lock_guard<std::mutex> lock(g->lazy_mu);
  HaloInverse &hi = g->halo_inv;
  if (hi.ptr) {
    // one stride per handle: a table another caller's enqueued kernel may still read is never rebuilt
    NGPDE_REQUIRE(hi.stride == stride, NGPDE_ERR_STATE, "graph_halo_inverse: the handle's table has stride %d, %d asked for", hi.stride, stride);
    *out = &hi;
    return NGPDE_OK;
  }
  NGPDE_REQUIRE(g->by_t.halo && g->by_t.tile_info && g->by_t.halo_ok && g->by_t.max_halo <= stride, NGPDE_ERR_STATE,
                "graph_halo_inverse: the handle has no halo lists of at most %d rows", stride);
  const size_t nt = (size_t)g->n_sched / kTileRows, n = (size_t)g->n_nodes;
  NGPDE_REQUIRE(nt * (size_t)stride < (1ull << 31), NGPDE_ERR_UNSUPPORTED, "graph_halo_inverse: too many tiles for int32 entries");
  std::vector<int2> halo(nt * kHaloCap), info(nt);
  NGPDE_HIP_CHECK(hipMemcpy(halo.data(), g->by_t.halo, halo.size() * sizeof(int2), hipMemcpyDeviceToHost));
  NGPDE_HIP_CHECK(hipMemcpy(info.data(), g->by_t.tile_info, info.size() * sizeof(int2), hipMemcpyDeviceToHost));
  auto each = [&](auto &&f) {   // (tile, slot, node) of every FOREIGN halo entry (own rows are slots 0 .. kTileRows - 1), tiles ascending
    for (size_t tl = 0; tl < nt; ++tl)
      for (int k = kTileRows; k < info[tl].x; ++k) f(tl, k, halo[tl * kHaloCap + k].x);
  };
  std::vector<int32_t> cnt(n, 0);
  each([&](size_t, int, int32_t v) { cnt[(size_t)v]++; });
  std::vector<int32_t> node, ptr(1, 0), where(n, -1);
  for (size_t v = 0; v < n; ++v)
    if (cnt[v]) {
      where[v] = (int32_t)node.size();
      node.push_back((int32_t)v);
      ptr.push_back(ptr.back() + cnt[v]);
    }
  std::vector<int32_t> ent((size_t)ptr.back()), cur(ptr.begin(), ptr.end() - 1);
  each([&](size_t tl, int k, int32_t v) { ent[(size_t)cur[(size_t)where[v]]++] = (int32_t)(tl * (size_t)stride + k); });
  HaloInverse built;
  int32_t st;
  if ((st = upload(&built.node, node.data(), node.size())) || (st = upload(&built.ptr, ptr.data(), ptr.size())) ||
      (st = upload(&built.ent, ent.data(), ent.size()))) {
    if (built.node) (void)hipFree(built.node);
    if (built.ptr) (void)hipFree(built.ptr);
    return st;
  }
  built.n_listed = (int32_t)node.size();
  built.stride = stride;
  hi = built;
  *out = &hi;
  return NGPDE_OK;
}

}  // namespace ngpde

using namespace ngpde;

extern "C" {

const char *ngpde_version(void) { return NGPDE_VERSION_STRING; }
const char *ngpde_last_error(void) { return last_error().c_str(); }

int32_t ngpde_graph_create(int64_t n_nodes, int64_t n_edges, const int64_t *s, const int64_t *t,
                           int32_t index_base, int32_t n_graphs, ngpde_graph_t **out) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(out != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_create: out is NULL");
  *out = nullptr;
  NGPDE_REQUIRE(n_nodes >= 0 && n_edges >= 0, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_graph_create: negative size (n_nodes=%lld, n_edges=%lld)", (long long)n_nodes,
                (long long)n_edges);
  NGPDE_REQUIRE(n_nodes < (1ll << 31) - 1 && n_edges + n_nodes < (1ll << 31) - 1, NGPDE_ERR_UNSUPPORTED,
                "ngpde_graph_create: graph too large for int32 device indices");
  NGPDE_REQUIRE(n_edges == 0 || (s && t), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_create: s/t is NULL");
  NGPDE_REQUIRE(index_base == 0 || index_base == 1, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_graph_create: index_base must be 0 or 1");
  NGPDE_REQUIRE(n_graphs >= 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_create: n_graphs < 0");
  std::vector<int32_t> s32((size_t)n_edges), t32((size_t)n_edges);
  for (int64_t e = 0; e < n_edges; ++e) {
    int64_t a = s[e] - index_base, b = t[e] - index_base;
    if (a < 0 || a >= n_nodes || b < 0 || b >= n_nodes)
      return fail(NGPDE_ERR_DIMENSION_MISMATCH,
                  "DimensionMismatch: edge %lld (%lld -> %lld) references a node outside 1:%lld",
                  (long long)e + index_base, (long long)s[e], (long long)t[e], (long long)n_nodes);
    s32[e] = (int32_t)a;
    t32[e] = (int32_t)b;
  }
  ngpde_graph *g = new (std::nothrow) ngpde_graph();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "out of host memory");
  g->n_nodes = n_nodes;
  g->n_edges = n_edges;
  g->n_graphs = n_graphs;
  build_csr(n_nodes, n_edges, t32, s32, g->by_t);
  build_csr(n_nodes, n_edges, s32, t32, g->by_s);
  for (int64_t i = 0; i < n_nodes; ++i) {
    g->max_in_degree = std::max(g->max_in_degree, g->by_t.h_rowptr[i + 1] - g->by_t.h_rowptr[i]);
    g->max_out_degree = std::max(g->max_out_degree, g->by_s.h_rowptr[i + 1] - g->by_s.h_rowptr[i]);
  }
  {  // cross positions: entry q of one list and entry p of the other that are the same COO edge
    std::vector<int32_t> pos_t((size_t)n_edges), pos_s((size_t)n_edges), xt((size_t)n_edges), xs((size_t)n_edges);
    for (int64_t p = 0; p < n_edges; ++p) pos_t[g->by_t.h_eid[p]] = (int32_t)p;
    for (int64_t q = 0; q < n_edges; ++q) pos_s[g->by_s.h_eid[q]] = (int32_t)q;
    for (int64_t q = 0; q < n_edges; ++q) xs[q] = pos_t[g->by_s.h_eid[q]];
    for (int64_t p = 0; p < n_edges; ++p) xt[p] = pos_s[g->by_t.h_eid[p]];
    int32_t stx;
    if ((stx = upload(&g->by_s.xpos, xs.data(), xs.size())) || (stx = upload(&g->by_t.xpos, xt.data(), xt.size()))) {
      ngpde_graph_destroy(g);
      return stx;
    }
  }
  g->h_order = locality_order_host(n_nodes, g->by_t.h_rowptr, g->by_t.h_col, g->by_s.h_rowptr, g->by_s.h_col, kTileRows);
  g->n_sched = (int32_t)(((n_nodes + kTileRows - 1) / kTileRows) * kTileRows);
  int32_t st;
  if ((st = upload_csr(g->by_t, n_nodes, n_edges)) || (st = upload_csr(g->by_s, n_nodes, n_edges)) ||
      (st = upload(&g->order, g->h_order.data(), g->h_order.size()))) {
    ngpde_graph_destroy(g);
    return st;
  }
  *out = g;
  return NGPDE_OK;
}

int32_t ngpde_graph_destroy(ngpde_graph_t *g) {
  NGPDE_RANGE();
  if (!g) return NGPDE_OK;
  free_csr(g->by_t);
  free_csr(g->by_s);
  if (g->c) (void)hipFree(g->c);
  if (g->w_coo) (void)hipFree(g->w_coo);
  if (g->order) (void)hipFree(g->order);
  if (g->halo_inv.ptr) (void)hipFree(g->halo_inv.ptr);
  if (g->halo_inv.node) (void)hipFree(g->halo_inv.node);
  if (g->halo_inv.ent) (void)hipFree(g->halo_inv.ent);
  delete g;
  return NGPDE_OK;
}

int32_t ngpde_graph_create_device(int64_t n_nodes, int64_t n_edges, const void *s, const void *t, int32_t index_bits,
                                  int32_t index_base, int32_t n_graphs, const int32_t *order, ngpde_stream_t stream,
                                  ngpde_graph_t **out) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(out != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_create_device: out is NULL");
  *out = nullptr;
  NGPDE_REQUIRE(n_nodes >= 0 && n_edges >= 0, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_graph_create_device: negative size (n_nodes=%lld, n_edges=%lld)", (long long)n_nodes, (long long)n_edges);
  NGPDE_REQUIRE(n_nodes < (1ll << 31) - 1 && n_edges + n_nodes < (1ll << 31) - 1, NGPDE_ERR_UNSUPPORTED,
                "ngpde_graph_create_device: graph too large for int32 device indices");
  NGPDE_REQUIRE(n_edges == 0 || (s && t), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_create_device: s/t is NULL");
  NGPDE_REQUIRE(index_bits == 32 || index_bits == 64, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_graph_create_device: index_bits must be 32 or 64");
  NGPDE_REQUIRE(index_base == 0 || index_base == 1, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_graph_create_device: index_base must be 0 or 1");
  NGPDE_REQUIRE(n_graphs >= 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_create_device: n_graphs < 0");
  if (index_bits == 32)
    return ngpde::graph_create_device<int32_t>(n_nodes, n_edges, (const int32_t *)s, (const int32_t *)t, index_base, n_graphs,
                                               order, (hipStream_t)stream, out);
  return ngpde::graph_create_device<int64_t>(n_nodes, n_edges, (const int64_t *)s, (const int64_t *)t, index_base, n_graphs, order,
                                             (hipStream_t)stream, out);
}

int32_t ngpde_graph_node_order(const ngpde_graph_t *g, int32_t *order_out) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr && (order_out != nullptr || g->n_nodes == 0), NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_graph_node_order: NULL argument");
  if (g->n_nodes == 0) return NGPDE_OK;
  if (!g->h_order.empty()) {
    std::memcpy(order_out, g->h_order.data(), (size_t)g->n_nodes * sizeof(int32_t));
  } else {
    NGPDE_HIP_CHECK(hipMemcpy(order_out, g->order, (size_t)g->n_nodes * sizeof(int32_t), hipMemcpyDeviceToHost));
  }
  return NGPDE_OK;
}

int32_t ngpde_graph_set_gcn_norm_device(ngpde_graph_t *g, int32_t add_self_loops, const float *edge_weight,
                                        int32_t weighted_degree, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_set_gcn_norm_device: graph is NULL");
  NGPDE_REQUIRE(!(weighted_degree && !edge_weight && g->n_edges > 0), NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_graph_set_gcn_norm_device: weighted_degree requires edge_weight");
  return ngpde::set_gcn_norm_device(g, add_self_loops, edge_weight, weighted_degree, (hipStream_t)stream);
}

int32_t ngpde_graph_array(const ngpde_graph_t *g, int32_t direction, int32_t which, const void **ptr, size_t *bytes) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g && ptr && bytes, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_array: NULL argument");
  NGPDE_REQUIRE(direction == 0 || direction == 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_array: direction must be 0 (by target) or 1 (by source)");
  const ngpde::Csr &c = direction == 0 ? g->by_t : g->by_s;
  const size_t n = (size_t)g->n_nodes, m = (size_t)g->n_edges, ns = (size_t)g->n_sched, nt = ns / ngpde::kTileRows;
  switch (which) {
    case NGPDE_GRAPH_ROWPTR: *ptr = c.rowptr; *bytes = (n + 1) * 4; break;
    case NGPDE_GRAPH_COL: *ptr = c.col; *bytes = m * 4; break;
    case NGPDE_GRAPH_EID: *ptr = c.eid; *bytes = m * 4; break;
    case NGPDE_GRAPH_XPOS: *ptr = c.xpos; *bytes = m * 4; break;
    case NGPDE_GRAPH_ENT: *ptr = c.ent; *bytes = c.ent ? m * 8 : 0; break;
    case NGPDE_GRAPH_SCHED: *ptr = c.sched; *bytes = c.sched ? ns * 16 : 0; break;
    case NGPDE_GRAPH_ELL: *ptr = c.ell; *bytes = c.ell ? ns * ngpde::kEllWidth * 8 : 0; break;
    case NGPDE_GRAPH_HALO: *ptr = c.halo; *bytes = c.halo ? nt * ngpde::kHaloCap * 8 : 0; break;
    case NGPDE_GRAPH_TILE_INFO: *ptr = c.tile_info; *bytes = c.tile_info ? nt * 8 : 0; break;
    case NGPDE_GRAPH_SLOTS: *ptr = c.slots; *bytes = c.slots ? ns * ngpde::kSlotWidth : 0; break;
    case NGPDE_GRAPH_SLOT_W: *ptr = c.slot_w; *bytes = c.slot_w ? ns * ngpde::kSlotWidth * 4 : 0; break;
    case NGPDE_GRAPH_C: *ptr = g->c; *bytes = g->c ? n * 4 : 0; break;
    case NGPDE_GRAPH_ORDER: *ptr = g->order; *bytes = g->order ? n * 4 : 0; break;
    case NGPDE_GRAPH_HALO_OK: *ptr = nullptr; *bytes = c.halo_ok ? 1 : 0; break;
    default: return ngpde::fail(NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_array: unknown array code %d", which);
  }
  return NGPDE_OK;
}

int32_t ngpde_graph_info(const ngpde_graph_t *g, int64_t *n_nodes, int64_t *n_edges, int32_t *n_graphs) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_info: graph is NULL");
  if (n_nodes) *n_nodes = g->n_nodes;
  if (n_edges) *n_edges = g->n_edges;
  if (n_graphs) *n_graphs = g->n_graphs;
  return NGPDE_OK;
}

int32_t ngpde_graph_csr_by_target(const ngpde_graph_t *g, const int32_t **rowptr, const int32_t **col,
                                  const int32_t **eid) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_csr_by_target: graph is NULL");
  if (rowptr) *rowptr = g->by_t.rowptr;
  if (col) *col = g->by_t.col;
  if (eid) *eid = g->by_t.eid;
  return NGPDE_OK;
}

int32_t ngpde_graph_csr_by_source(const ngpde_graph_t *g, const int32_t **rowptr, const int32_t **col,
                                  const int32_t **eid) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_csr_by_source: graph is NULL");
  if (rowptr) *rowptr = g->by_s.rowptr;
  if (col) *col = g->by_s.col;
  if (eid) *eid = g->by_s.eid;
  return NGPDE_OK;
}

int32_t ngpde_graph_set_gcn_norm(ngpde_graph_t *g, int32_t add_self_loops, const float *edge_weight,
                                 int32_t weighted_degree) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_graph_set_gcn_norm: graph is NULL");
  NGPDE_REQUIRE(!(weighted_degree && !edge_weight && g->n_edges > 0), NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_graph_set_gcn_norm: weighted_degree requires edge_weight");
  if (g->device_built) {   // no host copies of the lists: stage the weights and build on the device
    float *w_dev = nullptr;
    if (edge_weight) {
      NGPDE_HIP_CHECK(hipMalloc((void **)&w_dev, std::max<size_t>((size_t)g->n_edges, 1) * sizeof(float)));
      if (hipMemcpy(w_dev, edge_weight, (size_t)g->n_edges * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(w_dev);
        return ngpde::fail(NGPDE_ERR_HIP, "ngpde_graph_set_gcn_norm: upload of the edge weights failed");
      }
    }
    const int32_t st_dev = ngpde::set_gcn_norm_device(g, add_self_loops, w_dev, weighted_degree, nullptr);
    if (w_dev) (void)hipFree(w_dev);
    return st_dev;
  }
  const int64_t n = g->n_nodes, m = g->n_edges;
  // d = degree(g; dir=:in[, edge_weight]) after add_self_loops (weights padded with ones), :210-224
  std::vector<float> deg((size_t)n, add_self_loops ? 1.0f : 0.0f);
  for (int64_t i = 0; i < n; ++i) {
    // accumulate in COO order within the row, in float, as scatter(+) does
    float acc = 0.f;
    for (int32_t p = g->by_t.h_rowptr[i]; p < g->by_t.h_rowptr[i + 1]; ++p)
      acc += weighted_degree ? edge_weight[g->by_t.h_eid[p]] : 1.0f;
    deg[i] = acc + (add_self_loops ? 1.0f : 0.0f);
  }
  std::vector<float> c((size_t)n);
  for (int64_t i = 0; i < n; ++i) c[i] = 1.0f / std::sqrt(deg[i]);  // :225 (Inf for isolated nodes, as the reference)
  auto fill = [&](Csr &csr, std::vector<int2> &ent) {
    ent.resize((size_t)m);
    for (int64_t p = 0; p < m; ++p) {
      float w = edge_weight ? edge_weight[csr.h_eid[p]] : 1.0f;
      float coef = w * c[csr.h_col[p]];
      int2 v;
      v.x = csr.h_col[p];
      std::memcpy(&v.y, &coef, 4);
      ent[p] = v;
    }
  };
  std::vector<int2> et, es;
  fill(g->by_t, et);
  fill(g->by_s, es);
  auto fill_sched = [&](const Csr &csr, std::vector<int4> &sc) {
    sc.assign((size_t)g->n_sched, make_int4(-1, 0, 0, 0));
    for (int64_t k = 0; k < n; ++k) {
      const int32_t v = g->h_order[k];
      int4 e;
      e.x = v;
      e.y = csr.h_rowptr[v];
      e.z = csr.h_rowptr[v + 1] - csr.h_rowptr[v];
      std::memcpy(&e.w, &c[v], 4);
      sc[k] = e;
    }
  };
  std::vector<int4> st_, ss_;
  fill_sched(g->by_t, st_);
  fill_sched(g->by_s, ss_);
  auto fill_ell = [&](const Csr &csr, const std::vector<int2> &ent, std::vector<int2> &ell) {
    ell.assign((size_t)g->n_sched * kEllWidth, make_int2(0, 0));
    for (int64_t k = 0; k < n; ++k) {
      const int32_t v = g->h_order[k];
      const int32_t rs = csr.h_rowptr[v], deg = csr.h_rowptr[v + 1] - rs;
      for (int j = 0; j < std::min(deg, kEllWidth); ++j) ell[(size_t)k * kEllWidth + j] = ent[rs + j];
    }
  };
  std::vector<int2> lt_, ls_;
  fill_ell(g->by_t, et, lt_);
  fill_ell(g->by_s, es, ls_);
  // halo lists + slot bytes for the LDS-staged aggregation
  const int64_t n_tiles = g->n_sched / kTileRows;
  struct HaloOut { std::vector<int2> halo, info; std::vector<uint8_t> slots; std::vector<float> w; };
  auto fill_halo = [&](const Csr &csr, HaloOut &o) {
    o.halo.assign((size_t)n_tiles * kHaloCap, make_int2(0, 0));
    o.info.assign((size_t)n_tiles, make_int2(0, 0));
    o.slots.assign((size_t)g->n_sched * kSlotWidth, (uint8_t)kHaloCap);
    if (edge_weight) o.w.assign((size_t)g->n_sched * kSlotWidth, 0.f);
    std::vector<int32_t> slot_of((size_t)n, -1), stamp((size_t)n, -1);
    for (int64_t tl = 0; tl < n_tiles; ++tl) {
      int count = 0;
      bool fits = true;
      auto slot = [&](int32_t v) {
        if (stamp[v] != (int32_t)tl) {
          stamp[v] = (int32_t)tl;
          slot_of[v] = count;
          if (count < kHaloCap) {
            int2 e;
            e.x = v;
            std::memcpy(&e.y, &c[v], 4);
            o.halo[(size_t)tl * kHaloCap + count] = e;
          }
          ++count;
        }
        return slot_of[v];
      };
      for (int k = 0; k < kTileRows; ++k) {          // own rows first: slot k = k-th row (padding rows keep {0, 0})
        const int64_t pos = tl * kTileRows + k;
        if (pos < n) slot(g->h_order[pos]);
        else ++count;
      }
      for (int k = 0; k < kTileRows && fits; ++k) {
        const int64_t pos = tl * kTileRows + k;
        if (pos >= n) break;
        const int32_t v = g->h_order[pos];
        const int32_t rs = csr.h_rowptr[v], deg = csr.h_rowptr[v + 1] - rs;
        if (deg > kSlotWidth) { fits = false; break; }
        for (int j = 0; j < deg; ++j) {
          const int sl = slot(csr.h_col[rs + j]);
          if (count > kHaloCap) { fits = false; break; }
          o.slots[(size_t)pos * kSlotWidth + j] = (uint8_t)sl;
          if (edge_weight) o.w[(size_t)pos * kSlotWidth + j] = edge_weight[csr.h_eid[rs + j]];
        }
      }
      o.info[tl] = make_int2(fits ? count : 0, 0);
    }
  };
  HaloOut ht_, hs_;
  fill_halo(g->by_t, ht_);
  fill_halo(g->by_s, hs_);
  auto all_fit = [&](const HaloOut &o) {
    for (const int2 &i : o.info)
      if (i.x == 0) return false;
    return n_tiles > 0;
  };
  g->by_t.halo_ok = all_fit(ht_);
  g->by_s.halo_ok = all_fit(hs_);
  auto max_halo = [&](const HaloOut &o) {
    int32_t mx = 0;
    for (const int2 &i : o.info) mx = std::max(mx, i.x);
    return mx;
  };
  const int32_t mh_t = max_halo(ht_), mh_s = max_halo(hs_);
  for (Csr *c2 : {&g->by_t, &g->by_s}) {
    if (c2->halo) { (void)hipFree(c2->halo); c2->halo = nullptr; }
    if (c2->tile_info) { (void)hipFree(c2->tile_info); c2->tile_info = nullptr; }
    if (c2->slots) { (void)hipFree(c2->slots); c2->slots = nullptr; }
    if (c2->slot_w) { (void)hipFree(c2->slot_w); c2->slot_w = nullptr; }
  }
  if (g->by_t.ell) { (void)hipFree(g->by_t.ell); g->by_t.ell = nullptr; }
  if (g->by_s.ell) { (void)hipFree(g->by_s.ell); g->by_s.ell = nullptr; }
  if (g->by_t.ent) { (void)hipFree(g->by_t.ent); g->by_t.ent = nullptr; }
  if (g->by_s.ent) { (void)hipFree(g->by_s.ent); g->by_s.ent = nullptr; }
  if (g->by_t.sched) { (void)hipFree(g->by_t.sched); g->by_t.sched = nullptr; }
  if (g->by_s.sched) { (void)hipFree(g->by_s.sched); g->by_s.sched = nullptr; }
  if (g->c) { (void)hipFree(g->c); g->c = nullptr; }
  if (g->w_coo) { (void)hipFree(g->w_coo); g->w_coo = nullptr; }
  g->has_norm = false;
  int32_t st;
  // (the weights themselves are kept only where the hub geometry of the persistent solver could want them: some tile does not fit the halo lists)
  if (edge_weight && m > 0 && !(g->by_t.halo_ok && g->by_s.halo_ok) && (st = upload(&g->w_coo, edge_weight, (size_t)m))) return st;
  if ((st = upload(&g->by_t.ent, et.data(), (size_t)m))) return st;
  if ((st = upload(&g->by_s.ent, es.data(), (size_t)m))) return st;
  if ((st = upload(&g->by_t.sched, st_.data(), st_.size()))) return st;
  if ((st = upload(&g->by_s.sched, ss_.data(), ss_.size()))) return st;
  if ((st = upload(&g->by_t.ell, lt_.data(), lt_.size()))) return st;
  if ((st = upload(&g->by_s.ell, ls_.data(), ls_.size()))) return st;
  if ((st = upload(&g->by_t.halo, ht_.halo.data(), ht_.halo.size()))) return st;
  if ((st = upload(&g->by_s.halo, hs_.halo.data(), hs_.halo.size()))) return st;
  if ((st = upload(&g->by_t.tile_info, ht_.info.data(), ht_.info.size()))) return st;
  if ((st = upload(&g->by_s.tile_info, hs_.info.data(), hs_.info.size()))) return st;
  if ((st = upload(&g->by_t.slots, ht_.slots.data(), ht_.slots.size()))) return st;
  if ((st = upload(&g->by_s.slots, hs_.slots.data(), hs_.slots.size()))) return st;
  if (edge_weight) {
    if ((st = upload(&g->by_t.slot_w, ht_.w.data(), ht_.w.size()))) return st;
    if ((st = upload(&g->by_s.slot_w, hs_.w.data(), hs_.w.size()))) return st;
  }
  if ((st = upload(&g->c, c.data(), (size_t)n))) return st;
  g->by_t.max_halo = mh_t;
  g->by_s.max_halo = mh_s;
  g->self_loops = add_self_loops ? 1 : 0;
  g->has_norm = true;
  return NGPDE_OK;
}

// host only (include/ngpde.h): the numbering of a batch whose members are padded to whole tiles
int32_t ngpde_batch_pad_host(int32_t n_members, const int64_t *sizes, int64_t *padded_offsets, int64_t *index, const int32_t *order,
                             int32_t *order_padded) {
  NGPDE_REQUIRE(n_members >= 0 && (n_members == 0 || sizes) && padded_offsets && (!order || order_padded), NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_batch_pad_host: sizes, padded_offsets (and order_padded with order) are required");
  int64_t off = 0, poff = 0;
  std::vector<int64_t> offs((size_t)n_members + 1, 0);
  for (int32_t k = 0; k < n_members; ++k) {
    NGPDE_REQUIRE(sizes[k] >= 0, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_batch_pad_host: member %d has %lld nodes", k, (long long)sizes[k]);
    padded_offsets[k] = poff;
    if (index)
      for (int64_t i = 0; i < sizes[k]; ++i) index[off + i] = poff + i;
    off += sizes[k];
    offs[(size_t)k + 1] = off;
    poff += (sizes[k] + kTileRows - 1) / kTileRows * kTileRows;
  }
  padded_offsets[n_members] = poff;
  NGPDE_REQUIRE(poff <= INT32_MAX, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_batch_pad_host: %lld padded nodes exceed the 32-bit node ids", (long long)poff);
  if (order) {
    std::vector<uint8_t> seen((size_t)off, 0);
    int64_t w = 0;
    for (int32_t k = 0; k < n_members; ++k) {
      for (int64_t i = offs[(size_t)k]; i < offs[(size_t)k + 1]; ++i) {
        const int64_t v = order[i];
        NGPDE_REQUIRE(v >= offs[(size_t)k] && v < offs[(size_t)k + 1] && !seen[(size_t)v], NGPDE_ERR_INVALID_ARGUMENT,
                      "ngpde_batch_pad_host: order[%lld] = %lld is not a permutation inside member %d", (long long)i, (long long)v, k);
        seen[(size_t)v] = 1;
        order_padded[w++] = (int32_t)(padded_offsets[k] + (v - offs[(size_t)k]));
      }
      for (int64_t v = padded_offsets[k] + sizes[k]; v < padded_offsets[k + 1]; ++v) order_padded[w++] = (int32_t)v;
    }
  }
  return NGPDE_OK;
}

}  // extern "C"
