// AddressSanitizer / UBSan run of the host-side code on the path (GPU ASan is not available on the pool):
//   * the host builder of the derived-graph handle (neuralgraphpde.jl_amd/csrc/graph.hip: counting sorts, cross positions, the
//     BFS-grown locality order, GCN coefficients, tile schedule, halo lists, slot bytes) compiled against tests/asan/shim (device
//     memory = host memory) -- every array it builds is read back in full through ngpde_graph_array;
//   * the C restatement oracle/ngpde_oracle.c (layer forward / backward, two-step solves with the discrete adjoint).
// Exit code 0 and an empty sanitizer report = pass (tests/test_asan_cpu.py).  Graphs: the reference's 3-node fixture
// (/root/reference/test/runtests.jl:11-13), empty and edgeless graphs, self loops and duplicate edges, a hub row beyond the slot
// width, isolated nodes, block-diagonal batches, 1-based Int64 lists with an out-of-range index (must be REJECTED, not read).
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../include/ngpde.h"
#include "hip/hip_runtime.h"

// the two device-side entries graph.hip links against (out of reach of a host-only build)
struct ngpde_graph;
namespace ngpde {
int32_t set_gcn_norm_device(ngpde_graph *, int32_t, const float *, int32_t, hipStream_t) { return NGPDE_ERR_UNSUPPORTED; }
template <class I>
int32_t graph_create_device(int64_t, int64_t, const I *, const I *, int32_t, int32_t, const int32_t *, hipStream_t, ngpde_graph **) {
  return NGPDE_ERR_UNSUPPORTED;
}
template int32_t graph_create_device<int64_t>(int64_t, int64_t, const int64_t *, const int64_t *, int32_t, int32_t, const int32_t *, hipStream_t,
                                             ngpde_graph **);
template int32_t graph_create_device<int32_t>(int64_t, int64_t, const int32_t *, const int32_t *, int32_t, int32_t, const int32_t *, hipStream_t,
                                             ngpde_graph **);
}  // namespace ngpde

extern "C" {
void ngo_gcn_forward(int64_t n, int64_t e, const int64_t *s, const int64_t *t, int self_loops, int din, int dout, int act, const float *x,
                     const float *wt, const float *bias, float *y, float *x3_out, float *z_out);
void ngo_gcn_backward(int64_t n, int64_t e, const int64_t *s, const int64_t *t, int self_loops, int din, int dout, int act, const float *wt,
                      const float *z, const float *x3, const float *dy, float *dx, float *dwt, float *db);
int ngo_node_gcn2(int64_t n, int64_t e, const int64_t *s, const int64_t *t, int d, int act, int tableau, int nsteps, float dt, int with_grad,
                  const float *u0, const float *w1, const float *b1, const float *w2, const float *b2, float *uT, float *du0, float *dw1,
                  float *db1, float *dw2, float *db2);
}

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {   // splitmix64
  uint64_t z = (rng_state += 0x9E3779B97F4A7C15ull);
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}
static float rndf() { return (float)((rnd() >> 40) / 16777216.0) - 0.5f; }

static int failures = 0;
#define CHECK(cond)                                                        \
  do {                                                                     \
    if (!(cond)) { std::fprintf(stderr, "CHECK failed: %s (line %d): %s\n", #cond, __LINE__, ngpde_last_error()); ++failures; } \
  } while (0)

// build the handle, normalise it, read every derived array in full (a checksum keeps the reads alive)
static uint64_t exercise(int64_t n, const std::vector<int64_t> &s, const std::vector<int64_t> &t, int base, int n_graphs, bool self_loops,
                         bool weights) {
  ngpde_graph_t *g = nullptr;
  const int32_t st = ngpde_graph_create(n, (int64_t)s.size(), s.data(), t.data(), base, n_graphs, &g);
  CHECK(st == NGPDE_OK);
  if (st != NGPDE_OK) return 0;
  std::vector<float> w(s.size());
  for (float &v : w) v = 0.5f + (float)(rnd() % 100) / 100.f;
  CHECK(ngpde_graph_set_gcn_norm(g, self_loops ? 1 : 0, weights ? w.data() : nullptr, weights ? 1 : 0) == NGPDE_OK);
  uint64_t sum = 0;
  for (int dir = 0; dir < 2; ++dir)
    for (int which = NGPDE_GRAPH_ROWPTR; which <= NGPDE_GRAPH_ORDER; ++which) {
      const void *ptr = nullptr;
      size_t bytes = 0;
      if (ngpde_graph_array(g, dir, which, &ptr, &bytes) != NGPDE_OK || !ptr) continue;
      const unsigned char *b = static_cast<const unsigned char *>(ptr);
      for (size_t i = 0; i < bytes; ++i) sum += b[i];
    }
  std::vector<int32_t> order((size_t)(n ? n : 1));
  CHECK(ngpde_graph_node_order(g, order.data()) == NGPDE_OK);
  int64_t nn = 0, ne = 0;
  int32_t ng = 0;
  CHECK(ngpde_graph_info(g, &nn, &ne, &ng) == NGPDE_OK && nn == n && ne == (int64_t)s.size());
  CHECK(ngpde_graph_destroy(g) == NGPDE_OK);
  return sum;
}

int main() {
  uint64_t sum = 0;
  // the reference's fixture, 1-based
  sum += exercise(3, {1, 1, 2, 3}, {2, 3, 1, 1}, 1, 1, true, false);
  sum += exercise(3, {1, 1, 2, 3}, {2, 3, 1, 1}, 1, 1, false, true);
  // empty and edgeless graphs
  sum += exercise(0, {}, {}, 0, 1, true, false);
  sum += exercise(5, {}, {}, 0, 1, true, false);
  // random graphs: sizes around the 32-row tile, duplicates, self loops, isolated nodes, a hub, batches
  for (int trial = 0; trial < 40; ++trial) {
    const int64_t n = 1 + (int64_t)(rnd() % 200);
    const int64_t e = (int64_t)(rnd() % (6 * n + 1));
    std::vector<int64_t> s((size_t)e), t((size_t)e);
    for (int64_t k = 0; k < e; ++k) {
      s[(size_t)k] = (int64_t)(rnd() % n);
      t[(size_t)k] = (trial % 5 == 0) ? 0 : (int64_t)(rnd() % n);   // every fifth graph: one hub target with degree > 32
    }
    if (e > 2) { s[1] = s[0]; t[1] = t[0]; s[2] = t[2]; }             // a duplicate edge and a self loop
    sum += exercise(n, s, t, 0, 1, trial % 2 == 0, trial % 3 == 0);
  }
  {   // block-diagonal batch of 3 graphs of 40 nodes
    std::vector<int64_t> s, t;
    for (int gph = 0; gph < 3; ++gph)
      for (int k = 0; k < 150; ++k) { s.push_back(gph * 40 + (int64_t)(rnd() % 40)); t.push_back(gph * 40 + (int64_t)(rnd() % 40)); }
    sum += exercise(120, s, t, 0, 3, true, false);
  }
  {   // an index outside 1..n must be rejected before anything is read or written with it
    ngpde_graph_t *g = nullptr;
    const int64_t s[3] = {1, 2, 7}, t[3] = {2, 3, 1};
    CHECK(ngpde_graph_create(3, 3, s, t, 1, 1, &g) != NGPDE_OK && g == nullptr);
    const int64_t s0[2] = {0, 1}, t0[2] = {1, 2};
    CHECK(ngpde_graph_create(3, 2, s0, t0, 1, 1, &g) != NGPDE_OK && g == nullptr);   // 0 in a 1-based list
  }
  // the C restatement: one layer forward + backward, and Euler / Tsit5 solves with the adjoint
  for (int trial = 0; trial < 6; ++trial) {
    const int64_t n = 5 + (int64_t)(rnd() % 60), e = (int64_t)(rnd() % (5 * n));
    const int din = 1 + (int)(rnd() % 9), dout = 1 + (int)(rnd() % 9), d = 4 + 4 * (int)(rnd() % 3);
    std::vector<int64_t> s((size_t)e), t((size_t)e);
    for (int64_t k = 0; k < e; ++k) { s[(size_t)k] = (int64_t)(rnd() % n); t[(size_t)k] = (int64_t)(rnd() % n); }
    std::vector<float> x((size_t)n * din), wt((size_t)din * dout), b((size_t)dout), y((size_t)n * dout), x3((size_t)n * din), z((size_t)n * dout);
    for (float &v : x) v = rndf();
    for (float &v : wt) v = rndf();
    for (float &v : b) v = rndf();
    ngo_gcn_forward(n, e, s.data(), t.data(), 1, din, dout, trial % 3, x.data(), wt.data(), b.data(), y.data(), x3.data(), z.data());
    std::vector<float> dy((size_t)n * dout, 1.f), dx((size_t)n * din), dwt((size_t)din * dout), db((size_t)dout);
    ngo_gcn_backward(n, e, s.data(), t.data(), 1, din, dout, trial % 3, wt.data(), z.data(), x3.data(), dy.data(), dx.data(), dwt.data(), db.data());
    std::vector<float> u0((size_t)n * d), w1((size_t)d * d), b1((size_t)d), w2((size_t)d * d), b2((size_t)d), uT((size_t)n * d), du0((size_t)n * d),
        dw1((size_t)d * d), db1((size_t)d), dw2((size_t)d * d), db2((size_t)d);
    for (float &v : u0) v = rndf();
    for (float &v : w1) v = 0.3f * rndf();
    for (float &v : w2) v = 0.3f * rndf();
    CHECK(ngo_node_gcn2(n, e, s.data(), t.data(), d, 1, trial % 2, 2, 0.1f, 1, u0.data(), w1.data(), b1.data(), w2.data(), b2.data(), uT.data(),
                        du0.data(), dw1.data(), db1.data(), dw2.data(), db2.data()) == 0);
    for (float v : uT) sum += (uint64_t)(v != v);   // NaN count
  }
  std::printf("asan driver: checksum %llu, %d failed checks\n", (unsigned long long)sum, failures);
  return failures ? 1 : 0;
}
