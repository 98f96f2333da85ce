/* A second plain-C caller of the drop-in boundary (include/ngpde.h), for the entries beyond GCNConv: the fixed-step solver
 * plan (ngpde_node_gcn2_*), the message path of the edge-function layers (ngpde_dense_forward, ngpde_edge_mlp_forward next to
 * the primitives ngpde_edge_combine_forward / ngpde_segment_reduce_forward), the reassociated and the literal GNOConv message
 * (ngpde_gno_apply_forward / ngpde_gno_contract_forward), the one-launch GAT layer next to its composition
 * (ngpde_gat_layer_forward vs ngpde_dense_forward + ngpde_gat_forward + ngpde_bias_act_forward), ngpde_rk_stage_combine, the
 * device-resident NeuralODE(VMHConv) plan with saveat (ngpde_node_vmh_*), and the
 * node-level Dense pair / chain launches (ngpde_dense_pair_forward / _backward, ngpde_dense_chain2_forward) at a streaming size.
 * No Python, torch or C++ on the calling side.  Checkers: the C restatement of the reference solver (oracle/ngpde_oracle.c) and
 * plain double-precision loops over the CSR lists the library hands out.  Exit code 0 = every comparison within tolerance.
 * Built and run by tests/test_c_abi_gpu.py. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ngpde.h"

int ngo_node_gcn2(int64_t n, int64_t e, const int64_t *s, const int64_t *t, int d, int act, int tableau, int nsteps, float dt,
                  int with_grad, const float *u0, const float *w1, const float *b1, const float *w2, const float *b2, float *uT,
                  float *du0, float *dw1, float *db1, float *dw2, float *db2);

#define CHECK_HIP(x)                                                                  \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
  } while (0)
#define CHECK_NG(x)                                                                   \
  do {                                                                                \
    int32_t s_ = (x);                                                                 \
    if (s_ != NGPDE_OK) { fprintf(stderr, "%s -> %d: %s\n", #x, s_, ngpde_last_error()); return 3; } \
  } while (0)

static uint64_t rng_state = 0x2545F4914F6CDD1Dull;
static float rnd(void) {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (float)((double)(rng_state >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}
static float *host_rand(size_t n, float scale) {
  float *h = malloc(sizeof(float) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) h[i] = rnd() * scale;
  return h;
}
static float *dev_copy(const float *h, size_t n) {
  float *d = NULL;
  if (hipMalloc((void **)&d, (n ? n : 1) * sizeof(float)) != hipSuccess) return NULL;
  if (h && n && hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}
static float *host_copy(const float *d, size_t n) {
  float *h = malloc(sizeof(float) * (n ? n : 1));
  if (hipMemcpy(h, d, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return NULL;
  return h;
}
static double max_rel(const float *a, const float *b, size_t n) {
  double err = 0, ref = 0;
  for (size_t i = 0; i < n; ++i) {
    const double d = fabs((double)a[i] - (double)b[i]);
    if (d > err) err = d;
    if (fabs((double)b[i]) > ref) ref = fabs((double)b[i]);
  }
  return err / (ref > 1e-30 ? ref : 1e-30);
}
static int fails = 0;
static void report(const char *what, double err, double tol) {
  printf("%-34s %.2e (tol %.0e)%s\n", what, err, tol, err <= tol ? "" : "   <-- FAIL");
  if (!(err <= tol)) ++fails;
}

int main(void) {
  const int64_t n = 1024, e = 5 * n;
  int64_t *s = malloc(sizeof(int64_t) * e), *t = malloc(sizeof(int64_t) * e);
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {   /* ring with chords; the last one directed */
    const int64_t a = (i + 1) % n, b = (i + 5) % n, c = (i + 9) % n;
    s[m] = i; t[m++] = a; s[m] = a; t[m++] = i;
    s[m] = i; t[m++] = b; s[m] = b; t[m++] = i;
    s[m] = i; t[m++] = c;
  }
  printf("%s\n", ngpde_version());
  ngpde_graph_t *g = NULL;
  CHECK_NG(ngpde_graph_create(n, e, s, t, /*index_base=*/0, 1, &g));
  CHECK_NG(ngpde_graph_set_gcn_norm(g, /*add_self_loops=*/1, NULL, 0));

  /* ---- 1. solver plan: 3 Tsit5 steps of du/dt = GCNConv(GCNConv(u)), forward + discrete adjoint of sum(u(T)) ---- */
  for (int variant = 0; variant < 2; ++variant) {   /* d = 64, and d = 32: the state zero-padded onto the 64-wide persistent kernels */
    const int d = variant ? 32 : 64, steps = 3;
    const float dt = 0.1f;
    float *u0 = host_rand(n * d, 1.f), *w1 = host_rand(d * d, 0.2f), *w2 = host_rand(d * d, 0.2f), *b1 = host_rand(d, 0.1f),
          *b2 = host_rand(d, 0.1f), *ones = malloc(sizeof(float) * n * d);
    for (int64_t i = 0; i < n * d; ++i) ones[i] = 1.f;
    float *u0_d = dev_copy(u0, n * d), *w1_d = dev_copy(w1, d * d), *w2_d = dev_copy(w2, d * d), *b1_d = dev_copy(b1, d),
          *b2_d = dev_copy(b2, d), *ones_d = dev_copy(ones, n * d), *uT_d = dev_copy(NULL, n * d), *du0_d = dev_copy(NULL, n * d),
          *dw1_d = dev_copy(NULL, d * d), *dw2_d = dev_copy(NULL, d * d), *db1_d = dev_copy(NULL, d), *db2_d = dev_copy(NULL, d);
    ngpde_node_t *plan = NULL;
    CHECK_NG(ngpde_node_gcn2_create(g, d, NGPDE_ACT_RELU, NGPDE_TABLEAU_TSIT5, steps, dt, 1, &plan));
    int32_t fl = 0, bl = 0, flags = 0, pending = 0, fault = 0;
    uint64_t gen = 0;
    CHECK_NG(ngpde_node_launch_count(plan, &fl, &bl));
    CHECK_NG(ngpde_node_flags(plan, &flags));
    CHECK_NG(ngpde_node_gcn2_forward(plan, u0_d, w1_d, b1_d, w2_d, b2_d, uT_d, NULL));
    CHECK_NG(ngpde_node_generation(plan, &gen, &pending));
    if (gen != 1 || !pending) { fprintf(stderr, "generation %llu pending %d\n", (unsigned long long)gen, pending); return 4; }
    CHECK_NG(ngpde_node_expect_generation(plan, gen));
    if (ngpde_node_expect_generation(plan, gen + 1) != NGPDE_ERR_STATE) return 4;      /* the error convention */
    CHECK_NG(ngpde_node_gcn2_backward(plan, ones_d, du0_d, dw1_d, db1_d, dw2_d, db2_d, NULL));
    CHECK_NG(ngpde_node_fault(plan, NULL, &fault));
    CHECK_HIP(hipDeviceSynchronize());
    printf("plan (d = %d): %d + %d launches, flags 0x%x, tape %.1f MB, fault %d\n", d, fl, bl, flags, ngpde_node_tape_bytes(plan) / 1e6, fault);
    if (fault) return 4;
    if ((flags & NGPDE_NODE_PERSISTENT_FWD) && ((flags & NGPDE_NODE_WIDENED) != 0) != (d != 64)) {
      fprintf(stderr, "d = %d: unexpected plan flags 0x%x\n", d, flags);
      return 4;
    }
    float *uT = host_copy(uT_d, n * d), *du0 = host_copy(du0_d, n * d), *dw1 = host_copy(dw1_d, d * d), *dw2 = host_copy(dw2_d, d * d),
          *db1 = host_copy(db1_d, d), *db2 = host_copy(db2_d, d);
    float *uTo = malloc(sizeof(float) * n * d), *du0o = malloc(sizeof(float) * n * d), *dw1o = malloc(sizeof(float) * d * d),
          *dw2o = malloc(sizeof(float) * d * d), *db1o = malloc(sizeof(float) * d), *db2o = malloc(sizeof(float) * d);
    if (ngo_node_gcn2(n, e, s, t, d, NGPDE_ACT_RELU, 1, steps, dt, 1, u0, w1, b1, w2, b2, uTo, du0o, dw1o, db1o, dw2o, db2o)) return 5;
    report("node_gcn2 u(T)", max_rel(uT, uTo, n * d), 2e-4);
    report("node_gcn2 du0", max_rel(du0, du0o, n * d), 1e-3);
    report("node_gcn2 dW1", max_rel(dw1, dw1o, d * d), 1e-3);
    report("node_gcn2 dW2", max_rel(dw2, dw2o, d * d), 1e-3);
    report("node_gcn2 db1", max_rel(db1, db1o, d), 1e-3);
    report("node_gcn2 db2", max_rel(db2, db2o, d), 1e-3);
    CHECK_NG(ngpde_node_destroy(plan));
  }

  /* ---- 1b. the same plan on a graph with a hub (docs/src/tutorials/graph_node.md:14-23: Cora has nodes of degree > 100): a row of 122
   * entries does not fit the handle's 32-entry rows, so ngpde_node_gcn2_create takes the persistent kernels' hub geometry (flag
   * NGPDE_NODE_HUB_GEOMETRY) -- nothing for the caller to do.  tanh: no relu kinks between the float32 port and the kernels. ---- */
  {
    const int64_t nh = 1500, hub = 120;
    const int64_t eh = 2 * (nh + hub);
    int64_t *sh = malloc(sizeof(int64_t) * eh), *th = malloc(sizeof(int64_t) * eh);
    int64_t mh = 0;
    for (int64_t i = 0; i < nh; ++i) { sh[mh] = i; th[mh++] = (i + 1) % nh; sh[mh] = (i + 1) % nh; th[mh++] = i; }   /* ring */
    for (int64_t k = 2; k < hub + 2; ++k) { sh[mh] = 0; th[mh++] = 3 * k; sh[mh] = 3 * k; th[mh++] = 0; }             /* node 0 <-> 120 others */
    ngpde_graph_t *gh = NULL;
    CHECK_NG(ngpde_graph_create(nh, mh, sh, th, 0, 1, &gh));
    CHECK_NG(ngpde_graph_set_gcn_norm(gh, 1, NULL, 0));
    const int d = 64, steps = 3;
    const float dt = 0.1f;
    float *u0 = host_rand(nh * d, 1.f), *w1 = host_rand(d * d, 0.2f), *w2 = host_rand(d * d, 0.2f), *b1 = host_rand(d, 0.1f),
          *b2 = host_rand(d, 0.1f), *ones = malloc(sizeof(float) * nh * d);
    for (int64_t i = 0; i < nh * d; ++i) ones[i] = 1.f;
    float *u0_d = dev_copy(u0, nh * d), *w1_d = dev_copy(w1, d * d), *w2_d = dev_copy(w2, d * d), *b1_d = dev_copy(b1, d),
          *b2_d = dev_copy(b2, d), *ones_d = dev_copy(ones, nh * d), *uT_d = dev_copy(NULL, nh * d), *du0_d = dev_copy(NULL, nh * d),
          *dw1_d = dev_copy(NULL, d * d), *dw2_d = dev_copy(NULL, d * d), *db1_d = dev_copy(NULL, d), *db2_d = dev_copy(NULL, d);
    ngpde_node_t *plan = NULL;
    CHECK_NG(ngpde_node_gcn2_create(gh, d, NGPDE_ACT_TANH, NGPDE_TABLEAU_TSIT5, steps, dt, 1, &plan));
    int32_t flags = 0, fault = 0;
    CHECK_NG(ngpde_node_flags(plan, &flags));
    CHECK_NG(ngpde_node_gcn2_forward(plan, u0_d, w1_d, b1_d, w2_d, b2_d, uT_d, NULL));
    CHECK_NG(ngpde_node_gcn2_backward(plan, ones_d, du0_d, dw1_d, db1_d, dw2_d, db2_d, NULL));
    CHECK_NG(ngpde_node_fault(plan, NULL, &fault));
    CHECK_HIP(hipDeviceSynchronize());
    printf("plan on a graph with a hub of degree %lld: flags 0x%x%s, fault %d\n", (long long)(hub + 2), flags,
           (flags & NGPDE_NODE_HUB_GEOMETRY) ? " (hub geometry)" : "", fault);
    if (fault) return 4;
    if ((flags & NGPDE_NODE_PERSISTENT_FWD) && !(flags & NGPDE_NODE_HUB_GEOMETRY)) {   /* (NGPDE_NO_PERSISTENT etc.: the replayed plan is fine too) */
      fprintf(stderr, "hub graph: a persistent plan that is not the hub geometry, flags 0x%x\n", flags);
      return 4;
    }
    float *uT = host_copy(uT_d, nh * d), *du0 = host_copy(du0_d, nh * d), *dw1 = host_copy(dw1_d, d * d), *dw2 = host_copy(dw2_d, d * d),
          *db1 = host_copy(db1_d, d), *db2 = host_copy(db2_d, d);
    float *uTo = malloc(sizeof(float) * nh * d), *du0o = malloc(sizeof(float) * nh * d), *dw1o = malloc(sizeof(float) * d * d),
          *dw2o = malloc(sizeof(float) * d * d), *db1o = malloc(sizeof(float) * d), *db2o = malloc(sizeof(float) * d);
    if (ngo_node_gcn2(nh, mh, sh, th, d, NGPDE_ACT_TANH, 1, steps, dt, 1, u0, w1, b1, w2, b2, uTo, du0o, dw1o, db1o, dw2o, db2o)) return 5;
    report("node_gcn2 (hub geometry) u(T)", max_rel(uT, uTo, nh * d), 2e-4);
    report("node_gcn2 (hub geometry) du0", max_rel(du0, du0o, nh * d), 1e-3);
    report("node_gcn2 (hub geometry) dW1", max_rel(dw1, dw1o, d * d), 1e-3);
    report("node_gcn2 (hub geometry) dW2", max_rel(dw2, dw2o, d * d), 1e-3);
    report("node_gcn2 (hub geometry) db1", max_rel(db1, db1o, d), 1e-3);
    report("node_gcn2 (hub geometry) db2", max_rel(db2, db2o, d), 1e-3);
    CHECK_NG(ngpde_node_destroy(plan));
    CHECK_NG(ngpde_graph_destroy(gh));
  }

  /* the lists by target (p order) as the library holds them, for the loop checkers below */
  const int32_t *rp_d = NULL, *col_d = NULL, *eid_d = NULL;
  CHECK_NG(ngpde_graph_csr_by_target(g, &rp_d, &col_d, &eid_d));
  int32_t *rp = malloc(sizeof(int32_t) * (n + 1)), *col = malloc(sizeof(int32_t) * e);
  CHECK_HIP(hipMemcpy(rp, rp_d, sizeof(int32_t) * (n + 1), hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(col, col_d, sizeof(int32_t) * e, hipMemcpyDeviceToHost));

  /* ---- 2. message path of an edge-function layer: m_i = mean_e tanh(W2 tanh(P[t] + Q[s]) + b2), P = Dense(x), Q = Dense(x) ---- */
  {
    const int din = 12, h1 = 32, dw = 16;
    float *x = host_rand(n * din, 1.f), *wp = host_rand(din * h1, 0.3f), *wq = host_rand(din * h1, 0.3f), *bp = host_rand(h1, 0.1f),
          *w2 = host_rand(h1 * dw, 0.3f), *b2 = host_rand(dw, 0.1f);
    float *x_d = dev_copy(x, n * din), *wp_d = dev_copy(wp, din * h1), *wq_d = dev_copy(wq, din * h1), *bp_d = dev_copy(bp, h1),
          *w2_d = dev_copy(w2, h1 * dw), *b2_d = dev_copy(b2, dw), *P_d = dev_copy(NULL, n * h1), *Q_d = dev_copy(NULL, n * h1),
          *out_d = dev_copy(NULL, n * dw), *a1_d = dev_copy(NULL, e * h1), *m_d = dev_copy(NULL, e * dw), *out2_d = dev_copy(NULL, n * dw);
    const float *seg[1] = {x_d};
    const int32_t wdt[1] = {din}, rdv[1] = {1};
    CHECK_NG(ngpde_dense_forward(n, 1, seg, wdt, rdv, h1, NGPDE_ACT_IDENTITY, wp_d, bp_d, P_d, NULL, NULL));
    CHECK_NG(ngpde_dense_forward(n, 1, seg, wdt, rdv, h1, NGPDE_ACT_IDENTITY, wq_d, NULL, Q_d, NULL, NULL));
    const int32_t tdout[1] = {dw}, tact[1] = {NGPDE_ACT_TANH};
    const float *tw[1] = {w2_d}, *tb[1] = {b2_d};
    float *saves[2] = {NULL, NULL};
    if (!ngpde_edge_mlp_supported(g, h1, 1, tdout)) { fprintf(stderr, "fused message path not available\n"); return 6; }
    CHECK_NG(ngpde_edge_mlp_forward(g, h1, NGPDE_ACT_TANH, P_d, Q_d, NULL, 1, tdout, tact, tw, tb, NGPDE_AGGR_MEAN, out_d, saves, NULL));
    /* the same through the primitives */
    CHECK_NG(ngpde_edge_combine_forward(g, h1, NGPDE_ACT_TANH, P_d, Q_d, NULL, a1_d, NULL, NULL));
    const float *seg2[1] = {a1_d};
    const int32_t wdt2[1] = {h1};
    CHECK_NG(ngpde_dense_forward(e, 1, seg2, wdt2, rdv, dw, NGPDE_ACT_TANH, w2_d, b2_d, m_d, NULL, NULL));
    CHECK_NG(ngpde_segment_reduce_forward(g, dw, NGPDE_AGGR_MEAN, m_d, out2_d, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float *out = host_copy(out_d, n * dw), *out2 = host_copy(out2_d, n * dw), *ref = calloc(n * dw, sizeof(float));
    double *P = calloc(n * h1, sizeof(double)), *Q = calloc(n * h1, sizeof(double));
    for (int64_t i = 0; i < n; ++i)
      for (int o = 0; o < h1; ++o) {
        double sp = bp[o], sq = 0;
        for (int k = 0; k < din; ++k) { sp += (double)x[i * din + k] * wp[k * h1 + o]; sq += (double)x[i * din + k] * wq[k * h1 + o]; }
        P[i * h1 + o] = sp; Q[i * h1 + o] = sq;
      }
    for (int64_t i = 0; i < n; ++i) {
      double acc[64] = {0};
      for (int32_t p = rp[i]; p < rp[i + 1]; ++p) {
        double a1[64];
        for (int o = 0; o < h1; ++o) a1[o] = tanh(P[i * h1 + o] + Q[(int64_t)col[p] * h1 + o]);
        for (int o = 0; o < dw; ++o) {
          double z = b2[o];
          for (int k = 0; k < h1; ++k) z += a1[k] * w2[k * dw + o];
          acc[o] += tanh(z);
        }
      }
      for (int o = 0; o < dw; ++o) ref[i * dw + o] = (float)(rp[i + 1] > rp[i] ? acc[o] / (rp[i + 1] - rp[i]) : 0.0);
    }
    report("edge_mlp_forward (one launch)", max_rel(out, ref, n * dw), 2e-4);
    report("primitives (combine/dense/reduce)", max_rel(out2, ref, n * dw), 2e-4);
  }

  /* ---- 3. GNOConv message: m_p = reshape(K_p, out, in) h[s_p]; reassociated form m_p = T[s_p] z_p + Bh[s_p] ---- */
  {
    const int cin = 8, cout = 8, kdim = 16;
    float *h = host_rand(n * cin, 1.f), *z = host_rand(e * kdim, 1.f), *w2 = host_rand(kdim * cin * cout, 0.3f), *b2 = host_rand(cin * cout, 0.2f);
    /* K_p = W2^T z_p + b2 (element o + cout*i), T_j[o][k] = sum_i W2[k][o + cout*i] h_j[i], Bh_j[o] = sum_i b2[o + cout*i] h_j[i] */
    float *K = malloc(sizeof(float) * e * cin * cout), *T = malloc(sizeof(float) * n * cout * kdim), *Bh = malloc(sizeof(float) * n * cout);
    for (int64_t p = 0; p < e; ++p)
      for (int r = 0; r < cin * cout; ++r) {
        double v = b2[r];
        for (int k = 0; k < kdim; ++k) v += (double)z[p * kdim + k] * w2[k * cin * cout + r];
        K[p * cin * cout + r] = (float)v;
      }
    for (int64_t j = 0; j < n; ++j)
      for (int o = 0; o < cout; ++o) {
        double bh = 0;
        for (int i = 0; i < cin; ++i) bh += (double)b2[o + cout * i] * h[j * cin + i];
        Bh[j * cout + o] = (float)bh;
        for (int k = 0; k < kdim; ++k) {
          double v = 0;
          for (int i = 0; i < cin; ++i) v += (double)w2[k * cin * cout + o + cout * i] * h[j * cin + i];
          T[(j * cout + o) * kdim + k] = (float)v;
        }
      }
    float *h_d = dev_copy(h, n * cin), *z_d = dev_copy(z, e * kdim), *K_d = dev_copy(K, e * cin * cout), *T_d = dev_copy(T, n * cout * kdim),
          *Bh_d = dev_copy(Bh, n * cout), *m1_d = dev_copy(NULL, e * cout), *m2_d = dev_copy(NULL, e * cout);
    if (!ngpde_gno_apply_supported(cout, kdim)) return 7;
    CHECK_NG(ngpde_gno_apply_forward(g, cout, kdim, T_d, Bh_d, z_d, m1_d, NULL));
    CHECK_NG(ngpde_gno_contract_forward(g, cin, cout, K_d, h_d, m2_d, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float *m1 = host_copy(m1_d, e * cout), *m2 = host_copy(m2_d, e * cout), *ref = malloc(sizeof(float) * e * cout);
    for (int64_t p = 0; p < e; ++p)
      for (int o = 0; o < cout; ++o) {
        double v = 0;
        for (int i = 0; i < cin; ++i) v += (double)K[p * cin * cout + o + cout * i] * h[(int64_t)col[p] * cin + i];
        ref[p * cout + o] = (float)v;
      }
    report("gno_contract_forward (literal)", max_rel(m2, ref, e * cout), 2e-4);
    report("gno_apply_forward (reassociated)", max_rel(m1, ref, e * cout), 2e-4);
  }

  /* ---- 4. GAT layer in one launch vs its composition, and the Runge-Kutta combination ---- */
  {
    const int d = 64, heads = 4, c = 16;
    /* self loops are edges of the GAT graph: a second handle with them appended */
    int64_t *s2 = malloc(sizeof(int64_t) * (e + n)), *t2 = malloc(sizeof(int64_t) * (e + n));
    memcpy(s2, s, sizeof(int64_t) * e); memcpy(t2, t, sizeof(int64_t) * e);
    for (int64_t i = 0; i < n; ++i) { s2[e + i] = i; t2[e + i] = i; }
    ngpde_graph_t *g2 = NULL;
    CHECK_NG(ngpde_graph_create(n, e + n, s2, t2, 0, 1, &g2));
    CHECK_NG(ngpde_graph_set_gcn_norm(g2, 0, NULL, 0));
    float *x = host_rand(n * d, 1.f), *w = host_rand(d * d, 0.2f), *a = host_rand(2 * c * heads, 0.3f), *b = host_rand(d, 0.1f);
    float *x_d = dev_copy(x, n * d), *w_d = dev_copy(w, d * d), *a_d = dev_copy(a, 2 * c * heads), *b_d = dev_copy(b, d);
    float *y1_d = dev_copy(NULL, n * d), *y2_d = dev_copy(NULL, n * d), *wx_d = dev_copy(NULL, n * d), *agg_d = dev_copy(NULL, n * d),
          *alpha_d = dev_copy(NULL, (e + n) * heads), *al_d = dev_copy(NULL, n * heads), *ar_d = dev_copy(NULL, n * heads);
    if (!ngpde_gat_layer_supported(g2, d, heads, c)) { fprintf(stderr, "one-launch GAT layer not available\n"); return 8; }
    CHECK_NG(ngpde_gat_layer_forward(g2, d, heads, c, 0.2f, NGPDE_ACT_RELU, x_d, w_d, a_d, b_d, y1_d, NULL, NULL, NULL));
    const float *seg[1] = {x_d};
    const int32_t wdt[1] = {d}, rdv[1] = {1};
    CHECK_NG(ngpde_dense_forward(n, 1, seg, wdt, rdv, d, NGPDE_ACT_IDENTITY, w_d, NULL, wx_d, NULL, NULL));
    CHECK_NG(ngpde_gat_forward(g2, heads, c, 0.2f, wx_d, a_d, agg_d, alpha_d, al_d, ar_d, NULL));
    CHECK_NG(ngpde_bias_act_forward(n, d, NGPDE_ACT_RELU, agg_d, NULL, b_d, y2_d, NULL, NULL));
    /* u + 0.5 y1 - 0.25 y2 by ngpde_rk_stage_combine */
    float *comb_d = dev_copy(NULL, n * d);
    const float *terms[2] = {y1_d, y2_d};
    const float coefs[2] = {0.5f, -0.25f};
    CHECK_NG(ngpde_rk_stage_combine(n * d, 1.0f, x_d, 2, terms, coefs, comb_d, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float *y1 = host_copy(y1_d, n * d), *y2 = host_copy(y2_d, n * d), *cb = host_copy(comb_d, n * d), *ref = malloc(sizeof(float) * n * d);
    for (int64_t i = 0; i < n * d; ++i) ref[i] = x[i] + 0.5f * y1[i] - 0.25f * y2[i];
    report("gat_layer_forward vs composition", max_rel(y1, y2, n * d), 1e-4);
    report("rk_stage_combine", max_rel(cb, ref, n * d), 1e-6);
    /* the device-resident solver over the same layer (ngpde_node_gat_*): ONE Euler step of size dt is u + dt * layer(u), bit for
     * bit what ngpde_rk_stage_combine makes of the layer's output; its adjoint of loss = sum(u(T)) must be finite */
    if (ngpde_node_gat_supported(g2, d, heads, c)) {
      const double dtg = 0.125;
      ngpde_node_gat_t *gp = NULL;
      CHECK_NG(ngpde_node_gat_create(g2, heads, c, 0.2f, NGPDE_ACT_RELU, NGPDE_TABLEAU_EULER, 1, dtg, 1, &gp));
      float *uT_d = dev_copy(NULL, n * d), *eul_d = dev_copy(NULL, n * d), *du0_d = dev_copy(NULL, n * d), *dw_d = dev_copy(NULL, d * d),
            *da_d = dev_copy(NULL, 2 * c * heads), *db_d = dev_copy(NULL, d);
      float *ones = malloc(sizeof(float) * n * d);
      for (int64_t i = 0; i < n * d; ++i) ones[i] = 1.0f;
      float *ones_d = dev_copy(ones, n * d);
      CHECK_NG(ngpde_node_gat_forward(gp, x_d, w_d, a_d, b_d, uT_d, NULL));
      const float *t1[1] = {y1_d};
      const float c1[1] = {(float)dtg};
      CHECK_NG(ngpde_rk_stage_combine(n * d, 1.0f, x_d, 1, t1, c1, eul_d, NULL));
      CHECK_NG(ngpde_node_gat_backward(gp, w_d, a_d, ones_d, du0_d, dw_d, da_d, db_d, NULL));
      int32_t fault = 1;
      CHECK_NG(ngpde_node_gat_fault(gp, NULL, &fault));
      float *uT = host_copy(uT_d, n * d), *eul = host_copy(eul_d, n * d), *du0 = host_copy(du0_d, n * d), *dwh = host_copy(dw_d, d * d);
      report("node_gat_forward (one Euler step) vs layer + combine", max_rel(uT, eul, n * d), 0.0);
      int finite = !fault;
      for (int64_t i = 0; i < n * d; ++i) finite &= isfinite(du0[i]) != 0;
      for (int64_t i = 0; i < d * d; ++i) finite &= isfinite(dwh[i]) != 0;
      report("node_gat_backward finite, no fault", finite ? 0.0 : 1.0, 0.5);
      if (ngpde_node_gat_tape_bytes(gp) == 0) { fprintf(stderr, "node_gat: empty tape\n"); return 10; }
      CHECK_NG(ngpde_node_gat_destroy(gp));
    } else {
      fprintf(stderr, "device-resident GAT solver not available on this graph\n");
      return 11;
    }
    CHECK_NG(ngpde_graph_destroy(g2));
  }
  /* ---- 5. node-level Dense layers of the edge-function layers at a streaming size: the pair (P, Q from one pass over h) and its
   *         pullback in one launch, the two-layer chain (psi); checked on a sample of rows / all weights against double loops ---- */
  {
    const int64_t nn = 70001;
    const int dh = 64, dn = 2, din = dh + dn, hw = 64;
    float *h = host_rand(nn * dh, 1.0f), *dd = host_rand(nn * dn, 1.0f), *wp = host_rand(din * hw, 0.12f), *wq = host_rand(din * hw, 0.12f),
          *bp = host_rand(hw, 0.3f), *w2 = host_rand(hw * hw, 0.12f), *b2 = host_rand(hw, 0.3f), *gP = host_rand(nn * hw, 1.0f),
          *gQ = host_rand(nn * hw, 1.0f);
    float *h_d = dev_copy(h, nn * dh), *d_d = dev_copy(dd, nn * dn), *wp_d = dev_copy(wp, din * hw), *wq_d = dev_copy(wq, din * hw),
          *bp_d = dev_copy(bp, hw), *w2_d = dev_copy(w2, hw * hw), *b2_d = dev_copy(b2, hw), *P_d = dev_copy(NULL, nn * hw),
          *Q_d = dev_copy(NULL, nn * hw), *y_d = dev_copy(NULL, nn * hw), *gP_d = dev_copy(gP, nn * hw), *gQ_d = dev_copy(gQ, nn * hw),
          *dh_d = dev_copy(NULL, nn * dh), *dwp_d = dev_copy(NULL, din * hw), *dwq_d = dev_copy(NULL, din * hw), *dbp_d = dev_copy(NULL, hw);
    const float *seg[2] = {h_d, d_d};
    const int32_t wdt[2] = {dh, dn}, rdv[2] = {1, 1};
    CHECK_NG(ngpde_dense_pair_forward(nn, 2, seg, wdt, rdv, hw, NGPDE_ACT_IDENTITY, wp_d, bp_d, P_d, NULL, 2, seg, wdt, rdv, hw,
                                      NGPDE_ACT_IDENTITY, wq_d, NULL, Q_d, NULL, NULL));
    /* psi-like chain on [h | d]: swish then identity; inference form (no a1 / z1 / z2 buffers) when the library fuses it */
    const int fused = ngpde_dense_chain2_fused(nn, 2, seg, wdt, rdv, hw, hw);
    float *a1_d = fused ? NULL : dev_copy(NULL, nn * hw);
    CHECK_NG(ngpde_dense_chain2_forward(nn, 2, seg, wdt, rdv, hw, NGPDE_ACT_SWISH, wp_d, bp_d, a1_d, NULL, hw, NGPDE_ACT_IDENTITY, w2_d, b2_d,
                                        y_d, NULL, NULL));
    const size_t wsb = ngpde_dense_pair_backward_workspace_bytes(nn, 2, seg, wdt, rdv, 2, seg, wdt, rdv, hw);
    if (!wsb) { fprintf(stderr, "pair pullback not available at this size\n"); return 9; }
    void *ws_d = NULL;
    CHECK_HIP(hipMalloc(&ws_d, wsb));
    CHECK_NG(ngpde_dense_pair_backward(nn, 2, seg, wdt, rdv, wp_d, gP_d, dwp_d, dbp_d, 2, seg, wdt, rdv, wq_d, gQ_d, dwq_d, NULL, hw, dh_d,
                                       NULL, ws_d, wsb, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float *P = host_copy(P_d, nn * hw), *Q = host_copy(Q_d, nn * hw), *y = host_copy(y_d, nn * hw), *dhh = host_copy(dh_d, nn * dh),
          *dwp = host_copy(dwp_d, din * hw), *dwq = host_copy(dwq_d, din * hw), *dbp = host_copy(dbp_d, hw);
    const int64_t ns = 2048;   /* sampled rows: every 34th + the ragged tail */
    float *rP = malloc(sizeof(float) * ns * hw), *rQ = malloc(sizeof(float) * ns * hw), *ry = malloc(sizeof(float) * ns * hw),
          *rdh = malloc(sizeof(float) * ns * dh), *oP = malloc(sizeof(float) * ns * hw), *oQ = malloc(sizeof(float) * ns * hw),
          *oy = malloc(sizeof(float) * ns * hw), *odh = malloc(sizeof(float) * ns * dh);
    for (int64_t q = 0; q < ns; ++q) {
      const int64_t i = q < ns - 64 ? q * 34 : nn - (ns - q);
      double x[66], a1[64];
      for (int k = 0; k < dh; ++k) x[k] = h[i * dh + k];
      for (int k = 0; k < dn; ++k) x[dh + k] = dd[i * dn + k];
      for (int o = 0; o < hw; ++o) {
        double sp = bp[o], sq = 0;
        for (int k = 0; k < din; ++k) { sp += x[k] * wp[k * hw + o]; sq += x[k] * wq[k * hw + o]; }
        rP[q * hw + o] = (float)sp; rQ[q * hw + o] = (float)sq;
        a1[o] = sp / (1.0 + exp(-sp));
        oP[q * hw + o] = P[i * hw + o]; oQ[q * hw + o] = Q[i * hw + o]; oy[q * hw + o] = y[i * hw + o];
      }
      for (int o = 0; o < hw; ++o) {
        double z = b2[o];
        for (int k = 0; k < hw; ++k) z += a1[k] * w2[k * hw + o];
        ry[q * hw + o] = (float)z;
      }
      for (int k = 0; k < dh; ++k) {
        double sg = 0;
        for (int o = 0; o < hw; ++o) sg += (double)gP[i * hw + o] * wp[k * hw + o] + (double)gQ[i * hw + o] * wq[k * hw + o];
        rdh[q * dh + k] = (float)sg; odh[q * dh + k] = dhh[i * dh + k];
      }
    }
    double *aw = calloc((size_t)din * hw * 2 + hw, sizeof(double));
    for (int64_t i = 0; i < nn; ++i)
      for (int k = 0; k < din; ++k) {
        const double xv = k < dh ? h[i * dh + k] : dd[i * dn + (k - dh)];
        for (int o = 0; o < hw; ++o) {
          aw[(size_t)k * hw + o] += xv * gP[i * hw + o];
          aw[(size_t)din * hw + (size_t)k * hw + o] += xv * gQ[i * hw + o];
          if (k == 0) aw[(size_t)2 * din * hw + o] += gP[i * hw + o];
        }
      }
    float *rwp = malloc(sizeof(float) * din * hw), *rwq = malloc(sizeof(float) * din * hw), *rbp = malloc(sizeof(float) * hw);
    for (int k = 0; k < din * hw; ++k) { rwp[k] = (float)aw[k]; rwq[k] = (float)aw[(size_t)din * hw + k]; }
    for (int o = 0; o < hw; ++o) rbp[o] = (float)aw[(size_t)2 * din * hw + o];
    report("dense_pair_forward P", max_rel(oP, rP, ns * hw), 1e-4);
    report("dense_pair_forward Q", max_rel(oQ, rQ, ns * hw), 1e-4);
    report(fused ? "dense_chain2_forward (one launch)" : "dense_chain2_forward (two launches)", max_rel(oy, ry, ns * hw), 1e-4);
    report("dense_pair_backward dh", max_rel(odh, rdh, ns * dh), 2e-4);
    report("dense_pair_backward dWp", max_rel(dwp, rwp, din * hw), 3e-4);
    report("dense_pair_backward dWq", max_rel(dwq, rwq, din * hw), 3e-4);
    report("dense_pair_backward dbp", max_rel(dbp, rbp, hw), 3e-4);
  }
  /* ---- NeuralODE(VMHConv(phi, gamma)) device-resident (ngpde_node_vmh_*; docs/src/tutorials/VMH.md:75-89, src/layers.jl:308-332): two
   * Euler steps on a scalar state with positions as node data, saved after every step (saveat); values against a double-precision
   * loop over the edge list, du0 of L = sum of all saved states against central differences of that loop ---- */
  {
    const int pd = 1, hp = 8, mw = 4, hg = 8, steps = 2;
    const float dtv = 0.05f;
    const int32_t phi_dims[3] = {2 + pd, hp, mw}, gam_dims[3] = {1 + mw, hg, 1};
    const int32_t acts[2] = {NGPDE_ACT_TANH, NGPDE_ACT_IDENTITY};
    ngpde_graph_t *gv = NULL;   /* the plain handle of the edge-function layers: no self loops */
    CHECK_NG(ngpde_graph_create(n, e, s, t, /*index_base=*/0, 1, &gv));
    CHECK_NG(ngpde_graph_set_gcn_norm(gv, /*add_self_loops=*/0, NULL, 0));
    if (ngpde_node_vmh_supported(gv, 1, pd, 2, phi_dims, acts, 2, gam_dims, acts, NGPDE_AGGR_MEAN)) {
      float *pos = malloc(sizeof(float) * n), *u0 = host_rand(n, 1.f);
      for (int64_t i = 0; i < n; ++i) pos[i] = (float)i / (float)n;
      float *wp1 = host_rand((size_t)phi_dims[0] * hp, 0.5f), *bp1 = host_rand(hp, 0.2f), *wp2 = host_rand((size_t)hp * mw, 0.5f),
            *bp2 = host_rand(mw, 0.2f), *wg1 = host_rand((size_t)gam_dims[0] * hg, 0.5f), *bg1 = host_rand(hg, 0.2f),
            *wg2 = host_rand(hg, 0.5f), *bg2 = host_rand(1, 0.2f);
      float *pos_d = dev_copy(pos, n), *u0_d = dev_copy(u0, n), *us_d = dev_copy(NULL, (size_t)(steps + 1) * n), *du0_d = dev_copy(NULL, n);
      const float *pw[2] = {dev_copy(wp1, (size_t)phi_dims[0] * hp), dev_copy(wp2, (size_t)hp * mw)}, *pb[2] = {dev_copy(bp1, hp), dev_copy(bp2, mw)};
      const float *gw[2] = {dev_copy(wg1, (size_t)gam_dims[0] * hg), dev_copy(wg2, hg)}, *gb[2] = {dev_copy(bg1, hg), dev_copy(bg2, 1)};
      float *dpw[2] = {dev_copy(NULL, (size_t)phi_dims[0] * hp), dev_copy(NULL, (size_t)hp * mw)}, *dpb[2] = {dev_copy(NULL, hp), dev_copy(NULL, mw)};
      float *dgw[2] = {dev_copy(NULL, (size_t)gam_dims[0] * hg), dev_copy(NULL, hg)}, *dgb[2] = {dev_copy(NULL, hg), dev_copy(NULL, 1)};
      float *ones = malloc(sizeof(float) * (steps + 1) * n);
      for (int64_t i = 0; i < (steps + 1) * n; ++i) ones[i] = 1.f;
      float *ones_d = dev_copy(ones, (size_t)(steps + 1) * n);
      ngpde_node_vmh_t *vp = NULL;
      CHECK_NG(ngpde_node_vmh_create(gv, 1, pd, pos_d, 2, phi_dims, acts, 2, gam_dims, acts, NGPDE_AGGR_MEAN, NGPDE_TABLEAU_EULER, steps, dtv, 1, &vp));
      CHECK_NG(ngpde_node_vmh_forward_saveat(vp, u0_d, pw, pb, gw, gb, 1, 1, us_d, NULL));
      CHECK_NG(ngpde_node_vmh_backward_saveat(vp, pw, gw, 1, 1, ones_d, du0_d, dpw, dpb, dgw, dgb, NULL));
      int32_t fault = 1;
      CHECK_NG(ngpde_node_vmh_fault(vp, NULL, &fault));
      CHECK_HIP(hipDeviceSynchronize());
      if (fault || ngpde_node_vmh_tape_bytes(vp) == 0) { fprintf(stderr, "node_vmh: fault %d\n", fault); return 11; }
      /* the same in double precision: loss(u0) = sum over the saved states (u0 itself, u1, u2) */
      double *deg = calloc(n, sizeof(double)), *cur = malloc(sizeof(double) * n), *nxt = malloc(sizeof(double) * n),
             *msg = malloc(sizeof(double) * n * mw), *ref = malloc(sizeof(double) * (steps + 1) * n);
      for (int64_t k = 0; k < e; ++k) deg[t[k]] += 1.0;
      for (int pass = 0; pass < 7; ++pass) {   /* pass 0: the values; 1..6: u0[probe] +- eps for three probes */
        const int64_t probe[3] = {3, n / 2, n - 7};
        const double eps = 1e-3;
        for (int64_t i = 0; i < n; ++i) cur[i] = u0[i];
        if (pass > 0) cur[probe[(pass - 1) / 2]] += ((pass - 1) % 2 ? -eps : eps);
        double loss = 0;
        for (int64_t i = 0; i < n; ++i) { loss += cur[i]; if (pass == 0) ref[i] = cur[i]; }
        for (int st = 0; st < steps; ++st) {
          memset(msg, 0, sizeof(double) * n * mw);
          for (int64_t k = 0; k < e; ++k) {
            const int64_t a = t[k], b = s[k];
            const double in[3] = {cur[a], cur[b] - cur[a], (double)pos[b] - (double)pos[a]};
            double hid[8];
            for (int o = 0; o < hp; ++o) {
              double z = bp1[o];
              for (int c2 = 0; c2 < 3; ++c2) z += in[c2] * wp1[c2 * hp + o];
              hid[o] = tanh(z);
            }
            for (int o = 0; o < mw; ++o) {
              double z = bp2[o];
              for (int c2 = 0; c2 < hp; ++c2) z += hid[c2] * wp2[c2 * mw + o];
              msg[a * mw + o] += z;
            }
          }
          for (int64_t i = 0; i < n; ++i) {
            double gin[5] = {cur[i], 0, 0, 0, 0}, hid[8], z2 = bg2[0];
            for (int o = 0; o < mw; ++o) gin[1 + o] = deg[i] > 0 ? msg[i * mw + o] / deg[i] : 0.0;
            for (int o = 0; o < hg; ++o) {
              double z = bg1[o];
              for (int c2 = 0; c2 < 5; ++c2) z += gin[c2] * wg1[c2 * hg + o];
              hid[o] = tanh(z);
            }
            for (int c2 = 0; c2 < hg; ++c2) z2 += hid[c2] * wg2[c2];
            nxt[i] = cur[i] + (double)dtv * z2;
          }
          for (int64_t i = 0; i < n; ++i) { cur[i] = nxt[i]; loss += cur[i]; if (pass == 0) ref[(st + 1) * n + i] = cur[i]; }
        }
        if (pass == 0) {
          float *us = host_copy(us_d, (size_t)(steps + 1) * n), *reff = malloc(sizeof(float) * (steps + 1) * n);
          for (int64_t i = 0; i < (steps + 1) * n; ++i) reff[i] = (float)ref[i];
          report("node_vmh_forward_saveat (2 Euler steps)", max_rel(us, reff, (size_t)(steps + 1) * n), 1e-4);
        } else {
          static double lp;
          if (pass % 2) lp = loss;
          else {
            const float *du0 = host_copy(du0_d, n);
            const double fd = (lp - loss) / (2 * eps), got = du0[probe[(pass - 1) / 2]];
            report("node_vmh_backward_saveat du0 vs central difference", fabs(got - fd) / (fabs(fd) > 1e-30 ? fabs(fd) : 1e-30), 5e-3);
          }
        }
      }
      CHECK_NG(ngpde_node_vmh_destroy(vp));
    } else {
      printf("node_vmh: plan not supported on this graph (skipped)\n");
    }
    CHECK_NG(ngpde_graph_destroy(gv));
  }
  CHECK_NG(ngpde_graph_destroy(g));
  return fails ? 1 : 0;
}
