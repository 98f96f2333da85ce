/* A plain-C caller of the solver level's ONE create call (include/ngpde.h: ngpde_ode_create / _forward / _backward; csrc/api_ode.hip):
 * what a Julia host binds for  NeuralODE(model, tspan, Tsit5(); saveat)  (/root/reference/docs/src/tutorials/graph_node.md:44-66,
 * docs/src/tutorials/VMH.md:85-89, :104-108).  No Python, no torch, no C++ on the calling side.
 *   1. NeuralODE(VMHConv(phi, gamma)) with saveat on a point cloud: the saved states and every gradient are, bit for bit, those of the
 *      plan's own entries (ngpde_node_vmh_*_saveat, which tests/test_node_vmh_gpu.py holds against the float64 oracle); slot 0 is u0,
 *      the last slot the u(T) of the solve without saveat; one Euler step is checked against a double-precision loop written here.
 *   2. NeuralODE(Chain(GCNConv, GCNConv)): u(T) and the gradients bit for bit those of ngpde_node_gcn2_*.
 *   3. What the entry refuses: stacks that do not chain as the layer feeds them (DimensionMismatch, src/layers.jl:316, :328), a GAT
 *      right-hand side outside the plan's shape (ERR_UNSUPPORTED: the host steps it with ngpde_rk_stage_combine), saveat on GCN2.
 * Exit code 0 = all of it.  Built and run by tests/test_c_abi_gpu.py. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ngpde.h"

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
#define EXPECT(cond, what)                                                            \
  do {                                                                                \
    if (!(cond)) { printf("FAIL: %s\n", what); ++failures; } else printf("ok: %s\n", what); \
  } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float rnd(void) {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (float)((double)(rng_state >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}
static float *dev_copy(const float *h, size_t n) {
  float *d = NULL;
  if (hipMalloc((void **)&d, (n ? n : 1) * sizeof(float)) != hipSuccess) return NULL;
  if (h && n && hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
}
static int same_bits(const float *da, const float *db, size_t n) {
  float *a = malloc(n * sizeof(float)), *b = malloc(n * sizeof(float));
  int ok = hipMemcpy(a, da, n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess && hipMemcpy(b, db, n * sizeof(float), hipMemcpyDeviceToHost) == hipSuccess &&
           memcmp(a, b, n * sizeof(float)) == 0;
  free(a); free(b);
  return ok;
}

int main(void) {
  int failures = 0;
  printf("%s\n", ngpde_version());
  /* ---------------- 1. VMHConv on a point cloud: 640 points on a wavy ring, four neighbours each ---------------- */
  const int64_t n = 640, e = 4 * n;
  int64_t *s = malloc(sizeof(int64_t) * e), *t = malloc(sizeof(int64_t) * e);
  float *pos = malloc(sizeof(float) * n * 2);
  for (int64_t i = 0; i < n; ++i) {
    const double a = 6.283185307179586 * (double)i / (double)n;
    pos[2 * i] = (float)((1.0 + 0.1 * sin(5 * a)) * cos(a));
    pos[2 * i + 1] = (float)((1.0 + 0.1 * sin(5 * a)) * sin(a));
    const int64_t nb[4] = {(i + 1) % n, (i + n - 1) % n, (i + 2) % n, (i + n - 2) % n};
    for (int k = 0; k < 4; ++k) { s[4 * i + k] = nb[k]; t[4 * i + k] = i; }
  }
  ngpde_graph_t *g = NULL;
  CHECK_NG(ngpde_graph_create(n, e, s, t, /*index_base=*/0, /*n_graphs=*/1, &g));
  CHECK_NG(ngpde_graph_set_gcn_norm(g, /*add_self_loops=*/0, NULL, 0));   /* (the plain handle: this call also builds the tile schedule) */
  /* phi: [h_i; h_j - h_i; x_j - x_i] (4) => 16 (tanh) => 8;  gamma: [h_i; m_i] (9) => 16 (tanh) => 1 */
  const int phi_d[3] = {4, 16, 8}, gam_d[3] = {9, 16, 1};
  float *hw[4], *hb[4];          /* host copies: phi layer 1, 2, gamma layer 1, 2 */
  const int din[4] = {4, 16, 9, 16}, dout[4] = {16, 8, 16, 1};
  float *w_d[4], *b_d[4], *gw_a[4], *gb_a[4], *gw_b[4], *gb_b[4];
  for (int l = 0; l < 4; ++l) {
    hw[l] = malloc(sizeof(float) * din[l] * dout[l]);
    hb[l] = malloc(sizeof(float) * dout[l]);
    for (int k = 0; k < din[l] * dout[l]; ++k) hw[l][k] = 0.4f * rnd();
    for (int k = 0; k < dout[l]; ++k) hb[l][k] = 0.1f * rnd();
    w_d[l] = dev_copy(hw[l], (size_t)din[l] * dout[l]);
    b_d[l] = dev_copy(hb[l], dout[l]);
    gw_a[l] = dev_copy(NULL, (size_t)din[l] * dout[l]); gb_a[l] = dev_copy(NULL, dout[l]);
    gw_b[l] = dev_copy(NULL, (size_t)din[l] * dout[l]); gb_b[l] = dev_copy(NULL, dout[l]);
  }
  float *u0 = malloc(sizeof(float) * n);
  for (int64_t i = 0; i < n; ++i) u0[i] = rnd();
  float *u0_d = dev_copy(u0, n), *pos_d = dev_copy(pos, 2 * n);
  const int steps = 6, save_every = 2, slots = steps / save_every + 1;
  float *dus = malloc(sizeof(float) * slots * n);
  for (int64_t i = 0; i < slots * n; ++i) dus[i] = rnd();
  float *us_a = dev_copy(NULL, slots * n), *us_b = dev_copy(NULL, slots * n), *dus_d = dev_copy(dus, slots * n);
  float *du0_a = dev_copy(NULL, n), *du0_b = dev_copy(NULL, n), *uT_a = dev_copy(NULL, n);

  ngpde_ode_desc_t d;
  memset(&d, 0, sizeof d);
  d.rhs = NGPDE_RHS_VMH; d.tableau = NGPDE_TABLEAU_TSIT5; d.n_steps = steps; d.with_backward = 1; d.members = 1; d.dt = 0.05;
  d.width = 1; d.pos_width = 2; d.aggr = NGPDE_AGGR_MEAN; d.pos = pos_d;
  d.n_phi = 2; d.n_gamma = 2;
  for (int l = 0; l < 3; ++l) { d.phi_dims[l] = phi_d[l]; d.gamma_dims[l] = gam_d[l]; }
  d.phi_acts[0] = NGPDE_ACT_TANH; d.phi_acts[1] = NGPDE_ACT_IDENTITY; d.gamma_acts[0] = NGPDE_ACT_TANH; d.gamma_acts[1] = NGPDE_ACT_IDENTITY;
  ngpde_ode_t *ode = NULL;
  int32_t flags = 0;
  CHECK_NG(ngpde_ode_create(g, &d, &ode, &flags));
  EXPECT((flags & NGPDE_NODE_PERSISTENT_FWD) && (flags & NGPDE_NODE_PERSISTENT_BWD), "VMH: the create call chose the device-resident plan");
  ngpde_ode_params_t prm;
  ngpde_ode_grads_t gr;
  memset(&prm, 0, sizeof prm);
  memset(&gr, 0, sizeof gr);
  for (int l = 0; l < 2; ++l) {
    prm.first.weight[l] = w_d[l]; prm.first.bias[l] = b_d[l]; prm.second.weight[l] = w_d[2 + l]; prm.second.bias[l] = b_d[2 + l];
    gr.first.dweight[l] = gw_a[l]; gr.first.dbias[l] = gb_a[l]; gr.second.dweight[l] = gw_a[2 + l]; gr.second.dbias[l] = gb_a[2 + l];
  }
  CHECK_NG(ngpde_ode_forward(ode, u0_d, &prm, save_every, 1, us_a, NULL));
  CHECK_NG(ngpde_ode_backward(ode, &prm, save_every, 1, dus_d, du0_a, &gr, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  /* the plan's own entries on a second plan */
  ngpde_node_vmh_t *vmh = NULL;
  CHECK_NG(ngpde_node_vmh_create(g, 1, 2, pos_d, 2, phi_d, d.phi_acts, 2, gam_d, d.gamma_acts, NGPDE_AGGR_MEAN, NGPDE_TABLEAU_TSIT5, steps, 0.05, 1, &vmh));
  const float *pw[2] = {w_d[0], w_d[1]}, *pb[2] = {b_d[0], b_d[1]}, *gw[2] = {w_d[2], w_d[3]}, *gb[2] = {b_d[2], b_d[3]};
  float *dpw[2] = {gw_b[0], gw_b[1]}, *dpb[2] = {gb_b[0], gb_b[1]}, *dgw[2] = {gw_b[2], gw_b[3]}, *dgb[2] = {gb_b[2], gb_b[3]};
  CHECK_NG(ngpde_node_vmh_forward_saveat(vmh, u0_d, pw, pb, gw, gb, save_every, 1, us_b, NULL));
  CHECK_NG(ngpde_node_vmh_backward_saveat(vmh, pw, gw, save_every, 1, dus_d, du0_b, dpw, dpb, dgw, dgb, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  EXPECT(same_bits(us_a, us_b, slots * n), "VMH saveat: saved states = ngpde_node_vmh_forward_saveat's, bit for bit");
  EXPECT(same_bits(du0_a, du0_b, n), "VMH saveat: du0 bit for bit");
  int all = 1;
  for (int l = 0; l < 4; ++l) all = all && same_bits(gw_a[l], gw_b[l], (size_t)din[l] * dout[l]) && same_bits(gb_a[l], gb_b[l], dout[l]);
  EXPECT(all, "VMH saveat: every weight and bias gradient bit for bit");
  EXPECT(same_bits(us_a, u0_d, n), "VMH saveat: slot 0 is u0");
  CHECK_NG(ngpde_ode_forward(ode, u0_d, &prm, 0, 0, uT_a, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  EXPECT(same_bits(uT_a, us_a + (size_t)(slots - 1) * n, n), "VMH: the last saved state is the u(T) of the solve without saveat");
  { /* one Euler step against a double-precision loop of src/layers.jl:308-332 written here */
    ngpde_ode_desc_t de = d;
    de.tableau = NGPDE_TABLEAU_EULER; de.n_steps = 1; de.dt = 0.1; de.with_backward = 0; de.pos = pos_d;
    ngpde_ode_t *oe = NULL;
    CHECK_NG(ngpde_ode_create(g, &de, &oe, NULL));
    CHECK_NG(ngpde_ode_forward(oe, u0_d, &prm, 0, 0, uT_a, NULL));
    CHECK_HIP(hipDeviceSynchronize());
    float *got = malloc(sizeof(float) * n);
    CHECK_HIP(hipMemcpy(got, uT_a, n * sizeof(float), hipMemcpyDeviceToHost));
    double err = 0, ref = 0;
    double *msum = calloc((size_t)n * 8, sizeof(double));
    int *cnt = calloc(n, sizeof(int));
    for (int64_t k = 0; k < e; ++k) {
      const int64_t i = t[k], j = s[k];
      const double in[4] = {u0[i], (double)u0[j] - u0[i], (double)pos[2 * j] - pos[2 * i], (double)pos[2 * j + 1] - pos[2 * i + 1]};
      double h1[16];
      for (int o = 0; o < 16; ++o) {
        double a = hb[0][o];
        for (int q = 0; q < 4; ++q) a += in[q] * hw[0][q * 16 + o];
        h1[o] = tanh(a);
      }
      for (int o = 0; o < 8; ++o) {
        double a = hb[1][o];
        for (int q = 0; q < 16; ++q) a += h1[q] * hw[1][q * 8 + o];
        msum[i * 8 + o] += a;
      }
      cnt[i] += 1;
    }
    for (int64_t i = 0; i < n; ++i) {
      double in[9] = {u0[i]};
      for (int o = 0; o < 8; ++o) in[1 + o] = cnt[i] ? msum[i * 8 + o] / cnt[i] : 0.0;
      double h1[16], y = hb[3][0];
      for (int o = 0; o < 16; ++o) {
        double a = hb[2][o];
        for (int q = 0; q < 9; ++q) a += in[q] * hw[2][q * 16 + o];
        h1[o] = tanh(a);
      }
      for (int q = 0; q < 16; ++q) y += h1[q] * hw[3][q];
      const double want = u0[i] + 0.1 * y;
      if (fabs(want - got[i]) > err) err = fabs(want - got[i]);
      if (fabs(want) > ref) ref = fabs(want);
    }
    printf("VMH Euler step against the double-precision loop: max err %.3e (max |u| %.3f)\n", err, ref);
    EXPECT(err <= 1e-4 * ref + 1e-5, "VMH: one Euler step within 1e-4 of the double-precision loop");
    CHECK_NG(ngpde_ode_destroy(oe));
    free(got); free(msum); free(cnt);
  }
  /* ---------------- 3a. what the entry refuses ---------------- */
  {
    ngpde_ode_desc_t bad = d;
    bad.pos = pos_d;
    bad.phi_dims[0] = 5;
    ngpde_ode_t *ob = NULL;
    EXPECT(ngpde_ode_create(g, &bad, &ob, NULL) == NGPDE_ERR_DIMENSION_MISMATCH && strstr(ngpde_last_error(), "phi.layer_1 takes 5 inputs, the layer feeds it 4"),
           "VMH: phi's first layer must take [h_i; h_j - h_i; x_j - x_i] (DimensionMismatch)");
    bad = d; bad.pos = pos_d; bad.gamma_dims[0] = 8;
    EXPECT(ngpde_ode_create(g, &bad, &ob, NULL) == NGPDE_ERR_DIMENSION_MISMATCH, "VMH: gamma's first layer must take [h_i; m_i] (DimensionMismatch)");
    bad = d; bad.pos = pos_d; bad.gamma_dims[2] = 2;
    EXPECT(ngpde_ode_create(g, &bad, &ob, NULL) == NGPDE_ERR_DIMENSION_MISMATCH, "VMH: gamma must return the state's width (DimensionMismatch)");
    bad = d; bad.rhs = NGPDE_RHS_GAT; bad.width = 48; bad.heads = 3; bad.head_width = 16;
    EXPECT(ngpde_ode_create(g, &bad, &ob, NULL) == NGPDE_ERR_UNSUPPORTED && ob == NULL, "GAT 48 => 3 x 16: no device-resident plan (ERR_UNSUPPORTED: the host steps it)");
    bad = d; bad.rhs = 9;
    EXPECT(ngpde_ode_create(g, &bad, &ob, NULL) == NGPDE_ERR_INVALID_ARGUMENT, "unknown right-hand side (ERR_INVALID_ARGUMENT)");
  }
  CHECK_NG(ngpde_ode_destroy(ode));
  CHECK_NG(ngpde_node_vmh_destroy(vmh));
  /* ---------------- 2. Chain(GCNConv(32 => 32, relu), GCNConv(32 => 32, relu)), Tsit5 x 3 ---------------- */
  {
    const int w = 32;
    ngpde_graph_t *gn = NULL;   /* the same structure with GCNConv's normalisation (add_self_loops, src/layers.jl:210-226) */
    CHECK_NG(ngpde_graph_create(n, e, s, t, 0, 1, &gn));
    CHECK_NG(ngpde_graph_set_gcn_norm(gn, 1, NULL, 0));
    float *x = malloc(sizeof(float) * n * w), *w1 = malloc(sizeof(float) * w * w), *w2 = malloc(sizeof(float) * w * w), *bb = malloc(sizeof(float) * w);
    for (int64_t i = 0; i < n * w; ++i) x[i] = rnd();
    for (int i = 0; i < w * w; ++i) { w1[i] = 0.2f * rnd(); w2[i] = 0.2f * rnd(); }
    for (int i = 0; i < w; ++i) bb[i] = 0.05f * rnd();
    float *x_d = dev_copy(x, n * w), *w1_d = dev_copy(w1, w * w), *w2_d = dev_copy(w2, w * w), *b_dv = dev_copy(bb, w), *seed_d = dev_copy(x, n * w);
    float *o[2][6];
    for (int v = 0; v < 2; ++v) {
      o[v][0] = dev_copy(NULL, n * w); o[v][1] = dev_copy(NULL, n * w);
      o[v][2] = dev_copy(NULL, w * w); o[v][3] = dev_copy(NULL, w); o[v][4] = dev_copy(NULL, w * w); o[v][5] = dev_copy(NULL, w);
    }
    ngpde_ode_desc_t dg;
    memset(&dg, 0, sizeof dg);
    dg.rhs = NGPDE_RHS_GCN2; dg.tableau = NGPDE_TABLEAU_TSIT5; dg.n_steps = 3; dg.with_backward = 1; dg.members = 1; dg.dt = 0.1; dg.width = w; dg.act = NGPDE_ACT_RELU;
    ngpde_ode_t *og = NULL;
    CHECK_NG(ngpde_ode_create(gn, &dg, &og, &flags));
    printf("GCN2 plan flags 0x%x\n", flags);
    ngpde_ode_params_t pg;
    ngpde_ode_grads_t gg;
    memset(&pg, 0, sizeof pg);
    memset(&gg, 0, sizeof gg);
    pg.first.weight[0] = w1_d; pg.first.weight[1] = w2_d; pg.first.bias[0] = b_dv; pg.first.bias[1] = b_dv;
    gg.first.dweight[0] = o[0][2]; gg.first.dbias[0] = o[0][3]; gg.first.dweight[1] = o[0][4]; gg.first.dbias[1] = o[0][5];
    CHECK_NG(ngpde_ode_forward(og, x_d, &pg, 0, 0, o[0][0], NULL));
    CHECK_NG(ngpde_ode_backward(og, &pg, 0, 0, seed_d, o[0][1], &gg, NULL));
    EXPECT(ngpde_ode_forward(og, x_d, &pg, 2, 1, o[0][0], NULL) == NGPDE_ERR_UNSUPPORTED, "GCN2: saveat is the VMH plan's (ERR_UNSUPPORTED)");
    ngpde_node_t *pl = NULL;
    CHECK_NG(ngpde_node_gcn2_create(gn, w, NGPDE_ACT_RELU, NGPDE_TABLEAU_TSIT5, 3, 0.1f, 1, &pl));
    CHECK_NG(ngpde_node_gcn2_forward(pl, x_d, w1_d, b_dv, w2_d, b_dv, o[1][0], NULL));
    CHECK_NG(ngpde_node_gcn2_backward(pl, seed_d, o[1][1], o[1][2], o[1][3], o[1][4], o[1][5], NULL));
    CHECK_HIP(hipDeviceSynchronize());
    const size_t len[6] = {(size_t)n * w, (size_t)n * w, (size_t)w * w, (size_t)w, (size_t)w * w, (size_t)w};
    all = 1;
    for (int k = 0; k < 6; ++k) all = all && same_bits(o[0][k], o[1][k], len[k]);
    EXPECT(all, "GCN2: u(T), du0, dW1, db1, dW2, db2 = ngpde_node_gcn2_*'s, bit for bit");
    int32_t fault = 1;
    CHECK_NG(ngpde_ode_fault(og, NULL, &fault));
    EXPECT(fault == 0 && ngpde_ode_tape_bytes(og) > 0, "GCN2: no fault, the plan owns a tape");
    CHECK_NG(ngpde_ode_destroy(og));
    CHECK_NG(ngpde_node_destroy(pl));
    CHECK_NG(ngpde_graph_destroy(gn));
  }
  CHECK_NG(ngpde_graph_destroy(g));
  if (failures) { printf("%d FAILED\n", failures); return 1; }
  printf("all comparisons within tolerance\n");
  return 0;
}
