/* A plain-C caller of the drop-in boundary (include/ngpde.h): no Python, no torch, no C++ -- what a ccall / cgo / JNI
 * binding sees.  Builds a graph handle from the 1-based COO vectors a Julia GNNGraph holds, runs GCNConv forward and its
 * pullback on device buffers it allocated itself with the HIP runtime, and checks both against the C restatement of the
 * reference algorithm (oracle/ngpde_oracle.c, the checker).  Exit code 0 = parity within the float32 tolerance.
 * Built and run by tests/test_c_abi_gpu.py. */
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "ngpde.h"

void ngo_gcn_forward(int64_t n, int64_t e, const int64_t *s, const int64_t *t, int self_loops, int din, int dout, int act,
                     const float *x, const float *wt, const float *bias, float *y, float *x3_out, float *z_out);
void ngo_gcn_backward(int64_t n, int64_t e, const int64_t *s, const int64_t *t, int self_loops, int din, int dout, int act,
                      const float *wt, const float *z, const float *x3, const float *dy, float *dx, float *dwt, float *db);

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

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static float rnd(void) {   /* uniform in (-1, 1) */
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (float)((double)(rng_state >> 11) / 9007199254740992.0 * 2.0 - 1.0);
}
static float *dev_copy(const float *h, size_t n) {
  float *d = NULL;
  if (hipMalloc((void **)&d, (n ? n : 1) * sizeof(float)) != hipSuccess) return NULL;
  if (n && hipMemcpy(d, h, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
  return d;
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

int main(int argc, char **argv) {
  const int64_t n = 1500;
  const int d = argc > 1 ? atoi(argv[1]) : 64;   /* 64: the fused kernels; 24: the any-width path */
  /* ring with chords: i <-> i+1, i <-> i+7, i -> i+31 (the last one directed: in- and out-lists differ) */
  const int64_t e = 5 * n;
  int64_t *s = malloc(sizeof(int64_t) * e), *t = malloc(sizeof(int64_t) * e);
  int64_t m = 0;
  for (int64_t i = 0; i < n; ++i) {
    const int64_t a = (i + 1) % n, b = (i + 7) % n, c = (i + 31) % n;
    s[m] = i + 1; t[m++] = a + 1; s[m] = a + 1; t[m++] = i + 1;
    s[m] = i + 1; t[m++] = b + 1; s[m] = b + 1; t[m++] = i + 1;
    s[m] = i + 1; t[m++] = c + 1;
  }
  float *x = malloc(sizeof(float) * n * d), *w = malloc(sizeof(float) * d * d), *bias = malloc(sizeof(float) * d);
  float *dy = malloc(sizeof(float) * n * d);
  for (int64_t i = 0; i < n * d; ++i) { x[i] = rnd(); dy[i] = rnd(); }
  for (int i = 0; i < d * d; ++i) w[i] = rnd() * 0.3f;
  for (int i = 0; i < d; ++i) bias[i] = rnd() * 0.1f;

  /* ---- the library, through its C ABI ---- */
  printf("%s\n", ngpde_version());
  ngpde_graph_t *g = NULL;
  CHECK_NG(ngpde_graph_create(n, e, s, t, /*index_base=*/1, /*n_graphs=*/1, &g));
  CHECK_NG(ngpde_graph_set_gcn_norm(g, /*add_self_loops=*/1, NULL, 0));
  float *dx_ = dev_copy(x, n * d), *dw_ = dev_copy(w, d * d), *db_ = dev_copy(bias, d), *ddy = dev_copy(dy, n * d);
  float *y_d = dev_copy(x, n * d), *agg_d = dev_copy(x, n * d), *z_d = dev_copy(x, n * d);
  float *gx_d = dev_copy(x, n * d), *gw_d = dev_copy(w, d * d), *gb_d = dev_copy(bias, d);
  if (!dx_ || !dw_ || !db_ || !ddy || !y_d || !agg_d || !z_d || !gx_d || !gw_d || !gb_d) return 2;
  size_t ws_f = ngpde_gcn_workspace_bytes(g, d, d, 0), ws_b = ngpde_gcn_workspace_bytes(g, d, d, 1);
  void *ws = NULL;
  CHECK_HIP(hipMalloc(&ws, (ws_f > ws_b ? ws_f : ws_b) + 16));
  CHECK_NG(ngpde_gcn_forward(g, d, d, NGPDE_ACT_TANH, dx_, dw_, db_, y_d, agg_d, z_d, ws, ws_f, NULL));
  CHECK_NG(ngpde_gcn_backward(g, d, d, NGPDE_ACT_TANH, dx_, dw_, z_d, agg_d, ddy, gx_d, gw_d, gb_d, ws, ws_b, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  float *y = malloc(sizeof(float) * n * d), *gx = malloc(sizeof(float) * n * d), *gw = malloc(sizeof(float) * d * d),
        *gb = malloc(sizeof(float) * d);
  CHECK_HIP(hipMemcpy(y, y_d, sizeof(float) * n * d, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gx, gx_d, sizeof(float) * n * d, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gw, gw_d, sizeof(float) * d * d, hipMemcpyDeviceToHost));
  CHECK_HIP(hipMemcpy(gb, gb_d, sizeof(float) * d, hipMemcpyDeviceToHost));
  /* error convention: a status code and a message, no abort across the boundary */
  if (ngpde_gcn_forward(g, d, d, 99, dx_, dw_, db_, y_d, NULL, NULL, ws, ws_f, NULL) == NGPDE_OK) return 4;
  if (ngpde_last_error()[0] == 0) return 4;
  CHECK_NG(ngpde_graph_destroy(g));

  /* ---- the checker: C restatement of the reference algorithm on the host (0-based vectors) ---- */
  for (int64_t k = 0; k < e; ++k) { s[k] -= 1; t[k] -= 1; }
  float *yo = malloc(sizeof(float) * n * d), *x3 = malloc(sizeof(float) * n * d), *zo = malloc(sizeof(float) * n * d);
  float *gxo = malloc(sizeof(float) * n * d), *gwo = calloc(d * d, sizeof(float)), *gbo = calloc(d, sizeof(float));
  ngo_gcn_forward(n, e, s, t, 1, d, d, NGPDE_ACT_TANH, x, w, bias, yo, x3, zo);
  ngo_gcn_backward(n, e, s, t, 1, d, d, NGPDE_ACT_TANH, w, zo, x3, dy, gxo, gwo, gbo);
  const double ey = max_rel(y, yo, n * d), ex = max_rel(gx, gxo, n * d), ew = max_rel(gw, gwo, d * d), eb = max_rel(gb, gbo, d);
  printf("d=%d  y %.2e  dx %.2e  dW %.2e  db %.2e\n", d, ey, ex, ew, eb);
  /* forward 1e-4, gradients 2e-4 relative (float32 on both sides here: twice that) */
  return (ey <= 2e-4 && ex <= 4e-4 && ew <= 4e-4 && eb <= 4e-4) ? 0 : 1;
}
