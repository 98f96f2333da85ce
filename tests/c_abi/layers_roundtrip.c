/* Plain-C caller of the layer-level entries (include/ngpde.h: ngpde_edge_layer_*, ngpde_gno_layer_*): an MPPDEConv, a VMHConv, an
 * ExplicitEdgeConv and a GNOConv layer, forward and pullback with ONE call each -- what the Julia shim of INTEGRATION.md binds.
 * Checker: double-precision loops written from the reference's definitions, on the CONCATENATED message inputs exactly as
 * /root/reference/src/layers.jl:106, :316-328, :409-418, :523-536 build them (so the library's split of phi's first weight into signed
 * row blocks is checked against the unsplit form), for the values; for the gradients, central differences of those loops along
 * random directions in (state, every weight, every bias) against the inner product of the library's gradients with the direction.
 * No Python, torch or C++ on the calling side.  Exit code 0 = every comparison within tolerance.  Run by tests/test_c_abi_gpu.py. */
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

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static double rnd(void) {
  rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17;
  return (double)(rng_state >> 11) / 9007199254740992.0 * 2.0 - 1.0;
}
static int fails = 0;
static void report(const char *what, double err, double tol) {
  printf("%-44s %.2e (tol %.0e)%s\n", what, err, tol, err <= tol ? "" : "   <-- FAIL");
  if (!(err <= tol)) ++fails;
}

/* ---- the model in double precision -------------------------------------------------------------------------------------- */
#define MAXL 4
typedef struct {
  int n_layers, dims[MAXL + 1], act[MAXL];
  double *w[MAXL], *b[MAXL];   /* w[l]: [dims[l]][dims[l + 1]] (= the reference's (out x in) column-major) */
} Mlp;
typedef struct {
  int kind, aggr, n, e, hw, fw, pw, ew, tw, n_graphs;   /* widths: state, node_feat, pos, edge_feat, theta */
  const int64_t *s, *t;
  double *h, *feat, *pos, *efeat, *theta;               /* h [n][hw] is the state (one block) */
  Mlp phi, upd;
} Model;

static double act_f(int a, double z) {
  switch (a) {
    case NGPDE_ACT_TANH: return tanh(z);
    case NGPDE_ACT_SWISH: return z / (1.0 + exp(-z));
    case NGPDE_ACT_RELU: return z > 0 ? z : 0;
    default: return z;
  }
}
static void mlp_apply(const Mlp *m, const double *in, double *out, double *tmp) {   /* one column; tmp: 2 * 256 doubles */
  const double *cur = in;
  for (int l = 0; l < m->n_layers; ++l) {
    double *dst = (l + 1 == m->n_layers) ? out : tmp + 256 * (l & 1);
    for (int o = 0; o < m->dims[l + 1]; ++o) {
      double z = m->b[l] ? m->b[l][o] : 0.0;
      for (int i = 0; i < m->dims[l]; ++i) z += m->w[l][(size_t)i * m->dims[l + 1] + o] * cur[i];
      dst[o] = act_f(m->act[l], z);
    }
    cur = dst;
  }
}
/* y [n][out] of the layer, from the definitions */
static void model_forward(const Model *M, double *y) {
  const int mw = M->phi.dims[M->phi.n_layers];
  double *agg = calloc((size_t)M->n * mw, sizeof(double));
  int *deg = calloc(M->n, sizeof(int));
  double in[512], out[256], tmp[512];
  const int epg = M->e / (M->n_graphs > 0 ? M->n_graphs : 1), npg = M->n / (M->n_graphs > 0 ? M->n_graphs : 1);
  for (int k = 0; k < M->e; ++k) {
    const int i = (int)M->t[k], j = (int)M->s[k];   /* xi = gather at the target, xj = at the source */
    int c = 0;
    if (M->kind == NGPDE_LAYER_MPPDE) {   /* [hi; hj; di - dj; e; theta]  (:409-410) */
      for (int f = 0; f < M->hw; ++f) in[c++] = M->h[(size_t)i * M->hw + f];
      for (int f = 0; f < M->hw; ++f) in[c++] = M->h[(size_t)j * M->hw + f];
      for (int f = 0; f < M->fw; ++f) in[c++] = M->feat[(size_t)i * M->fw + f] - M->feat[(size_t)j * M->fw + f];
      for (int f = 0; f < M->ew; ++f) in[c++] = M->efeat[(size_t)k * M->ew + f];
      for (int f = 0; f < M->tw; ++f) in[c++] = M->theta[(size_t)(k / epg) * M->tw + f];
    } else {                              /* [hi...; hj... (VMH: hj - hi); xj - xi] with the fixed node features behind the state (:106, :316) */
      for (int f = 0; f < M->hw; ++f) in[c++] = M->h[(size_t)i * M->hw + f];
      for (int f = 0; f < M->fw; ++f) in[c++] = M->feat[(size_t)i * M->fw + f];
      const int vmh = M->kind == NGPDE_LAYER_VMH;
      for (int f = 0; f < M->hw; ++f) in[c++] = M->h[(size_t)j * M->hw + f] - (vmh ? M->h[(size_t)i * M->hw + f] : 0.0);
      for (int f = 0; f < M->fw; ++f) in[c++] = M->feat[(size_t)j * M->fw + f] - (vmh ? M->feat[(size_t)i * M->fw + f] : 0.0);
      for (int f = 0; f < M->pw; ++f) in[c++] = M->pos[(size_t)j * M->pw + f] - M->pos[(size_t)i * M->pw + f];
    }
    mlp_apply(&M->phi, in, out, tmp);
    for (int f = 0; f < mw; ++f) agg[(size_t)i * mw + f] += out[f];
    ++deg[i];
  }
  for (int i = 0; i < M->n; ++i) {
    if (M->aggr == NGPDE_AGGR_MEAN)
      for (int f = 0; f < mw; ++f) agg[(size_t)i * mw + f] = deg[i] ? agg[(size_t)i * mw + f] / deg[i] : 0.0;
    if (M->upd.n_layers == 0) {
      memcpy(y + (size_t)i * mw, agg + (size_t)i * mw, sizeof(double) * mw);
      continue;
    }
    int c = 0;
    for (int f = 0; f < M->hw; ++f) in[c++] = M->h[(size_t)i * M->hw + f];          /* gamma([h; m]) (:328) / psi([h; m; theta]) (:418) */
    for (int f = 0; f < mw; ++f) in[c++] = agg[(size_t)i * mw + f];
    if (M->kind == NGPDE_LAYER_MPPDE)
      for (int f = 0; f < M->tw; ++f) in[c++] = M->theta[(size_t)(i / npg) * M->tw + f];
    mlp_apply(&M->upd, in, y + (size_t)i * M->upd.dims[M->upd.n_layers], tmp);
  }
  free(agg);
  free(deg);
}

/* ---- GNOConv (:509-547): K_e = reshape(phi([s_i; s_j; e]), out, in) column-major, m_i = mean_j K_e h_j, y = act.(W h + m + b) -------- */
typedef struct {
  int n, e, cin, cout, sw, ew, act;
  const int64_t *s, *t;
  double *h, *feat, *efeat, *W, *b;   /* W [in][out] */
  Mlp phi;
} Gno;
static void gno_forward(const Gno *M, double *y) {
  const int kw = M->cin * M->cout;
  double *agg = calloc((size_t)M->n * M->cout, sizeof(double));
  int *deg = calloc(M->n, sizeof(int));
  double in[64], *K = malloc(sizeof(double) * kw), tmp[512];
  for (int k = 0; k < M->e; ++k) {
    const int i = (int)M->t[k], j = (int)M->s[k];
    int c = 0;
    for (int f = 0; f < M->sw; ++f) in[c++] = M->feat[(size_t)i * M->sw + f];
    for (int f = 0; f < M->sw; ++f) in[c++] = M->feat[(size_t)j * M->sw + f];
    for (int f = 0; f < M->ew; ++f) in[c++] = M->efeat[(size_t)k * M->ew + f];
    mlp_apply(&M->phi, in, K, tmp);
    for (int o = 0; o < M->cout; ++o) {
      double v = 0;
      for (int q = 0; q < M->cin; ++q) v += K[o + M->cout * q] * M->h[(size_t)j * M->cin + q];   /* K_e[o, q] = phi_out[o + out * q] */
      agg[(size_t)i * M->cout + o] += v;
    }
    ++deg[i];
  }
  for (int i = 0; i < M->n; ++i)
    for (int o = 0; o < M->cout; ++o) {
      double z = (deg[i] ? agg[(size_t)i * M->cout + o] / deg[i] : 0.0) + M->b[o];
      for (int q = 0; q < M->cin; ++q) z += M->W[(size_t)q * M->cout + o] * M->h[(size_t)i * M->cin + q];
      y[(size_t)i * M->cout + o] = act_f(M->act, z);
    }
  free(agg); free(deg); free(K);
}

/* ---- plumbing -------------------------------------------------------------------------------------------------------------- */
static float *dev_from_double(const double *h, size_t n) {
  float *tmp = malloc(sizeof(float) * (n ? n : 1)), *d = NULL;
  for (size_t i = 0; i < n; ++i) tmp[i] = (float)h[i];
  if (hipMalloc((void **)&d, (n ? n : 1) * sizeof(float)) != hipSuccess) return NULL;
  if (n && hipMemcpy(d, tmp, n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) return NULL;
  free(tmp);
  return d;
}
static double *rand_d(size_t n, double scale) {
  double *p = malloc(sizeof(double) * (n ? n : 1));
  for (size_t i = 0; i < n; ++i) p[i] = (double)(float)(rnd() * scale);   /* float-representable: both sides see the same inputs */
  return p;
}
static void mlp_init(Mlp *m, int n_layers, const int *dims, const int *act) {
  m->n_layers = n_layers;
  for (int l = 0; l <= n_layers; ++l) m->dims[l] = dims[l];
  for (int l = 0; l < n_layers; ++l) {
    m->act[l] = act[l];
    m->w[l] = rand_d((size_t)dims[l] * dims[l + 1], 1.5 / sqrt((double)dims[l]));
    m->b[l] = rand_d(dims[l + 1], 0.2);
  }
}

typedef struct { double *ptr; size_t n; float *grad_dev; } Var;   /* a differentiable array, its length, the library's gradient */

static int run_case(const char *name, ngpde_graph_t *g, Model *M) {
  const int mw = M->phi.dims[M->phi.n_layers];
  const int ow = M->upd.n_layers ? M->upd.dims[M->upd.n_layers] : mw;
  const size_t ny = (size_t)M->n * ow;
  ngpde_edge_layer_t L;
  memset(&L, 0, sizeof L);
  L.kind = M->kind; L.aggr = M->aggr; L.n_state = 1;
  float *h_d = dev_from_double(M->h, (size_t)M->n * M->hw);
  L.state[0] = h_d; L.state_width[0] = M->hw;
  if (M->fw) { L.node_feat = dev_from_double(M->feat, (size_t)M->n * M->fw); L.node_feat_width = M->fw; }
  if (M->pw) { L.pos = dev_from_double(M->pos, (size_t)M->n * M->pw); L.pos_width = M->pw; }
  if (M->tw) { L.theta = dev_from_double(M->theta, (size_t)M->n_graphs * M->tw); L.theta_width = M->tw; }
  if (M->ew) {   /* edge features: COO order -> p order */
    float *coo = dev_from_double(M->efeat, (size_t)M->e * M->ew), *p = NULL;
    CHECK_HIP(hipMalloc((void **)&p, sizeof(float) * (size_t)M->e * M->ew));
    CHECK_NG(ngpde_edge_permute(g, M->ew, 0, coo, p, NULL));
    L.edge_feat = p; L.edge_feat_width = M->ew;
  }
  Var vars[1 + 4 * MAXL];
  int nv = 0;
  ngpde_mlp_grad_t gphi, gupd;
  memset(&gphi, 0, sizeof gphi);
  memset(&gupd, 0, sizeof gupd);
  float *dh_d = NULL;
  CHECK_HIP(hipMalloc((void **)&dh_d, sizeof(float) * (size_t)M->n * M->hw));
  vars[nv++] = (Var){M->h, (size_t)M->n * M->hw, dh_d};
  for (int which = 0; which < 2; ++which) {
    Mlp *m = which ? &M->upd : &M->phi;
    ngpde_mlp_t *d = which ? &L.update : &L.phi;
    ngpde_mlp_grad_t *gd = which ? &gupd : &gphi;
    d->n_layers = m->n_layers;
    for (int l = 0; l <= m->n_layers; ++l) d->dims[l] = m->dims[l];
    for (int l = 0; l < m->n_layers; ++l) {
      const size_t nw = (size_t)m->dims[l] * m->dims[l + 1];
      d->act[l] = m->act[l];
      d->weight[l] = dev_from_double(m->w[l], nw);
      d->bias[l] = dev_from_double(m->b[l], m->dims[l + 1]);
      CHECK_HIP(hipMalloc((void **)&gd->dweight[l], sizeof(float) * nw));
      CHECK_HIP(hipMalloc((void **)&gd->dbias[l], sizeof(float) * m->dims[l + 1]));
      vars[nv++] = (Var){m->w[l], nw, gd->dweight[l]};
      vars[nv++] = (Var){m->b[l], (size_t)m->dims[l + 1], gd->dbias[l]};
    }
  }
  /* ---- forward + pullback, one call each */
  const size_t wsb = ngpde_edge_layer_workspace_bytes(g, &L, 1);
  if (!wsb) { fprintf(stderr, "%s: workspace query failed: %s\n", name, ngpde_last_error()); return 4; }
  void *ws = NULL;
  float *y_d = NULL, *dy_d = NULL;
  CHECK_HIP(hipMalloc(&ws, wsb));
  CHECK_HIP(hipMalloc((void **)&y_d, sizeof(float) * ny));
  double *R = rand_d(ny, 1.0);
  dy_d = dev_from_double(R, ny);
  CHECK_NG(ngpde_edge_layer_forward(g, &L, 1, y_d, ws, wsb, NULL));
  float *dstate[4] = {dh_d, NULL, NULL, NULL};
  CHECK_NG(ngpde_edge_layer_backward(g, &L, dy_d, dstate, &gphi, &gupd, ws, wsb, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  /* ---- values */
  float *y = malloc(sizeof(float) * ny);
  CHECK_HIP(hipMemcpy(y, y_d, sizeof(float) * ny, hipMemcpyDeviceToHost));
  double *yo = malloc(sizeof(double) * ny);
  model_forward(M, yo);
  double err = 0, ref = 0;
  for (size_t i = 0; i < ny; ++i) {
    if (fabs(y[i] - yo[i]) > err) err = fabs(y[i] - yo[i]);
    if (fabs(yo[i]) > ref) ref = fabs(yo[i]);
  }
  char label[128];
  snprintf(label, sizeof label, "%s forward (%zu KB workspace)", name, wsb >> 10);
  report(label, err / ref, 1e-4);
  /* ---- gradients: d/d eps of sum(R . y) along random directions, by central differences of the double loops */
  double *yp = malloc(sizeof(double) * ny), *ym = malloc(sizeof(double) * ny);
  for (int trial = 0; trial < 4; ++trial) {   /* 0: everything; 1: the state only; 2: phi only; 3: the update only */
    if (trial == 3 && M->upd.n_layers == 0) continue;
    const int lo = trial == 0 ? 0 : (trial == 1 ? 0 : (trial == 2 ? 1 : 1 + 2 * M->phi.n_layers));
    const int hi = trial == 0 ? nv : (trial == 1 ? 1 : (trial == 2 ? 1 + 2 * M->phi.n_layers : nv));
    const double eps = 1e-5;
    double analytic = 0, scale = 0;
    double *dir[1 + 4 * MAXL];
    for (int v = lo; v < hi; ++v) {
      dir[v] = malloc(sizeof(double) * vars[v].n);
      float *gh = malloc(sizeof(float) * vars[v].n);
      CHECK_HIP(hipMemcpy(gh, vars[v].grad_dev, sizeof(float) * vars[v].n, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < vars[v].n; ++i) {
        dir[v][i] = rnd();
        analytic += (double)gh[i] * dir[v][i];
        scale += fabs((double)gh[i] * dir[v][i]);
      }
      free(gh);
    }
    for (int sgn = 1; sgn >= -1; sgn -= 2) {
      for (int v = lo; v < hi; ++v)
        for (size_t i = 0; i < vars[v].n; ++i) vars[v].ptr[i] += sgn * eps * dir[v][i];
      model_forward(M, sgn > 0 ? yp : ym);
      for (int v = lo; v < hi; ++v)
        for (size_t i = 0; i < vars[v].n; ++i) vars[v].ptr[i] -= sgn * eps * dir[v][i];
    }
    double fd = 0;
    for (size_t i = 0; i < ny; ++i) fd += R[i] * (yp[i] - ym[i]) / (2 * eps);
    for (int v = lo; v < hi; ++v) free(dir[v]);
    static const char *what[4] = {"all gradients", "d state", "d phi", "d update"};
    snprintf(label, sizeof label, "%s pullback, %s", name, what[trial]);
    /* relative to the sum of the magnitudes of the terms (the inner product may cancel) */
    report(label, fabs(analytic - fd) / (scale > 1e-30 ? scale : 1e-30), 2e-4);
  }
  free(y); free(yo); free(yp); free(ym); free(R);
  CHECK_HIP(hipFree(ws));
  return 0;
}

static int run_gno(ngpde_graph_t *g, Gno *M) {
  const size_t ny = (size_t)M->n * M->cout;
  ngpde_gno_layer_t L;
  memset(&L, 0, sizeof L);
  L.in_chs = M->cin; L.out_chs = M->cout; L.aggr = NGPDE_AGGR_MEAN; L.act = M->act;
  L.h = dev_from_double(M->h, (size_t)M->n * M->cin);
  L.node_feat = dev_from_double(M->feat, (size_t)M->n * M->sw); L.node_feat_width = M->sw;
  {
    float *coo = dev_from_double(M->efeat, (size_t)M->e * M->ew), *p = NULL;
    CHECK_HIP(hipMalloc((void **)&p, sizeof(float) * (size_t)M->e * M->ew));
    CHECK_NG(ngpde_edge_permute(g, M->ew, 0, coo, p, NULL));
    L.edge_feat = p; L.edge_feat_width = M->ew;
  }
  L.weight = dev_from_double(M->W, (size_t)M->cin * M->cout);
  L.bias = dev_from_double(M->b, M->cout);
  Var vars[3 + 2 * MAXL];
  int nv = 0;
  float *dh_d = NULL, *dW_d = NULL, *db_d = NULL;
  CHECK_HIP(hipMalloc((void **)&dh_d, sizeof(float) * (size_t)M->n * M->cin));
  CHECK_HIP(hipMalloc((void **)&dW_d, sizeof(float) * (size_t)M->cin * M->cout));
  CHECK_HIP(hipMalloc((void **)&db_d, sizeof(float) * M->cout));
  vars[nv++] = (Var){M->h, (size_t)M->n * M->cin, dh_d};
  vars[nv++] = (Var){M->W, (size_t)M->cin * M->cout, dW_d};
  vars[nv++] = (Var){M->b, (size_t)M->cout, db_d};
  ngpde_mlp_grad_t gphi;
  memset(&gphi, 0, sizeof gphi);
  L.phi.n_layers = M->phi.n_layers;
  for (int l = 0; l <= M->phi.n_layers; ++l) L.phi.dims[l] = M->phi.dims[l];
  for (int l = 0; l < M->phi.n_layers; ++l) {
    const size_t nw = (size_t)M->phi.dims[l] * M->phi.dims[l + 1];
    L.phi.act[l] = M->phi.act[l];
    L.phi.weight[l] = dev_from_double(M->phi.w[l], nw);
    L.phi.bias[l] = dev_from_double(M->phi.b[l], M->phi.dims[l + 1]);
    CHECK_HIP(hipMalloc((void **)&gphi.dweight[l], sizeof(float) * nw));
    CHECK_HIP(hipMalloc((void **)&gphi.dbias[l], sizeof(float) * M->phi.dims[l + 1]));
    vars[nv++] = (Var){M->phi.w[l], nw, gphi.dweight[l]};
    vars[nv++] = (Var){M->phi.b[l], (size_t)M->phi.dims[l + 1], gphi.dbias[l]};
  }
  const size_t wsb = ngpde_gno_layer_workspace_bytes(g, &L, 1);
  if (!wsb) { fprintf(stderr, "GNOConv: workspace query failed: %s\n", ngpde_last_error()); return 4; }
  void *ws = NULL;
  float *y_d = NULL;
  CHECK_HIP(hipMalloc(&ws, wsb));
  CHECK_HIP(hipMalloc((void **)&y_d, sizeof(float) * ny));
  double *R = rand_d(ny, 1.0);
  float *dy_d = dev_from_double(R, ny);
  CHECK_NG(ngpde_gno_layer_forward(g, &L, 1, y_d, ws, wsb, NULL));
  CHECK_NG(ngpde_gno_layer_backward(g, &L, dy_d, dh_d, &gphi, dW_d, db_d, ws, wsb, NULL));
  CHECK_HIP(hipDeviceSynchronize());
  float *y = malloc(sizeof(float) * ny);
  CHECK_HIP(hipMemcpy(y, y_d, sizeof(float) * ny, hipMemcpyDeviceToHost));
  double *yo = malloc(sizeof(double) * ny), *yp = malloc(sizeof(double) * ny), *ym = malloc(sizeof(double) * ny);
  gno_forward(M, yo);
  double err = 0, ref = 0;
  for (size_t i = 0; i < ny; ++i) {
    if (fabs(y[i] - yo[i]) > err) err = fabs(y[i] - yo[i]);
    if (fabs(yo[i]) > ref) ref = fabs(yo[i]);
  }
  char label[128];
  snprintf(label, sizeof label, "GNOConv forward (%zu KB workspace)", wsb >> 10);
  report(label, err / ref, 1e-4);
  for (int trial = 0; trial < 3; ++trial) {   /* 0: everything; 1: h only; 2: phi only */
    const int lo = trial == 2 ? 3 : 0, hi = trial == 1 ? 1 : nv;
    const double eps = 1e-5;
    double analytic = 0, scale = 0;
    double *dir[3 + 2 * MAXL];
    for (int v = lo; v < hi; ++v) {
      dir[v] = malloc(sizeof(double) * vars[v].n);
      float *gh = malloc(sizeof(float) * vars[v].n);
      CHECK_HIP(hipMemcpy(gh, vars[v].grad_dev, sizeof(float) * vars[v].n, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < vars[v].n; ++i) {
        dir[v][i] = rnd();
        analytic += (double)gh[i] * dir[v][i];
        scale += fabs((double)gh[i] * dir[v][i]);
      }
      free(gh);
    }
    for (int sgn = 1; sgn >= -1; sgn -= 2) {
      for (int v = lo; v < hi; ++v)
        for (size_t i = 0; i < vars[v].n; ++i) vars[v].ptr[i] += sgn * eps * dir[v][i];
      gno_forward(M, sgn > 0 ? yp : ym);
      for (int v = lo; v < hi; ++v)
        for (size_t i = 0; i < vars[v].n; ++i) vars[v].ptr[i] -= sgn * eps * dir[v][i];
    }
    double fd = 0;
    for (size_t i = 0; i < ny; ++i) fd += R[i] * (yp[i] - ym[i]) / (2 * eps);
    for (int v = lo; v < hi; ++v) free(dir[v]);
    static const char *what[3] = {"all gradients", "d h", "d phi"};
    snprintf(label, sizeof label, "GNOConv pullback, %s", what[trial]);
    report(label, fabs(analytic - fd) / (scale > 1e-30 ? scale : 1e-30), 2e-4);
  }
  free(y); free(yo); free(yp); free(ym); free(R);
  CHECK_HIP(hipFree(ws));
  return 0;
}

int main(void) {
  printf("%s\n", ngpde_version());
  /* ---- MPPDEConv on a batch of 3 periodic meshes of 96 nodes (reach 2), one node feature, two edge features, theta of width 2 */
  {
    const int traj = 3, n1 = 96, reach = 2, n = traj * n1, e = traj * n1 * 2 * reach;
    int64_t *s = malloc(sizeof(int64_t) * e), *t = malloc(sizeof(int64_t) * e);
    int k = 0;
    for (int gph = 0; gph < traj; ++gph)
      for (int off = -reach; off <= reach; ++off) {
        if (!off) continue;
        for (int i = 0; i < n1; ++i) { s[k] = gph * n1 + i; t[k] = gph * n1 + (i + off + n1) % n1; ++k; }
      }
    ngpde_graph_t *g = NULL;
    CHECK_NG(ngpde_graph_create(n, e, s, t, 0, traj, &g));
    Model M;
    memset(&M, 0, sizeof M);
    M.kind = NGPDE_LAYER_MPPDE; M.aggr = NGPDE_AGGR_MEAN; M.n = n; M.e = e; M.s = s; M.t = t; M.n_graphs = traj;
    M.hw = 16; M.fw = 2; M.ew = 2; M.tw = 2;
    M.h = rand_d((size_t)n * M.hw, 1.0); M.feat = rand_d((size_t)n * M.fw, 1.0); M.efeat = rand_d((size_t)e * M.ew, 1.0);
    M.theta = rand_d((size_t)traj * M.tw, 1.0);
    const int pd[3] = {2 * 16 + 2 + 2 + 2, 24, 12}, pa[2] = {NGPDE_ACT_SWISH, NGPDE_ACT_SWISH};
    const int ud[3] = {16 + 12 + 2, 20, 16}, ua[2] = {NGPDE_ACT_SWISH, NGPDE_ACT_IDENTITY};
    mlp_init(&M.phi, 2, pd, pa);
    mlp_init(&M.upd, 2, ud, ua);
    int rc = run_case("MPPDEConv", g, &M);
    if (rc) return rc;
    CHECK_NG(ngpde_graph_destroy(g));
  }
  /* ---- VMHConv and ExplicitEdgeConv on a ring with chords, 2-d positions, one fixed node feature */
  for (int kind = NGPDE_LAYER_VMH; kind >= NGPDE_LAYER_EDGECONV; --kind) {
    const int n = 200, e = 5 * n;
    int64_t *s = malloc(sizeof(int64_t) * e), *t = malloc(sizeof(int64_t) * e);
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const int a = (i + 1) % n, b = (i + 7) % n, c = (i + 31) % n;
      s[m] = i; t[m++] = a; s[m] = a; t[m++] = i;
      s[m] = i; t[m++] = b; s[m] = b; t[m++] = i;
      s[m] = i; t[m++] = c;
    }
    ngpde_graph_t *g = NULL;
    CHECK_NG(ngpde_graph_create(n, e, s, t, 0, 1, &g));
    Model M;
    memset(&M, 0, sizeof M);
    M.kind = kind; M.aggr = kind == NGPDE_LAYER_VMH ? NGPDE_AGGR_MEAN : NGPDE_AGGR_SUM; M.n = n; M.e = e; M.s = s; M.t = t; M.n_graphs = 1;
    M.hw = 3; M.fw = 1; M.pw = 2;
    M.h = rand_d((size_t)n * M.hw, 1.0); M.feat = rand_d((size_t)n * M.fw, 1.0); M.pos = rand_d((size_t)n * M.pw, 1.0);
    const int pd[4] = {2 * (3 + 1) + 2, 20, 20, 8}, pa[3] = {NGPDE_ACT_TANH, NGPDE_ACT_TANH, NGPDE_ACT_IDENTITY};
    mlp_init(&M.phi, 3, pd, pa);
    if (kind == NGPDE_LAYER_VMH) {
      const int ud[4] = {3 + 8, 16, 16, 3}, ua[3] = {NGPDE_ACT_TANH, NGPDE_ACT_TANH, NGPDE_ACT_IDENTITY};
      mlp_init(&M.upd, 3, ud, ua);
    }
    int rc = run_case(kind == NGPDE_LAYER_VMH ? "VMHConv" : "ExplicitEdgeConv", g, &M);
    if (rc) return rc;
    CHECK_NG(ngpde_graph_destroy(g));
  }
  /* ---- GNOConv 8 => 16 on the same ring: 2-d node coordinates, one edge feature, phi = Dense(5 => 16, relu) -> Dense(16 => 128) */
  {
    const int n = 200, e = 5 * n;
    int64_t *s = malloc(sizeof(int64_t) * e), *t = malloc(sizeof(int64_t) * e);
    int m = 0;
    for (int i = 0; i < n; ++i) {
      const int a = (i + 1) % n, b = (i + 7) % n, c = (i + 31) % n;
      s[m] = i; t[m++] = a; s[m] = a; t[m++] = i;
      s[m] = i; t[m++] = b; s[m] = b; t[m++] = i;
      s[m] = i; t[m++] = c;
    }
    ngpde_graph_t *g = NULL;
    CHECK_NG(ngpde_graph_create(n, e, s, t, 0, 1, &g));
    Gno M;
    memset(&M, 0, sizeof M);
    M.n = n; M.e = e; M.s = s; M.t = t; M.cin = 8; M.cout = 16; M.sw = 2; M.ew = 1; M.act = NGPDE_ACT_TANH;
    M.h = rand_d((size_t)n * M.cin, 1.0); M.feat = rand_d((size_t)n * M.sw, 1.0); M.efeat = rand_d((size_t)e * M.ew, 1.0);
    M.W = rand_d((size_t)M.cin * M.cout, 0.4); M.b = rand_d(M.cout, 0.2);
    const int pd[3] = {2 * 2 + 1, 16, 8 * 16}, pa[2] = {NGPDE_ACT_RELU, NGPDE_ACT_IDENTITY};
    mlp_init(&M.phi, 2, pd, pa);
    for (size_t i = 0; i < (size_t)pd[1] * pd[2]; ++i) M.phi.w[1][i] *= 0.3;
    int rc = run_gno(g, &M);
    if (rc) return rc;
    CHECK_NG(ngpde_graph_destroy(g));
  }
  printf(fails ? "%d comparison(s) FAILED\n" : "all comparisons within tolerance\n", fails);
  return fails ? 1 : 0;
}
