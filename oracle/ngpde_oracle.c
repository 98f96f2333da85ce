/*
 * oracle/ngpde_oracle.c -- plain-C restatement of the GCNConv hot path of NeuralGraphPDE.jl and of
 * the fixed-step neural graph ODE around it, in the REFERENCE-FAITHFUL form the Julia package
 * executes on CPU.  TEST INFRASTRUCTURE ONLY: used by tests/ (cross-check of the numpy oracle) and
 * by bench.py's `cpu_baseline` leg (kind "port").  The product never links or loads this file.
 *
 * PARITY UNPINNED for GCNConv values and gradients: the reference's tests assert shapes only
 * (/root/reference/test/runtests.jl:16-25); this file is cross-checked against the independent numpy
 * restatement (oracle/ngpde_oracle.py, itself finite-difference checked) in tests/test_oracle_c.py.
 *
 * What "reference-faithful" means here -- per layer call, exactly as src/layers.jl:200-239 does:
 *   :211  add_self_loops: re-allocate s,t with 1:N appended              (every call)
 *   :224  degree: scatter(+) of ones over t                               (every call)
 *   :225-226  c = 1/sqrt(d); x .* c'                                      (temporary)
 *   :232  propagate(copy_xj, g, +): on CPU GraphNeuralNetworks.jl takes the fused path
 *         x * adjacency_matrix(g): builds a SparseMatrixCSC from the COO list (counting sort,
 *         every call), then a serial dense x sparse product
 *   :234  x .* c'                                                         (temporary)
 *   :236  W * x (BLAS gemm; here a register-blocked loop nest, -O3 -march=native)
 *   :238  act.(x .+ b)
 * The pullback is what Zygote derives: dy .* act'(z), dW = dz * x3', W' * dz, and the transposed
 * sparse product.  All arithmetic in float (T = Float32 in the reference's tests and tutorials).
 * Features are (D x N) column-major == row-major [N][D].
 *
 * Threads: single-threaded unless built with -fopenmp (then the row loops are parallel).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

enum { ACT_IDENTITY = 0, ACT_RELU = 1, ACT_TANH = 2, ACT_SIGMOID = 3, ACT_SWISH = 4 };

/* bench.py's CPU-baseline variant (i): NNlib 0.8's gather / scatter and the sparse product are serial loops while BLAS gemm is
 * threaded -- with this flag set the OpenMP build keeps the sparse products on one thread and threads only the dense products */
static int g_serial_sparse = 0;
void ngo_set_serial_sparse(int on) { g_serial_sparse = on; }

static float actf(int a, float z) {
  switch (a) {
    case ACT_RELU: return z > 0.f ? z : 0.f;
    case ACT_TANH: return tanhf(z);
    case ACT_SIGMOID: return 1.f / (1.f + expf(-z));
    case ACT_SWISH: return z / (1.f + expf(-z));
    default: return z;
  }
}

static float dactf(int a, float z) {
  switch (a) {
    case ACT_RELU: return z > 0.f ? 1.f : 0.f;
    case ACT_TANH: { float t = tanhf(z); return 1.f - t * t; }
    case ACT_SIGMOID: { float s = 1.f / (1.f + expf(-z)); return s * (1.f - s); }
    case ACT_SWISH: { float s = 1.f / (1.f + expf(-z)); return s * (1.f + z * (1.f - s)); }
    default: return 1.f;
  }
}

/* y[n][o] = sum_i x[n][i] * wt[i][o]   (wt = Julia (out x in) column-major) */
static void gemm_nn(int64_t n, int din, int dout, const float *x, const float *wt, float *y) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    float *yr = y + r * dout;
    for (int o = 0; o < dout; ++o) yr[o] = 0.f;
    for (int i = 0; i < din; ++i) {
      const float xv = x[r * din + i];
      const float *w = wt + (size_t)i * dout;
      for (int o = 0; o < dout; ++o) yr[o] += xv * w[o];
    }
  }
}

/* dx[n][i] = sum_o dz[n][o] * wt[i][o] */
static void gemm_nt(int64_t n, int din, int dout, const float *dz, const float *wt, float *dx) {
#pragma omp parallel for schedule(static)
  for (int64_t r = 0; r < n; ++r) {
    for (int i = 0; i < din; ++i) {
      const float *w = wt + (size_t)i * dout;
      float acc = 0.f;
      for (int o = 0; o < dout; ++o) acc += dz[r * dout + o] * w[o];
      dx[r * din + i] = acc;
    }
  }
}

/* dwt[i][o] += sum_n x[n][i] * dz[n][o] */
static void gemm_tn_acc(int64_t n, int din, int dout, const float *x, const float *dz, float *dwt) {
  for (int64_t r = 0; r < n; ++r)
    for (int i = 0; i < din; ++i) {
      const float xv = x[r * din + i];
      float *w = dwt + (size_t)i * dout;
      for (int o = 0; o < dout; ++o) w[o] += xv * dz[r * dout + o];
    }
}

typedef struct {
  int64_t n, m;        /* nodes, edges incl. self loops */
  int64_t *colptr;     /* CSC of A[s,t]: column t lists sources s */
  int64_t *rowval;
  int64_t *rowptr_s;   /* CSR (by source) for the transposed product */
  int64_t *colval_s;
  float *c;            /* 1/sqrt(in-degree) */
} sparse_t;

/* the per-call preprocessing of src/layers.jl:211,224 + adjacency_matrix(g) */
static void build_sparse(int64_t n, int64_t e, const int64_t *s, const int64_t *t, int self_loops, int need_t,
                         sparse_t *A) {
  const int64_t m = e + (self_loops ? n : 0);
  int64_t *ss = (int64_t *)malloc(sizeof(int64_t) * (m ? m : 1));
  int64_t *tt = (int64_t *)malloc(sizeof(int64_t) * (m ? m : 1));
  memcpy(ss, s, sizeof(int64_t) * e);
  memcpy(tt, t, sizeof(int64_t) * e);
  if (self_loops)
    for (int64_t i = 0; i < n; ++i) ss[e + i] = tt[e + i] = i;
  A->n = n; A->m = m;
  float *deg = (float *)calloc(n ? n : 1, sizeof(float));
  for (int64_t k = 0; k < m; ++k) deg[tt[k]] += 1.f;                     /* degree(g; dir=:in) */
  A->c = (float *)malloc(sizeof(float) * (n ? n : 1));
  for (int64_t i = 0; i < n; ++i) A->c[i] = 1.f / sqrtf(deg[i]);
  free(deg);
  A->colptr = (int64_t *)calloc(n + 1, sizeof(int64_t));
  A->rowval = (int64_t *)malloc(sizeof(int64_t) * (m ? m : 1));
  for (int64_t k = 0; k < m; ++k) A->colptr[tt[k] + 1]++;
  for (int64_t i = 0; i < n; ++i) A->colptr[i + 1] += A->colptr[i];
  int64_t *cur = (int64_t *)malloc(sizeof(int64_t) * (n ? n : 1));
  memcpy(cur, A->colptr, sizeof(int64_t) * n);
  for (int64_t k = 0; k < m; ++k) A->rowval[cur[tt[k]]++] = ss[k];
  A->rowptr_s = NULL; A->colval_s = NULL;
  if (need_t) {
    A->rowptr_s = (int64_t *)calloc(n + 1, sizeof(int64_t));
    A->colval_s = (int64_t *)malloc(sizeof(int64_t) * (m ? m : 1));
    for (int64_t k = 0; k < m; ++k) A->rowptr_s[ss[k] + 1]++;
    for (int64_t i = 0; i < n; ++i) A->rowptr_s[i + 1] += A->rowptr_s[i];
    memcpy(cur, A->rowptr_s, sizeof(int64_t) * n);
    for (int64_t k = 0; k < m; ++k) A->colval_s[cur[ss[k]]++] = tt[k];
  }
  free(cur); free(ss); free(tt);
}

static void free_sparse(sparse_t *A) {
  free(A->colptr); free(A->rowval); free(A->c);
  if (A->rowptr_s) free(A->rowptr_s);
  if (A->colval_s) free(A->colval_s);
}

/* out = ((x .* c') * A) .* c'   with A given by (ptr, idx): out[:, j] = c_j * sum_{k in col j} c_k x[:, k] */
static void norm_spmm(int64_t n, int d, const int64_t *ptr, const int64_t *idx, const float *c, const float *x,
                      float *out) {
  float *x1 = (float *)malloc(sizeof(float) * (size_t)(n ? n : 1) * d);       /* x .* c' temporary (:226) */
  for (int64_t i = 0; i < n; ++i)
    for (int f = 0; f < d; ++f) x1[i * d + f] = x[i * d + f] * c[i];
#pragma omp parallel for schedule(static) if (!g_serial_sparse)
  for (int64_t j = 0; j < n; ++j) {
    float *o = out + j * d;
    for (int f = 0; f < d; ++f) o[f] = 0.f;
    for (int64_t p = ptr[j]; p < ptr[j + 1]; ++p) {
      const float *xr = x1 + idx[p] * d;
      for (int f = 0; f < d; ++f) o[f] += xr[f];
    }
    for (int f = 0; f < d; ++f) o[f] *= c[j];                               /* (:234) */
  }
  free(x1);
}

/* (l::GCNConv)(x, ps, st), dout >= din branch (all BASELINE configs have din == dout) */
void ngo_gcn_forward(int64_t n, int64_t e, const int64_t *s, const int64_t *t, int self_loops, int din, int dout,
                     int act, const float *x, const float *wt, const float *bias, float *y, float *x3_out,
                     float *z_out) {
  sparse_t A;
  build_sparse(n, e, s, t, self_loops, 0, &A);
  float *x3 = x3_out ? x3_out : (float *)malloc(sizeof(float) * (size_t)(n ? n : 1) * din);
  norm_spmm(n, din, A.colptr, A.rowval, A.c, x, x3);
  gemm_nn(n, din, dout, x3, wt, y);
  for (int64_t r = 0; r < n; ++r)
    for (int o = 0; o < dout; ++o) {
      const float z = y[r * dout + o] + (bias ? bias[o] : 0.f);
      if (z_out) z_out[r * dout + o] = z;
      y[r * dout + o] = actf(act, z);
    }
  if (!x3_out) free(x3);
  free_sparse(&A);
}

/* pullback; dwt / db are ACCUMULATED (+=) so a solver can sum over stage evaluations */
void ngo_gcn_backward(int64_t n, int64_t e, const int64_t *s, const int64_t *t, int self_loops, int din, int dout,
                      int act, const float *wt, const float *z, const float *x3, const float *dy, float *dx,
                      float *dwt, float *db) {
  sparse_t A;
  build_sparse(n, e, s, t, self_loops, 1, &A);
  float *dz = (float *)malloc(sizeof(float) * (size_t)(n ? n : 1) * dout);
  for (int64_t k = 0; k < n * dout; ++k) dz[k] = dy[k] * dactf(act, z[k]);
  if (db)
    for (int64_t r = 0; r < n; ++r)
      for (int o = 0; o < dout; ++o) db[o] += dz[r * dout + o];
  gemm_tn_acc(n, din, dout, x3, dz, dwt);
  float *dx3 = (float *)malloc(sizeof(float) * (size_t)(n ? n : 1) * din);
  gemm_nt(n, din, dout, dz, wt, dx3);
  norm_spmm(n, din, A.rowptr_s, A.colval_s, A.c, dx3, dx);
  free(dz); free(dx3);
  free_sparse(&A);
}

/* ---- fixed-step explicit RK over Chain(GCNConv(d=>d,act), GCNConv(d=>d,act)), loss = sum(u(T)) ---- */

static const double TS_A[6][5] = {
    {0, 0, 0, 0, 0},
    {0.161, 0, 0, 0, 0},
    {-0.008480655492356989, 0.335480655492357, 0, 0, 0},
    {2.8971530571054935, -6.359448489975075, 4.3622954328695815, 0, 0},
    {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525, 0},
    {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383}};
static const double TS_B[6] = {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742,
                               -3.290069515436081, 2.324710524099774};

/* tableau: 0 = Euler, 1 = Tsit5.  with_grad != 0 also runs the discrete adjoint of loss = sum(uT).
 * Returns 0 on success.  Gradient outputs may be NULL when with_grad == 0. */
int ngo_node_gcn2(int64_t n, int64_t e, const int64_t *s, const int64_t *t, int d, int act, int tableau, int nsteps,
                  float dt, int with_grad, const float *u0, const float *w1, const float *b1, const float *w2,
                  const float *b2, float *uT, float *du0, float *dw1, float *db1, float *dw2, float *db2) {
  const int S = tableau == 0 ? 1 : 6;
  double a[6][5], b[6];
  memset(a, 0, sizeof a);
  if (tableau == 0) { b[0] = 1.0; } else { memcpy(a, TS_A, sizeof a); memcpy(b, TS_B, sizeof b); }
  const size_t ne = (size_t)n * d;
  /* tape: per (step, stage): x3_1, z1, y1, x3_2, z2, k */
  const size_t per = 6 * ne;
  float *tape = (float *)malloc(sizeof(float) * per * (size_t)(with_grad ? nsteps : 1) * S);
  float *u = (float *)malloc(sizeof(float) * ne), *U = (float *)malloc(sizeof(float) * ne);
  if (!tape || !u || !U) return -1;
  memcpy(u, u0, sizeof(float) * ne);
  for (int st = 0; st < nsteps; ++st) {
    float *base = tape + (size_t)(with_grad ? st : 0) * S * per;
    for (int i = 0; i < S; ++i) {
      float *T = base + (size_t)i * per;
      memcpy(U, u, sizeof(float) * ne);
      for (int j = 0; j < i; ++j) {
        const float cf = (float)(dt * a[i][j]);
        if (cf == 0.f) continue;
        const float *kj = base + (size_t)j * per + 5 * ne;
        for (size_t q = 0; q < ne; ++q) U[q] += cf * kj[q];
      }
      ngo_gcn_forward(n, e, s, t, 1, d, d, act, U, w1, b1, T + 2 * ne, T + 0 * ne, T + 1 * ne);
      ngo_gcn_forward(n, e, s, t, 1, d, d, act, T + 2 * ne, w2, b2, T + 5 * ne, T + 3 * ne, T + 4 * ne);
    }
    for (int i = 0; i < S; ++i) {
      const float cf = (float)(dt * b[i]);
      const float *ki = base + (size_t)i * per + 5 * ne;
      for (size_t q = 0; q < ne; ++q) u[q] += cf * ki[q];
    }
  }
  memcpy(uT, u, sizeof(float) * ne);
  if (with_grad) {
    float *lam = (float *)malloc(sizeof(float) * ne), *kbar = (float *)malloc(sizeof(float) * ne);
    float *dy1 = (float *)malloc(sizeof(float) * ne);
    float *ubar[6];
    for (int i = 0; i < S; ++i) ubar[i] = (float *)malloc(sizeof(float) * ne);
    for (size_t q = 0; q < ne; ++q) lam[q] = 1.f;
    memset(dw1, 0, sizeof(float) * d * d); memset(dw2, 0, sizeof(float) * d * d);
    memset(db1, 0, sizeof(float) * d); memset(db2, 0, sizeof(float) * d);
    for (int st = nsteps - 1; st >= 0; --st) {
      float *base = tape + (size_t)st * S * per;
      for (int i = S - 1; i >= 0; --i) {
        float *T = base + (size_t)i * per;
        const float cb = (float)(dt * b[i]);
        for (size_t q = 0; q < ne; ++q) kbar[q] = cb * lam[q];
        for (int j = i + 1; j < S; ++j) {
          const float cf = (float)(dt * a[j][i]);
          if (cf == 0.f) continue;
          for (size_t q = 0; q < ne; ++q) kbar[q] += cf * ubar[j][q];
        }
        ngo_gcn_backward(n, e, s, t, 1, d, d, act, w2, T + 4 * ne, T + 3 * ne, kbar, dy1, dw2, db2);
        ngo_gcn_backward(n, e, s, t, 1, d, d, act, w1, T + 1 * ne, T + 0 * ne, dy1, ubar[i], dw1, db1);
      }
      for (int i = 0; i < S; ++i)
        for (size_t q = 0; q < ne; ++q) lam[q] += ubar[i][q];
    }
    memcpy(du0, lam, sizeof(float) * ne);
    for (int i = 0; i < S; ++i) free(ubar[i]);
    free(lam); free(kbar); free(dy1);
  }
  free(tape); free(u); free(U);
  return 0;
}
