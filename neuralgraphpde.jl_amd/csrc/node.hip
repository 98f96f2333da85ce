// node.hip -- device-resident fixed-step neural graph ODE over Chain(GCNConv(d=>d), GCNConv(d=>d)),
// forward solve + discrete adjoint, each replayed from one HIP graph.
//
// Caller of the hot path in the reference: the tutorial's NeuralODE wrapper evaluates
//   dudt(u, p, t) = Chain(GCNConv(nhidden => nhidden, relu), GCNConv(nhidden => nhidden, relu))(u, p, st)
// once per Runge-Kutta stage (/root/reference/docs/src/tutorials/graph_node.md:59-66, :78) and its
// pullback once per stage of the reverse pass (:54, :127).  Here every stage is two fused launches:
//   forward  : [aggregate + MFMA + act] for layer 1, [aggregate + MFMA + act + next-stage input
//              u_n + dt*sum_j a_ij k_j (or the step update with b_j)] for layer 2;
//   backward : [A^T-aggregate + mask + MFMA(dZ W, X^T dZ)] for layer 1, and one launch that finishes
//              stage i (A^T-aggregate -> U-bar_i), forms the next stage's K-bar = dt*b*lambda +
//              dt*sum_j a_ji U-bar_j (or the lambda update at the step boundary) and runs layer 2's
//              dense backward for that next stage.
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"

using namespace ngpde;

namespace {

struct Tableau {
  int S;
  std::vector<std::vector<double>> a;  // a[i][j], j < i
  std::vector<double> b;
};

Tableau make_tableau(int which) {
  Tableau t;
  if (which == NGPDE_TABLEAU_EULER) {
    t.S = 1;
    t.a = {{}};
    t.b = {1.0};
    return t;
  }
  // Tsitouras 5(4) as used by OrdinaryDiffEq.Tsit5 (graph_node.md:48); fixed step: only the 5th-order
  // weights are needed, and they equal the 7th stage row (FSAL), so this is a 6-stage explicit scheme.
  t.S = 6;
  t.a = {{},
         {0.161},
         {-0.008480655492356989, 0.335480655492357},
         {2.8971530571054935, -6.359448489975075, 4.3622954328695815},
         {5.325864828439257, -11.748883564062828, 7.4955393428898365, -0.09249506636175525},
         {5.86145544294642, -12.92096931784711, 8.159367898576159, -0.071584973281401, -0.028269050394068383}};
  t.b = {0.09646076681806523, 0.01, 0.4798896504144996, 1.379008574103742, -3.290069515436081, 2.324710524099774};
  return t;
}

// dst[i][:] = src[i][:] * c[i]   (invert: / c[i]) -- entry to / exit from the pre-scaled form of the pipeline
// What rides along with an entry / exit kernel of a plan (one launch instead of three): a raw copy of the source rows (the plan keeps u0
// for ngpde_node_profile) and the fault latch of the persistent launch that ran just before (abort word of that launch -> the plan's
// sticky fault word; node_persistent.hip launched a kernel of its own for it until round 5)
struct RowsExtra {
  float4 *keep = nullptr;
  const unsigned *abort_word = nullptr;
  unsigned *fault = nullptr;
};
__global__ void scale_rows_kernel(size_t n4, int lpr, int invert, const float4 *__restrict__ src, const float *__restrict__ c,
                                  size_t n_nodes, float4 *__restrict__ dst, RowsExtra x) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && x.abort_word && *x.abort_word != 0) *x.fault = 1u;
  if (i >= n4) return;
  const float ci = c[(i / lpr) % n_nodes];   // (a batch of identical graphs repeats the member's coefficients)
  const float f = invert ? 1.0f / ci : ci;
  const float4 v = src[i];
  if (x.keep) x.keep[i] = v;
  dst[i] = make_float4(v.x * f, v.y * f, v.z * f, v.w * f);
}

int32_t launch_scale_rows(const float *src, const float *c, float *dst, int64_t n, int d, bool invert, hipStream_t stream,
                          int members = 1, RowsExtra x = RowsExtra()) {
  const size_t n4 = (size_t)n * members * d / 4;
  if (n4 == 0) return NGPDE_OK;
  hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, n4, d / 4, invert ? 1 : 0,
                     reinterpret_cast<const float4 *>(src), c, (size_t)n, reinterpret_cast<float4 *>(dst), x);
  NGPDE_LAUNCH_CHECK("scale_rows_kernel");
  return NGPDE_OK;
}

// The same with a change of row width: dst rows are `d_out` wide, src rows `d_in` (columns beyond the narrower of the two are
// written as zeros / dropped) -- entry to / exit from a plan that runs a narrow state on the 64-wide persistent kernels.
__global__ void scale_rows_width_kernel(size_t n4, int lpr_in, int lpr_out, int invert, const float4 *__restrict__ src,
                                        const float *__restrict__ c, size_t n_nodes, float4 *__restrict__ dst, RowsExtra x) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;   // one float4 of dst
  if (i == 0 && x.abort_word && *x.abort_word != 0) *x.fault = 1u;
  if (i >= n4) return;
  const size_t row = i / lpr_out;
  const int q = (int)(i - row * lpr_out);
  float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
  if (q < lpr_in) {
    const float ci = c[row % n_nodes];
    const float f = invert ? 1.0f / ci : ci;
    const float4 s = src[row * lpr_in + q];
    if (x.keep) x.keep[row * lpr_in + q] = s;   // (keep has the SOURCE's row width: asked for on the way in, lpr_in <= lpr_out)
    v = make_float4(s.x * f, s.y * f, s.z * f, s.w * f);
  }
  dst[i] = v;
}

int32_t launch_scale_rows_width(const float *src, int d_in, const float *c, float *dst, int d_out, int64_t n, bool invert,
                                hipStream_t stream, int members = 1, RowsExtra x = RowsExtra()) {
  if (d_in == d_out) return launch_scale_rows(src, c, dst, n, d_in, invert, stream, members, x);
  const size_t n4 = (size_t)n * members * d_out / 4;
  if (n4 == 0) return NGPDE_OK;
  hipLaunchKernelGGL(scale_rows_width_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, n4, d_in / 4, d_out / 4,
                     invert ? 1 : 0, reinterpret_cast<const float4 *>(src), c, (size_t)n, reinterpret_cast<float4 *>(dst), x);
  NGPDE_LAUNCH_CHECK("scale_rows_width_kernel");
  return NGPDE_OK;
}

// The caller's parameters -> the plan's [d][d] / [d] copies in ONE launch (they were four copies, or two 2-D copies and two copies for a
// widened plan: the du x du block of the zero-padded d x d matrix, whose padding is written once at creation); a NULL bias is zeros
__global__ void pack_params_kernel(const float *__restrict__ w1u, const float *__restrict__ b1u, const float *__restrict__ w2u,
                                   const float *__restrict__ b2u, int du, int d, float *__restrict__ w1, float *__restrict__ b1,
                                   float *__restrict__ w2, float *__restrict__ b2) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, mm = du * du;
  if (i < mm) w1[(i / du) * d + i % du] = w1u[i];
  else if (i < 2 * mm) w2[((i - mm) / du) * d + (i - mm) % du] = w2u[i - mm];
  else if (i < 2 * mm + du) b1[i - 2 * mm] = b1u ? b1u[i - 2 * mm] : 0.f;
  else if (i < 2 * mm + 2 * du) b2[i - 2 * mm - du] = b2u ? b2u[i - 2 * mm - du] : 0.f;
}
int32_t launch_pack_params(const float *w1u, const float *b1u, const float *w2u, const float *b2u, int du, int d, float *w1, float *b1,
                           float *w2, float *b2, hipStream_t stream) {
  const int total = 2 * du * du + 2 * du;
  hipLaunchKernelGGL(pack_params_kernel, dim3((total + 255) / 256), dim3(256), 0, stream, w1u, b1u, w2u, b2u, du, d, w1, b1, w2, b2);
  NGPDE_LAUNCH_CHECK("pack_params_kernel");
  return NGPDE_OK;
}

// The four slab reductions of an adjoint (dW1, db1, dW2, db2) in ONE launch, written where the caller wants them -- its own du x du /
// du arrays (the leading block of the plan's d x d accumulation when the plan is widened) -- with reduce_slabs_kernel's sums in its
// order (gcn_fused.hip: bitwise the same numbers); block (0, 0) latches the fault word of the adjoint launch in front of it
struct Reduce4 {
  const float *slab[4];
  float *out[4];        // NULL: not asked for
  int len[4], ct[4];    // ct > 0: a [D][D] matrix in MFMA tile order (D = 16 ct), ct == 0: a vector
  int n_slabs, du;
  const unsigned *abort_word;
  unsigned *fault;
};
__global__ __launch_bounds__(1024) void reduce4_slabs_kernel(const Reduce4 r) {
  __shared__ float part[16][64];
  if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0 && r.abort_word && *r.abort_word != 0) *r.fault = 1u;
  const int job = blockIdx.y;
  const float *__restrict__ slab = r.slab[job];
  float *__restrict__ out = r.out[job];
  const int len = r.len[job], ct = r.ct[job], n_slabs = r.n_slabs;
  if (out == nullptr || (int)blockIdx.x * 64 >= len) return;   // (block-uniform)
  const int el = threadIdx.x & 63, part_id = threadIdx.x >> 6;
  const int e = blockIdx.x * 64 + el;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (e < len) {
    int b = part_id;
    for (; b + 48 < n_slabs; b += 64) {
      s0 += slab[(size_t)b * len + e];
      s1 += slab[(size_t)(b + 16) * len + e];
      s2 += slab[(size_t)(b + 32) * len + e];
      s3 += slab[(size_t)(b + 48) * len + e];
    }
    for (; b < n_slabs; b += 16) s0 += slab[(size_t)b * len + e];
  }
  part[part_id][el] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (part_id == 0 && e < len) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += part[k][el];
    if (ct > 0) {  // e = (tt * 64 + lane) * 4 + reg  ->  dWt[(mt*16 + 4*kq + reg)][nt*16 + i]
      const int reg = e & 3, ln = (e >> 2) & 63, tt = e >> 8;
      const int mt = tt / ct, nt = tt % ct;
      const int row = mt * 16 + 4 * (ln >> 4) + reg, col = nt * 16 + (ln & 15);
      if (row < r.du && col < r.du) out[row * r.du + col] = v;
    } else if (e < r.du) {
      out[e] = v;
    }
  }
}

// [h][w] block between matrices of row pitch spitch / dpitch (elements): the parameters of a widened plan
int32_t copy_block(float *dst, int dpitch, const float *src, int spitch, int w, int h, hipStream_t stream) {
  NGPDE_HIP_CHECK(hipMemcpy2DAsync(dst, (size_t)dpitch * sizeof(float), src, (size_t)spitch * sizeof(float), (size_t)w * sizeof(float),
                                   (size_t)h, hipMemcpyDeviceToDevice, stream));
  return NGPDE_OK;
}

bool act_needs_z(int act) {
  return !(act == NGPDE_ACT_IDENTITY || act == NGPDE_ACT_RELU || act == NGPDE_ACT_LEAKYRELU);
}

}  // namespace

struct ngpde_node {
  const ngpde_graph *g = nullptr;
  int d = 0, act = 0, n_steps = 0;
  // du < d: a WIDENED plan -- the caller's state and parameters are du wide (16 or 32) and run zero-padded on the 64-wide persistent
  // kernels (a phase of those is a latency floor, not a byte count, so the padding is free where a native narrow kernel would sit on the
  // same floor); padded columns of u stay decoupled from the real ones because the padded rows AND columns of W are zero
  int du = 0;
  float dt = 0.f;
  bool with_bwd = false, needs_z = false, eager = false;
  Tableau tb;
  int64_t n = 0;
  size_t row_elems = 0;  // n * d
  int members = 1;       // > 1: a block-diagonal batch of `members` graphs with this structure, solved one after the other by the
                         // persistent launches (u0 / uT / du0 are [members * n][d])
  size_t all_elems = 0;  // members * row_elems
  int nb = 0;            // workgroups of the fused kernels (= slabs)
  int slots = 0;         // tape slots per stage evaluation

  float *u = nullptr, *ustage = nullptr, *u0keep = nullptr;
  float *w1 = nullptr, *b1 = nullptr, *w2 = nullptr, *b2 = nullptr;
  float *tape = nullptr;
  size_t tape_bytes = 0;
  // relu with backward: the pullback needs only the sign of z, so each evaluation keeps two aggregated inputs on the tape
  // plus two 4-bit-per-value masks; the layer outputs live in 2 S buffers that every step re-uses
  bool mask_mode = false;
  // pre-scaled form: u, the stage inputs, the layer outputs and the adjoint products g1 / g2 are held multiplied by c[row]
  // (u~ = c .* u), so that no kernel scales a row while staging it and the halo rows go to LDS by DMA; lambda and the
  // stage adjoints are derivatives with respect to u~.  u0 is scaled on entry, u(T) and du0 on exit.
  bool pre = false;
  float *ybuf = nullptr;         // [S][2][row_elems]
  uint8_t *masks = nullptr;      // [n_steps][S][2][mask_bytes]
  size_t mask_bytes = 0;
  float *lam = nullptr, *g1 = nullptr, *g2 = nullptr;
  std::vector<float *> ubar;
  float *slabs = nullptr;
  size_t slab_bytes = 0;
  float *slab_dw1 = nullptr, *slab_db1 = nullptr, *slab_dw2 = nullptr, *slab_db2 = nullptr;
  float *dw1 = nullptr, *db1 = nullptr, *dw2 = nullptr, *db2 = nullptr;

  // Persistent form (node_persistent.hip): the whole forward solve / the whole adjoint as ONE launch each, tiles
  // synchronised by per-tile phase flags.  Chosen when the graph is one co-resident wave of tiles (<= 2 per CU), d = 64,
  // relu (adjoint), unweighted, pre-scaled form available; NGPDE_NO_PERSISTENT=1 or NGPDE_PERSISTENT=fwd|bwd restrict it.
  bool persist_fwd = false, persist_bwd = false;
  bool hub = false;          // ... in the hub geometry (graphs whose tiles do not fit the handle's halo lists)
  // persistent adjoint of an activation other than relu: the tape holds the aggregated inputs (first half) and the pre-activations
  // (second half, `ztape`), two rows per stage evaluation each, indexed like the relu plan's tape
  bool ztape_mode = false;
  float *ztape = nullptr;
  NodePersist persist;
  OwnFirst of;               // the plan's own-first slot tables (common.h): every kernel of the plan, persistent or replayed, reads them
  bool has_of = false;
  float *pbuf = nullptr;     // layer-1 output exchanged between tiles in the persistent forward
  // a batch (members > 1) runs two members at a time per workgroup (node_persistent.hip, "slots"): the exchanged arrays
  // (ustage, pbuf, g1, g2) hold one [N][d] array per slot, pubar is the adjoint's stage-adjoint scratch [2][5][N][d]
  bool interleave = false;
  bool pair = false;         // ONE member, two tiles per workgroup (graphs of up to twice the co-resident tile count)
  int ktiles = 0;            // > 0: tile rounds -- ktiles tiles per workgroup taking turns (larger graphs still); kstate = their u, k_j rows
  float *kstate = nullptr;
  float *pubar = nullptr;

  hipStream_t cap_stream = nullptr;
  hipGraph_t fwd_graph = nullptr, bwd_graph = nullptr;
  hipGraphExec_t fwd_exec = nullptr, bwd_exec = nullptr;
  bool forward_done = false;
  // The plan owns ONE tape: a second forward overwrites what the first one's backward needs.  Every forward stamps a new
  // generation; a caller that may interleave solves (autograd) records it and has it checked before the backward.
  uint64_t generation = 0;
  bool backward_pending = false;
  int fwd_launches = 0, bwd_launches = 0;

  // tape slot k of (step, stage): 0 = A1 (aggregated input of layer 1), 1 = Y1, 2 = A2, 3 = Y2 = k_i,
  // 4 = Z1, 5 = Z2 (only for activations whose derivative needs the pre-activation)
  float *slot(int step, int stage, int k) const {
    const int s = with_bwd ? step : 0;
    if (mask_mode) {
      if (k == 1 || k == 3) return ybuf + ((size_t)stage * 2 + (k == 3 ? 1 : 0)) * row_elems;
      return tape + ((size_t)(s * tb.S + stage) * 2 + (k == 2 ? 1 : 0)) * row_elems;
    }
    return tape + ((size_t)(s * tb.S + stage) * slots + k) * row_elems;
  }
  uint8_t *mask_slot(int step, int stage, int layer) const {
    return masks + ((size_t)(step * tb.S + stage) * 2 + (layer - 1)) * mask_bytes;
  }
};

namespace {

// profiling pass: start/stop events on every stride-th launch, tagged by kernel role
struct Prof {
  int stride = 1, counter = 0;
  std::vector<hipEvent_t> ev0, ev1;
  std::vector<int> role;
  bool want(int r, hipEvent_t *a, hipEvent_t *b) {
    if ((counter++ % stride) != 0) return false;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return false;
    ev0.push_back(e0); ev1.push_back(e1); role.push_back(r);
    *a = e0; *b = e1;
    return true;
  }
  ~Prof() {
    for (hipEvent_t e : ev0) (void)hipEventDestroy(e);
    for (hipEvent_t e : ev1) (void)hipEventDestroy(e);
  }
};

int32_t dev_alloc(float **p, size_t elems) {
  *p = nullptr;
  NGPDE_HIP_CHECK(hipMalloc((void **)p, std::max<size_t>(elems, 1) * sizeof(float)));
  return NGPDE_OK;
}

int32_t enqueue_forward(ngpde_node *p, hipStream_t stream, int *launches, Prof *prof = nullptr) {
  const Tableau &tb = p->tb;
  int32_t st;
  int count = 0;
  for (int n = 0; n < p->n_steps; ++n) {
    for (int i = 0; i < tb.S; ++i) {
      FusedFwdArgs f1;
      f1.g = p->g; f1.d = p->d; f1.act = p->act;
      f1.x = (i == 0) ? p->u : p->ustage;
      f1.wt = p->w1; f1.bias = p->b1;
      f1.y = p->slot(n, i, 1);
      f1.save_agg = p->with_bwd ? p->slot(n, i, 0) : nullptr;
      f1.save_z = (p->with_bwd && p->needs_z) ? p->slot(n, i, 4) : nullptr;
      f1.save_mask = p->mask_mode ? p->mask_slot(n, i, 1) : nullptr;
      f1.pre = p->pre;
      f1.of = p->has_of ? &p->of : nullptr;
      if (prof) prof->want(0, &f1.ev_start, &f1.ev_stop);
      if ((st = launch_fused_fwd(f1, stream))) return st;
      FusedFwdArgs f2;
      f2.g = p->g; f2.d = p->d; f2.act = p->act;
      f2.x = p->slot(n, i, 1);
      f2.wt = p->w2; f2.bias = p->b2;
      f2.y = p->slot(n, i, 3);
      f2.save_agg = p->with_bwd ? p->slot(n, i, 2) : nullptr;
      f2.save_z = (p->with_bwd && p->needs_z) ? p->slot(n, i, 5) : nullptr;
      f2.save_mask = p->mask_mode ? p->mask_slot(n, i, 2) : nullptr;
      f2.pre = p->pre;
      f2.of = p->has_of ? &p->of : nullptr;
      // epilogue: next stage input, or the step update after the last stage
      const bool last = (i == tb.S - 1);
      const std::vector<double> &row = last ? tb.b : tb.a[i + 1];
      f2.has_comb = true;
      f2.comb_out = last ? p->u : p->ustage;
      f2.comb.n = 0;
      f2.comb.ptr[f2.comb.n] = p->u;
      f2.comb.coef[f2.comb.n++] = 1.0f;
      for (int j = 0; j < i; ++j) {
        if (row[j] == 0.0) continue;
        f2.comb.ptr[f2.comb.n] = p->slot(n, j, 3);
        f2.comb.coef[f2.comb.n++] = (float)(p->dt * row[j]);
      }
      f2.comb.coef_self = (float)(p->dt * row[i]);
      if (prof) prof->want(1, &f2.ev_start, &f2.ev_stop);
      if ((st = launch_fused_fwd(f2, stream))) return st;
      count += 2;
    }
  }
  if (launches) *launches = count;
  return NGPDE_OK;
}

void fill_dense(const ngpde_node *p, FusedBwdArgs &a, int layer, int step, int stage) {
  a.do_dense = true;
  a.pre = p->pre;
  a.of = p->has_of ? &p->of : nullptr;
  if (p->mask_mode) a.mask = p->mask_slot(step, stage, layer);
  if (layer == 2) {
    a.z = p->needs_z ? p->slot(step, stage, 5) : p->slot(step, stage, 3);
    a.saved_agg = p->slot(step, stage, 2);
    a.wt = p->w2; a.g_out = p->g2; a.slab_dw = p->slab_dw2; a.slab_db = p->slab_db2;
  } else {
    a.z = p->needs_z ? p->slot(step, stage, 4) : p->slot(step, stage, 1);
    a.saved_agg = p->slot(step, stage, 0);
    a.wt = p->w1; a.g_out = p->g1; a.slab_dw = p->slab_dw1; a.slab_db = p->slab_db1;
  }
}

int32_t enqueue_backward(ngpde_node *p, hipStream_t stream, int *launches, Prof *prof = nullptr) {
  const Tableau &tb = p->tb;
  const int S = tb.S;
  int32_t st;
  int count = 0;
  // zero the slabs with a kernel, not hipMemsetAsync: a memset node captured into the graph is not reliably ordered before
  // the first kernel node on replays after the first (observed: dW drifting in the 7th digit from the second replay on,
  // garbage with unpaired workgroups; eager launches were always right)
  if ((st = launch_zero(p->slabs, p->slab_bytes, stream))) return st;
  {  // K-bar of the last stage of the last step, then layer 2's dense backward
    FusedBwdArgs a;
    a.g = p->g; a.d = p->d; a.act = p->act;
    a.aggregate = false; a.g_in = p->lam;
    a.has_comb = true; a.comb.n = 0; a.comb.coef_self = (float)(p->dt * tb.b[S - 1]);
    fill_dense(p, a, 2, p->n_steps - 1, S - 1);
    if ((st = launch_fused_bwd(a, stream))) return st;
    ++count;
  }
  for (int n = p->n_steps - 1; n >= 0; --n) {
    for (int i = S - 1; i >= 0; --i) {
      FusedBwdArgs m;  // layer 1 of stage i: dY1 = A^T g2
      m.g = p->g; m.d = p->d; m.act = p->act;
      m.aggregate = true; m.g_in = p->g2;
      fill_dense(p, m, 1, n, i);
      if (prof) prof->want(2, &m.ev_start, &m.ev_stop);
      if ((st = launch_fused_bwd(m, stream))) return st;
      FusedBwdArgs e;  // U-bar_i = A^T g1, then the next stage's K-bar and layer-2 dense backward
      e.g = p->g; e.d = p->d; e.act = p->act;
      e.aggregate = true; e.g_in = p->g1;
      e.has_comb = true; e.comb.n = 0;
      if (i >= 1) {
        e.store_t = p->ubar[i];
        e.comb.ptr[e.comb.n] = p->lam;
        e.comb.coef[e.comb.n++] = (float)(p->dt * tb.b[i - 1]);
        for (int j = i + 1; j < S; ++j) {
          if (tb.a[j][i - 1] == 0.0) continue;
          e.comb.ptr[e.comb.n] = p->ubar[j];
          e.comb.coef[e.comb.n++] = (float)(p->dt * tb.a[j][i - 1]);
        }
        e.comb.coef_self = (float)(p->dt * tb.a[i][i - 1]);
        fill_dense(p, e, 2, n, i - 1);
      } else {
        // lambda_n = lambda_{n+1} + sum_j U-bar_j
        e.comb.ptr[e.comb.n] = p->lam;
        e.comb.coef[e.comb.n++] = 1.0f;
        for (int j = 1; j < S; ++j) {
          e.comb.ptr[e.comb.n] = p->ubar[j];
          e.comb.coef[e.comb.n++] = 1.0f;
        }
        e.comb.coef_self = 1.0f;
        e.store_v = p->lam;
        if (n > 0) {
          e.v_scale = (float)(p->dt * tb.b[S - 1]);
          fill_dense(p, e, 2, n - 1, S - 1);
        } else {
          e.do_dense = false;
          e.pre = p->pre;
          e.of = p->has_of ? &p->of : nullptr;
        }
      }
      if (prof && e.do_dense) prof->want(3, &e.ev_start, &e.ev_stop);
      if ((st = launch_fused_bwd(e, stream))) return st;
      count += 2;
    }
  }
  const int dd = p->d * p->d;
  const int ns = fused_num_slabs(p->n, p->d);   // the slabs beyond it stay zero
  if ((st = launch_reduce_slabs(p->slab_dw1, ns, dd, p->d / 16, p->dw1, stream))) return st;
  if ((st = launch_reduce_slabs(p->slab_db1, ns, p->d, 0, p->db1, stream))) return st;
  if ((st = launch_reduce_slabs(p->slab_dw2, ns, dd, p->d / 16, p->dw2, stream))) return st;
  if ((st = launch_reduce_slabs(p->slab_db2, ns, p->d, 0, p->db2, stream))) return st;
  count += 4;
  if (launches) *launches = count;
  return NGPDE_OK;
}

// the persistent forms: ONE launch for the whole solve / the whole adjoint (node_persistent.hip)
// fold_latch: the caller's next kernel on the stream latches the fault word (ngpde_node_gcn2_forward: the exit scaling)
int32_t enqueue_forward_persistent(ngpde_node *p, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr, bool fold_latch = false) {
  NodePersistFwd a;
  a.no_latch = fold_latch;
  a.g = p->g; a.ps = &p->persist; a.n_steps = p->n_steps; a.S = p->tb.S; a.act = p->act; a.n_members = p->members;
  a.u_in = p->u; a.u_out = p->u;     // a tile writes its rows of u(T) only after all its readers are past phase 1
  a.bufA = p->ustage; a.bufB = p->pbuf;
  a.w1 = p->w1; a.b1 = p->b1; a.w2 = p->w2; a.b2 = p->b2;
  a.row_elems = p->row_elems;
  if (p->with_bwd) {
    a.tape = p->tape; a.masks = p->masks; a.mask_bytes = p->mask_bytes; a.ztape = p->ztape;
  }
  a.interleave = p->interleave; a.pair = p->pair; a.k_tiles = p->ktiles; a.state = p->kstate;
  a.of = p->has_of ? &p->of : nullptr;
  a.ev_start = ev0; a.ev_stop = ev1;
  return launch_node_fwd_persistent(a, stream);
}

// out[4] = where dW1, db1, dW2, db2 go (du x du / du arrays; NULL entries are skipped); nullptr: the plan's own d x d / d buffers
int32_t enqueue_backward_persistent(ngpde_node *p, hipStream_t stream, hipEvent_t ev0 = nullptr, hipEvent_t ev1 = nullptr,
                                    float *const *out = nullptr) {
  NodePersistBwd a;
  a.no_latch = true;   // (the reduction below latches)
  a.g = p->g; a.ps = &p->persist; a.n_steps = p->n_steps; a.S = p->tb.S; a.n_members = p->members; a.act = p->act;
  a.lam = p->lam; a.g1 = p->g1; a.g2 = p->g2; a.w1 = p->w1; a.w2 = p->w2;
  a.tape = p->tape; a.masks = p->masks; a.row_elems = p->row_elems; a.mask_bytes = p->mask_bytes; a.ztape = p->ztape;
  a.slab_dw1 = p->slab_dw1; a.slab_db1 = p->slab_db1; a.slab_dw2 = p->slab_dw2; a.slab_db2 = p->slab_db2;
  a.interleave = p->interleave; a.pair = p->pair; a.ubar = p->pubar; a.k_tiles = p->ktiles;
  a.of = p->has_of ? &p->of : nullptr;
  a.ev_start = ev0; a.ev_stop = ev1;
  int32_t st;
  if ((st = launch_node_bwd_persistent(a, stream))) return st;
  const int dd = p->d * p->d;
  Reduce4 r;
  r.n_slabs = (p->pair || p->ktiles) ? p->persist.pair_wgs : p->persist.n_tiles;   // one slab per workgroup, each written once at the end of the launch
  r.slab[0] = p->slab_dw1; r.slab[1] = p->slab_db1; r.slab[2] = p->slab_dw2; r.slab[3] = p->slab_db2;
  r.len[0] = r.len[2] = dd; r.len[1] = r.len[3] = p->d;
  r.ct[0] = r.ct[2] = p->d / 16; r.ct[1] = r.ct[3] = 0;
  if (out) {
    for (int j = 0; j < 4; ++j) r.out[j] = out[j];
    r.du = p->du;
  } else {
    r.out[0] = p->dw1; r.out[1] = p->db1; r.out[2] = p->dw2; r.out[3] = p->db2;
    r.du = p->d;
  }
  r.abort_word = node_persistent_abort_word(&p->persist); r.fault = p->persist.fault;
  hipLaunchKernelGGL(reduce4_slabs_kernel, dim3((dd + 63) / 64, 4), dim3(1024), 0, stream, r);
  NGPDE_LAUNCH_CHECK("reduce4_slabs_kernel");
  return NGPDE_OK;
}

int32_t capture(ngpde_node *p, bool backward) {
  hipGraph_t graph = nullptr;
  NGPDE_HIP_CHECK(hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeRelaxed));
  int count = 0;
  int32_t st = backward ? enqueue_backward(p, p->cap_stream, &count) : enqueue_forward(p, p->cap_stream, &count);
  hipError_t e = hipStreamEndCapture(p->cap_stream, &graph);
  if (st) {
    if (graph) (void)hipGraphDestroy(graph);
    return st;
  }
  if (e != hipSuccess) {
    if (graph) (void)hipGraphDestroy(graph);
    return fail(NGPDE_ERR_HIP, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
  }
  hipGraphExec_t exec = nullptr;
  e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
  if (e != hipSuccess) {
    (void)hipGraphDestroy(graph);
    return fail(NGPDE_ERR_HIP, "hipGraphInstantiate failed: %s", hipGetErrorString(e));
  }
  if (backward) {
    p->bwd_graph = graph; p->bwd_exec = exec; p->bwd_launches = count;
  } else {
    p->fwd_graph = graph; p->fwd_exec = exec; p->fwd_launches = count;
  }
  return NGPDE_OK;
}

}  // namespace

extern "C" {

int32_t ngpde_node_destroy(ngpde_node_t *p) {
  NGPDE_RANGE();
  if (!p) return NGPDE_OK;
  if (p->fwd_exec) (void)hipGraphExecDestroy(p->fwd_exec);
  if (p->bwd_exec) (void)hipGraphExecDestroy(p->bwd_exec);
  if (p->fwd_graph) (void)hipGraphDestroy(p->fwd_graph);
  if (p->bwd_graph) (void)hipGraphDestroy(p->bwd_graph);
  if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
  float *bufs[] = {p->u0keep, p->u, p->ustage, p->w1, p->b1, p->w2, p->b2, p->tape, p->lam, p->g1, p->g2, p->slabs,
                   p->dw1, p->db1, p->dw2, p->db2};
  for (float *b : bufs)
    if (b) (void)hipFree(b);
  for (float *b : p->ubar)
    if (b) (void)hipFree(b);
  if (p->ybuf) (void)hipFree(p->ybuf);
  if (p->masks) (void)hipFree(p->masks);
  if (p->pbuf) (void)hipFree(p->pbuf);
  if (p->pubar) (void)hipFree(p->pubar);
  if (p->kstate) (void)hipFree(p->kstate);
  node_persistent_free(&p->persist);
  own_first_tables_free(&p->of);
  delete p;
  return NGPDE_OK;
}

int32_t ngpde_node_gcn2_create(const ngpde_graph_t *g, int32_t d, int32_t act, int32_t tableau, int32_t n_steps,
                               float dt, int32_t with_backward, ngpde_node_t **out) {
  NGPDE_RANGE();
  return ngpde_node_gcn2_create_batch(g, 1, d, act, tableau, n_steps, dt, with_backward, out);
}

// d: the width the kernels run at; du <= d: the caller's width (see ngpde_node::du)
static int32_t node_create(const ngpde_graph_t *g, int32_t members, int32_t d, int32_t du, int32_t act, int32_t tableau,
                           int32_t n_steps, float dt, int32_t with_backward, ngpde_node_t **out) {
  NGPDE_REQUIRE(out != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gcn2_create: out is NULL");
  NGPDE_REQUIRE(members >= 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gcn2_create_batch: members must be >= 1");
  *out = nullptr;
  NGPDE_REQUIRE(g != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gcn2_create: graph is NULL");
  NGPDE_REQUIRE(g->has_norm, NGPDE_ERR_STATE, "ngpde_node_gcn2_create: GCN normalisation not set");
  NGPDE_REQUIRE(fused_supported(d, d), NGPDE_ERR_UNSUPPORTED,
                "ngpde_node_gcn2_create: d must be one of 16, 32, 64, 128 (got %d)", d);
  NGPDE_REQUIRE(act >= NGPDE_ACT_IDENTITY && act <= NGPDE_ACT_SOFTPLUS, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_node_gcn2_create: unknown activation code %d", act);
  NGPDE_REQUIRE(tableau == NGPDE_TABLEAU_EULER || tableau == NGPDE_TABLEAU_TSIT5, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_node_gcn2_create: unknown tableau %d", tableau);
  NGPDE_REQUIRE(n_steps >= 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gcn2_create: n_steps must be >= 1");
  NGPDE_REQUIRE(g->n_nodes >= 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gcn2_create: empty graph");
  ngpde_node *p = new (std::nothrow) ngpde_node();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "out of host memory");
  p->g = g; p->d = d; p->du = du; p->act = act; p->n_steps = n_steps; p->dt = dt;
  p->with_bwd = with_backward != 0;
  p->needs_z = p->with_bwd && act_needs_z(act);
  p->tb = make_tableau(tableau);
  p->n = g->n_nodes;
  p->members = members;
  p->row_elems = (size_t)p->n * d;
  p->all_elems = (size_t)members * p->row_elems;
  p->nb = fused_num_blocks(p->n);
  p->mask_mode = p->with_bwd && act == NGPDE_ACT_RELU && std::getenv("NGPDE_NO_MASK") == nullptr;
  p->pre = fused_prescaled_supported(g, d) && std::getenv("NGPDE_NO_PRESCALE") == nullptr;
  // The persistent form (node_persistent.hip) is decided here, before the tape is sized: with an activation other than relu its
  // adjoint reads the pre-activations from a tape of its own layout.  (Interleaved batches: relu only.)
  int pmode = p->pre ? node_persistent_mode(g, d, act, p->with_bwd) : 0;
  // Graphs with hubs (a tile beyond the handle's 96-row halo lists, BASELINE config 1): the hub geometry of the persistent kernels, on
  // pre-scaled arrays like every persistent plan -- both directions or not at all (there is no pre-scaled replayed plan for them)
  bool hub = !p->pre && pmode == 0 && std::getenv("NGPDE_NO_PRESCALE") == nullptr && node_persistent_hub_possible(g, d) &&
             (!p->with_bwd || p->mask_mode || act != NGPDE_ACT_RELU);
  if (hub) {
    pmode = 4;
    p->pre = true;
  }
  p->pair = pmode == 2 && members == 1 && (!p->with_bwd || p->mask_mode);
  p->ktiles = (pmode == 3 && members == 1) ? node_persistent_rounds(g) : 0;
  bool want_persist = (pmode == 1 && (!p->with_bwd || p->mask_mode || (act != NGPDE_ACT_RELU && members == 1))) || p->pair || p->ktiles > 0 || hub;
  const int S = p->tb.S;
  int32_t st = NGPDE_OK;
  if (want_persist) {
    const char *only = std::getenv("NGPDE_PERSISTENT");
    p->persist_fwd = !(only && std::strcmp(only, "bwd") == 0);
    p->persist_bwd = p->with_bwd && !(only && std::strcmp(only, "fwd") == 0);
    // stage-indexed coefficient tables (device memory): forward cf[i][j], adjoint dtb[j], cu[i][j]
    float coef[90] = {0};   // forward: cf[i][j] (j < i) at i * 6 + j, self weights at 36 + i; adjoint: dt b at 42 + j,
                            // cu[i][j] (j > i >= 1) at 48 + i * 6 + j, self weights at 84 + i  (node_persistent.hip)
    const Tableau &tb = p->tb;
    for (int i = 0; i < S; ++i) {
      const std::vector<double> &row = (i == S - 1) ? tb.b : tb.a[i + 1];
      for (int j = 0; j < i; ++j) coef[i * 6 + j] = (float)(dt * row[j]);
      coef[36 + i] = (float)(dt * row[i]);
      coef[42 + i] = (float)(dt * tb.b[i]);
    }
    for (int i = 1; i < S; ++i) {
      for (int j = i + 1; j < S; ++j) coef[48 + i * 6 + j] = (float)(dt * tb.a[j][i - 1]);
      coef[84 + i] = (float)(dt * tb.a[i][i - 1]);
    }
    st = node_persistent_setup(g, coef, &p->persist, p->pair, hub);
    if (st == NGPDE_ERR_UNSUPPORTED) {   // a wait list too long for one polling wave (or neighbouring tile pairs): the replayed plan
      st = NGPDE_OK;
      p->persist_fwd = p->persist_bwd = false;
    }
    if (p->pair && !(p->persist_fwd && (p->persist_bwd || !p->with_bwd))) {   // tile pairs: both directions or none
      p->persist_fwd = p->persist_bwd = false;
    }
    if (!p->persist_fwd) p->pair = false;
    if (p->ktiles > 0) {   // tile rounds: both directions or none; the grid is ceil(tiles / K)
      if (p->persist_fwd && (p->persist_bwd || !p->with_bwd)) p->persist.pair_wgs = (p->persist.n_tiles + p->ktiles - 1) / p->ktiles;
      else { p->persist_fwd = p->persist_bwd = false; p->ktiles = 0; }
    }
  }
  if (!p->persist_fwd) p->ktiles = 0;
  // activations other than relu: the persistent pair needs BOTH directions persistent (the tapes' layouts differ from the replayed plan's)
  if (p->with_bwd && !p->mask_mode && !(p->persist_fwd && p->persist_bwd)) p->persist_fwd = p->persist_bwd = false;
  if (hub && !(p->persist_fwd && (p->persist_bwd || !p->with_bwd))) {   // hub geometry refused (a cap, NGPDE_PERSISTENT=fwd / bwd): the unscaled replayed plan
    p->persist_fwd = p->persist_bwd = false;
    p->pre = false;
    hub = false;
    node_persistent_free(&p->persist);
  }
  p->hub = hub;
  // own-first slot tables (common.h: OwnFirst): for every pre-scaled plan on the handle's own tile lists -- persistent in any form or
  // replayed, one member or a batch, weighted or not -- so that all of them sum a row's neighbours in ONE order and stay bitwise
  // comparable; NGPDE_NO_OWN_FIRST=1 keeps the handle's order (A/B runs)
  // Only for graphs of one wave of workgroups (<= two 32-row tiles per CU): the forms for larger graphs -- tile pairs, tile rounds -- have
  // no exposed wait to fill and would only pay the padding rounds (32 768 nodes - 1.2 %, 65 536 - 0.7 %); the rule looks at the graph
  // and the device alone, never at a plan-selecting switch, so a graph's persistent and replayed plans always agree on it.
  int dev_ = 0, cus_ = 0;
  const bool one_wave = hipGetDevice(&dev_) == hipSuccess && hipDeviceGetAttribute(&cus_, hipDeviceAttributeMultiprocessorCount, dev_) == hipSuccess &&
                        p->nb <= 2 * cus_;
  if (st == NGPDE_OK && p->pre && !hub && one_wave) {
    const char *nof = std::getenv("NGPDE_NO_OWN_FIRST");
    if (!(nof && nof[0] == '1')) {
      st = own_first_tables_build(g, &p->of, nullptr);
      p->has_of = st == NGPDE_OK;
    }
  }
  p->ztape_mode = p->with_bwd && !p->mask_mode && p->persist_fwd && p->persist_bwd;
  p->slots = p->mask_mode ? 2 : (p->ztape_mode ? 4 : (p->with_bwd ? (p->needs_z ? 6 : 4) : 4));
  p->mask_bytes = p->mask_mode ? fused_mask_bytes(p->n, d) : 0;
  const char *eager = std::getenv("NGPDE_NODE_EAGER");
  p->eager = eager && eager[0] == '1';
  p->interleave = members > 1 && !hub && node_persistent_interleave_env();   // (hub geometry: the members one after the other)
  const size_t xslots = p->interleave ? 2 : 1;   // [N][d] arrays per exchanged buffer
  auto A = [&](float **ptr, size_t elems) {
    if (st == NGPDE_OK) st = dev_alloc(ptr, elems);
  };
  A(&p->u, p->all_elems);
  A(&p->ustage, xslots * p->row_elems);
  A(&p->u0keep, (size_t)members * p->n * du);
  A(&p->w1, (size_t)d * d); A(&p->b1, d); A(&p->w2, (size_t)d * d); A(&p->b2, d);
  if (st == NGPDE_OK && du != d) {   // the padding of a widened plan's parameters is written once, here
    for (float *w : {p->w1, p->w2})
      if (hipMemset(w, 0, (size_t)d * d * sizeof(float)) != hipSuccess) st = fail(NGPDE_ERR_HIP, "hipMemset failed");
    for (float *b : {p->b1, p->b2})
      if (hipMemset(b, 0, (size_t)d * sizeof(float)) != hipSuccess) st = fail(NGPDE_ERR_HIP, "hipMemset failed");
  }
  const size_t tape_elems = (size_t)(p->with_bwd ? n_steps : 1) * S * p->slots * p->all_elems;
  p->tape_bytes = tape_elems * sizeof(float);
  A(&p->tape, tape_elems);
  if (p->mask_mode) {
    A(&p->ybuf, (size_t)S * 2 * p->row_elems);
    if (st == NGPDE_OK) {
      const size_t mb = (size_t)members * n_steps * S * 2 * p->mask_bytes;
      if (hipMalloc((void **)&p->masks, std::max<size_t>(mb, 1)) != hipSuccess) st = fail(NGPDE_ERR_HIP, "hipMalloc of the sign-bit masks failed");
      else p->tape_bytes += mb + (size_t)S * 2 * p->row_elems * sizeof(float);
    }
  }
  if (p->with_bwd) {
    A(&p->lam, p->all_elems); A(&p->g1, xslots * p->row_elems); A(&p->g2, xslots * p->row_elems);
    if (p->interleave || p->pair || p->ktiles) A(&p->pubar, 2 * 5 * p->row_elems);
    p->ubar.assign(S, nullptr);
    for (int j = 1; j < S; ++j) A(&p->ubar[j], p->row_elems);
    const size_t dd = (size_t)d * d;
    const size_t per = (size_t)p->nb * (dd + d);
    p->slab_bytes = 2 * per * sizeof(float);
    A(&p->slabs, 2 * per);
    if (st == NGPDE_OK) {
      p->slab_dw1 = p->slabs;
      p->slab_db1 = p->slab_dw1 + (size_t)p->nb * dd;
      p->slab_dw2 = p->slab_db1 + (size_t)p->nb * d;
      p->slab_db2 = p->slab_dw2 + (size_t)p->nb * dd;
    }
    A(&p->dw1, dd); A(&p->db1, d); A(&p->dw2, dd); A(&p->db2, d);
  }
  if (st == NGPDE_OK && p->persist_fwd) A(&p->pbuf, xslots * p->row_elems);
  if (st == NGPDE_OK && p->ktiles) A(&p->kstate, 7 * p->row_elems);
  if (st == NGPDE_OK && p->ztape_mode) p->ztape = p->tape + (size_t)n_steps * S * 2 * p->all_elems;
  if (st == NGPDE_OK && members > 1 && !(p->persist_fwd && (p->persist_bwd || !p->with_bwd)))
    st = fail(NGPDE_ERR_UNSUPPORTED, "ngpde_node_gcn2_create_batch: the member-by-member solve exists for the persistent plan only "
                                     "(d = 64, relu, unweighted, at most two 32-row tiles per CU); batch the graphs into one handle instead");
  if (st == NGPDE_OK && !p->eager && !(p->persist_fwd && (p->persist_bwd || !p->with_bwd))) {
    hipError_t e = hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking);
    if (e != hipSuccess) st = fail(NGPDE_ERR_HIP, "hipStreamCreate failed: %s", hipGetErrorString(e));
    if (st == NGPDE_OK && !p->persist_fwd) st = capture(p, false);
    if (st == NGPDE_OK && p->with_bwd && !p->persist_bwd) st = capture(p, true);
  } else if (st == NGPDE_OK) {
    p->fwd_launches = 2 * S * n_steps;
    p->bwd_launches = p->with_bwd ? 1 + 2 * S * n_steps + 4 : 0;
  }
  if (p->persist_fwd) p->fwd_launches = 2;                    // flag reset, the solve (the fault latch rides on the exit scaling)
  if (p->persist_bwd) p->bwd_launches = 3;                    // flag reset, the adjoint, ONE reduction of the four slab sets (it latches too)
  if (st != NGPDE_OK) {
    std::string keep = last_error();
    ngpde_node_destroy(p);
    last_error() = keep;
    return st;
  }
  *out = p;
  return NGPDE_OK;
}

int32_t ngpde_node_gcn2_create_batch(const ngpde_graph_t *g, int32_t members, int32_t d, int32_t act, int32_t tableau,
                                     int32_t n_steps, float dt, int32_t with_backward, ngpde_node_t **out) {
  NGPDE_RANGE();
  // d = 16 / 32 where the 64-wide persistent solver can take the graph: run widened (NGPDE_NO_WIDEN=1: the native-width replayed plan)
  const char *nw = std::getenv("NGPDE_NO_WIDEN");
  if (out && g && (d == 16 || d == 32) && !(nw && nw[0] == '1') && g->has_norm && g->n_nodes >= 1 &&
      (node_persistent_mode(g, 64, act, with_backward != 0) != 0 || node_persistent_hub_possible(g, 64))) {
    ngpde_node_t *w = nullptr;
    if (node_create(g, members, 64, d, act, tableau, n_steps, dt, with_backward, &w) == NGPDE_OK) {
      if (w->persist_fwd && (w->persist_bwd || !w->with_bwd)) {
        *out = w;
        return NGPDE_OK;
      }
      ngpde_node_destroy(w);
    }
  }
  return node_create(g, members, d, d, act, tableau, n_steps, dt, with_backward, out);
}

size_t ngpde_node_tape_bytes(const ngpde_node_t *p) { return p ? p->tape_bytes : 0; }

int32_t ngpde_node_launch_count(const ngpde_node_t *p, int32_t *forward, int32_t *backward) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_launch_count: plan is NULL");
  if (forward) *forward = p->fwd_launches;
  if (backward) *backward = p->bwd_launches;
  return NGPDE_OK;
}

int32_t ngpde_node_flags(const ngpde_node_t *p, int32_t *flags) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr && flags != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_flags: NULL argument");
  *flags = (p->pre ? NGPDE_NODE_PRESCALED : 0) | (p->mask_mode ? NGPDE_NODE_SIGN_MASKS : 0) | (p->eager ? NGPDE_NODE_EAGER : 0) |
           (p->persist_fwd ? NGPDE_NODE_PERSISTENT_FWD : 0) | (p->persist_bwd ? NGPDE_NODE_PERSISTENT_BWD : 0) |
           (p->pair ? NGPDE_NODE_TILE_PAIRS : 0) | (p->ktiles ? NGPDE_NODE_TILE_ROUNDS : 0) | (p->du != p->d ? NGPDE_NODE_WIDENED : 0) |
           (p->hub ? NGPDE_NODE_HUB_GEOMETRY : 0) | (p->has_of ? NGPDE_NODE_OWN_FIRST : 0);
  return NGPDE_OK;
}

int32_t ngpde_node_fault(ngpde_node_t *p, ngpde_stream_t stream_, int32_t *fault) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr && fault != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_fault: NULL argument");
  *fault = 0;
  if (!p->persist.fault_host) return NGPDE_OK;
  NGPDE_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  *fault = *p->persist.fault_host ? 1 : 0;
  return NGPDE_OK;
}

int32_t ngpde_node_pipeline_stats(ngpde_node_t *p, ngpde_stream_t stream_, int64_t *ahead_forward, int64_t *ahead_backward,
                                  int64_t *slot_phases) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_pipeline_stats: plan is NULL");
  if (ahead_forward) *ahead_forward = 0;
  if (ahead_backward) *ahead_backward = 0;
  if (slot_phases) *slot_phases = 0;
  if (!p->persist.stats) return NGPDE_OK;
  const int nt = p->persist.n_tiles;
  std::vector<int> h((size_t)nt * 2);
  NGPDE_HIP_CHECK(hipMemcpyAsync(h.data(), p->persist.stats, h.size() * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream_));
  NGPDE_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  int64_t f = 0, b = 0;
  for (int t = 0; t < nt; ++t) f += h[2 * t], b += h[2 * t + 1];
  if (ahead_forward) *ahead_forward = f;
  if (ahead_backward) *ahead_backward = b;
  if (slot_phases) *slot_phases = (int64_t)nt * p->members * p->n_steps * p->tb.S * 2;
  return NGPDE_OK;
}

int32_t ngpde_node_gcn2_forward(ngpde_node_t *p, const float *u0, const float *w1, const float *b1, const float *w2,
                                const float *b2, float *uT, ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gcn2_forward: plan is NULL");
  NGPDE_REQUIRE(u0 && w1 && w2 && uT, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gcn2_forward: NULL argument");
  NGPDE_REQUIRE(!(p->persist.fault_host && *p->persist.fault_host), NGPDE_ERR_STATE,
                "ngpde_node_gcn2_forward: an earlier persistent launch of this plan gave up waiting for its neighbours (its outputs are NaN): "
                "another kernel held the device's compute units, or two persistent solves of different processes shared the device; "
                "create a new plan (NGPDE_NO_PERSISTENT=1 selects the replayed plan)");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t user_bytes = (size_t)p->members * p->n * p->du * sizeof(float);
  // entry: three launches where there were up to nine stream operations (round 5) -- u0 scaled into the plan and kept raw by one kernel,
  // the four parameter arrays by one
  if (p->pre) {
    RowsExtra x;
    x.keep = reinterpret_cast<float4 *>(p->u0keep);
    int32_t st = launch_scale_rows_width(u0, p->du, p->g->c, p->u, p->d, p->n, false, stream, p->members, x);
    if (st) return st;
  } else {
    NGPDE_HIP_CHECK(hipMemcpyAsync(p->u, u0, p->all_elems * sizeof(float), hipMemcpyDeviceToDevice, stream));
    NGPDE_HIP_CHECK(hipMemcpyAsync(p->u0keep, u0, user_bytes, hipMemcpyDeviceToDevice, stream));
  }
  {
    int32_t st = launch_pack_params(w1, b1, w2, b2, p->du, p->d, p->w1, p->b1, p->w2, p->b2, stream);
    if (st) return st;
  }
  const bool fold_latch = p->persist_fwd && p->pre;   // (the exit scaling latches the launch's fault word)
  if (p->persist_fwd) {
    int32_t st = enqueue_forward_persistent(p, stream, nullptr, nullptr, fold_latch);
    if (st) return st;
  } else if (p->eager) {
    int32_t st = enqueue_forward(p, stream, nullptr);
    if (st) return st;
  } else {
    NGPDE_HIP_CHECK(hipGraphLaunch(p->fwd_exec, stream));
  }
  if (p->pre) {
    RowsExtra x;
    if (fold_latch) { x.abort_word = node_persistent_abort_word(&p->persist); x.fault = p->persist.fault; }
    int32_t st = launch_scale_rows_width(p->u, p->d, p->g->c, uT, p->du, p->n, true, stream, p->members, x);
    if (st) return st;
  } else {
    NGPDE_HIP_CHECK(hipMemcpyAsync(uT, p->u, p->all_elems * sizeof(float), hipMemcpyDeviceToDevice, stream));
  }
  p->forward_done = true;
  ++p->generation;
  p->backward_pending = p->with_bwd;
  return NGPDE_OK;
}

int32_t ngpde_node_generation(const ngpde_node_t *p, uint64_t *generation, int32_t *backward_pending) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_generation: plan is NULL");
  if (generation) *generation = p->generation;
  if (backward_pending) *backward_pending = p->backward_pending ? 1 : 0;
  return NGPDE_OK;
}

int32_t ngpde_node_expect_generation(const ngpde_node_t *p, uint64_t generation) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_expect_generation: plan is NULL");
  NGPDE_REQUIRE(p->generation == generation, NGPDE_ERR_STATE,
                "the plan's tape belongs to solve %llu, not %llu: another forward ran on this plan before this backward "
                "(one solve in flight per plan; use one plan per outstanding solve)",
                (unsigned long long)p->generation, (unsigned long long)generation);
  return NGPDE_OK;
}

int32_t ngpde_node_gcn2_backward(ngpde_node_t *p, const float *duT, float *du0, float *dw1, float *db1, float *dw2,
                                 float *db2, ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gcn2_backward: plan is NULL");
  NGPDE_REQUIRE(p->with_bwd, NGPDE_ERR_STATE, "ngpde_node_gcn2_backward: plan was created without backward");
  NGPDE_REQUIRE(p->forward_done, NGPDE_ERR_STATE, "ngpde_node_gcn2_backward: forward has not been run");
  NGPDE_REQUIRE(duT != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gcn2_backward: duT is NULL");
  NGPDE_REQUIRE(!(p->persist.fault_host && *p->persist.fault_host), NGPDE_ERR_STATE,
                "ngpde_node_gcn2_backward: a persistent launch of this plan gave up waiting for its neighbours (outputs are NaN); "
                "create a new plan (NGPDE_NO_PERSISTENT=1 selects the replayed plan)");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t dd = (size_t)p->d * p->d * sizeof(float), db = (size_t)p->du * sizeof(float);
  if (p->pre) {   // u(T) = u~(T) ./ c  =>  dL/du~(T) = duT ./ c
    int32_t st = launch_scale_rows_width(duT, p->du, p->g->c, p->lam, p->d, p->n, true, stream, p->members);
    if (st) return st;
  } else {
    NGPDE_HIP_CHECK(hipMemcpyAsync(p->lam, duT, p->all_elems * sizeof(float), hipMemcpyDeviceToDevice, stream));
  }
  if (p->persist_bwd) {   // (its one reduction launch writes the four gradients where the caller wants them)
    float *const outs[4] = {dw1, db1, dw2, db2};
    int32_t st = enqueue_backward_persistent(p, stream, nullptr, nullptr, outs);
    if (st) return st;
  } else if (p->eager) {
    int32_t st = enqueue_backward(p, stream, nullptr);
    if (st) return st;
  } else {
    NGPDE_HIP_CHECK(hipGraphLaunch(p->bwd_exec, stream));
  }
  if (du0 && p->pre) {   // u~0 = c .* u0  =>  du0 = c .* dL/du~0
    int32_t st = launch_scale_rows_width(p->lam, p->d, p->g->c, du0, p->du, p->n, false, stream, p->members);
    if (st) return st;
  } else if (du0) {
    NGPDE_HIP_CHECK(hipMemcpyAsync(du0, p->lam, p->all_elems * sizeof(float), hipMemcpyDeviceToDevice, stream));
  }
  p->backward_pending = false;
  if (p->persist_bwd) return NGPDE_OK;
  if (p->du != p->d) {
    int32_t st;
    if (dw1 && (st = copy_block(dw1, p->du, p->dw1, p->d, p->du, p->du, stream))) return st;
    if (dw2 && (st = copy_block(dw2, p->du, p->dw2, p->d, p->du, p->du, stream))) return st;
  } else {
    if (dw1) NGPDE_HIP_CHECK(hipMemcpyAsync(dw1, p->dw1, dd, hipMemcpyDeviceToDevice, stream));
    if (dw2) NGPDE_HIP_CHECK(hipMemcpyAsync(dw2, p->dw2, dd, hipMemcpyDeviceToDevice, stream));
  }
  if (db1) NGPDE_HIP_CHECK(hipMemcpyAsync(db1, p->db1, db, hipMemcpyDeviceToDevice, stream));
  if (db2) NGPDE_HIP_CHECK(hipMemcpyAsync(db2, p->db2, db, hipMemcpyDeviceToDevice, stream));
  return NGPDE_OK;
}

int32_t ngpde_node_profile(ngpde_node_t *p, int32_t stride, float *out_us, int32_t *out_count,
                           ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr && out_us != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_profile: NULL argument");
  NGPDE_REQUIRE(p->forward_done, NGPDE_ERR_STATE, "ngpde_node_profile: run a forward solve first");
  NGPDE_REQUIRE(stride >= 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_profile: stride must be >= 1");
  hipStream_t stream = (hipStream_t)stream_;
  Prof prof;
  prof.stride = stride;
  int32_t st;
  if (p->pre) {
    if ((st = launch_scale_rows_width(p->u0keep, p->du, p->g->c, p->u, p->d, p->n, false, stream, p->members))) return st;
  } else {
    NGPDE_HIP_CHECK(hipMemcpyAsync(p->u, p->u0keep, p->all_elems * sizeof(float), hipMemcpyDeviceToDevice, stream));
  }
  hipEvent_t pe[4] = {nullptr, nullptr, nullptr, nullptr};
  if (p->persist_fwd) {
    prof.want(0, &pe[0], &pe[1]);
    if ((st = enqueue_forward_persistent(p, stream, pe[0], pe[1]))) return st;
  } else if ((st = enqueue_forward(p, stream, nullptr, &prof))) return st;
  if (p->with_bwd) {
    // adjoint seed of loss = sum(u(T)): ones
    std::vector<float> ones(p->all_elems, 1.0f);
    NGPDE_HIP_CHECK(hipMemcpyAsync(p->lam, ones.data(), p->all_elems * sizeof(float), hipMemcpyHostToDevice, stream));
    NGPDE_HIP_CHECK(hipStreamSynchronize(stream));
    if (p->persist_bwd) {
      prof.counter = 0;
      prof.want(2, &pe[2], &pe[3]);
      if ((st = enqueue_backward_persistent(p, stream, pe[2], pe[3]))) return st;
    } else if ((st = enqueue_backward(p, stream, nullptr, &prof))) return st;
  }
  NGPDE_HIP_CHECK(hipStreamSynchronize(stream));
  double sum[4] = {0, 0, 0, 0};
  int cnt[4] = {0, 0, 0, 0};
  for (size_t k = 0; k < prof.role.size(); ++k) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, prof.ev0[k], prof.ev1[k]) != hipSuccess) continue;
    sum[prof.role[k]] += ms * 1000.0;
    cnt[prof.role[k]]++;
  }
  for (int r = 0; r < 4; ++r) {
    out_us[r] = cnt[r] ? (float)(sum[r] / cnt[r]) : 0.f;
    if (out_count) out_count[r] = cnt[r];
  }
  return NGPDE_OK;
}


/* ---- GAT-style layer as right-hand side: device-resident solve + discrete adjoint (gat_fused.hip) ---------------------------- */
struct ngpde_node_gat {
  const ngpde_graph *g = nullptr;
  int heads = 0, c = 0, act = 0, S = 0, n_steps = 0, members = 1;
  float slope = 0.2f;
  bool with_bwd = false, solved = false;
  NodePersist persist;
  float *cf = nullptr, *cb = nullptr;       // device coefficient tables [(S + 1)][8], [S][8]
  float *xs = nullptr, *yz = nullptr, *alpha = nullptr, *kbuf = nullptr;
  float *ubar = nullptr, *dzbuf = nullptr, *dscore = nullptr, *dal = nullptr, *slabs = nullptr;
  int *xpad = nullptr;
  size_t tape_bytes = 0, row_elems = 0, alpha_elems = 0;
};

static void node_gat_free(ngpde_node_gat *p) {
  if (!p) return;
  void *bufs[] = {p->cf, p->cb, p->xs, p->yz, p->alpha, p->kbuf, p->ubar, p->dzbuf, p->dscore, p->dal, p->slabs, p->xpad};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  node_persistent_free(&p->persist);
  delete p;
}

int32_t ngpde_node_gat_supported(const ngpde_graph_t *g, int32_t din, int32_t heads, int32_t c) {
  return (g && din == 64 && gat_node_persistent_supported(g, heads, c)) ? 1 : 0;
}

int32_t ngpde_node_gat_create(const ngpde_graph_t *g, int32_t heads, int32_t c, float negative_slope, int32_t act, int32_t tableau,
                              int32_t n_steps, double dt, int32_t with_backward, ngpde_node_gat_t **out) {
  return ngpde_node_gat_create_batch(g, 1, heads, c, negative_slope, act, tableau, n_steps, dt, with_backward, out);
}

int32_t ngpde_node_gat_create_batch(const ngpde_graph_t *g, int32_t members, int32_t heads, int32_t c, float negative_slope, int32_t act,
                                    int32_t tableau, int32_t n_steps, double dt, int32_t with_backward, ngpde_node_gat_t **out) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr && out != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gat_create: NULL argument");
  *out = nullptr;
  NGPDE_REQUIRE(members >= 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gat_create_batch: members >= 1 required (got %d)", members);
  NGPDE_REQUIRE(tableau == NGPDE_TABLEAU_EULER || tableau == NGPDE_TABLEAU_TSIT5, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_node_gat_create: unknown tableau %d", tableau);
  NGPDE_REQUIRE(n_steps >= 1 && act >= NGPDE_ACT_IDENTITY && act <= NGPDE_ACT_SOFTPLUS, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_node_gat_create: n_steps >= 1 and a known activation required (got %d, %d)", n_steps, act);
  NGPDE_REQUIRE(gat_node_persistent_supported(g, heads, c), NGPDE_ERR_UNSUPPORTED,
                "ngpde_node_gat_create: needs 64 => heads x c = 64 with heads in {1, 2, 4}, tiles that fit the LDS halo in both directions "
                "and at most as many tiles as the device keeps resident (use the generic solver otherwise)");
  ngpde_node_gat *p = new (std::nothrow) ngpde_node_gat();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gat_create: out of host memory");
  p->g = g; p->heads = heads; p->c = c; p->act = act; p->slope = negative_slope; p->n_steps = n_steps; p->with_bwd = with_backward != 0;
  p->members = members;
  const Tableau tb = make_tableau(tableau);
  p->S = tb.S;
  const int S = tb.S;
  // the coefficients exactly as the generic solver passes them to ngpde_rk_stage_combine: float(dt * a_ij) formed in double
  std::vector<float> cf((size_t)(S + 1) * 8, 0.f), cb((size_t)S * 8, 0.f);
  for (int i = 0; i < S; ++i) {
    for (int j = 0; j < i; ++j) cf[(size_t)i * 8 + j] = (float)(dt * tb.a[i][j]);
    cf[(size_t)S * 8 + i] = (float)(dt * tb.b[i]);
    cb[(size_t)i * 8 + i] = (float)(dt * tb.b[i]);
    for (int j = i + 1; j < S; ++j) cb[(size_t)i * 8 + j] = (float)(dt * tb.a[j][i]);
  }
  p->row_elems = (size_t)g->n_nodes * 64;
  p->alpha_elems = (size_t)std::max<int64_t>(g->n_edges, 1) * heads;
  const size_t phases = (size_t)n_steps * S, nt = (size_t)g->n_sched / kTileRows, M = (size_t)members, slots = members > 1 ? 2 : 1;
  auto alloc = [&](auto **ptr, size_t bytes) -> int32_t {
    NGPDE_HIP_CHECK(hipMalloc((void **)ptr, std::max<size_t>(bytes, 256)));
    return NGPDE_OK;
  };
  int32_t st = NGPDE_OK;
  const float coef_unused[90] = {0.f};
  auto step = [&](int32_t r) { if (st == NGPDE_OK) st = r; };
  step(node_persistent_setup(g, coef_unused, &p->persist, false));
  if (st == NGPDE_OK) step(alloc(&p->cf, cf.size() * 4));
  if (st == NGPDE_OK) step(alloc(&p->cb, std::max<size_t>(cb.size(), 64) * 4));
  if (st == NGPDE_OK) step(alloc(&p->kbuf, slots * 7 * p->row_elems * 4));   // per slot: k_0..k_5 and (batches) u
  if (st == NGPDE_OK) step(alloc(&p->xs, M * (p->with_bwd ? phases : 2) * p->row_elems * 4));
  p->tape_bytes = M * (p->with_bwd ? phases : 2) * p->row_elems * 4;
  if (st == NGPDE_OK && p->with_bwd) {
    if (act != NGPDE_ACT_IDENTITY) { step(alloc(&p->yz, M * phases * p->row_elems * 4)); p->tape_bytes += M * phases * p->row_elems * 4; }
    if (st == NGPDE_OK) step(alloc(&p->alpha, M * phases * p->alpha_elems * 4));
    p->tape_bytes += M * phases * p->alpha_elems * 4;
    if (st == NGPDE_OK) step(alloc(&p->ubar, slots * 6 * p->row_elems * 4));
    if (st == NGPDE_OK) step(alloc(&p->dzbuf, slots * 2 * p->row_elems * 4));
    if (st == NGPDE_OK) step(alloc(&p->dscore, slots * 2 * gat_node_dscore_elems(g) * 4));
    if (st == NGPDE_OK) step(alloc(&p->dal, slots * (size_t)g->n_nodes * heads * 4));
    if (st == NGPDE_OK) step(alloc(&p->slabs, nt * (64 * 64 + 64 + 128) * 4));
    if (st == NGPDE_OK) step(alloc(&p->xpad, (size_t)std::max<int64_t>(g->n_edges, 1) * 4));
    if (st == NGPDE_OK) {
      int *scratch = nullptr;
      step(alloc(&scratch, (size_t)std::max<int64_t>(g->n_edges, 1) * 4));
      if (st == NGPDE_OK) step(launch_gat_node_xpad(g, scratch, p->xpad, nullptr));
      if (st == NGPDE_OK && hipStreamSynchronize(nullptr) != hipSuccess) st = NGPDE_ERR_HIP;
      if (scratch) (void)hipFree(scratch);
    }
  }
  if (st == NGPDE_OK && (hipMemcpy(p->cf, cf.data(), cf.size() * 4, hipMemcpyHostToDevice) != hipSuccess ||
                         hipMemcpy(p->cb, cb.data(), cb.size() * 4, hipMemcpyHostToDevice) != hipSuccess)) {
    last_error() = "ngpde_node_gat_create: copying the coefficient tables failed";
    st = NGPDE_ERR_HIP;
  }
  if (st != NGPDE_OK) {
    const std::string keep = last_error();
    node_gat_free(p);
    last_error() = keep;
    return st;
  }
  *out = p;
  return NGPDE_OK;
}

int32_t ngpde_node_gat_destroy(ngpde_node_gat_t *p) {
  NGPDE_RANGE();
  node_gat_free(p);
  return NGPDE_OK;
}

size_t ngpde_node_gat_tape_bytes(const ngpde_node_gat_t *p) { return p ? p->tape_bytes : 0; }

int32_t ngpde_node_gat_fault(ngpde_node_gat_t *p, ngpde_stream_t stream_, int32_t *fault) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr && fault != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gat_fault: NULL argument");
  NGPDE_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  *fault = (p->persist.fault_host && *p->persist.fault_host) ? 1 : 0;
  return NGPDE_OK;
}

int32_t ngpde_node_gat_forward(ngpde_node_gat_t *p, const float *u0, const float *weight, const float *a, const float *bias, float *uT,
                               ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gat_forward: plan is NULL");
  if (p->g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(u0 && weight && a && uT, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gat_forward: NULL argument");
  NGPDE_REQUIRE(!(p->persist.fault_host && *p->persist.fault_host), NGPDE_ERR_STATE,
                "ngpde_node_gat_forward: an earlier launch of this plan gave up waiting for its neighbours (ngpde_node_gat_fault); destroy the plan");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t phases = (size_t)p->n_steps * p->S, xs_stride = (p->with_bwd ? phases : 2) * p->row_elems;
  for (int mb = 0; mb < p->members; ++mb)   // slot 0 of every member's stage-input array = the input of its phase 1
    NGPDE_HIP_CHECK(hipMemcpyAsync(p->xs + (size_t)mb * xs_stride, u0 + (size_t)mb * p->row_elems, p->row_elems * 4, hipMemcpyDeviceToDevice, stream));
  GatNodeFwd f;
  f.g = p->g; f.ps = &p->persist; f.heads = p->heads; f.act = p->act; f.n_steps = p->n_steps; f.S = p->S; f.slope = p->slope;
  f.taped = p->with_bwd; f.u_in = u0; f.wt = weight; f.a = a; f.bias = bias; f.u_out = uT; f.xs = p->xs; f.yz = p->yz; f.alpha = p->alpha;
  f.kbuf = p->kbuf; f.cf = p->cf;
  f.n_members = p->members; f.xs_stride = xs_stride; f.yz_stride = phases * p->row_elems; f.alpha_stride = phases * p->alpha_elems;
  const int32_t st = launch_gat_node_fwd(f, stream);
  if (st == NGPDE_OK) p->solved = true;
  return st;
}

int32_t ngpde_node_gat_backward(ngpde_node_gat_t *p, const float *weight, const float *a, const float *duT, float *du0, float *dweight,
                                float *da, float *dbias, ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gat_backward: plan is NULL");
  NGPDE_REQUIRE(p->with_bwd, NGPDE_ERR_STATE, "ngpde_node_gat_backward: the plan was created without a backward pass");
  NGPDE_REQUIRE(p->solved, NGPDE_ERR_STATE, "ngpde_node_gat_backward: no forward solve has filled the tape");
  NGPDE_REQUIRE(weight && a && duT && du0 && dweight && da, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_gat_backward: NULL argument");
  NGPDE_REQUIRE(!(p->persist.fault_host && *p->persist.fault_host), NGPDE_ERR_STATE,
                "ngpde_node_gat_backward: an earlier launch of this plan gave up waiting for its neighbours (ngpde_node_gat_fault); destroy the plan");
  const size_t nt = (size_t)p->g->n_sched / kTileRows;
  GatNodeBwd b;
  b.g = p->g; b.ps = &p->persist; b.heads = p->heads; b.act = p->act; b.n_steps = p->n_steps; b.S = p->S; b.slope = p->slope;
  b.wt = weight; b.a = a; b.xs = p->xs; b.yz = p->yz; b.alpha = p->alpha; b.duT = duT; b.lam = du0; b.ubar = p->ubar; b.dzbuf = p->dzbuf;
  b.dscore = p->dscore; b.dal = p->dal; b.slab_dw = p->slabs; b.slab_db = p->slabs + nt * 64 * 64; b.slab_u = b.slab_db + nt * 64;
  b.xpad = p->xpad; b.cb = p->cb; b.dwt = dweight; b.da = da; b.db = dbias;
  {
    const size_t phases = (size_t)p->n_steps * p->S;
    b.n_members = p->members; b.xs_stride = phases * p->row_elems; b.yz_stride = phases * p->row_elems; b.alpha_stride = phases * p->alpha_elems;
  }
  return launch_gat_node_bwd(b, (hipStream_t)stream_);
}


/* ---- VMHConv as right-hand side: device-resident solve + discrete adjoint (node_vmh.hip) -------------------------------------- */
}  // extern "C"

struct ngpde_node_vmh {
  const ngpde_graph *g = nullptr;
  VmhShape shape;
  int S = 0, n_steps = 0;
  bool with_bwd = false, solved = false;
  NodePersist persist;
  float *pos = nullptr, *cf = nullptr, *cb = nullptr, *x = nullptr;         // x: [2][N] exchanged stage input
  float *state = nullptr;                                                     // [16][N] tile rounds: per-node state between turns
  int *srcpos = nullptr, *srcdeg = nullptr;                                   // schedule-ordered by-source positions (launch_vmh_srcpos)
  float *tape_phi = nullptr, *tape_gam = nullptr, *dz_phi = nullptr, *dz_gam = nullptr, *dsrc = nullptr;
  float *partial = nullptr, *dwpad = nullptr;                                // weight-pullback workspace; [64 x 64 + 64] padded result
  size_t tape_bytes = 0, partial_floats = 0;
  size_t tape_cap[4] = {0, 0, 0, 0};   // capacity in floats of tape_phi, tape_gam, dz_phi, dz_gam (a block from the pool may be larger than the need)
  int tape_dev = 0;                    // the device they were allocated on
};

// The tapes of a VMH plan are tens of GB at the tutorial's minibatch size, and a training loop that batches its point clouds in a new
// order every epoch (VMH.md:120, DataLoader(shuffle = true)) builds a new plan per step: allocating and freeing such blocks each time
// costs 0.02 - 6 s per step (measured: hipFree of 124 GB 0.8 s, the hipMalloc after it up to 6 s).  Freed tapes are therefore parked
// (at most kTapePoolMax blocks) and handed to the next plan whose need they fit.  A parked block is not zeroed again: it was zeroed
// when allocated, what it holds since is finite, and the 16-float blocks of a row a solve does not write feed only the rows /
// columns of the padded 64 x 64 weight-pullback result that launch_vmh_copy_block never copies out.
namespace {
constexpr size_t kTapePoolMax = 8;
struct TapeBlock {
  float *ptr;
  size_t floats;
  int device;
};
int tape_device() {
  int dev = 0;
  (void)hipGetDevice(&dev);
  return dev;
}
std::mutex g_tape_mu;
std::vector<TapeBlock> g_tape_pool;

// (the block's CAPACITY and the device it was allocated on travel with it: a block that came from the pool larger than the plan's
// need goes back with its full size, on its own device)
float *tape_pool_take(size_t floats, size_t *capacity) {   // the smallest parked block that fits without wasting more than half of itself
  const int dev = tape_device();
  std::lock_guard<std::mutex> lock(g_tape_mu);
  int best = -1;
  for (int i = 0; i < (int)g_tape_pool.size(); ++i)
    if (g_tape_pool[i].device == dev && g_tape_pool[i].floats >= floats && g_tape_pool[i].floats <= 2 * floats + 1024 &&
        (best < 0 || g_tape_pool[i].floats < g_tape_pool[best].floats))
      best = i;
  if (best < 0) return nullptr;
  float *ptr = g_tape_pool[best].ptr;
  *capacity = g_tape_pool[best].floats;
  g_tape_pool.erase(g_tape_pool.begin() + best);
  return ptr;
}
size_t tape_pool_release_all() {   // (before a fresh allocation that would not fit beside the parked blocks)
  std::lock_guard<std::mutex> lock(g_tape_mu);
  size_t n = g_tape_pool.size();
  for (auto &b : g_tape_pool) (void)hipFree(b.ptr);
  g_tape_pool.clear();
  return n;
}
void tape_pool_give(float *ptr, size_t floats, int device) {
  if (!ptr) return;
  std::lock_guard<std::mutex> lock(g_tape_mu);
  if (floats < (64u << 20) / 4 || g_tape_pool.size() >= kTapePoolMax) {   // small blocks and an overfull pool: back to the device
    (void)hipFree(ptr);
    return;
  }
  g_tape_pool.push_back({ptr, floats, device});
}
}  // namespace

static void node_vmh_free(ngpde_node_vmh *p) {
  if (!p) return;
  void *bufs[] = {p->pos, p->cf, p->cb, p->x, p->state, p->dsrc, p->partial, p->dwpad, p->srcpos, p->srcdeg};
  for (void *b : bufs)
    if (b) (void)hipFree(b);
  tape_pool_give(p->tape_phi, p->tape_cap[0], p->tape_dev);
  tape_pool_give(p->tape_gam, p->tape_cap[1], p->tape_dev);
  tape_pool_give(p->dz_phi, p->tape_cap[2], p->tape_dev);
  tape_pool_give(p->dz_gam, p->tape_cap[3], p->tape_dev);
  node_persistent_free(&p->persist);
  delete p;
}

static int32_t vmh_shape(int32_t hd, int32_t pd, int32_t n_phi, const int32_t *phi_dims, const int32_t *phi_acts, int32_t n_gamma,
                         const int32_t *gamma_dims, const int32_t *gamma_acts, int32_t aggr, VmhShape *s) {
  NGPDE_REQUIRE(phi_dims && phi_acts && gamma_dims && gamma_acts, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh: NULL layer table");
  NGPDE_REQUIRE(n_phi >= 1 && n_phi <= kVmhMaxL && n_gamma >= 1 && n_gamma <= kVmhMaxL, NGPDE_ERR_UNSUPPORTED,
                "ngpde_node_vmh: 1 to %d Dense layers per MLP", kVmhMaxL);
  s->hd = hd; s->pd = pd; s->aggr = aggr; s->n_phi = n_phi; s->n_gam = n_gamma;
  for (int l = 0; l <= n_phi; ++l) s->phi_dims[l] = phi_dims[l];
  for (int l = 0; l <= n_gamma; ++l) s->gam_dims[l] = gamma_dims[l];
  for (int l = 0; l < n_phi; ++l) s->phi_act[l] = phi_acts[l];
  for (int l = 0; l < n_gamma; ++l) s->gam_act[l] = gamma_acts[l];
  return NGPDE_OK;
}

extern "C" {

int32_t ngpde_node_vmh_supported(const ngpde_graph_t *g, int32_t hd, int32_t pd, int32_t n_phi, const int32_t *phi_dims, const int32_t *phi_acts,
                                 int32_t n_gamma, const int32_t *gamma_dims, const int32_t *gamma_acts, int32_t aggr) {
  VmhShape s;
  if (!g || vmh_shape(hd, pd, n_phi, phi_dims, phi_acts, n_gamma, gamma_dims, gamma_acts, aggr, &s) != NGPDE_OK) return 0;
  return node_vmh_supported(g, s) ? 1 : 0;
}

int32_t ngpde_node_vmh_create(const ngpde_graph_t *g, int32_t hd, int32_t pd, const float *pos, int32_t n_phi, const int32_t *phi_dims,
                              const int32_t *phi_acts, int32_t n_gamma, const int32_t *gamma_dims, const int32_t *gamma_acts, int32_t aggr,
                              int32_t tableau, int32_t n_steps, double dt, int32_t with_backward, ngpde_node_vmh_t **out) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(g != nullptr && out != nullptr && pos != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_create: NULL argument");
  *out = nullptr;
  NGPDE_REQUIRE(tableau == NGPDE_TABLEAU_EULER || tableau == NGPDE_TABLEAU_TSIT5, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_create: unknown tableau %d", tableau);
  NGPDE_REQUIRE(n_steps >= 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_create: n_steps >= 1 required");
  VmhShape shape;
  int32_t st = vmh_shape(hd, pd, n_phi, phi_dims, phi_acts, n_gamma, gamma_dims, gamma_acts, aggr, &shape);
  if (st) return st;
  NGPDE_REQUIRE(node_vmh_supported(g, shape), NGPDE_ERR_UNSUPPORTED,
                "ngpde_node_vmh_create: needs a scalar state, 1-3 position coordinates, MLPs of 2-4 Dense layers up to 64 wide with identity / relu / "
                "tanh / sigmoid hidden and identity output layers, + or mean aggregation, tiles that fit the LDS halo in both directions and "
                "at most one 16-row half tile per CU (use the generic solver otherwise)");
  ngpde_node_vmh *p = new (std::nothrow) ngpde_node_vmh();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_create: out of host memory");
  p->g = g; p->shape = shape; p->n_steps = n_steps; p->with_bwd = with_backward != 0;
  const Tableau tb = make_tableau(tableau);
  const int S = p->S = tb.S;
  float cf[48] = {0}, cb[64] = {0};
  for (int i = 0; i < S; ++i) {   // forward: node_persistent.hip's table; adjoint: cb[i][i] = dt b_i, cb[i][j] (j > i) = dt a[j][i]
    const std::vector<double> &row = (i == S - 1) ? tb.b : tb.a[i + 1];
    for (int j = 0; j < i; ++j) cf[i * 6 + j] = (float)(dt * row[j]);
    cf[36 + i] = (float)(dt * row[i]);
    cb[i * 8 + i] = (float)(dt * tb.b[i]);
    for (int j = i + 1; j < S; ++j) cb[i * 8 + j] = (float)(dt * tb.a[j][i]);
  }
  const size_t N = (size_t)g->n_nodes, E = (size_t)std::max<int64_t>(g->n_edges, 1), evals = (size_t)n_steps * S;
  auto alloc = [&](float **ptr, size_t floats, bool zero) -> int32_t {
    NGPDE_HIP_CHECK(hipMalloc((void **)ptr, std::max<size_t>(floats * 4, 256)));
    if (zero) NGPDE_HIP_CHECK(hipMemset(*ptr, 0, std::max<size_t>(floats * 4, 256)));
    return NGPDE_OK;
  };
  auto step = [&](int32_t r) { if (st == NGPDE_OK) st = r; };
  const float coef_unused[90] = {0.f};
  step(node_persistent_setup(g, coef_unused, &p->persist, false));
  if (st == NGPDE_OK) step(alloc(&p->pos, N * pd, false));
  if (st == NGPDE_OK && hipMemcpy(p->pos, pos, N * pd * 4, hipMemcpyDeviceToDevice) != hipSuccess) st = fail(NGPDE_ERR_HIP, "ngpde_node_vmh_create: copying the positions failed");
  if (st == NGPDE_OK) step(alloc(&p->cf, 48, false));
  if (st == NGPDE_OK) step(alloc(&p->cb, 64, false));
  if (st == NGPDE_OK) step(alloc(&p->x, 2 * N, true));
  if (st == NGPDE_OK) step(alloc(&p->state, 16 * N, true));
  if (st == NGPDE_OK && p->with_bwd) {
    float *sp = nullptr, *sd = nullptr;
    step(alloc(&sp, (size_t)g->n_sched * kSlotWidth, false));
    if (st == NGPDE_OK) step(alloc(&sd, (size_t)g->n_sched, false));
    p->srcpos = reinterpret_cast<int *>(sp); p->srcdeg = reinterpret_cast<int *>(sd);
    if (st == NGPDE_OK) step(launch_vmh_srcpos(g, p->srcpos, p->srcdeg, nullptr));
    if (st == NGPDE_OK && hipStreamSynchronize(nullptr) != hipSuccess) st = fail(NGPDE_ERR_HIP, "ngpde_node_vmh_create: building the by-source positions failed");
  }
  if (st == NGPDE_OK && p->with_bwd) {
    // (zeroed once: the padded columns of a layer's rows are never written, and the weight-pullback GEMMs read whole 64-wide rows)
    const size_t tp = (size_t)n_phi * evals * E * 64, tg = (size_t)n_gamma * evals * N * 64;
    p->tape_dev = tape_device();
    auto tape = [&](float **ptr, size_t floats, size_t *cap) -> int32_t {   // a parked block that fits, else a fresh zeroed one
      floats = std::max<size_t>(floats, 64);
      *cap = floats;
      if ((*ptr = tape_pool_take(floats, cap)) != nullptr) return NGPDE_OK;
      if (hipMalloc((void **)ptr, floats * 4) != hipSuccess) {
        (void)hipGetLastError();
        *ptr = nullptr;
        if (tape_pool_release_all() == 0 || hipMalloc((void **)ptr, floats * 4) != hipSuccess) {
          (void)hipGetLastError();
          *ptr = nullptr;
          return fail(NGPDE_ERR_HIP, "ngpde_node_vmh_create: %.1f GB of tape do not fit the device", floats * 4 / 1e9);
        }
      }
      NGPDE_HIP_CHECK(hipMemset(*ptr, 0, floats * 4));
      return NGPDE_OK;
    };
    step(tape(&p->tape_phi, tp, &p->tape_cap[0]));
    if (st == NGPDE_OK) step(tape(&p->tape_gam, tg, &p->tape_cap[1]));
    if (st == NGPDE_OK) step(tape(&p->dz_phi, tp, &p->tape_cap[2]));
    if (st == NGPDE_OK) step(tape(&p->dz_gam, tg, &p->tape_cap[3]));
    if (st == NGPDE_OK) step(alloc(&p->dsrc, 2 * E, true));
    p->tape_bytes = 2 * (tp + tg) * 4;
    p->partial_floats = (size_t)dense_weight_chunks((int64_t)(evals * E), 64, 64) * 65 * 64 + 64;
    p->partial_floats = std::max(p->partial_floats, (size_t)dense_weight_chunks((int64_t)(evals * N), 64, 64) * 65 * 64 + 64);
    if (st == NGPDE_OK) step(alloc(&p->partial, p->partial_floats, false));
    if (st == NGPDE_OK) step(alloc(&p->dwpad, 64 * 64 + 64, false));
  }
  if (st == NGPDE_OK && (hipMemcpy(p->cf, cf, sizeof cf, hipMemcpyHostToDevice) != hipSuccess ||
                         hipMemcpy(p->cb, cb, sizeof cb, hipMemcpyHostToDevice) != hipSuccess))
    st = fail(NGPDE_ERR_HIP, "ngpde_node_vmh_create: copying the coefficient tables failed");
  if (st != NGPDE_OK) {
    const std::string keep = last_error();
    node_vmh_free(p);
    (void)tape_pool_release_all();   // (a plan that did not fit: nothing of it stays parked)
    last_error() = keep;
    return st;
  }
  *out = p;
  return NGPDE_OK;
}

int32_t ngpde_node_vmh_destroy(ngpde_node_vmh_t *p) {
  NGPDE_RANGE();
  node_vmh_free(p);
  return NGPDE_OK;
}

size_t ngpde_node_vmh_tape_bytes(const ngpde_node_vmh_t *p) { return p ? p->tape_bytes : 0; }

size_t ngpde_release_cached_memory(void) {
  size_t bytes = 0;
  {
    std::lock_guard<std::mutex> lock(g_tape_mu);
    for (const auto &b : g_tape_pool) bytes += b.floats * 4;
  }
  (void)tape_pool_release_all();
  return bytes;
}

int32_t ngpde_node_vmh_fault(ngpde_node_vmh_t *p, ngpde_stream_t stream_, int32_t *fault) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr && fault != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_fault: NULL argument");
  NGPDE_HIP_CHECK(hipStreamSynchronize((hipStream_t)stream_));
  *fault = (p->persist.fault_host && *p->persist.fault_host) ? 1 : 0;
  return NGPDE_OK;
}

static void vmh_fill(const ngpde_node_vmh *p, VmhLaunch &a, const float *const *phi_w, const float *const *phi_b, const float *const *gam_w,
                     const float *const *gam_b) {
  a.g = p->g; a.ps = &p->persist; a.shape = p->shape; a.n_steps = p->n_steps; a.S = p->S; a.pos = p->pos;
  for (int l = 0; l < p->shape.n_phi; ++l) { a.phi_w[l] = phi_w[l]; a.phi_b[l] = phi_b ? phi_b[l] : nullptr; }
  for (int l = 0; l < p->shape.n_gam; ++l) { a.gam_w[l] = gam_w[l]; a.gam_b[l] = gam_b ? gam_b[l] : nullptr; }
  a.x0 = p->x; a.x1 = p->x + p->g->n_nodes; a.state = p->state; a.srcpos = p->srcpos; a.srcdeg = p->srcdeg; a.tape_phi = p->tape_phi; a.tape_gam = p->tape_gam; a.dz_phi = p->dz_phi; a.dz_gam = p->dz_gam;
  a.dsrc = p->dsrc; a.cf = p->cf; a.cb = p->cb;
}

// the number of saved states of a plan under saveat, or -1 (save_every must divide the plan's steps)
static int vmh_saved_states(const ngpde_node_vmh *p, int32_t save_every, int32_t save_start) {
  if (save_every < 1 || p->n_steps % save_every) return -1;
  return p->n_steps / save_every + (save_start ? 1 : 0);
}

int32_t ngpde_node_vmh_forward_saveat(ngpde_node_vmh_t *p, const float *u0, const float *const *phi_weight, const float *const *phi_bias,
                                      const float *const *gamma_weight, const float *const *gamma_bias, int32_t save_every,
                                      int32_t save_start, float *usave, ngpde_stream_t stream_);

int32_t ngpde_node_vmh_forward(ngpde_node_vmh_t *p, const float *u0, const float *const *phi_weight, const float *const *phi_bias,
                               const float *const *gamma_weight, const float *const *gamma_bias, float *uT, ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_forward: plan is NULL");
  // u(T) alone = the one state saved after the last step
  return ngpde_node_vmh_forward_saveat(p, u0, phi_weight, phi_bias, gamma_weight, gamma_bias, p->n_steps, 0, uT, stream_);
}

int32_t ngpde_node_vmh_forward_saveat(ngpde_node_vmh_t *p, const float *u0, const float *const *phi_weight, const float *const *phi_bias,
                                      const float *const *gamma_weight, const float *const *gamma_bias, int32_t save_every,
                                      int32_t save_start, float *uT, ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_forward: plan is NULL");
  const int T = vmh_saved_states(p, save_every, save_start);
  NGPDE_REQUIRE(T >= 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_forward_saveat: save_every = %d must be >= 1 and divide the plan's %d steps",
                (int)save_every, p->n_steps);
  if (p->g->n_nodes == 0) return NGPDE_OK;
  NGPDE_REQUIRE(u0 && uT && phi_weight && gamma_weight && u0 != uT, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_forward: NULL argument (or uT aliasing u0)");
  for (int l = 0; l < p->shape.n_phi; ++l) NGPDE_REQUIRE(phi_weight[l], NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_forward: phi weight %d is NULL", l);
  for (int l = 0; l < p->shape.n_gam; ++l) NGPDE_REQUIRE(gamma_weight[l], NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_forward: gamma weight %d is NULL", l);
  NGPDE_REQUIRE(!(p->persist.fault_host && *p->persist.fault_host), NGPDE_ERR_STATE,
                "ngpde_node_vmh_forward: an earlier launch of this plan gave up waiting for its neighbours (ngpde_node_vmh_fault); destroy the plan");
  VmhLaunch a;
  vmh_fill(p, a, phi_weight, phi_bias, gamma_weight, gamma_bias);
  a.u_in = u0;
  if (T == 1 && !save_start) {
    a.u_out = uT;
  } else {
    a.save = uT; a.save_every = save_every; a.save_off = save_start ? 1 : 0;
  }
  const int32_t st = launch_node_vmh_fwd(a, (hipStream_t)stream_);
  if (st == NGPDE_OK) p->solved = true;
  return st;
}

int32_t ngpde_node_vmh_backward_saveat(ngpde_node_vmh_t *p, const float *const *phi_weight, const float *const *gamma_weight, int32_t save_every,
                                       int32_t save_start, const float *dusave, float *du0, float *const *dphi_weight,
                                       float *const *dphi_bias, float *const *dgamma_weight, float *const *dgamma_bias, ngpde_stream_t stream_);

int32_t ngpde_node_vmh_backward(ngpde_node_vmh_t *p, const float *const *phi_weight, const float *const *gamma_weight, const float *duT,
                                float *du0, float *const *dphi_weight, float *const *dphi_bias, float *const *dgamma_weight,
                                float *const *dgamma_bias, ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_backward: plan is NULL");
  return ngpde_node_vmh_backward_saveat(p, phi_weight, gamma_weight, p->n_steps, 0, duT, du0, dphi_weight, dphi_bias, dgamma_weight, dgamma_bias,
                                        stream_);
}

int32_t ngpde_node_vmh_backward_saveat(ngpde_node_vmh_t *p, const float *const *phi_weight, const float *const *gamma_weight, int32_t save_every,
                                       int32_t save_start, const float *duT, float *du0, float *const *dphi_weight,
                                       float *const *dphi_bias, float *const *dgamma_weight, float *const *dgamma_bias, ngpde_stream_t stream_) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(p != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_backward: plan is NULL");
  const int T = vmh_saved_states(p, save_every, save_start);
  NGPDE_REQUIRE(T >= 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_backward_saveat: save_every = %d must be >= 1 and divide the plan's %d steps",
                (int)save_every, p->n_steps);
  NGPDE_REQUIRE(p->with_bwd, NGPDE_ERR_STATE, "ngpde_node_vmh_backward: the plan was created without a backward pass");
  NGPDE_REQUIRE(p->solved, NGPDE_ERR_STATE, "ngpde_node_vmh_backward: no forward solve has filled the tape");
  NGPDE_REQUIRE(phi_weight && gamma_weight && duT && du0 && dphi_weight && dgamma_weight, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_backward: NULL argument");
  NGPDE_REQUIRE(!(p->persist.fault_host && *p->persist.fault_host), NGPDE_ERR_STATE,
                "ngpde_node_vmh_backward: an earlier launch of this plan gave up waiting for its neighbours (ngpde_node_vmh_fault); destroy the plan");
  hipStream_t stream = (hipStream_t)stream_;
  const size_t N = (size_t)p->g->n_nodes, E = (size_t)p->g->n_edges, evals = (size_t)p->n_steps * p->S;
  // lambda starts as the cotangent of the last saved state (= u(T)); the kernel adds the earlier ones at their times
  const float *last = duT + (size_t)(T - 1) * N;
  if (du0 != last) NGPDE_HIP_CHECK(hipMemcpyAsync(du0, last, N * 4, hipMemcpyDeviceToDevice, stream));
  VmhLaunch a;
  vmh_fill(p, a, phi_weight, nullptr, gamma_weight, nullptr);
  a.lam = du0;
  if (T > 1) {
    NGPDE_REQUIRE(du0 < duT || du0 >= duT + (size_t)T * N, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_node_vmh_backward_saveat: du0 lies inside the cotangent array");
    a.dsave = duT; a.save_every = save_every; a.save_off = save_start ? 1 : 0;
  }
  int32_t st;
  if ((st = launch_node_vmh_bwd(a, stream))) return st;
  // dW_l = A_l^T dZ_l, db_l = column sums of dZ_l over the rows of ALL evaluations: one weight-pullback GEMM per layer on the tapes
  auto layer_grad = [&](const float *tape, const float *dz, size_t rows, int din, int dout, float *dw, float *db) -> int32_t {
    SegTable segs;
    segs.n = 1; segs.ptr[0] = tape; segs.width[0] = 64; segs.row_div[0] = 1; segs.vec[0] = 1;
    for (int k = 1; k <= 4; ++k) segs.offset[k] = 64;
    int32_t s2;
    if ((s2 = launch_dense_seg_bwd_weight((int64_t)rows, segs, 64, 64, dz, p->dwpad, p->dwpad + 64 * 64, p->partial, stream))) return s2;
    if (dw && (s2 = launch_vmh_copy_block(p->dwpad, 64, dw, dout, din, dout, stream))) return s2;
    if (db && (s2 = launch_vmh_copy_block(p->dwpad + 64 * 64, 64, db, dout, 1, dout, stream))) return s2;
    return NGPDE_OK;
  };
  for (int l = 0; l < p->shape.n_phi; ++l)
    if ((st = layer_grad(p->tape_phi + (size_t)l * evals * E * 64, p->dz_phi + (size_t)l * evals * E * 64, evals * E, p->shape.phi_dims[l],
                         p->shape.phi_dims[l + 1], dphi_weight[l], dphi_bias ? dphi_bias[l] : nullptr)))
      return st;
  for (int l = 0; l < p->shape.n_gam; ++l)
    if ((st = layer_grad(p->tape_gam + (size_t)l * evals * N * 64, p->dz_gam + (size_t)l * evals * N * 64, evals * N, p->shape.gam_dims[l],
                         p->shape.gam_dims[l + 1], dgamma_weight[l], dgamma_bias ? dgamma_bias[l] : nullptr)))
      return st;
  return NGPDE_OK;
}

}  // extern "C"
