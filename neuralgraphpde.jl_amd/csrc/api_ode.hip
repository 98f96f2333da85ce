// api_ode.hip -- ONE entry that builds the device-resident fixed-step neural-ODE plan for a right-hand side, and one pair that runs it.
// What a Lux / DiffEqFlux host calls where the tutorials write  NeuralODE(model, tspan, Tsit5(); saveat)  and  solve + adjoint:
//   /root/reference/docs/src/tutorials/graph_node.md:44-66, :78   (Chain(GCNConv, GCNConv))
//   /root/reference/docs/src/tutorials/VMH.md:85-89, :104-108      (VMHConv(phi, gamma), saveat)
//   BASELINE config 3                                              (one GAT-style layer; src/NeuralGraphPDE.jl:7)
// The choice of plan -- which of the three solvers takes the right-hand side, whether a batch of identical structures runs on the member's
// handle, what the shapes must satisfy -- was host code (node.py) until round 6; a Julia host would have had to write it again.  The
// plans themselves are ngpde_node_gcn2_* / ngpde_node_gat_* / ngpde_node_vmh_* (node.hip, gat_fused.hip, node_vmh.hip), unchanged.
#include <cstring>
#include <new>

#include "common.h"

using namespace ngpde;

struct ngpde_ode {
  int32_t rhs = 0;
  ngpde_node_t *gcn = nullptr;
  ngpde_node_gat_t *gat = nullptr;
  ngpde_node_vmh_t *vmh = nullptr;
  ngpde_ode_desc_t desc;
  int32_t flags = 0;
};

namespace {

int32_t check_mlp_chain(const char *name, int32_t n, const int32_t *dims, const int32_t *acts, int32_t first, int32_t last) {
  NGPDE_REQUIRE(n >= 1 && n <= NGPDE_MLP_MAX_LAYERS, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_ode_create: %s has %d layers (1 .. %d)", name, n, NGPDE_MLP_MAX_LAYERS);
  for (int l = 0; l <= n; ++l)
    NGPDE_REQUIRE(dims[l] > 0, NGPDE_ERR_DIMENSION_MISMATCH, "DimensionMismatch: NeuralODE(VMHConv): %s.dims[%d] = %d", name, l, dims[l]);
  for (int l = 0; l < n; ++l)
    NGPDE_REQUIRE(acts[l] >= NGPDE_ACT_IDENTITY && acts[l] <= NGPDE_ACT_SOFTPLUS, NGPDE_ERR_INVALID_ARGUMENT,
                  "ngpde_ode_create: %s.layer_%d: unknown activation %d", name, l + 1, acts[l]);
  // the plan's entries take pointers only: a parameter tree that does not chain would make the kernels read past the weight arrays --
  // the reference fails in the matrix product with a DimensionMismatch (src/layers.jl:316, :328)
  NGPDE_REQUIRE(dims[0] == first, NGPDE_ERR_DIMENSION_MISMATCH, "DimensionMismatch: NeuralODE(VMHConv): %s.layer_1 takes %d inputs, the layer feeds it %d",
                name, dims[0], first);
  NGPDE_REQUIRE(last < 0 || dims[n] == last, NGPDE_ERR_DIMENSION_MISMATCH, "DimensionMismatch: NeuralODE(VMHConv): %s returns %d rows, the state has %d", name,
                dims[n], last);
  return NGPDE_OK;
}

}  // namespace

extern "C" {

int32_t ngpde_ode_create(const ngpde_graph_t *g, const ngpde_ode_desc_t *d, ngpde_ode_t **out, int32_t *flags) {
  NGPDE_REQUIRE(g != nullptr && d != nullptr && out != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_ode_create: NULL argument");
  *out = nullptr;
  if (flags) *flags = 0;
  NGPDE_REQUIRE(d->tableau == NGPDE_TABLEAU_EULER || d->tableau == NGPDE_TABLEAU_TSIT5, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_ode_create: unknown tableau %d",
                d->tableau);
  NGPDE_REQUIRE(d->n_steps >= 1 && d->members >= 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_ode_create: n_steps = %d, members = %d", d->n_steps, d->members);
  ngpde_ode *o = new (std::nothrow) ngpde_ode();
  NGPDE_REQUIRE(o != nullptr, NGPDE_ERR_HIP, "ngpde_ode_create: out of host memory");
  o->rhs = d->rhs;
  o->desc = *d;
  int32_t st = NGPDE_OK, fl = 0;
  switch (d->rhs) {
    case NGPDE_RHS_GCN2: {
      if (d->members > 1) st = ngpde_node_gcn2_create_batch(g, d->members, d->width, d->act, d->tableau, d->n_steps, (float)d->dt, d->with_backward, &o->gcn);
      else st = ngpde_node_gcn2_create(g, d->width, d->act, d->tableau, d->n_steps, (float)d->dt, d->with_backward, &o->gcn);
      if (st == NGPDE_OK) (void)ngpde_node_flags(o->gcn, &fl);
      break;
    }
    case NGPDE_RHS_GAT: {
      // (ngpde_node_gat_supported compares the two directions' schedules on the device and synchronises: once per create)
      if (d->width != 64 || d->heads * d->head_width != 64 || ngpde_node_gat_supported(g, 64, d->heads, d->head_width) != 1) {
        st = fail(NGPDE_ERR_UNSUPPORTED, "ngpde_ode_create: no device-resident plan for this GAT right-hand side (64 => heads x c = 64, tiles within "
                                         "the LDS halo): step the layer with ngpde_gat_layer_* and ngpde_rk_stage_combine");
        break;
      }
      if (d->members > 1)
        st = ngpde_node_gat_create_batch(g, d->members, d->heads, d->head_width, d->negative_slope, d->act, d->tableau, d->n_steps, d->dt, d->with_backward, &o->gat);
      else st = ngpde_node_gat_create(g, d->heads, d->head_width, d->negative_slope, d->act, d->tableau, d->n_steps, d->dt, d->with_backward, &o->gat);
      fl = NGPDE_NODE_PERSISTENT_FWD | NGPDE_NODE_PERSISTENT_BWD;
      break;
    }
    case NGPDE_RHS_VMH: {
      NGPDE_REQUIRE(d->members == 1, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_ode_create: a batch of point clouds is ONE block-diagonal graph to the VMH plan (members = 1)");
      const int hd = d->width;
      if ((st = check_mlp_chain("phi", d->n_phi, d->phi_dims, d->phi_acts, 2 * hd + d->pos_width, -1))) break;
      if ((st = check_mlp_chain("gamma", d->n_gamma, d->gamma_dims, d->gamma_acts, hd + d->phi_dims[d->n_phi], hd))) break;
      if (d->pos == nullptr || ngpde_node_vmh_supported(g, hd, d->pos_width, d->n_phi, d->phi_dims, d->phi_acts, d->n_gamma, d->gamma_dims, d->gamma_acts, d->aggr) != 1) {
        st = fail(NGPDE_ERR_UNSUPPORTED, "ngpde_ode_create: no device-resident plan for this VMHConv right-hand side (scalar state, 1 - 3 coordinates, MLPs of "
                                         "2 - 4 Dense layers up to 64 wide, + / mean, tiles within the LDS halo): step the layer with ngpde_edge_layer_* and "
                                         "ngpde_rk_stage_combine");
        break;
      }
      st = ngpde_node_vmh_create(g, hd, d->pos_width, d->pos, d->n_phi, d->phi_dims, d->phi_acts, d->n_gamma, d->gamma_dims, d->gamma_acts, d->aggr, d->tableau,
                                 d->n_steps, d->dt, d->with_backward, &o->vmh);
      fl = NGPDE_NODE_PERSISTENT_FWD | NGPDE_NODE_PERSISTENT_BWD;
      break;
    }
    default: st = fail(NGPDE_ERR_INVALID_ARGUMENT, "ngpde_ode_create: unknown right-hand side %d", d->rhs);
  }
  if (st != NGPDE_OK) {
    delete o;
    return st;
  }
  o->desc.pos = nullptr;   // (copied by the plan: the caller's array need not outlive the call)
  o->flags = fl;
  if (flags) *flags = fl;
  *out = o;
  return NGPDE_OK;
}

int32_t ngpde_ode_destroy(ngpde_ode_t *o) {
  if (!o) return NGPDE_OK;
  int32_t st = NGPDE_OK;
  if (o->gcn) st = ngpde_node_destroy(o->gcn);
  if (o->gat) st = ngpde_node_gat_destroy(o->gat);
  if (o->vmh) st = ngpde_node_vmh_destroy(o->vmh);
  delete o;
  return st;
}

size_t ngpde_ode_tape_bytes(const ngpde_ode_t *o) {
  if (!o) return 0;
  if (o->gcn) return ngpde_node_tape_bytes(o->gcn);
  if (o->gat) return ngpde_node_gat_tape_bytes(o->gat);
  return o->vmh ? ngpde_node_vmh_tape_bytes(o->vmh) : 0;
}

int32_t ngpde_ode_fault(ngpde_ode_t *o, ngpde_stream_t stream, int32_t *fault) {
  NGPDE_REQUIRE(o != nullptr && fault != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_ode_fault: NULL argument");
  if (o->gcn) return ngpde_node_fault(o->gcn, stream, fault);
  if (o->gat) return ngpde_node_gat_fault(o->gat, stream, fault);
  return ngpde_node_vmh_fault(o->vmh, stream, fault);
}

int32_t ngpde_ode_forward(ngpde_ode_t *o, const float *u0, const ngpde_ode_params_t *p, int32_t save_every, int32_t save_start, float *out,
                          ngpde_stream_t stream) {
  NGPDE_REQUIRE(o != nullptr && u0 != nullptr && p != nullptr && out != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_ode_forward: NULL argument");
  NGPDE_REQUIRE(save_every == 0 || o->rhs == NGPDE_RHS_VMH, NGPDE_ERR_UNSUPPORTED,
                "ngpde_ode_forward: saveat is served by the VMH plan only (the other right-hand sides return u(T)): step them with ngpde_rk_stage_combine");
  switch (o->rhs) {
    case NGPDE_RHS_GCN2:
      NGPDE_REQUIRE(p->first.weight[0] && p->first.weight[1], NGPDE_ERR_INVALID_ARGUMENT, "ngpde_ode_forward: layer_1 / layer_2 weight is NULL");
      return ngpde_node_gcn2_forward(o->gcn, u0, p->first.weight[0], p->first.bias[0], p->first.weight[1], p->first.bias[1], out, stream);
    case NGPDE_RHS_GAT:
      NGPDE_REQUIRE(p->first.weight[0] && p->attention, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_ode_forward: weight or attention vector is NULL");
      return ngpde_node_gat_forward(o->gat, u0, p->first.weight[0], p->attention, p->first.bias[0], out, stream);
    default:
      if (save_every > 0)
        return ngpde_node_vmh_forward_saveat(o->vmh, u0, p->first.weight, p->first.bias, p->second.weight, p->second.bias, save_every, save_start, out, stream);
      return ngpde_node_vmh_forward(o->vmh, u0, p->first.weight, p->first.bias, p->second.weight, p->second.bias, out, stream);
  }
}

int32_t ngpde_ode_backward(ngpde_ode_t *o, const ngpde_ode_params_t *p, int32_t save_every, int32_t save_start, const float *dout, float *du0,
                           const ngpde_ode_grads_t *gr, ngpde_stream_t stream) {
  NGPDE_REQUIRE(o != nullptr && p != nullptr && dout != nullptr && du0 != nullptr && gr != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_ode_backward: NULL argument");
  NGPDE_REQUIRE(save_every == 0 || o->rhs == NGPDE_RHS_VMH, NGPDE_ERR_UNSUPPORTED, "ngpde_ode_backward: saveat is served by the VMH plan only");
  switch (o->rhs) {
    case NGPDE_RHS_GCN2:
      return ngpde_node_gcn2_backward(o->gcn, dout, du0, gr->first.dweight[0], gr->first.dbias[0], gr->first.dweight[1], gr->first.dbias[1], stream);
    case NGPDE_RHS_GAT:
      return ngpde_node_gat_backward(o->gat, p->first.weight[0], p->attention, dout, du0, gr->first.dweight[0], gr->dattention, gr->first.dbias[0], stream);
    default:
      if (save_every > 0)
        return ngpde_node_vmh_backward_saveat(o->vmh, p->first.weight, p->second.weight, save_every, save_start, dout, du0, gr->first.dweight, gr->first.dbias,
                                              gr->second.dweight, gr->second.dbias, stream);
      return ngpde_node_vmh_backward(o->vmh, p->first.weight, p->second.weight, dout, du0, gr->first.dweight, gr->first.dbias, gr->second.dweight,
                                     gr->second.dbias, stream);
  }
}

}  // extern "C"
