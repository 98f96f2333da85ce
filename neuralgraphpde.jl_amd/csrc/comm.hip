// comm.hip -- the data-parallel collective behind the C ABI: an all-reduce(sum, fp32) of the flat parameter-gradient vector over
// RCCL (xGMI inside a node), on the caller's stream, optionally with the fused Adam step behind it.
// [new functionality: the reference has no multi-GPU code (SURVEY.md 2.3, 5 "Distributed communication backend", 8e); the unit that
// shards is the trajectory / the graph of a batch, test/runtests.jl:89-102, src/layers.jl:359-361; the optimiser step it feeds is
// docs/src/tutorials/graph_node.md:122-129]
//
// One process per GPU: every rank creates its communicator on ITS current device from the unique id rank 0 made
// (ngpde_comm_unique_id) and the host shipped to the others (Julia: MPI.bcast / a file; Python: torch.distributed's store).
// RCCL is looked up at the first call (dlopen), so the library loads -- and every single-GPU entry works -- on a box without it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <mutex>
#include <cstring>
#include <new>

#include "common.h"

struct ngpde_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
};

namespace ngpde {
namespace {

struct RcclApi {
  ncclResult_t (*get_unique_id)(ncclUniqueId *) = nullptr;
  ncclResult_t (*comm_init_rank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*comm_destroy)(ncclComm_t) = nullptr;
  ncclResult_t (*all_reduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char *(*error_string)(ncclResult_t) = nullptr;
  ncclResult_t (*comm_count)(const ncclComm_t, int *) = nullptr;
  ncclResult_t (*comm_user_rank)(const ncclComm_t, int *) = nullptr;
  bool ok = false;
};

const RcclApi &rccl() {
  static const RcclApi api = [] {
    RcclApi a;
    void *h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return a;
    a.get_unique_id = reinterpret_cast<decltype(a.get_unique_id)>(dlsym(h, "ncclGetUniqueId"));
    a.comm_init_rank = reinterpret_cast<decltype(a.comm_init_rank)>(dlsym(h, "ncclCommInitRank"));
    a.comm_destroy = reinterpret_cast<decltype(a.comm_destroy)>(dlsym(h, "ncclCommDestroy"));
    a.all_reduce = reinterpret_cast<decltype(a.all_reduce)>(dlsym(h, "ncclAllReduce"));
    a.error_string = reinterpret_cast<decltype(a.error_string)>(dlsym(h, "ncclGetErrorString"));
    a.comm_count = reinterpret_cast<decltype(a.comm_count)>(dlsym(h, "ncclCommCount"));
    a.comm_user_rank = reinterpret_cast<decltype(a.comm_user_rank)>(dlsym(h, "ncclCommUserRank"));
    a.ok = a.get_unique_id && a.comm_init_rank && a.comm_destroy && a.all_reduce && a.error_string && a.comm_count && a.comm_user_rank;
    return a;
  }();
  return api;
}

#define NGPDE_RCCL_CHECK(expr)                                                                                           \
  do {                                                                                                                   \
    ncclResult_t _r = (expr);                                                                                            \
    if (_r != ncclSuccess) return ::ngpde::fail(NGPDE_ERR_HIP, "%s failed: %s", #expr, rccl().error_string(_r));         \
  } while (0)

}  // namespace
}  // namespace ngpde

using namespace ngpde;

extern "C" {

int32_t ngpde_comm_unique_id(void *id_out, size_t id_bytes) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(id_out != nullptr && id_bytes >= NGPDE_COMM_ID_BYTES, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_comm_unique_id: needs a buffer of %d bytes",
                NGPDE_COMM_ID_BYTES);
  static_assert(sizeof(ncclUniqueId) <= NGPDE_COMM_ID_BYTES, "unique id fits the ABI's buffer");
  NGPDE_REQUIRE(rccl().ok, NGPDE_ERR_UNSUPPORTED, "ngpde_comm_unique_id: librccl.so not found");
  ncclUniqueId id;
  NGPDE_RCCL_CHECK(rccl().get_unique_id(&id));
  std::memset(id_out, 0, NGPDE_COMM_ID_BYTES);
  std::memcpy(id_out, &id, sizeof id);
  return NGPDE_OK;
}

int32_t ngpde_comm_create(const void *unique_id, int32_t rank, int32_t world, ngpde_comm_t **out) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(out != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_comm_create: out is NULL");
  *out = nullptr;
  NGPDE_REQUIRE(unique_id != nullptr && world >= 1 && rank >= 0 && rank < world, NGPDE_ERR_INVALID_ARGUMENT,
                "ngpde_comm_create: bad arguments (rank %d of %d)", rank, world);
  NGPDE_REQUIRE(rccl().ok, NGPDE_ERR_UNSUPPORTED, "ngpde_comm_create: librccl.so not found");
  ngpde_comm *c = new (std::nothrow) ngpde_comm();
  NGPDE_REQUIRE(c != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "out of host memory");
  ncclUniqueId id;
  std::memcpy(&id, unique_id, sizeof id);
  ncclResult_t r = rccl().comm_init_rank(&c->comm, world, id, rank);   // (collective over the ranks: every rank calls it)
  if (r != ncclSuccess) {
    delete c;
    return fail(NGPDE_ERR_HIP, "ncclCommInitRank failed: %s", rccl().error_string(r));
  }
  c->rank = rank;
  c->world = world;
  *out = c;
  return NGPDE_OK;
}

int32_t ngpde_comm_destroy(ngpde_comm_t *c) {
  NGPDE_RANGE();
  if (!c) return NGPDE_OK;
  if (c->comm && rccl().ok) (void)rccl().comm_destroy(c->comm);
  delete c;
  return NGPDE_OK;
}

int32_t ngpde_comm_info(const ngpde_comm_t *c, int32_t *rank, int32_t *world) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(c != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_comm_info: communicator is NULL");
  if (rank) *rank = c->rank;
  if (world) *world = c->world;
  return NGPDE_OK;
}

int32_t ngpde_comm_rccl_info(const ngpde_comm_t *c, int32_t *count, int32_t *user_rank) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(c != nullptr && c->comm != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_comm_rccl_info: communicator is NULL");
  int n = 0, r = 0;
  NGPDE_RCCL_CHECK(rccl().comm_count(c->comm, &n));
  NGPDE_RCCL_CHECK(rccl().comm_user_rank(c->comm, &r));
  if (count) *count = n;
  if (user_rank) *user_rank = r;
  return NGPDE_OK;
}

int32_t ngpde_grad_allreduce(ngpde_comm_t *c, float *flat, int64_t count, ngpde_stream_t stream) {
  NGPDE_RANGE();
  NGPDE_REQUIRE(c != nullptr && c->comm != nullptr, NGPDE_ERR_INVALID_ARGUMENT, "ngpde_grad_allreduce: communicator is NULL");
  NGPDE_REQUIRE(count >= 0 && (flat != nullptr || count == 0), NGPDE_ERR_INVALID_ARGUMENT, "ngpde_grad_allreduce: bad buffer");
  if (count == 0) return NGPDE_OK;
  NGPDE_RCCL_CHECK(rccl().all_reduce(flat, flat, (size_t)count, ncclFloat32, ncclSum, c->comm, (hipStream_t)stream));
  return NGPDE_OK;
}

int32_t ngpde_grad_allreduce_adam(ngpde_comm_t *c, int64_t n, float *x, float *grad, float *m, float *v, float eta, float beta1,
                                  float beta2, float eps, int64_t step, ngpde_stream_t stream) {
  NGPDE_RANGE();
  int32_t st = ngpde_grad_allreduce(c, grad, n, stream);
  if (st) return st;
  return ngpde_adam_step(n, x, grad, m, v, eta, beta1, beta2, eps, step, 1.0f / (float)c->world, stream);   // mean over the ranks
}

}  // extern "C"
