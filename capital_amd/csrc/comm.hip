// Communicator bundle over RCCL (xGMI inside one node), one process per GPU.
//
// Replaces topo::square / topo::rect (reference src/util/topology.h:16-143, MPI_Comm_split
// bundles) and the MPI collectives of SURVEY 2c.  The 1D world communicator is what the
// CholeskyQR2 1D path (cacqr.hpp:229, policy.h:22 MPI_Allreduce over `world`) and the
// 1 x P block-column Cholesky need.  The unique id is created on rank 0 and shipped by the
// host (torch.distributed broadcast in capital_amd/topo.py - plumbing only).
#include <rccl/rccl.h>

#include <cstring>
#include <new>

#include "common.h"

struct cap_comm {
  ncclComm_t nccl;
  int rank, size;
  bool self;
  // host-staged backend (tests: several ranks sharing ONE GPU, collectives run by the host through
  // torch.distributed/gloo callbacks).  The factorization schedule is identical; only these hooks differ.
  cap_allgather_fn cb_allgather; cap_bcast_fn cb_bcast; cap_allreduce_fn cb_allreduce; void* cb_ctx;
};

#define CAP_NCCL(x)                                                                          \
  do {                                                                                       \
    ncclResult_t r_ = (x);                                                                   \
    if (r_ != ncclSuccess) {                                                                 \
      fprintf(stderr, "capital_amd: RCCL error '%s' at %s:%d\n", ncclGetErrorString(r_),     \
              __FILE__, __LINE__);                                                           \
      return CAP_ERR_COMM;                                                                   \
    }                                                                                        \
  } while (0)

extern "C" {

int cap_comm_unique_id(void* id128) {
  if (!id128) return CAP_ERR_ARG;
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  CAP_NCCL(ncclGetUniqueId(&id));
  memcpy(id128, &id, sizeof(id));
  return CAP_OK;
}

int cap_comm_create(cap_comm** comm, const void* id128, int rank, int size, void* stream) {
  (void)stream;
  if (!comm || !id128 || size < 1 || rank < 0 || rank >= size) return CAP_ERR_ARG;
  cap_comm* c = new (std::nothrow) cap_comm();
  if (!c) return CAP_ERR_ALLOC;
  c->rank = rank; c->size = size; c->self = false; c->nccl = nullptr;
  c->cb_allgather = nullptr; c->cb_bcast = nullptr; c->cb_allreduce = nullptr; c->cb_ctx = nullptr;
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r = ncclCommInitRank(&c->nccl, size, id, rank);
  if (r != ncclSuccess) {
    fprintf(stderr, "capital_amd: ncclCommInitRank failed: %s\n", ncclGetErrorString(r));
    delete c;
    return CAP_ERR_COMM;
  }
  *comm = c;
  return CAP_OK;
}

int cap_comm_create_self(cap_comm** comm) {
  if (!comm) return CAP_ERR_ARG;
  cap_comm* c = new (std::nothrow) cap_comm();
  if (!c) return CAP_ERR_ALLOC;
  c->rank = 0; c->size = 1; c->self = true; c->nccl = nullptr;
  c->cb_allgather = nullptr; c->cb_bcast = nullptr; c->cb_allreduce = nullptr; c->cb_ctx = nullptr;
  *comm = c;
  return CAP_OK;
}

int cap_comm_create_callbacks(cap_comm** comm, int rank, int size, cap_allgather_fn ag, cap_bcast_fn bc,
                              cap_allreduce_fn ar, void* ctx) {
  if (!comm || size < 1 || rank < 0 || rank >= size || !ag || !bc || !ar) return CAP_ERR_ARG;
  cap_comm* c = new (std::nothrow) cap_comm();
  if (!c) return CAP_ERR_ALLOC;
  c->rank = rank; c->size = size; c->self = false; c->nccl = nullptr;
  c->cb_allgather = ag; c->cb_bcast = bc; c->cb_allreduce = ar; c->cb_ctx = ctx;
  *comm = c;
  return CAP_OK;
}

int cap_comm_destroy(cap_comm* c) {
  if (!c) return CAP_OK;
  if (c->nccl) (void)ncclCommDestroy(c->nccl);
  delete c;
  return CAP_OK;
}

int cap_comm_rank(const cap_comm* c) { return c ? c->rank : 0; }
int cap_comm_size(const cap_comm* c) { return c ? c->size : 1; }

// MPI_Allreduce(MPI_IN_PLACE, SUM) - summa.hpp:236, cacqr/policy.h:22,82
int cap_comm_allreduce_sum(cap_comm* c, double* buf, int64_t count, void* stream) {
  if (!c || c->size == 1 || count == 0) return CAP_OK;
  if (c->cb_allreduce) return c->cb_allreduce(c->cb_ctx, buf, count, stream) ? CAP_ERR_COMM : CAP_OK;
  CAP_NCCL(ncclAllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, c->nccl, cap_stream(stream)));
  return CAP_OK;
}

// MPI_Bcast - summa.hpp:185,193; policy.h:288-289
int cap_comm_bcast(cap_comm* c, double* buf, int64_t count, int root, void* stream) {
  if (!c || c->size == 1 || count == 0) return CAP_OK;
  if (root < 0 || root >= c->size) return CAP_ERR_ARG;
  if (c->cb_bcast) return c->cb_bcast(c->cb_ctx, buf, count, root, stream) ? CAP_ERR_COMM : CAP_OK;
  CAP_NCCL(ncclBroadcast(buf, buf, (size_t)count, ncclDouble, root, c->nccl, cap_stream(stream)));
  return CAP_OK;
}

// MPI_Allgather - policy.h:176-177
int cap_comm_allgather(cap_comm* c, const double* send, double* recv, int64_t count_per_rank, void* stream) {
  if (!c || c->size == 1) {
    if (send != recv && count_per_rank > 0)
      CAP_HIP(hipMemcpyAsync(recv, send, sizeof(double) * count_per_rank, hipMemcpyDeviceToDevice, cap_stream(stream)));
    return CAP_OK;
  }
  if (c->cb_allgather) return c->cb_allgather(c->cb_ctx, send, recv, count_per_rank, stream) ? CAP_ERR_COMM : CAP_OK;
  CAP_NCCL(ncclAllGather(send, recv, (size_t)count_per_rank, ncclDouble, c->nccl, cap_stream(stream)));
  return CAP_OK;
}

int cap_comm_barrier(cap_comm* c, void* stream) {
  if (!c || c->size == 1) return CAP_OK;
  if (c->cb_allreduce) {
    static double* tok = nullptr;
    if (!tok) { CAP_HIP(hipMalloc((void**)&tok, sizeof(double))); CAP_HIP(hipMemset(tok, 0, sizeof(double))); }
    return c->cb_allreduce(c->cb_ctx, tok, 1, stream) ? CAP_ERR_COMM : CAP_OK;
  }
  static double* token = nullptr;
  if (!token) { CAP_HIP(hipMalloc((void**)&token, sizeof(double))); CAP_HIP(hipMemset(token, 0, sizeof(double))); }
  CAP_NCCL(ncclAllReduce(token, token, 1, ncclDouble, ncclSum, c->nccl, cap_stream(stream)));
  CAP_HIP(hipStreamSynchronize(cap_stream(stream)));
  return CAP_OK;
}

}  // extern "C"
