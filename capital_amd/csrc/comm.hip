// Communicator bundle over RCCL (xGMI inside one node), one process per GPU.
//
// Replaces topo::square / topo::rect (reference src/util/topology.h:16-143, MPI_Comm_split
// bundles) and the MPI collectives of SURVEY 2c:
//   cap_comm       one communicator (MPI_Comm)             -> ncclComm_t
//   cap_comm_split MPI_Comm_split(color, key)              -> ncclCommSplit
//   cap_topo       the row/column/depth/slice/world bundle -> cap_comm handles + grid coordinates
// The unique id is created on rank 0 and shipped by the host (torch.distributed broadcast in
// capital_amd/topo.py - plumbing only).
//
// Three backends behind the same handle:
//   RCCL          cap_comm_create: every collective is an RCCL call, ALSO when size == 1 (so a one-GPU box
//                 exercises ncclCommInitRank / ncclBroadcast / ncclAllGather / ncclAllReduce for real);
//   self          cap_comm_create_self: size 1, no RCCL at all (copies);
//   host-staged   cap_comm_create_callbacks: the collectives are supplied by the caller (tests: several ranks
//                 sharing ONE GPU through gloo).  The callback receives the stream and must order itself
//                 behind it (and nothing else), so the schedules' event edges are really exercised.
#include <rccl/rccl.h>

#include <cmath>
#include <cstring>
#include <new>

#include "common.h"

struct cap_comm {
  ncclComm_t nccl;
  int rank, size;
  bool self;
  cap_allgather_fn cb_allgather; cap_bcast_fn cb_bcast; cap_allreduce_fn cb_allreduce; void* cb_ctx;
  cap_alltoallv_fn cb_alltoallv;    // optional (host-staged communicators): cap_comm_set_alltoallv_callback
  double* token;      // 1-double device scratch (barrier)
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

namespace {
cap_comm* new_comm(int rank, int size) {
  cap_comm* c = new (std::nothrow) cap_comm();
  if (!c) return nullptr;
  c->rank = rank; c->size = size; c->self = false; c->nccl = nullptr;
  c->cb_allgather = nullptr; c->cb_bcast = nullptr; c->cb_allreduce = nullptr; c->cb_ctx = nullptr; c->cb_alltoallv = nullptr;
  c->token = nullptr;
  return c;
}
}  // namespace

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
  cap_comm* c = new_comm(rank, size);
  if (!c) return CAP_ERR_ALLOC;
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
  cap_comm* c = new_comm(0, 1);
  if (!c) return CAP_ERR_ALLOC;
  c->self = true;
  *comm = c;
  return CAP_OK;
}

int cap_comm_create_callbacks(cap_comm** comm, int rank, int size, cap_allgather_fn ag, cap_bcast_fn bc,
                              cap_allreduce_fn ar, void* ctx) {
  if (!comm || size < 1 || rank < 0 || rank >= size || !ag || !bc || !ar) return CAP_ERR_ARG;
  cap_comm* c = new_comm(rank, size);
  if (!c) return CAP_ERR_ALLOC;
  c->cb_allgather = ag; c->cb_bcast = bc; c->cb_allreduce = ar; c->cb_ctx = ctx;
  *comm = c;
  return CAP_OK;
}

// MPI_Comm_split (topology.h:28-39, 84-126): ranks with the same color form a new communicator, ordered by key.
// Collective over `comm`.  color < 0 = this rank joins no group (*out = NULL).
int cap_comm_split(cap_comm* comm, int color, int key, cap_comm** out) {
  if (!out) return CAP_ERR_ARG;
  *out = nullptr;
  if (!comm || comm->self) {                       // a self communicator splits into itself
    if (color < 0) return CAP_OK;
    return cap_comm_create_self(out);
  }
  if (comm->cb_allgather) return CAP_ERR_UNSUPPORTED;   // host-staged groups are built by the host (capital_amd/topo.py)
  ncclComm_t sub = nullptr;
  ncclConfig_t cfg = NCCL_CONFIG_INITIALIZER;
  CAP_NCCL(ncclCommSplit(comm->nccl, color < 0 ? NCCL_SPLIT_NOCOLOR : color, key, &sub, &cfg));
  if (color < 0 || !sub) return CAP_OK;
  cap_comm* c = new_comm(0, 1);
  if (!c) { (void)ncclCommDestroy(sub); return CAP_ERR_ALLOC; }
  c->nccl = sub;
  CAP_NCCL(ncclCommUserRank(sub, &c->rank));
  CAP_NCCL(ncclCommCount(sub, &c->size));
  *out = c;
  return CAP_OK;
}

// MPI_Comm_dup (topology.h:54-59): same group, independent communicator - collectives on the copy may run
// concurrently (on another stream) with collectives on the original.
int cap_comm_dup(cap_comm* comm, cap_comm** out) {
  if (!out) return CAP_ERR_ARG;
  if (comm && comm->cb_allgather) {                 // host-staged: same callbacks (the host runs them in program order)
    cap_comm* c = new_comm(comm->rank, comm->size);
    if (!c) return CAP_ERR_ALLOC;
    c->cb_allgather = comm->cb_allgather; c->cb_bcast = comm->cb_bcast; c->cb_allreduce = comm->cb_allreduce; c->cb_ctx = comm->cb_ctx;
    c->cb_alltoallv = comm->cb_alltoallv;
    *out = c;
    return CAP_OK;
  }
  return cap_comm_split(comm, 0, comm ? comm->rank : 0, out);
}

int cap_comm_destroy(cap_comm* c) {
  if (!c) return CAP_OK;
  if (c->nccl) (void)ncclCommDestroy(c->nccl);
  if (c->token) (void)hipFree(c->token);
  delete c;
  return CAP_OK;
}

int cap_comm_rank(const cap_comm* c) { return c ? c->rank : 0; }
int cap_comm_size(const cap_comm* c) { return c ? c->size : 1; }
// 0 self / NULL, 1 RCCL, 2 host-staged
int cap_comm_backend(const cap_comm* c) { return (!c || c->self) ? 0 : (c->cb_allgather ? 2 : 1); }

// What the communication library itself reports (self-diagnosis of multi-GPU runs): ranks and this rank's index as seen by
// RCCL (ncclCommCount / ncclCommUserRank) and the device the communicator is bound to; for the other backends the stored values.
int cap_comm_query(const cap_comm* c, int* nranks, int* rank, int* device) {
  if (!nranks || !rank || !device) return CAP_ERR_ARG;
  *nranks = c ? c->size : 1; *rank = c ? c->rank : 0; *device = -1;
  (void)hipGetDevice(device);
  if (c && c->nccl) {
    CAP_NCCL(ncclCommCount(c->nccl, nranks));
    CAP_NCCL(ncclCommUserRank(c->nccl, rank));
    CAP_NCCL(ncclCommCuDevice(c->nccl, device));
  }
  return CAP_OK;
}

// MPI_Allreduce(MPI_IN_PLACE, SUM) - summa.hpp:236, cacqr/policy.h:22,82
int cap_comm_allreduce_sum(cap_comm* c, double* buf, int64_t count, void* stream) {
  if (!c || c->self || count == 0) return CAP_OK;
  if (c->cb_allreduce) return c->size == 1 ? CAP_OK : (c->cb_allreduce(c->cb_ctx, buf, count, stream) ? CAP_ERR_COMM : CAP_OK);
  CAP_NCCL(ncclAllReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, c->nccl, cap_stream(stream)));
  return CAP_OK;
}

// MPI_Reduce(SUM) to `root`, in place on the root - cacqr.hpp:98,147.  Non-root buffers are left untouched
// by the RCCL backend (recv == send is only written on the root); the host-staged backend all-reduces.
int cap_comm_reduce_sum(cap_comm* c, double* buf, int64_t count, int root, void* stream) {
  if (!c || c->self || count == 0) return CAP_OK;
  if (root < 0 || root >= c->size) return CAP_ERR_ARG;
  if (c->cb_allreduce) return c->size == 1 ? CAP_OK : (c->cb_allreduce(c->cb_ctx, buf, count, stream) ? CAP_ERR_COMM : CAP_OK);
  CAP_NCCL(ncclReduce(buf, buf, (size_t)count, ncclDouble, ncclSum, root, c->nccl, cap_stream(stream)));
  return CAP_OK;
}

// MPI_Bcast - summa.hpp:185,193; policy.h:288-289
int cap_comm_bcast(cap_comm* c, double* buf, int64_t count, int root, void* stream) {
  if (!c || c->self || count == 0) return CAP_OK;
  if (root < 0 || root >= c->size) return CAP_ERR_ARG;
  if (c->cb_bcast) return c->size == 1 ? CAP_OK : (c->cb_bcast(c->cb_ctx, buf, count, root, stream) ? CAP_ERR_COMM : CAP_OK);
  CAP_NCCL(ncclBroadcast(buf, buf, (size_t)count, ncclDouble, root, c->nccl, cap_stream(stream)));
  return CAP_OK;
}

// MPI_Allgather - policy.h:176-177
int cap_comm_allgather(cap_comm* c, const double* send, double* recv, int64_t count_per_rank, void* stream) {
  if (!c || c->self || (c->cb_allgather && c->size == 1)) {
    if (send != recv && count_per_rank > 0)
      CAP_HIP(hipMemcpyAsync(recv, send, sizeof(double) * count_per_rank, hipMemcpyDeviceToDevice, cap_stream(stream)));
    return CAP_OK;
  }
  if (count_per_rank == 0) return CAP_OK;
  if (c->cb_allgather) return c->cb_allgather(c->cb_ctx, send, recv, count_per_rank, stream) ? CAP_ERR_COMM : CAP_OK;
  CAP_NCCL(ncclAllGather(send, recv, (size_t)count_per_rank, ncclDouble, c->nccl, cap_stream(stream)));
  return CAP_OK;
}

// Personalised all-to-all (the data movement behind util::block_to_cyclic_* / cyclic_to_local, util.hpp:56-230, and the
// transpose-partner exchange util::transpose, util.hpp:232-247 - upstream builds both from MPI_Allgather / MPI_Sendrecv_replace):
// rank r receives sendcounts_q[r] doubles from every q.  counts / displacements are HOST arrays of `size` int64 (elements).
// RCCL: one group of ncclSend / ncclRecv pairs (each pair rides its own xGMI link on the fully connected node); the piece a
// rank sends to itself is a device copy.
int cap_comm_alltoallv(cap_comm* c, const double* send, const int64_t* sendcounts, const int64_t* sdispls, double* recv,
                       const int64_t* recvcounts, const int64_t* rdispls, void* stream) {
  if (!sendcounts || !sdispls || !recvcounts || !rdispls) return CAP_ERR_ARG;
  const int size = c ? c->size : 1, rank = c ? c->rank : 0;
  if (sendcounts[rank] != recvcounts[rank]) return CAP_ERR_ARG;
  if (sendcounts[rank] > 0)
    CAP_HIP(hipMemcpyAsync(recv + rdispls[rank], send + sdispls[rank], sizeof(double) * sendcounts[rank], hipMemcpyDeviceToDevice, cap_stream(stream)));
  if (!c || c->self || size == 1) return CAP_OK;
  if (c->cb_allgather) {
    if (!c->cb_alltoallv) return CAP_ERR_UNSUPPORTED;
    return c->cb_alltoallv(c->cb_ctx, send, sendcounts, sdispls, recv, recvcounts, rdispls, stream) ? CAP_ERR_COMM : CAP_OK;
  }
  CAP_NCCL(ncclGroupStart());
  for (int r = 0; r < size; r++) {
    if (r == rank) continue;
    if (sendcounts[r] > 0) CAP_NCCL(ncclSend(send + sdispls[r], (size_t)sendcounts[r], ncclDouble, r, c->nccl, cap_stream(stream)));
    if (recvcounts[r] > 0) CAP_NCCL(ncclRecv(recv + rdispls[r], (size_t)recvcounts[r], ncclDouble, r, c->nccl, cap_stream(stream)));
  }
  CAP_NCCL(ncclGroupEnd());
  return CAP_OK;
}

// host-staged communicators only: the caller's implementation of cap_comm_alltoallv (the piece a rank sends to itself has
// already been copied when the callback runs; it must skip it)
int cap_comm_set_alltoallv_callback(cap_comm* c, cap_alltoallv_fn fn) {
  if (!c || !c->cb_allgather) return CAP_ERR_ARG;
  c->cb_alltoallv = fn;
  return CAP_OK;
}

// MPI_Sendrecv_replace with one partner (util::transpose, util.hpp:232-247): `count` doubles of `buf` are swapped with the
// same range on rank `partner` (partner == my rank: nothing to do).  `tmp`: device scratch of `count` doubles.
int cap_comm_exchange(cap_comm* c, double* buf, double* tmp, int64_t count, int partner, void* stream) {
  const int size = c ? c->size : 1, rank = c ? c->rank : 0;
  if (partner < 0 || partner >= size || count < 0) return CAP_ERR_ARG;
  if (partner == rank || count == 0) return CAP_OK;
  if (!tmp) return CAP_ERR_ARG;
  if (c->cb_allgather) {
    if (!c->cb_alltoallv) return CAP_ERR_UNSUPPORTED;
    int64_t sc[64], sd[64], rc[64], rd[64];
    if (size > 64) return CAP_ERR_UNSUPPORTED;
    for (int r = 0; r < size; r++) { sc[r] = rc[r] = (r == partner) ? count : 0; sd[r] = rd[r] = 0; }
    if (c->cb_alltoallv(c->cb_ctx, buf, sc, sd, tmp, rc, rd, stream)) return CAP_ERR_COMM;
  } else {
    CAP_NCCL(ncclGroupStart());
    CAP_NCCL(ncclSend(buf, (size_t)count, ncclDouble, partner, c->nccl, cap_stream(stream)));
    CAP_NCCL(ncclRecv(tmp, (size_t)count, ncclDouble, partner, c->nccl, cap_stream(stream)));
    CAP_NCCL(ncclGroupEnd());
  }
  CAP_HIP(hipMemcpyAsync(buf, tmp, sizeof(double) * count, hipMemcpyDeviceToDevice, cap_stream(stream)));
  return CAP_OK;
}

// MPI_Barrier (bench/cholesky/cholinv.cpp:47): a 1-element all-reduce, then the stream is drained
int cap_comm_barrier(cap_comm* c, void* stream) {
  if (!c || c->self) { CAP_HIP(hipStreamSynchronize(cap_stream(stream))); return CAP_OK; }
  if (!c->token) { CAP_HIP(hipMalloc((void**)&c->token, sizeof(double))); CAP_HIP(hipMemset(c->token, 0, sizeof(double))); }
  CAP_TRY(cap_comm_allreduce_sum(c, c->token, 1, stream));
  CAP_HIP(hipStreamSynchronize(cap_stream(stream)));
  return CAP_OK;
}

// -------------------------------------------------------------------------------------------------
// Grid bundles (topology.h:16-143)
// -------------------------------------------------------------------------------------------------
// rank -> grid coordinates, pure (no communicator).  kind 0: topo::square layout 0 (topology.h:75-83):
// d = ceil(sqrt(size / c)), z = rank % c, y = rank / (d c), x = (rank % (d c)) / c.  kind 1: topo::rect
// (topology.h:44-50): d = size / c^2, z = rank % c, y = rank / c^2, x = (rank % c^2) / c.
int cap_topo_coords(int kind, int rank, int size, int c, int* d, int* x, int* y, int* z) {
  if (c < 1 || size < 1 || rank < 0 || rank >= size || !d || !x || !y || !z) return CAP_ERR_ARG;
  if (kind == 0) {
    int dd = (int)std::nearbyint(std::ceil(std::sqrt((double)size / (double)c) - 1e-12));
    if (dd < 1) dd = 1;
    if (c * dd * dd != size) return CAP_ERR_ARG;
    *d = dd; *z = rank % c; *y = rank / (dd * c); *x = (rank % (dd * c)) / c;
    return CAP_OK;
  }
  if (kind == 1) {
    if (size % (c * c)) return CAP_ERR_ARG;
    *d = size / (c * c); *z = rank % c; *y = rank / (c * c); *x = (rank % (c * c)) / c;
    return CAP_OK;
  }
  return CAP_ERR_ARG;
}

struct cap_topo {
  int kind, rank, size, c, d, x, y, z, layout, num_chunks;
  cap_comm* world;                 // not owned
  cap_comm *row, *column, *depth, *slice;           // square; rect: row, depth, slice
  cap_comm *column_contig, *column_alt, *cube;      // rect only
  bool owns;
};

// topo::square(comm, c, layout, num_chunks) / topo::rect(comm, c, layout, num_chunks).  Collective over `world`.
// Only layout 0 is accepted for squares (layouts 1-2 are numerically wrong upstream).
int cap_topo_create(cap_topo** topo, int kind, cap_comm* world, int c, int layout, int num_chunks) {
  if (!topo || c < 1) return CAP_ERR_ARG;
  if (kind == 0 && layout != 0) return CAP_ERR_UNSUPPORTED;
  cap_topo* t = new (std::nothrow) cap_topo();
  if (!t) return CAP_ERR_ALLOC;
  memset(t, 0, sizeof(*t));
  t->kind = kind; t->world = world; t->rank = cap_comm_rank(world); t->size = cap_comm_size(world);
  t->c = c; t->layout = layout; t->num_chunks = num_chunks; t->owns = true;
  int st = cap_topo_coords(kind, t->rank, t->size, c, &t->d, &t->x, &t->y, &t->z);
  if (st != CAP_OK) { delete t; return st; }
  const int r = t->rank;
  if (kind == 0) {
    // topology.h:84-94: depth = ranks sharing (x, y); slice = ranks sharing z; row / column split the slice
    st = cap_comm_split(world, r / c, r, &t->depth);
    if (st == CAP_OK) st = cap_comm_split(world, t->z, r, &t->slice);
    if (st == CAP_OK) st = cap_comm_split(t->slice, t->y, t->x, &t->row);
    if (st == CAP_OK) st = cap_comm_split(t->slice, t->x, t->y, &t->column);
  } else {
    // topology.h:24-39
    const int cube_sz = c * c * c, slice_sz = c * c;
    st = cap_comm_split(world, r / cube_sz, r, &t->cube);
    const int cube_rank = t->cube ? cap_comm_rank(t->cube) : 0;
    if (st == CAP_OK) st = cap_comm_split(t->cube, cube_rank / c, cube_rank, &t->depth);
    if (st == CAP_OK) st = cap_comm_split(t->cube, (cube_rank % c) + c * (cube_rank / slice_sz), cube_rank, &t->row);
    cap_comm* column = nullptr;
    if (st == CAP_OK) st = cap_comm_split(world, r % slice_sz, r, &column);
    if (st == CAP_OK) st = cap_comm_split(world, r % c, r, &t->slice);
    const int col_rank = column ? cap_comm_rank(column) : 0;
    if (st == CAP_OK) st = cap_comm_split(column, col_rank / c, col_rank, &t->column_contig);
    if (st == CAP_OK) st = cap_comm_split(column, col_rank % c, col_rank, &t->column_alt);
    if (column) cap_comm_destroy(column);
  }
  if (st != CAP_OK) { cap_topo_destroy(t); return st; }
  *topo = t;
  return CAP_OK;
}

// Bundle assembled from communicators the caller built (host-staged test groups).  which: see cap_topo_comm.
int cap_topo_create_from(cap_topo** topo, int kind, cap_comm* world, int c, int layout, int num_chunks, cap_comm** comms, int ncomms) {
  if (!topo || !comms || ncomms < 7) return CAP_ERR_ARG;
  cap_topo* t = new (std::nothrow) cap_topo();
  if (!t) return CAP_ERR_ALLOC;
  memset(t, 0, sizeof(*t));
  t->kind = kind; t->world = world; t->rank = cap_comm_rank(world); t->size = cap_comm_size(world);
  t->c = c; t->layout = layout; t->num_chunks = num_chunks; t->owns = false;
  int st = cap_topo_coords(kind, t->rank, t->size, c, &t->d, &t->x, &t->y, &t->z);
  if (st != CAP_OK) { delete t; return st; }
  t->row = comms[0]; t->column = comms[1]; t->depth = comms[2]; t->slice = comms[3];
  t->column_contig = comms[4]; t->column_alt = comms[5]; t->cube = comms[6];
  *topo = t;
  return CAP_OK;
}

int cap_topo_destroy(cap_topo* t) {
  if (!t) return CAP_OK;
  if (t->owns)
    for (cap_comm* c : {t->row, t->column, t->depth, t->slice, t->column_contig, t->column_alt, t->cube}) if (c) cap_comm_destroy(c);
  delete t;
  return CAP_OK;
}

// which: 0 world, 1 row, 2 column, 3 depth, 4 slice, 5 column_contig, 6 column_alt, 7 cube
cap_comm* cap_topo_comm(cap_topo* t, int which) {
  if (!t) return nullptr;
  switch (which) {
    case 0: return t->world; case 1: return t->row; case 2: return t->column; case 3: return t->depth;
    case 4: return t->slice; case 5: return t->column_contig; case 6: return t->column_alt; case 7: return t->cube;
  }
  return nullptr;
}

// field: 0 rank, 1 size, 2 c, 3 d, 4 x, 5 y, 6 z, 7 layout, 8 num_chunks, 9 kind
int cap_topo_get(const cap_topo* t, int field) {
  if (!t) return -1;
  const int v[10] = {t->rank, t->size, t->c, t->d, t->x, t->y, t->z, t->layout, t->num_chunks, t->kind};
  return (field >= 0 && field < 10) ? v[field] : -1;
}

}  // extern "C"
