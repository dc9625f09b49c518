// Multi-GPU blocked Cholesky (upper, A = R^T R), one process per GPU, RCCL over xGMI.
//
// Layout: 1 x P block-column-cyclic (the 2D block-cyclic descriptor with Pr = 1): global block
// column J (nb wide) lives on rank J % P as local block J / P; every rank stores all n rows of its
// columns (column-major, ld = n).  The reference distributes element-cyclically over a d x d x c grid
// and moves operands with MPI_Bcast / MPI_Allreduce / MPI_Allgather (summa.hpp:163-253,
// policy.h:160-305); here the same roles are played by
//   * one small ncclBroadcast per step:  msg(k+1) = [ R(k,k+1) | Dinv(k+1) ]  (2 nb^2 doubles),
//   * one ncclAllGather per step:        the solved block row k (nb x (n - (k+1) nb) in total).
//
// Per step k on every rank, three HIP streams + events (no host synchronisation):
//   panel stream (high priority)  solve my part of block row k with Dinv(k)           (1 GEMM)
//                                 owner(k+1): update + factor + invert diag block k+1 (cholinv recursion)
//                                 after msg(k+1): apply step k to my part of block row k+1 (look-ahead)
//   comm stream                   all-gather block row k; broadcast msg(k+1)
//   main stream                   rank-nb update of my columns, rows >= (k+2) nb, upper staircase only,
//                                 A operand read straight out of the gathered pieces (no repacking)
// so the bulk update of step k overlaps with the factorization / communication of step k+1.
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"

int cap_rec_cholinv_full(double* R, int64_t ldr, double* Ri, int64_t ldi, int64_t n, double* W, int64_t wcap, int* info,
                         hipStream_t s, int64_t info_base = 0);
int64_t cap_rec_work_size(int64_t n);

struct cap_dist_plan {
  int64_t n, nb, nblk;
  int P, p;
  cap_comm* comm;
  int64_t nloc_blocks, lc;       // local blocks / columns
  int64_t nmax0;                 // widest gathered piece (columns)
  double* R;                     // n x lc
  double* S[2];                  // my solved block row (nb x cols, ld = nb)
  double* G[2];                  // gathered block row: P pieces
  double* msg[2];                // [R(k-1,k) | Dinv(k)], 2 nb^2
  double* W; int64_t wcap;       // recursion scratch
  int* info_dev; double* info_red;
  hipStream_t s_panel, s_comm;
  std::vector<hipEvent_t> ev_msg, ev_solved, ev_gather, ev_update, ev_fact;
  hipEvent_t ev_init, ev_join_p, ev_join_c;
};

namespace {
__host__ __device__ inline int64_t lbfirst(int64_t r, int64_t k, int64_t P) { return k >= r ? (k - r) / P + 1 : 0; }  // blocks J <= k owned by r
__host__ __device__ inline int64_t nblocks_of(int64_t r, int64_t nblk, int64_t P) { return r < nblk ? (nblk - 1 - r) / P + 1 : 0; }

__global__ void fill_symmetric_bc_kernel(double* out, int64_t ld, int64_t n, int64_t nb, int P, int p, int dom) {
  // same closed form as fill_symmetric_kernel (aux.hip), local column -> global column through the block-cyclic map
  int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t lcol = blockIdx.y + (int64_t)blockIdx.z * 65535;
  int64_t nblk = (n + nb - 1) / nb;
  int64_t nl = nblocks_of(p, nblk, P);
  if (row >= n || lcol >= nl * nb) return;
  int64_t gcol = ((lcol / nb) * P + p) * nb + lcol % nb;
  double v = 0.0;
  if (gcol < n) {
    int64_t hi = gcol > row ? gcol : row, lo = gcol > row ? row : gcol;
    uint64_t seed = (uint64_t)(hi + n * lo);
    uint64_t x0 = ((seed & 0xFFFFFFFFull) << 16) | 0x330Eull;
    uint64_t x1 = (0x5DEECE66Dull * x0 + 0xBull) & ((1ull << 48) - 1);
    v = (double)x1 * (1.0 / 281474976710656.0);
    if (dom && gcol == row) v += (double)n;
  }
  out[row + lcol * ld] = v;
}

__global__ void info_to_double(const int* info, double* out) { *out = (double)*info; }

int ensure_events(cap_dist_plan* d) {
  if (!d->ev_msg.empty()) return CAP_OK;
  auto mk = [&](std::vector<hipEvent_t>& v, size_t cnt) -> int {
    v.resize(cnt);
    for (auto& e : v) CAP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return CAP_OK;
  };
  size_t cnt = (size_t)d->nblk + 2;
  CAP_TRY(mk(d->ev_msg, cnt)); CAP_TRY(mk(d->ev_solved, cnt)); CAP_TRY(mk(d->ev_gather, cnt));
  CAP_TRY(mk(d->ev_update, cnt)); CAP_TRY(mk(d->ev_fact, cnt));
  CAP_HIP(hipEventCreateWithFlags(&d->ev_init, hipEventDisableTiming));
  CAP_HIP(hipEventCreateWithFlags(&d->ev_join_p, hipEventDisableTiming));
  CAP_HIP(hipEventCreateWithFlags(&d->ev_join_c, hipEventDisableTiming));
  int lo = 0, hi = 0;
  CAP_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CAP_HIP(hipStreamCreateWithPriority(&d->s_panel, hipStreamNonBlocking, hi));
  CAP_HIP(hipStreamCreateWithPriority(&d->s_comm, hipStreamNonBlocking, hi));
  return CAP_OK;
}
}  // namespace

extern "C" {

int cap_bc_owner(int64_t J, int P) { return (int)(J % P); }
int64_t cap_bc_local_block(int64_t J, int P) { return J / P; }
int64_t cap_bc_num_local_cols(int64_t n, int64_t nb, int P, int p) {
  int64_t nblk = (n + nb - 1) / nb;
  return nblocks_of(p, nblk, P) * nb;
}

int cap_fill_symmetric_bc(double* local, int64_t ld, int64_t n, int64_t nb, int P, int p, int diagonally_dominant, void* stream) {
  if (!local || n <= 0 || nb <= 0 || P < 1 || p < 0 || p >= P || ld < n) return CAP_ERR_ARG;
  int64_t lc = cap_bc_num_local_cols(n, nb, P, p);
  if (lc == 0) return CAP_OK;
  dim3 grid((unsigned)cap_ceil_div(n, 256), (unsigned)std::min<int64_t>(lc, 65535), (unsigned)cap_ceil_div(lc, 65535));
  hipLaunchKernelGGL(fill_symmetric_bc_kernel, grid, dim3(256), 0, cap_stream(stream), local, ld, n, nb, P, p, diagonally_dominant);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

int cap_dist_plan_create(cap_dist_plan** plan, int64_t n, int64_t nb, cap_comm* comm) {
  if (!plan || n <= 0) return CAP_ERR_ARG;
  if (nb <= 0) nb = 512;
  if (nb % 128 || n % nb) return CAP_ERR_UNSUPPORTED;      // block width: multiple of the 128 MFMA tile, N divisible
  cap_dist_plan* d = new (std::nothrow) cap_dist_plan();
  if (!d) return CAP_ERR_ALLOC;
  d->n = n; d->nb = nb; d->nblk = n / nb; d->comm = comm;
  d->P = cap_comm_size(comm); d->p = cap_comm_rank(comm);
  if (d->P > 8) { delete d; return CAP_ERR_UNSUPPORTED; }
  d->nloc_blocks = nblocks_of(d->p, d->nblk, d->P); d->lc = d->nloc_blocks * nb;
  d->nmax0 = nblocks_of(0, d->nblk, d->P) * nb;            // rank 0 owns the most blocks
  d->R = nullptr; d->W = nullptr; d->info_dev = nullptr; d->info_red = nullptr;
  for (int i = 0; i < 2; i++) { d->S[i] = d->G[i] = d->msg[i] = nullptr; }
  d->s_panel = d->s_comm = nullptr;
  d->wcap = cap_rec_work_size(nb);
  hipError_t e = hipMalloc((void**)&d->R, sizeof(double) * std::max<int64_t>(n * d->lc, 2));
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc((void**)&d->S[i], sizeof(double) * nb * d->nmax0);
    if (e == hipSuccess) e = hipMalloc((void**)&d->G[i], sizeof(double) * nb * d->nmax0 * d->P);
    if (e == hipSuccess) e = hipMalloc((void**)&d->msg[i], sizeof(double) * 2 * nb * nb);
    if (e == hipSuccess) e = hipMemset(d->msg[i], 0, sizeof(double) * 2 * nb * nb);
    if (e == hipSuccess) e = hipMemset(d->S[i], 0, sizeof(double) * nb * d->nmax0);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&d->W, sizeof(double) * d->wcap);
  if (e == hipSuccess) e = hipMalloc((void**)&d->info_dev, sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&d->info_red, sizeof(double));
  if (e != hipSuccess) { cap_dist_plan_destroy(d); return CAP_ERR_ALLOC; }
  *plan = d;
  return CAP_OK;
}

int cap_dist_plan_destroy(cap_dist_plan* d) {
  if (!d) return CAP_OK;
  if (d->R) (void)hipFree(d->R);
  for (int i = 0; i < 2; i++) { if (d->S[i]) (void)hipFree(d->S[i]); if (d->G[i]) (void)hipFree(d->G[i]); if (d->msg[i]) (void)hipFree(d->msg[i]); }
  if (d->W) (void)hipFree(d->W);
  if (d->info_dev) (void)hipFree(d->info_dev);
  if (d->info_red) (void)hipFree(d->info_red);
  if (!d->ev_msg.empty()) {
    for (auto* v : {&d->ev_msg, &d->ev_solved, &d->ev_gather, &d->ev_update, &d->ev_fact}) for (auto e : *v) (void)hipEventDestroy(e);
    (void)hipEventDestroy(d->ev_init); (void)hipEventDestroy(d->ev_join_p); (void)hipEventDestroy(d->ev_join_c);
    (void)hipStreamDestroy(d->s_panel); (void)hipStreamDestroy(d->s_comm);
  }
  delete d;
  return CAP_OK;
}

int64_t cap_dist_local_cols(const cap_dist_plan* d) { return d ? d->lc : 0; }
double* cap_dist_R_ptr(cap_dist_plan* d, int64_t* ld) { if (!d) return nullptr; if (ld) *ld = d->n; return d->R; }

int cap_dist_factor(cap_dist_plan* d, const double* Aloc, int64_t lda, void* stream) {
  if (!d || (d->lc > 0 && (!Aloc || lda < d->n))) return CAP_ERR_ARG;
  CAP_TRY(ensure_events(d));
  hipStream_t s0 = cap_stream(stream), s1 = d->s_panel, sc = d->s_comm;
  const int64_t n = d->n, nb = d->nb, nblk = d->nblk, P = d->P, p = d->p, ld = n;
  const int64_t nb2 = nb * nb;
  CAP_HIP(hipMemsetAsync(d->info_dev, 0, sizeof(int), s0));
  if (d->lc > 0) CAP_TRY(cap_copy_rect(Aloc, lda, d->R, ld, n, d->lc, s0));
  CAP_HIP(hipEventRecord(d->ev_init, s0));
  CAP_HIP(hipStreamWaitEvent(s1, d->ev_init, 0));
  CAP_HIP(hipStreamWaitEvent(sc, d->ev_init, 0));

  // diagonal block 0 on its owner, then msg(0) = [ - | Dinv(0) ]
  if (p == 0) {
    CAP_TRY(cap_rec_cholinv_full(d->R, ld, d->msg[0] + nb2, nb, nb, d->W, d->wcap, d->info_dev, s1, 0));
    CAP_HIP(hipEventRecord(d->ev_fact[0], s1));
    CAP_HIP(hipStreamWaitEvent(sc, d->ev_fact[0], 0));
  }
  CAP_TRY(cap_comm_bcast(d->comm, d->msg[0], 2 * nb2, 0, (void*)sc));
  CAP_HIP(hipEventRecord(d->ev_msg[0], sc));

  for (int64_t k = 0; k < nblk; k++) {
    const int par = (int)(k & 1);
    const int64_t lb0 = lbfirst(p, k, P);                       // my first local block with J > k
    const int64_t nloc_k = (d->nloc_blocks - lb0) * nb;          // my columns right of block column k
    int64_t nmax_k = 0;
    for (int64_t r = 0; r < P; r++) nmax_k = std::max(nmax_k, (nblocks_of(r, nblk, P) - lbfirst(r, k, P)) * nb);
    const int64_t piece = nb * nmax_k;
    double* Dinv = d->msg[par] + nb2;

    // ---- panel: solve my part of block row k:  S = Dinv(k)^T * R[k, mine]
    CAP_HIP(hipStreamWaitEvent(s1, d->ev_msg[k], 0));
    if (nloc_k > 0) {
      if (k >= 2) CAP_HIP(hipStreamWaitEvent(s1, d->ev_gather[k - 2], 0));   // S[par] was the all-gather source of step k-2
      double* Rrow = d->R + k * nb + lb0 * nb * ld;
      CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, nloc_k, nb, 1.0, Dinv, nb, Rrow, ld, 0.0, d->S[par], nb, 0, s1, 2 | 16));
      CAP_TRY(cap_copy_rect(d->S[par], nb, Rrow, ld, nb, nloc_k, s1));
    }
    CAP_HIP(hipEventRecord(d->ev_solved[k], s1));

    // ---- comm: all-gather block row k (equal-sized padded pieces)
    if (nmax_k > 0) {
      CAP_HIP(hipStreamWaitEvent(sc, d->ev_solved[k], 0));
      if (k >= 2) CAP_HIP(hipStreamWaitEvent(sc, d->ev_update[k - 2], 0));   // G[par] was read by the bulk update of step k-2
      CAP_TRY(cap_comm_allgather(d->comm, d->S[par], d->G[par], piece, (void*)sc));
    }
    CAP_HIP(hipEventRecord(d->ev_gather[k], sc));

    if (k + 1 < nblk) {
      const int q = (int)((k + 1) % P);
      double* msg1 = d->msg[par ^ 1];
      // ---- panel, owner of block column k+1: bring its diagonal block up to date, factor + invert it
      if (p == q) {
        const int64_t lbq = (k + 1) / P;                           // == lb0 on this rank
        double* D = d->R + (k + 1) * nb + lbq * nb * ld;
        if (k >= 1) CAP_HIP(hipStreamWaitEvent(s1, d->ev_update[k - 1], 0));
        CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, nb, nb, -1.0, d->S[par], nb, d->S[par], nb, 1.0, D, ld, 1, s1));
        CAP_TRY(cap_rec_cholinv_full(D, ld, msg1 + nb2, nb, nb, d->W, d->wcap, d->info_dev, s1, (k + 1) * nb));
        CAP_TRY(cap_copy_rect(d->S[par], nb, msg1, nb, nb, nb, s1));   // R(k,k+1) rides along
        CAP_HIP(hipEventRecord(d->ev_fact[k + 1], s1));
        CAP_HIP(hipStreamWaitEvent(sc, d->ev_fact[k + 1], 0));
      }
      // ---- comm: msg(k+1) = [ R(k,k+1) | Dinv(k+1) ] from the owner
      CAP_TRY(cap_comm_bcast(d->comm, msg1, 2 * nb2, q, (void*)sc));
      CAP_HIP(hipEventRecord(d->ev_msg[k + 1], sc));

      // ---- panel: look-ahead - apply step k to my part of block row k+1 (columns J > k+1)
      const int64_t lb1 = lbfirst(p, k + 1, P);
      const int64_t nloc_1 = (d->nloc_blocks - lb1) * nb;
      if (nloc_1 > 0) {
        CAP_HIP(hipStreamWaitEvent(s1, d->ev_msg[k + 1], 0));
        if (k >= 1) CAP_HIP(hipStreamWaitEvent(s1, d->ev_update[k - 1], 0));
        CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, nloc_1, nb, -1.0, msg1, nb, d->S[par] + (lb1 - lb0) * nb2, nb, 1.0,
                                d->R + (k + 1) * nb + lb1 * nb * ld, ld, 0, s1));
      }

      // ---- main: bulk update of my columns J >= k+2, rows >= (k+2) nb, upper staircase
      const int64_t m2 = n - (k + 2) * nb;
      if (m2 > 0 && nloc_1 > 0) {
        int gstart[8];
        for (int64_t r = 0; r < 8; r++) gstart[r] = r < P ? (int)lbfirst(r, k, P) : 0;
        CAP_HIP(hipStreamWaitEvent(s0, d->ev_gather[k], 0));
        CAP_TRY(cap_dist_update_launch(m2, nloc_1, nb, d->G[par], piece, gstart, d->G[par] + p * piece + (lb1 - lb0) * nb2,
                                       d->R + (k + 2) * nb + lb1 * nb * ld, ld, (int)P, (int)p, (int)nb, (int)(k + 2), (int)lb1, s0));
      }
    }
    CAP_HIP(hipEventRecord(d->ev_update[k], s0));
  }
  // join the helper streams back into the caller's stream
  CAP_HIP(hipEventRecord(d->ev_join_p, s1));
  CAP_HIP(hipEventRecord(d->ev_join_c, sc));
  CAP_HIP(hipStreamWaitEvent(s0, d->ev_join_p, 0));
  CAP_HIP(hipStreamWaitEvent(s0, d->ev_join_c, 0));
  return CAP_OK;
}

int cap_dist_info(cap_dist_plan* d, void* stream, int64_t* info) {
  if (!d || !info) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  hipLaunchKernelGGL(info_to_double, dim3(1), dim3(1), 0, s, d->info_dev, d->info_red);
  CAP_TRY(cap_comm_allreduce_sum(d->comm, d->info_red, 1, stream));   // only a diagonal block's owner sets info
  double h = 0;
  CAP_HIP(hipMemcpyAsync(&h, d->info_red, sizeof(double), hipMemcpyDeviceToHost, s));
  CAP_HIP(hipStreamSynchronize(s));
  *info = (int64_t)h;
  return h == 0 ? CAP_OK : CAP_ERR_NOT_SPD;
}

}  // extern "C"
