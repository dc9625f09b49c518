// Distributed DGEMM: SUMMA on the reference's d x d x c process grid, RCCL over xGMI.
//
// Replaces matmult::summa::invoke (GEMM overload) + distribute + collect (reference
// src/alg/matmult/summa/summa.hpp:6-44, 163-253; driver bench/matmult/summa_gemm.cpp:7-55) for GPU-resident
// element-cyclic pieces (matrix.hpp:8-11): process (x, y, z) owns A[rows = y mod d, cols = x mod d] etc., the
// matrices are replicated over the c layers.
//
//   upstream (c == d only):  layer z broadcasts A's piece of process column z along `row` and B's piece of process
//                            row z along `column`, multiplies locally, MPI_Allreduce over `depth`.
//   here (any c dividing d): layer z walks the inner process indices k' = z, z + c, z + 2c, ... < d - one broadcast
//                            pair + one local MFMA GEMM per step (c == d: upstream's single step, c == 1: plain 2D
//                            SUMMA with d steps) - then ncclAllReduce over `depth` when c > 1.
//
// Overlap (what upstream's num_chunks pipelining is after, summa.hpp:195-215): the row and the column broadcast run on
// two HIP streams with their own communicators; B travels in `num_chunks` column chunks and the GEMM of chunk j runs
// on the caller's stream while chunk j + 1 (and, for c < d, the next step's A piece) is still in flight.  Events only,
// no host synchronisation.
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"

struct cap_summa_plan {
  cap_topo* topo;
  int c, d, x, y, z;
  cap_comm *row, *column, *depth;
  int64_t m, n, k, ml, nl, kl;          // global / local (ceil) dimensions
  int num_chunks;
  double* bufA[2]; double* bufB[2]; double* acc;
  hipStream_t s_row, s_col;
  std::vector<hipEvent_t> ev_a, ev_b, ev_free;   // [buf], [buf * num_chunks + chunk], [buf]
  hipEvent_t ev_start, ev_join_r, ev_join_c;
};

namespace {
__global__ void combine_kernel(double* C, int64_t ldc, const double* acc, int64_t lda, int64_t m, int64_t n, double beta) {
  const int64_t col = blockIdx.y;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < m; row += (int64_t)gridDim.x * blockDim.x) {
    double* p = C + row + col * ldc;
    const double a = acc[row + col * lda];
    *p = beta == 0.0 ? a : beta * (*p) + a;      // beta == 0: C's old content (possibly NaN) is not read into the result
  }
}
}  // namespace

extern "C" {

int cap_summa_plan_create(cap_summa_plan** plan, cap_topo* topo, int64_t m, int64_t n, int64_t k, int num_chunks) {
  if (!plan || !topo || m <= 0 || n <= 0 || k <= 0 || num_chunks < 0) return CAP_ERR_ARG;
  if (cap_topo_get(topo, 9) != 0) return CAP_ERR_ARG;                 // topo::square only
  cap_summa_plan* p = new (std::nothrow) cap_summa_plan();
  if (!p) return CAP_ERR_ALLOC;
  p->topo = topo;
  p->c = cap_topo_get(topo, 2); p->d = cap_topo_get(topo, 3);
  p->x = cap_topo_get(topo, 4); p->y = cap_topo_get(topo, 5); p->z = cap_topo_get(topo, 6);
  if (p->d % p->c) { delete p; return CAP_ERR_UNSUPPORTED; }          // every inner index needs exactly one layer
  p->row = cap_topo_comm(topo, 1); p->column = cap_topo_comm(topo, 2); p->depth = cap_topo_comm(topo, 3);
  p->m = m; p->n = n; p->k = k;
  p->ml = cap_ceil_div(m, p->d); p->nl = cap_ceil_div(n, p->d); p->kl = cap_ceil_div(k, p->d);
  p->num_chunks = std::max(1, num_chunks);
  if (p->num_chunks > p->nl) p->num_chunks = (int)p->nl;
  for (int i = 0; i < 2; i++) p->bufA[i] = p->bufB[i] = nullptr;
  p->acc = nullptr; p->s_row = p->s_col = nullptr;
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc((void**)&p->bufA[i], sizeof(double) * p->ml * p->kl);
    if (e == hipSuccess) e = hipMalloc((void**)&p->bufB[i], sizeof(double) * p->kl * p->nl);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&p->acc, sizeof(double) * p->ml * p->nl);
  int lo = 0, hi = 0;
  if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&p->s_row, hipStreamNonBlocking, hi);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&p->s_col, hipStreamNonBlocking, hi);
  auto mk = [&](std::vector<hipEvent_t>& v, size_t cnt) {
    v.resize(cnt);
    for (auto& ev : v) if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  };
  mk(p->ev_a, 2); mk(p->ev_b, 2 * (size_t)p->num_chunks); mk(p->ev_free, 2);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_start, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_join_r, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_join_c, hipEventDisableTiming);
  if (e != hipSuccess) { cap_summa_plan_destroy(p); return CAP_ERR_ALLOC; }
  *plan = p;
  return CAP_OK;
}

int cap_summa_plan_destroy(cap_summa_plan* p) {
  if (!p) return CAP_OK;
  for (int i = 0; i < 2; i++) { if (p->bufA[i]) (void)hipFree(p->bufA[i]); if (p->bufB[i]) (void)hipFree(p->bufB[i]); }
  if (p->acc) (void)hipFree(p->acc);
  for (auto* v : {&p->ev_a, &p->ev_b, &p->ev_free}) for (auto ev : *v) if (ev) (void)hipEventDestroy(ev);
  if (p->ev_start) { (void)hipEventDestroy(p->ev_start); (void)hipEventDestroy(p->ev_join_r); (void)hipEventDestroy(p->ev_join_c); }
  if (p->s_row) (void)hipStreamDestroy(p->s_row);
  if (p->s_col) (void)hipStreamDestroy(p->s_col);
  delete p;
  return CAP_OK;
}

void cap_summa_local_dims(const cap_summa_plan* p, int64_t* ml, int64_t* nl, int64_t* kl) {
  if (!p) return;
  if (ml) *ml = p->ml; if (nl) *nl = p->nl; if (kl) *kl = p->kl;
}

// C_local = alpha * (A B)_local + beta * C_local.  A_local: ml x kl (lda), B_local: kl x nl (ldb), C_local: ml x nl (ldc),
// all column-major element-cyclic pieces with zero padding (matrix.hpp:8-11).  NoTrans x NoTrans, like the reference's
// driver (bench/matmult/summa_gemm.cpp:38); the transposed forms need a partner exchange upstream does outside SUMMA.
int cap_summa_dgemm(cap_summa_plan* p, double alpha, const double* A, int64_t lda, const double* B, int64_t ldb, double beta,
                    double* C, int64_t ldc, void* stream) {
  if (!p || !A || !B || !C || lda < p->ml || ldb < p->kl || ldc < p->ml) return CAP_ERR_ARG;
  hipStream_t s0 = cap_stream(stream), sr = p->s_row, sc = p->s_col;
  const int64_t ml = p->ml, nl = p->nl, kl = p->kl;
  const int nch = p->num_chunks;
  CAP_HIP(hipEventRecord(p->ev_start, s0));
  CAP_HIP(hipStreamWaitEvent(sr, p->ev_start, 0));
  CAP_HIP(hipStreamWaitEvent(sc, p->ev_start, 0));
  std::vector<int> steps;
  for (int kp = p->z; kp < p->d; kp += p->c) steps.push_back(kp);
  for (size_t si = 0; si < steps.size(); si++) {
    const int kp = steps[si], buf = (int)(si & 1);
    // ---- row stream: A's piece of process column kp to everybody in my process row (summa.hpp:185)
    if (si >= 2) CAP_HIP(hipStreamWaitEvent(sr, p->ev_free[buf], 0));       // the GEMMs of step si-2 have read bufA[buf]
    if (p->x == kp) CAP_TRY(cap_copy_rect(A, lda, p->bufA[buf], ml, ml, kl, sr));
    CAP_TRY(cap_comm_bcast(p->row, p->bufA[buf], ml * kl, kp, (void*)sr));
    CAP_HIP(hipEventRecord(p->ev_a[buf], sr));
    // ---- column stream: B's piece of process row kp down my process column, in column chunks (summa.hpp:193, 201-214)
    if (si >= 2) CAP_HIP(hipStreamWaitEvent(sc, p->ev_free[buf], 0));
    if (p->y == kp) CAP_TRY(cap_copy_rect(B, ldb, p->bufB[buf], kl, kl, nl, sc));
    for (int ch = 0; ch < nch; ch++) {
      const int64_t c0 = nl * ch / nch, c1 = nl * (ch + 1) / nch;
      CAP_TRY(cap_comm_bcast(p->column, p->bufB[buf] + c0 * kl, kl * (c1 - c0), kp, (void*)sc));
      CAP_HIP(hipEventRecord(p->ev_b[buf * nch + ch], sc));
    }
    // ---- caller's stream: acc(:, chunk) (+)= A_piece * B_piece(:, chunk) as the chunks land
    CAP_HIP(hipStreamWaitEvent(s0, p->ev_a[buf], 0));
    for (int ch = 0; ch < nch; ch++) {
      const int64_t c0 = nl * ch / nch, c1 = nl * (ch + 1) / nch;
      CAP_HIP(hipStreamWaitEvent(s0, p->ev_b[buf * nch + ch], 0));
      CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, ml, c1 - c0, kl, alpha, p->bufA[buf], ml, p->bufB[buf] + c0 * kl, kl,
                              si == 0 ? 0.0 : 1.0, p->acc + c0 * ml, ml, 0, s0));
    }
    CAP_HIP(hipEventRecord(p->ev_free[buf], s0));
  }
  // ---- depth: sum the layers' partial products (summa.hpp:236), then C = beta C + acc (summa.hpp:32-35)
  if (p->c > 1) CAP_TRY(cap_comm_allreduce_sum(p->depth, p->acc, ml * nl, (void*)s0));
  if (nl > 65535) return CAP_ERR_UNSUPPORTED;
  hipLaunchKernelGGL(combine_kernel, dim3((unsigned)std::min<int64_t>(cap_ceil_div(ml, 256), 1024), (unsigned)nl), dim3(256), 0, s0, C, ldc,
                     p->acc, ml, ml, nl, beta);
  CAP_HIP(hipGetLastError());
  CAP_HIP(hipEventRecord(p->ev_join_r, sr));
  CAP_HIP(hipEventRecord(p->ev_join_c, sc));
  CAP_HIP(hipStreamWaitEvent(s0, p->ev_join_r, 0));
  CAP_HIP(hipStreamWaitEvent(s0, p->ev_join_c, 0));
  return CAP_OK;
}

}  // extern "C"
