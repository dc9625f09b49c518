// CholeskyQR / CholeskyQR2, 1D (row-cyclic) path.
//
// Replaces qr::cacqr::factor -> invoke_1d -> sweep_1d (reference
// src/alg/qr/cacqr/cacqr.hpp:217-248, 172-193, 5-29) and policy::cacqr::NoSerialize::
// compute_gram (policy.h:18-24, MPI_Allreduce over `world`).
//
//   sweep:  G = Q^T Q        local DSYRK, split-K over the 2^21 local rows (MFMA)
//           G = allreduce(G) RCCL all-reduce of the n x n Gram (512 KiB at n = 256)
//           R = chol(G), Rinv = R^-1   in-LDS / recursive cholinv (potrf; memcpy; trtri upstream)
//           Q = Q * Rinv      DGEMM "NN" streaming pass (upstream: dtrmm Right/Upper/NoTrans)
//   CholeskyQR2: second sweep on Q, then R = R2 * R1 (cacqr.hpp:180-188).
// Q ping-pongs between two device buffers so the streaming GEMM is never in place.
#include <cstring>
#include <new>

#include "common.h"

// from cholinv.hip
int cap_rec_cholinv_full(double* R, int64_t ldr, double* Ri, int64_t ldi, int64_t n, double* W, int64_t wcap, int* info,
                         hipStream_t s, int64_t info_base = 0);
int64_t cap_rec_work_size(int64_t n);

struct cap_cacqr_plan {
  int64_t m, n; int num_iter; cap_comm* comm;
  double* Q[2]; int cur; int64_t ldq;
  double* G; double* Gi; double* R1; double* R; double* W; int64_t wcap;
  int* info_dev;
};

namespace {
// one CholeskyQR sweep: Qout = Qin * chol(Qin^T Qin)^-1.  The first sweep reads the caller's A in place (no A -> Q copy:
// serialize<rect,rect>(A -> Q) of cacqr.hpp:226 only exists upstream because its TRMM is in place)
int sweep(cap_cacqr_plan* p, const double* Qin, int64_t ldin, double* Qout, hipStream_t s) {
  const int64_t m = p->m, n = p->n;
  // Gram: upper triangle of Q^T Q (cacqr.hpp:15), full square zero-initialised so the all-reduce moves
  // a dense n x n block like NoSerialize::compute_gram (policy.h:22)
  CAP_TRY(cap_zero_rect(p->G, n, n, n, s));
  CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, n, n, m, 1.0, Qin, ldin, Qin, ldin, 0.0, p->G, n, 1, s));
  CAP_TRY(cap_comm_allreduce_sum(p->comm, p->G, n * n, (void*)s));
  // R = chol(G) in place (upper), Gi = R^-1
  CAP_TRY(cap_zero_rect(p->Gi, n, n, n, s));
  CAP_TRY(cap_rec_cholinv_full(p->G, n, p->Gi, n, n, p->W, p->wcap, p->info_dev, s));
  // Q <- Q * R^-1 (cacqr.hpp:24-25)
  // tag 8: R^-1 is upper triangular -> a column tile only contracts the rows above its diagonal block
  CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, m, n, n, 1.0, Qin, ldin, p->Gi, n, 0.0, Qout, p->ldq, 0, s, 8));
  return CAP_OK;
}
}  // namespace

extern "C" {

int cap_cacqr_plan_create(cap_cacqr_plan** plan, int64_t m_local, int64_t n, int num_iter, cap_comm* comm) {
  if (!plan || m_local <= 0 || n <= 0 || num_iter < 1 || num_iter > 2) return CAP_ERR_ARG;
  cap_cacqr_plan* p = new (std::nothrow) cap_cacqr_plan();
  if (!p) return CAP_ERR_ALLOC;
  memset(p, 0, sizeof(*p));
  p->m = m_local; p->n = n; p->num_iter = num_iter; p->comm = comm;
  p->ldq = cap_round_up(m_local, 2);
  p->wcap = cap_rec_work_size(n);
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMalloc((void**)&p->Q[i], sizeof(double) * p->ldq * n);
  if (e == hipSuccess) e = hipMalloc((void**)&p->G, sizeof(double) * n * n * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&p->W, sizeof(double) * p->wcap);
  if (e == hipSuccess) e = hipMalloc((void**)&p->info_dev, sizeof(int));
  if (e != hipSuccess) { cap_cacqr_plan_destroy(p); return CAP_ERR_ALLOC; }
  p->Gi = p->G + n * n; p->R1 = p->G + 2 * n * n; p->R = p->G + 3 * n * n;
  *plan = p;
  return CAP_OK;
}

int cap_cacqr_plan_destroy(cap_cacqr_plan* p) {
  if (!p) return CAP_OK;
  for (int i = 0; i < 2; i++) if (p->Q[i]) (void)hipFree(p->Q[i]);
  if (p->G) (void)hipFree(p->G);
  if (p->W) (void)hipFree(p->W);
  if (p->info_dev) (void)hipFree(p->info_dev);
  delete p;
  return CAP_OK;
}

int cap_cacqr_factor(cap_cacqr_plan* p, const double* A, int64_t lda, void* stream) {
  if (!p || !A || lda < p->m) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  const int64_t n = p->n;
  CAP_HIP(hipMemsetAsync(p->info_dev, 0, sizeof(int), s));
  CAP_TRY(sweep(p, A, lda, p->Q[0], s));
  p->cur = 0;
  if (p->num_iter > 1) {
    CAP_TRY(cap_copy_rect(p->G, n, p->R1, n, n, n, s));            // save_R_1d, policy.h:27-30
    CAP_TRY(sweep(p, p->Q[0], p->ldq, p->Q[1], s));
    p->cur = 1;
    // R = R2 * R1 (dtrmm Right/Upper/NoTrans, cacqr.hpp:182-187); both are upper with zero lower parts
    CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, n, n, n, 1.0, p->G, n, p->R1, n, 0.0, p->R, n, 0, s));
  } else {
    CAP_TRY(cap_copy_rect(p->G, n, p->R, n, n, n, s));
  }
  return CAP_OK;
}

double* cap_cacqr_Q_ptr(cap_cacqr_plan* p, int64_t* ld) { if (!p) return nullptr; if (ld) *ld = p->ldq; return p->Q[p->cur]; }
double* cap_cacqr_R_ptr(cap_cacqr_plan* p, int64_t* ld) { if (!p) return nullptr; if (ld) *ld = p->n; return p->R; }

int cap_cacqr_info(cap_cacqr_plan* p, void* stream, int64_t* info) {
  if (!p || !info) return CAP_ERR_ARG;
  int h = 0;
  CAP_HIP(hipMemcpyAsync(&h, p->info_dev, sizeof(int), hipMemcpyDeviceToHost, cap_stream(stream)));
  CAP_HIP(hipStreamSynchronize(cap_stream(stream)));
  *info = h;
  return h == 0 ? CAP_OK : CAP_ERR_NOT_SPD;
}

}  // extern "C"
