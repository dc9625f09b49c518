// CholeskyQR / CholeskyQR2: the 1D (row-cyclic) path and the 3D / tunable c x d x c grid path.
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
//
// Grid path (cacqr.hpp:75-170 sweep_3d / sweep_tune, :44-73 solve, :195-215 invoke_3d) on a topo::rect bundle
// (c x d x c, c == d: the 3D cube; d > c: the tunable grid): A's piece on (x, y, z) holds rows = y mod d, columns =
// x mod c, replicated over the c layers.  Per sweep
//   Qz = bcast over `row` of the piece of process column z                 (MPI_Bcast, cacqr.hpp:92 / :141)
//   G[=z, =x] = Qz^T Q_mine, summed over all process rows y                (MPI_Reduce + MPI_Allreduce + MPI_Bcast upstream,
//                                                                           :98-99 / :147-149; here all-reduces over
//                                                                           column_contig and column_alt)
//   the c x c cyclic blocks are all-gathered over `row` and `depth` into the dense n x n Gram on EVERY rank, and the
//   Cholesky factor + inverse are computed redundantly per GPU (n is a few hundred: a distributed cholinv over the
//   c x c x c cube, as upstream calls it, would be pure latency on GPUs)
//   Q_new[=y, =x] = sum_z Qz * Rinv[=z, =x]: layer z contributes its term, ncclAllReduce over `depth`
//                                                                          (summa TRMM + depth all-reduce, :107-111)
// which is `A2 - Q1 R12` done right for every block at once (upstream's solve() has the sign flipped).
#include <cstdlib>
#include <cstring>
#include <new>

#include "common.h"

// from cholinv.hip
int cap_rec_cholinv_full(double* R, int64_t ldr, double* Ri, int64_t ldi, int64_t n, double* W, int64_t wcap, int* info,
                         hipStream_t s, int64_t info_base = 0);
int64_t cap_rec_work_size(int64_t n);
// cqr_kernels.hip: the n = 256 streaming kernels (whole-Gram workgroups; persistent row-streaming Q R^-1)
int64_t cap_gram256_work(int64_t m);
int cap_gram256_launch(const double* Q, int64_t ld, int64_t m, double* G, int64_t ldg, double* work, hipStream_t s);
int cap_qrapply256_launch(const double* Qin, int64_t ldin, const double* Ri, double* Qout, int64_t ldout, int64_t m, hipStream_t s);

struct cap_cacqr_plan {
  int64_t m, n; int num_iter; cap_comm* comm;
  double* Q[2]; int cur; int64_t ldq;
  double* G; double* Gi; double* R1; double* R; double* W; int64_t wcap;
  int* info_dev;
  double* gram_work;      // n == 256: one partial-Gram slab per workgroup of gram256
  bool gi_clean;          // Gi's never-written blocks are known to be zero (see sweep)
  // grid path: m, n above are the GLOBAL column count / local row count of the dense n x n work; nl = n / c local columns
  cap_topo* topo; int c, d, x, y, z; int64_t nl;
  double* Qz; double* Gblk; double* Gall; double* Rip; double* Rpiece;
};

namespace {
// one CholeskyQR sweep: Qout = Qin * chol(Qin^T Qin)^-1.  The first sweep reads the caller's A in place (no A -> Q copy:
// serialize<rect,rect>(A -> Q) of cacqr.hpp:226 only exists upstream because its TRMM is in place)
int sweep(cap_cacqr_plan* p, const double* Qin, int64_t ldin, double* Qout, hipStream_t s) {
  const int64_t m = p->m, n = p->n;
  // Gram: upper triangle of Q^T Q (cacqr.hpp:15), full square zero-initialised so the all-reduce moves
  // a dense n x n block like NoSerialize::compute_gram (policy.h:22)
  static const bool k256_env = CAP_ENV("CAP_CQR256") ? atoi(CAP_ENV("CAP_CQR256")) != 0 : true;
  const bool k256 = k256_env && p->gram_work && n == 256 && m % 128 == 0 && !(ldin & 1) && 128 * ldin * 8 < 0xfffffff0LL && !((uintptr_t)Qin & 15);
  {
    CapRange range("CQR::gram");                  // cacqr.hpp:83-101
    if (k256) {
      CAP_TRY(cap_gram256_launch(Qin, ldin, m, p->G, n, p->gram_work, s));     // its slab sum also zero-fills below the diagonal
    } else {
      CAP_TRY(cap_zero_rect(p->G, n, n, n, s));
      CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, n, n, m, 1.0, Qin, ldin, Qin, ldin, 0.0, p->G, n, 1, s));
    }
    CAP_TRY(cap_comm_allreduce_sum(p->comm, p->G, n * n, (void*)s));
  }
  // R = chol(G) in place (upper), Gi = R^-1.  The 64-blocked path (n = 256) rewrites every entry of Gi it ever wrote (diagonal
  // blocks with their zero lower parts, the off-diagonal blocks above them) and never touches the blocks below: one zero-fill
  // per plan is enough there
  if (!(k256 && p->gi_clean)) { CAP_TRY(cap_zero_rect(p->Gi, n, n, n, s)); p->gi_clean = k256; }
  CAP_TRY(cap_rec_cholinv_full(p->G, n, p->Gi, n, n, p->W, p->wcap, p->info_dev, s));
  // Q <- Q * R^-1 (cacqr.hpp:24-25)
  // tag 8: R^-1 is upper triangular -> a column tile only contracts the rows above its diagonal block
  CapRange range("CQR::formR");                   // cacqr.hpp:106-115 (the TRMM with R^-1)
  if (k256) CAP_TRY(cap_qrapply256_launch(Qin, ldin, p->Gi, Qout, p->ldq, m, s));
  else CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, m, n, n, 1.0, Qin, ldin, p->Gi, n, 0.0, Qout, p->ldq, 0, s, 8));
  return CAP_OK;
}
// gathered[z'][x'] blocks (nl x nl each, block (z', x') = G[rows = z' mod c, cols = x' mod c]) -> dense n x n
__global__ void blocks_to_dense_kernel(const double* blocks, double* dense, int64_t n, int64_t nl, int c) {
  const int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x, j = blockIdx.y;
  if (i >= n || j >= n) return;
  const int64_t zi = i % c, a = i / c, xj = j % c, b = j / c;
  dense[i + j * n] = blocks[(zi * c + xj) * nl * nl + a + b * nl];
}

// one sweep on the c x d x c grid; Qin / Qout are local pieces (ml x nl)
int sweep_grid(cap_cacqr_plan* p, const double* Qin, int64_t ldin, double* Qout, hipStream_t s) {
  const int64_t ml = p->m, n = p->n, nl = p->nl;
  cap_comm* row = cap_topo_comm(p->topo, 1); cap_comm* depth = cap_topo_comm(p->topo, 3);
  cap_comm* ccontig = cap_topo_comm(p->topo, 5); cap_comm* calt = cap_topo_comm(p->topo, 6);
  // Qz: the piece of process column z, broadcast along my process row (root = the rank with x == z)
  if (p->x == p->z) CAP_TRY(cap_copy_rect(Qin, ldin, p->Qz, ml, ml, nl, s));
  CAP_TRY(cap_comm_bcast(row, p->Qz, ml * nl, p->z, (void*)s));
  // partial Gram block over my rows, then the sum over all process rows
  CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nl, nl, ml, 1.0, p->Qz, ml, Qin, ldin, 0.0, p->Gblk, nl, 0, s));
  CAP_TRY(cap_comm_allreduce_sum(ccontig, p->Gblk, nl * nl, (void*)s));
  CAP_TRY(cap_comm_allreduce_sum(calt, p->Gblk, nl * nl, (void*)s));
  // dense Gram everywhere: blocks (z, x') over `row`, then (z', x') over `depth`
  CAP_TRY(cap_comm_allgather(row, p->Gblk, p->Gall + (int64_t)p->z * p->c * nl * nl, nl * nl, (void*)s));
  CAP_TRY(cap_comm_allgather(depth, p->Gall + (int64_t)p->z * p->c * nl * nl, p->Gall, p->c * nl * nl, (void*)s));
  cap_acc_r(p->Gall, 0, (int64_t)p->c * p->c * nl * nl, 1); cap_acc_w(p->G, n, n, n);
  hipLaunchKernelGGL(blocks_to_dense_kernel, dim3((unsigned)cap_ceil_div(n, 256), (unsigned)n), dim3(256), 0, s, p->Gall, p->G, n, nl, p->c);
  CAP_HIP(hipGetLastError());
  // R = chol(G) (upper, in place), Gi = R^-1 - redundantly on every GPU
  CAP_TRY(cap_copy_window(p->G, 0, n, 0, 0, p->G, 0, n, 0, 0, n, n, 1, 1, (void*)s));     // keep the upper triangle, zero the rest
  CAP_TRY(cap_zero_rect(p->Gi, n, n, n, s));
  CAP_TRY(cap_rec_cholinv_full(p->G, n, p->Gi, n, n, p->W, p->wcap, p->info_dev, s));
  // my layer's term of Q R^-1: Qz * Rinv[rows = z mod c, cols = x mod c], summed over the layers
  CAP_TRY(cap_cyclic_export(p->Gi, n, p->Rip, nl, n, n, p->x, p->z, p->c, p->c, (void*)s));
  CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, ml, nl, nl, 1.0, p->Qz, ml, p->Rip, nl, 0.0, Qout, p->ldq, 0, s));
  CAP_TRY(cap_comm_allreduce_sum(depth, Qout, p->ldq * nl, (void*)s));
  return CAP_OK;
}
}  // namespace

extern "C" {

// qr::cacqr on the c x d x c grid of a topo::rect bundle (cacqr.hpp:217-248 with c > 1): A_local is the
// ceil(M / d) x (N / c) element-cyclic piece; N must be a multiple of c (a zero-padded column would make the Gram singular).
int cap_cacqr_plan_create_grid(cap_cacqr_plan** plan, int64_t m_global, int64_t n_global, int num_iter, cap_topo* topo) {
  if (!plan || !topo || m_global <= 0 || n_global <= 0 || num_iter < 1 || num_iter > 2) return CAP_ERR_ARG;
  if (cap_topo_get(topo, 9) != 1) return CAP_ERR_ARG;                  // topo::rect
  const int c = cap_topo_get(topo, 2), d = cap_topo_get(topo, 3);
  if (n_global % c) return CAP_ERR_UNSUPPORTED;
  // the Gram block is summed by all-reduce(column_contig) then all-reduce(column_alt): that covers all d process rows
  // only when the column splits into whole groups of c (the same reduce + all-reduce pair as cacqr.hpp:147-148 over topology.h:38-39's splits)
  if (d % c) return CAP_ERR_UNSUPPORTED;
  cap_cacqr_plan* p = new (std::nothrow) cap_cacqr_plan();
  if (!p) return CAP_ERR_ALLOC;
  memset(p, 0, sizeof(*p));
  p->topo = topo; p->c = c; p->d = d;
  p->x = cap_topo_get(topo, 4); p->y = cap_topo_get(topo, 5); p->z = cap_topo_get(topo, 6);
  p->m = cap_ceil_div(m_global, d); p->n = n_global; p->nl = n_global / c; p->num_iter = num_iter; p->comm = nullptr;
  p->ldq = cap_round_up(p->m, 2);
  p->wcap = cap_rec_work_size(p->n);
  const int64_t n = p->n, nl = p->nl;
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMalloc((void**)&p->Q[i], sizeof(double) * p->ldq * nl);
  if (e == hipSuccess) e = hipMalloc((void**)&p->G, sizeof(double) * n * n * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&p->W, sizeof(double) * p->wcap);
  if (e == hipSuccess) e = hipMalloc((void**)&p->info_dev, sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&p->Qz, sizeof(double) * p->m * nl);
  if (e == hipSuccess) e = hipMalloc((void**)&p->Gblk, sizeof(double) * (nl * nl * 3 + (int64_t)c * c * nl * nl));
  if (e != hipSuccess) { cap_cacqr_plan_destroy(p); return CAP_ERR_ALLOC; }
  p->Gi = p->G + n * n; p->R1 = p->G + 2 * n * n; p->R = p->G + 3 * n * n;
  p->Rip = p->Gblk + nl * nl; p->Rpiece = p->Gblk + 2 * nl * nl; p->Gall = p->Gblk + 3 * nl * nl;
  // padding rows of the Q pieces stay zero
  for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMemset(p->Q[i], 0, sizeof(double) * p->ldq * nl);
  if (e != hipSuccess) { cap_cacqr_plan_destroy(p); return CAP_ERR_ALLOC; }
  *plan = p;
  return CAP_OK;
}

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
  if (e == hipSuccess && n == 256 && m_local % 128 == 0) e = hipMalloc((void**)&p->gram_work, sizeof(double) * cap_gram256_work(m_local));
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
  if (p->gram_work) (void)hipFree(p->gram_work);
  if (p->Qz) (void)hipFree(p->Qz);
  if (p->Gblk) (void)hipFree(p->Gblk);
  delete p;
  return CAP_OK;
}

int cap_cacqr_factor(cap_cacqr_plan* p, const double* A, int64_t lda, void* stream) {
  if (!p || !A || lda < p->m) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  const int64_t n = p->n;
  CAP_HIP(hipMemsetAsync(p->info_dev, 0, sizeof(int), s));
  if (p->topo) {
    // invoke_3d (cacqr.hpp:195-215): sweep, [save R1, sweep, R = R2 R1]; R stays dense and replicated, the caller's
    // cyclic piece is cut out by cap_cacqr_R_piece
    CAP_TRY(sweep_grid(p, A, lda, p->Q[0], s));
    p->cur = 0;
    if (p->num_iter > 1) {
      CAP_TRY(cap_copy_rect(p->G, n, p->R1, n, n, n, s));
      CAP_TRY(sweep_grid(p, p->Q[0], p->ldq, p->Q[1], s));
      p->cur = 1;
      CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, n, n, n, 1.0, p->G, n, p->R1, n, 0.0, p->R, n, 0, s));
    } else {
      CAP_TRY(cap_copy_rect(p->G, n, p->R, n, n, n, s));
    }
    return CAP_OK;
  }
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
int64_t cap_cacqr_local_cols(const cap_cacqr_plan* p) { return p ? (p->topo ? p->nl : p->n) : 0; }
// grid path: this rank's element-cyclic piece of R (rows = y mod c, columns = x mod c of the c x c face - what upstream's
// args.R holds, cacqr.hpp:214), nl x nl, into out (ld >= nl).  1D path: the whole replicated R.
int cap_cacqr_R_piece(cap_cacqr_plan* p, double* out, int64_t ld, void* stream) {
  if (!p || !out) return CAP_ERR_ARG;
  if (!p->topo) return cap_copy_window(p->R, 0, p->n, 0, 0, out, 0, ld, 0, 0, p->n, p->n, 1, 1, stream);
  if (ld < p->nl) return CAP_ERR_ARG;
  return cap_cyclic_export(p->R, p->n, out, ld, p->n, p->n, p->x, p->y % p->c, p->c, p->c, stream);
}

int cap_cacqr_info(cap_cacqr_plan* p, void* stream, int64_t* info) {
  if (!p || !info) return CAP_ERR_ARG;
  int h = 0;
  CAP_HIP(hipMemcpyAsync(&h, p->info_dev, sizeof(int), hipMemcpyDeviceToHost, cap_stream(stream)));
  CAP_HIP(hipStreamSynchronize(cap_stream(stream)));
  *info = h;
  return h == 0 ? CAP_OK : CAP_ERR_NOT_SPD;
}

}  // extern "C"
