// Mixed-precision Cholesky solve on P GPUs (BASELINE config 5: N = 131072 on 8 MI355X): the bf16-MFMA factorization of
// mixed.hip on the 1 x P block-column-cyclic layout of dist.hip, then fp64 iterative refinement with a distributed solve.
// Not in the reference (fp64 only, solve path a stub: trsm/diaginvert/diaginvert.hpp:7-10); the distributed fp64 path
// (dist.hip) and the single-GPU mixed path (mixed.hip) are its oracles.
//
//   factor  R32 (fp32, my block columns, all rows), block row k (width nb, default 1024 = K of the bf16 update):
//             owner(k)   diagonal block -> fp64, fused chain (R_kk, Dinv), back to fp32         | Dinv(k) broadcast (fp64, nb^2) |
//             every rank S_k = Dinv^T R[k, mine]  in fp64 (converted row slab, MFMA GEMM), stored as fp32 (factor) + bf16 (panel)
//             all-gather of the bf16 panel pieces: a quarter of the fp64 schedule's bytes per strip
//             C32[rows > k, mine] -= G16^T P16  on v_mfma_f32_32x32x16_bf16 (bf16_tn_kernel with the staircase mask + gathered A
//             operand): HEAD = block row k + 1 on the panel stream, bulk = the rows below on the caller's stream (look-ahead 1)
//   solve   V (n x w, replicated) <- R^-1 R^-T V with the fp64-promoted block columns:
//             forward  R^T y = v   block i on owner(i): t = v_i - R[0:i, i]^T y[0:i] (local: column i is mine), y_i = Dinv(i)^T t,
//                                  broadcast of y_i (nb x w)
//             backward R x = y     x_i = Dinv(i) (y_i - sum_r acc_r[i]) after an all-reduce of the nb x w partial sums;
//                                  owner(i) then adds R[0:i, i] x_i to its accumulator
//           r = b - A x: A is symmetric and I hold its block columns, so (A x)[my rows] = A_mine^T x is a local K = n MFMA GEMM;
//           the pieces are scattered into a zero n x w buffer and all-reduced.  Classical refinement x += A^-1 r until
//           ||r||_F / ||b||_F <= tol.  Every rank ends with the same X.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"
#include "mixed_kernels.h"

int cap_rec_cholinv_full(double* R, int64_t ldr, double* Ri, int64_t ldi, int64_t n, double* W, int64_t wcap, int* info,
                         hipStream_t s, int64_t info_base = 0);
int64_t cap_rec_work_size(int64_t n);
int cap_bf16_dist_update_launch(int64_t m, int64_t nloc, int64_t k, const void* G16, int64_t piece, const int* gstart, const void* B16,
                                float* C, int64_t ldc, int P, int p, int nb, int J0, int lb0, hipStream_t s);

struct cap_dmp_plan {
  int64_t n, npad, nb, nblk, w;     // w: internal right-hand-side width (multiple of 128)
  int P, p;
  cap_comm* comm;
  int64_t nloc_blocks, lc, lc_valid, nmax0;
  float* R32; double* R64; int64_t ld;          // npad x lc
  __bf16* P16[2]; __bf16* G16[2];               // my bf16 panel piece (nb x cols, ld = nb) / the gathered pieces
  double* msg[2];                               // Dinv(k), nb x nb
  double* D64; double* T64; double* S64; double* W; int64_t wcap;
  double* Dall;                                 // every diagonal-block inverse (nblk x nb x nb)
  double* V; double* Rw; double* Bw; double* Acc; double* Tmp; double* Q; double* norms;   // solve state (npad x w each)
  int* info_dev; double* info_red;
  hipStream_t s_panel, s_comm;
  std::vector<hipEvent_t> ev_fact, ev_msg, ev_solved, ev_gather, ev_head, ev_bulk;
  hipEvent_t ev_init, ev_join_p, ev_join_c;
  bool have_r64;
};

namespace {
inline int64_t lbfirst_(int64_t r, int64_t k, int64_t P) { return k >= r ? (k - r) / P + 1 : 0; }
inline int64_t nblocks_of_(int64_t r, int64_t nblk, int64_t P) { return r < nblk ? (nblk - 1 - r) / P + 1 : 0; }

// my block columns of A -> fp32, with the identity tail of the padded matrix (rows / global columns >= n)
__global__ void import_f32_bc_kernel(const double* A, int64_t lda, float* R, int64_t ld, int64_t n, int64_t npad, int64_t nb, int P, int p,
                                     int64_t lc) {
  const int64_t lcol = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (lcol >= lc) return;
  const int64_t gcol = ((lcol / nb) * P + p) * nb + lcol % nb;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < npad; row += (int64_t)gridDim.x * blockDim.x)
    R[row + lcol * ld] = (row < n && gcol < n) ? (float)A[row + lcol * lda] : (row == gcol ? 1.0f : 0.0f);
}
// Tmp <- Vblk - Tmp  (rows x cols; Vblk strided)
__global__ void sub_from_kernel(double* T, int64_t ldt, const double* V, int64_t ldv, int64_t rows, int64_t cols) {
  const int64_t col = blockIdx.y;
  if (col >= cols) return;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < rows; row += (int64_t)gridDim.x * blockDim.x)
    T[row + col * ldt] = V[row + col * ldv] - T[row + col * ldt];
}
// my rows of A x (lc_valid x w, compact) -> rows of the replicated buffer at their global positions
__global__ void scatter_rows_bc_kernel(const double* Q, int64_t ldq, double* Out, int64_t ldo, int64_t nb, int P, int p, int64_t lc_valid,
                                       int64_t cols) {
  const int64_t col = blockIdx.y;
  if (col >= cols) return;
  for (int64_t l = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; l < lc_valid; l += (int64_t)gridDim.x * blockDim.x) {
    const int64_t g = ((l / nb) * P + p) * nb + l % nb;
    Out[g + col * ldo] = Q[l + col * ldq];
  }
}
__global__ void info_to_double3(const int* info, double* out) { *out = (double)*info; }

int ensure_events3(cap_dmp_plan* d) {
  if (!d->ev_msg.empty()) return CAP_OK;
  auto mk = [&](std::vector<hipEvent_t>& v, size_t cnt) -> int {
    v.resize(cnt);
    for (auto& e : v) CAP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return CAP_OK;
  };
  const size_t cnt = (size_t)d->nblk + 2;
  CAP_TRY(mk(d->ev_fact, cnt)); CAP_TRY(mk(d->ev_msg, cnt)); CAP_TRY(mk(d->ev_solved, cnt));
  CAP_TRY(mk(d->ev_gather, cnt)); CAP_TRY(mk(d->ev_head, cnt)); CAP_TRY(mk(d->ev_bulk, cnt));
  CAP_HIP(hipEventCreateWithFlags(&d->ev_init, hipEventDisableTiming));
  CAP_HIP(hipEventCreateWithFlags(&d->ev_join_p, hipEventDisableTiming));
  CAP_HIP(hipEventCreateWithFlags(&d->ev_join_c, hipEventDisableTiming));
  int lo = 0, hi = 0;
  CAP_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CAP_HIP(hipStreamCreateWithPriority(&d->s_panel, hipStreamNonBlocking, hi));
  CAP_HIP(hipStreamCreateWithPriority(&d->s_comm, hipStreamNonBlocking, hi));
  // One completed command on each fresh stream before the schedule's first event record / wait names it.  Defensive (round 5): once,
  // on a one-rank plan whose FIRST factor call was followed at once by cap_dmp_info and a device synchronisation, both returned
  // while the plan's streams were still in their first step (tests/dist_worker.py, profiles/r05_flake_forensics.txt) - the
  // hardware queue behind a stream is created lazily with its first command.  Private streams with one 8-byte memset each: nothing
  // another rank could be waiting for.
  {
    CAP_HIP(hipMemsetAsync(d->info_red, 0, sizeof(double), d->s_panel));
    CAP_HIP(hipStreamSynchronize(d->s_panel));
    CAP_HIP(hipMemsetAsync(d->info_red, 0, sizeof(double), d->s_comm));
    CAP_HIP(hipStreamSynchronize(d->s_comm));
  }
  return CAP_OK;
}
}  // namespace

extern "C" {

// n % 128 == 0 (whole bf16 tiles); nb: block width = K of the bf16 update, a power of two >= 128 (0 = 1024); comm: P <= 8 ranks
int cap_dmp_plan_create(cap_dmp_plan** plan, int64_t n, int64_t nb, int64_t nrhs_max, cap_comm* comm) {
  if (!plan || n <= 0 || nrhs_max <= 0) return CAP_ERR_ARG;
  if (nb <= 0) nb = 1024;
  if (n % 128 || nb % 128 || (nb & (nb - 1))) return CAP_ERR_UNSUPPORTED;
  cap_dmp_plan* d = new (std::nothrow) cap_dmp_plan();
  if (!d) return CAP_ERR_ALLOC;
  d->n = n; d->nb = nb; d->nblk = cap_ceil_div(n, nb); d->npad = d->nblk * nb; d->w = cap_round_up(nrhs_max, 128);
  d->comm = comm; d->P = cap_comm_size(comm); d->p = cap_comm_rank(comm);
  if (d->P > 8) { delete d; return CAP_ERR_UNSUPPORTED; }
  d->nloc_blocks = nblocks_of_(d->p, d->nblk, d->P); d->lc = d->nloc_blocks * nb;
  d->nmax0 = nblocks_of_(0, d->nblk, d->P) * nb;
  d->lc_valid = 0;
  for (int64_t lb = 0; lb < d->nloc_blocks; lb++) d->lc_valid += std::min<int64_t>(nb, n - std::min<int64_t>(n, (lb * d->P + d->p) * nb));
  d->ld = d->npad;
  d->R32 = nullptr; d->R64 = nullptr; d->D64 = nullptr; d->Dall = nullptr; d->V = nullptr; d->info_dev = nullptr; d->info_red = nullptr;
  for (int i = 0; i < 2; i++) { d->P16[i] = d->G16[i] = nullptr; d->msg[i] = nullptr; }
  d->s_panel = d->s_comm = nullptr; d->have_r64 = false;
  d->wcap = cap_rec_work_size(nb);
  const int64_t cols = std::max<int64_t>(d->nmax0, nb), npad = d->npad, w = d->w;
  hipError_t e = hipMalloc((void**)&d->R32, sizeof(float) * npad * cols);
  if (e == hipSuccess) e = hipMalloc((void**)&d->R64, sizeof(double) * npad * cols);
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc((void**)&d->P16[i], sizeof(__bf16) * nb * (cols + nb));
    if (e == hipSuccess) e = hipMemset(d->P16[i], 0, sizeof(__bf16) * nb * (cols + nb));
    if (e == hipSuccess) e = hipMalloc((void**)&d->G16[i], sizeof(__bf16) * nb * cols * d->P);
    if (e == hipSuccess) e = hipMalloc((void**)&d->msg[i], sizeof(double) * nb * nb);
    if (e == hipSuccess) e = hipMemset(d->msg[i], 0, sizeof(double) * nb * nb);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&d->D64, sizeof(double) * (nb * nb + 2 * nb * cols + d->wcap));
  if (e == hipSuccess) e = hipMalloc((void**)&d->Dall, sizeof(double) * d->nblk * nb * nb);
  if (e == hipSuccess) e = hipMalloc((void**)&d->V, sizeof(double) * (4 * npad * w + 2 * nb * w + std::max<int64_t>(cols, 128) * w + 8));
  if (e == hipSuccess) e = hipMalloc((void**)&d->info_dev, sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&d->info_red, sizeof(double) * (d->P + 1));
  if (e != hipSuccess) { cap_dmp_plan_destroy(d); return CAP_ERR_ALLOC; }
  d->T64 = d->D64 + nb * nb; d->S64 = d->T64 + nb * cols; d->W = d->S64 + nb * cols;
  d->Rw = d->V + npad * w; d->Bw = d->Rw + npad * w; d->Acc = d->Bw + npad * w; d->Tmp = d->Acc + npad * w;
  d->Q = d->Tmp + 2 * nb * w; d->norms = d->Q + std::max<int64_t>(cols, 128) * w;
  // helper streams and events exist from here on (round 6: never created inside the first factor call)
  { const int st = ensure_events3(d); if (st != CAP_OK) { cap_dmp_plan_destroy(d); return st; } }
  *plan = d;
  return CAP_OK;
}

int cap_dmp_plan_destroy(cap_dmp_plan* d) {
  if (!d) return CAP_OK;
  for (void* q : {(void*)d->R32, (void*)d->R64, (void*)d->P16[0], (void*)d->P16[1], (void*)d->G16[0], (void*)d->G16[1], (void*)d->msg[0],
                  (void*)d->msg[1], (void*)d->D64, (void*)d->Dall, (void*)d->V, (void*)d->info_dev, (void*)d->info_red})
    if (q) (void)hipFree(q);
  if (!d->ev_msg.empty()) {
    for (auto* v : {&d->ev_fact, &d->ev_msg, &d->ev_solved, &d->ev_gather, &d->ev_head, &d->ev_bulk})
      for (auto e : *v) (void)hipEventDestroy(e);
    (void)hipEventDestroy(d->ev_init); (void)hipEventDestroy(d->ev_join_p); (void)hipEventDestroy(d->ev_join_c);
    cap_stream_destroy(d->s_panel); cap_stream_destroy(d->s_comm);
  }
  delete d;
  return CAP_OK;
}

int64_t cap_dmp_local_cols(const cap_dmp_plan* d) { return d ? d->lc_valid : 0; }
float* cap_dmp_R32_ptr(cap_dmp_plan* d, int64_t* ld) { if (!d) return nullptr; if (ld) *ld = d->ld; return d->R32; }

// Alocal: my block columns of the fp64 matrix (n rows, cap_dmp_local_cols columns, lda >= n); upper triangle consumed
int cap_dmp_factor(cap_dmp_plan* d, const double* Aloc, int64_t lda, void* stream) {
  if (!d || (d->lc_valid > 0 && (!Aloc || lda < d->n))) return CAP_ERR_ARG;
  CAP_TRY(ensure_events3(d));
  hipStream_t s0 = cap_stream(stream), s1 = d->s_panel, sc = d->s_comm;
  const int64_t n = d->n, npad = d->npad, nb = d->nb, nblk = d->nblk, P = d->P, p = d->p, ld = d->ld, nb2 = nb * nb;
  CAP_HIP(hipMemsetAsync(d->info_dev, 0, sizeof(int), s0));
  if (d->lc > 0) {
    cap_acc_r(Aloc, lda, n, d->lc_valid); cap_acc_w(d->R32, ld, npad, d->lc, 0, 4);
    hipLaunchKernelGGL(import_f32_bc_kernel, grid2(npad, d->lc), dim3(256), 0, s0, Aloc, lda, d->R32, ld, n, npad, nb, (int)P, (int)p, d->lc);
    CAP_HIP(hipGetLastError());
  }
  CAP_HIP(hipEventRecord(d->ev_init, s0));
  CAP_HIP(hipStreamWaitEvent(s1, d->ev_init, 0));
  CAP_HIP(hipStreamWaitEvent(sc, d->ev_init, 0));

  for (int64_t k = 0; k < nblk; k++) {
    const int par = (int)(k & 1);
    const int owner = (int)(k % P);
    double* Dinv = d->msg[par];
    __bf16* Pk = d->P16[par]; __bf16* Gk = d->G16[par];
    const int64_t lbk = lbfirst_(p, k, P);                       // my local blocks with J <= k
    const int64_t ncols = (d->nloc_blocks - lbk) * nb;           // my columns J > k
    // ---- diagonal block on its owner (block row k has every update: HEAD(k-1) on this stream, bulk(<= k-2) by event)
    if (k >= 2) CAP_HIP(hipStreamWaitEvent(s1, d->ev_bulk[k - 2], 0));
    if (p == owner) {
      CapRange range("CI::factor_diag");
      float* D32 = d->R32 + k * nb + (k / P) * nb * ld;
      launch_f32_to_f64(s1, D32, ld, d->D64, nb, nb, nb, 1);
      CAP_HIP(hipMemsetAsync(Dinv, 0, sizeof(double) * nb2, s1));
      CAP_TRY(cap_rec_cholinv_full(d->D64, nb, Dinv, nb, nb, d->W, d->wcap, d->info_dev, s1, k * nb));
      launch_f64_to_f32_bf16(s1, d->D64, nb, D32, ld, (__bf16*)nullptr, (int64_t)0, nb, nb, 1);
      CAP_HIP(hipGetLastError());
      CAP_HIP(hipEventRecord(d->ev_fact[k], s1));
      CAP_HIP(hipStreamWaitEvent(sc, d->ev_fact[k], 0));
    } else if (k >= 2) {
      CAP_HIP(hipStreamWaitEvent(sc, d->ev_solved[k - 2], 0));   // the broadcast overwrites the inverse block row k-2 was solved with
    }
    CAP_TRY(cap_comm_bcast(d->comm, Dinv, nb2, owner, (void*)sc));
    CAP_HIP(hipEventRecord(d->ev_msg[k], sc));
    // ---- block row k of my columns, in fp64: S = Dinv^T R[k, mine]; the factor keeps fp32, the update panel bf16
    CAP_HIP(hipStreamWaitEvent(s1, d->ev_msg[k], 0));
    CAP_TRY(cap_copy_rect(Dinv, nb, d->Dall + k * nb2, nb, nb, nb, s1));
    if (ncols > 0) {
      CapRange range("CI::trsm");
      float* Row32 = d->R32 + k * nb + lbk * nb * ld;
      launch_f32_to_f64(s1, Row32, ld, d->T64, nb, nb, ncols, 0);
      CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, ncols, nb, 1.0, Dinv, nb, d->T64, nb, 0.0, d->S64, nb, 0, s1, 2 | 16));
      launch_f64_to_f32_bf16(s1, d->S64, nb, Row32, ld, Pk, nb, nb, ncols, 0);
      CAP_HIP(hipGetLastError());
    }
    CAP_HIP(hipEventRecord(d->ev_solved[k], s1));
    if (k + 1 >= nblk) continue;

    // ---- all-gather of the bf16 panel pieces (equal-sized, padded): a quarter of the fp64 strip's bytes
    int gstart[8];
    int64_t nmax = 0;
    for (int64_t r = 0; r < 8; r++) gstart[r] = r < P ? (int)lbfirst_(r, k, P) : 0;
    for (int64_t r = 0; r < P; r++) nmax = std::max(nmax, (nblocks_of_(r, nblk, P) - lbfirst_(r, k, P)) * nb);
    const int64_t piece = nb * nmax;                             // bf16 elements per piece
    CAP_HIP(hipStreamWaitEvent(sc, d->ev_solved[k], 0));         // also: s1 is past HEAD(k-2), the panel stream's reader of Gk
    if (k >= 2) CAP_HIP(hipStreamWaitEvent(sc, d->ev_bulk[k - 2], 0));
    CAP_TRY(cap_comm_allgather(d->comm, reinterpret_cast<const double*>(Pk), reinterpret_cast<double*>(Gk), piece / 4, (void*)sc));
    CAP_HIP(hipEventRecord(d->ev_gather[k], sc));

    // ---- HEAD: block row k + 1 of my columns J >= k + 1 (panel stream: the next diagonal block and block row need it)
    CAP_HIP(hipStreamWaitEvent(s1, d->ev_gather[k], 0));
    if (k >= 1) CAP_HIP(hipStreamWaitEvent(s1, d->ev_bulk[k - 1], 0));      // bulk(k-1) updates block row k + 1 too: fixed order
    if (ncols > 0) {
      CapRange range("CI::tmu");
      CAP_TRY(cap_bf16_dist_update_launch(nb, ncols, nb, Gk, piece, gstart, Gk + p * piece, d->R32 + (k + 1) * nb + lbk * nb * ld, ld, (int)P, (int)p,
                                          (int)nb, (int)(k + 1), (int)lbk, s1));
    }
    CAP_HIP(hipEventRecord(d->ev_head[k], s1));
    // ---- bulk: rows from block k + 2 on, my columns J >= k + 2, caller's stream
    CAP_HIP(hipStreamWaitEvent(s0, d->ev_gather[k], 0));
    if (k + 2 < nblk) {
      const int64_t lb2 = lbfirst_(p, k + 1, P), ncols2 = (d->nloc_blocks - lb2) * nb;
      if (ncols2 > 0) {
        CapRange range("CI::tmu");
        CAP_TRY(cap_bf16_dist_update_launch(npad - (k + 2) * nb, ncols2, nb, Gk, piece, gstart, Gk + p * piece + (lb2 - lbk) * nb * nb,
                                            d->R32 + (k + 2) * nb + lb2 * nb * ld, ld, (int)P, (int)p, (int)nb, (int)(k + 2), (int)lb2, s0));
      }
    }
    CAP_HIP(hipEventRecord(d->ev_bulk[k], s0));
  }
  CAP_HIP(hipEventRecord(d->ev_join_p, s1));
  CAP_HIP(hipEventRecord(d->ev_join_c, sc));
  CAP_HIP(hipStreamWaitEvent(s0, d->ev_join_p, 0));
  CAP_HIP(hipStreamWaitEvent(s0, d->ev_join_c, 0));
  // fp64 promotion of my block columns for the refinement sweeps
  if (d->lc > 0) {
    launch_f32_to_f64(s0, d->R32, ld, d->R64, ld, npad, d->lc, 0);
    CAP_HIP(hipGetLastError());
  }
  d->have_r64 = true;
  return CAP_OK;
}

// 0, or the smallest 1-based failing pivot any owner reported.  Collective.
int cap_dmp_info(cap_dmp_plan* d, void* stream, int64_t* info) {
  if (!d || !info) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  CAP_TRY(cap_drain_streams({d->s_panel, d->s_comm}));
  double* mine = d->info_red + d->P;
  cap_acc_r(d->info_dev, 1, 1, 1, 0, 4); cap_acc_w(mine, 1, 1, 1);
  hipLaunchKernelGGL(info_to_double3, dim3(1), dim3(1), 0, s, d->info_dev, mine);
  CAP_HIP(hipGetLastError());
  CAP_TRY(cap_comm_allgather(d->comm, mine, d->info_red, 1, stream));
  double h[8] = {0};
  CAP_HIP(hipMemcpyAsync(h, d->info_red, sizeof(double) * d->P, hipMemcpyDeviceToHost, s));
  CAP_HIP(hipStreamSynchronize(s));
  int64_t best = 0;
  for (int r = 0; r < d->P; r++) { const int64_t v = (int64_t)h[r]; if (v > 0 && (best == 0 || v < best)) best = v; }
  *info = best;
  return best == 0 ? CAP_OK : CAP_ERR_NOT_SPD;
}

// Solve A X = B to fp64 accuracy on every rank (B, X: n x nrhs, the same on all ranks; Alocal: my block columns of the FULL
// symmetric A).  Returns sweeps used and the final ||B - A X||_F / ||B||_F.  Synchronises the stream once per sweep.
int cap_dmp_solve(cap_dmp_plan* d, const double* Aloc, int64_t lda, const double* B, int64_t ldb, double* X, int64_t ldx, int64_t nrhs,
                  int max_iter, double tol, int* iters, double* relres, void* stream) {
  if (!d || !B || !X || ldb < d->n || ldx < d->n || nrhs <= 0 || nrhs > d->w || !d->have_r64) return CAP_ERR_ARG;
  if (d->lc_valid > 0 && (!Aloc || lda < d->n)) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  const int64_t n = d->n, npad = d->npad, nb = d->nb, nblk = d->nblk, P = d->P, p = d->p, w = cap_round_up(nrhs, 128), nb2 = nb * nb;
  double* Tmp = d->Tmp;                                          // nb x w, contiguous (collective payload)
  auto rows_grid = [&](int64_t rows) { return dim3((unsigned)std::min<int64_t>(cap_ceil_div(rows, 256), 4096), (unsigned)w); };
  auto apply_Ainv = [&](double* V) -> int {                      // V (npad x w, replicated) <- R^-1 R^-T V
    for (int64_t i = 0; i < nblk; i++) {                         // forward: R^T y = v
      const int owner = (int)(i % P);
      if (p == owner) {
        const double* Rcol = d->R64 + (i / P) * nb * npad;       // rows 0 .. of my block column i
        if (i > 0) CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, w, i * nb, -1.0, Rcol, npad, V, npad, 1.0, V + i * nb, npad, 0, s));
        CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, w, nb, 1.0, d->Dall + i * nb2, nb, V + i * nb, npad, 0.0, Tmp, nb, 0, s, 16));
      }
      CAP_TRY(cap_comm_bcast(d->comm, Tmp, nb * w, owner, (void*)s));
      CAP_TRY(cap_copy_rect(Tmp, nb, V + i * nb, npad, nb, w, s));
    }
    CAP_HIP(hipMemsetAsync(d->Acc, 0, sizeof(double) * npad * w, s));
    for (int64_t i = nblk - 1; i >= 0; i--) {                    // backward: R x = y
      const int owner = (int)(i % P);
      CAP_TRY(cap_copy_rect(d->Acc + i * nb, npad, Tmp, nb, nb, w, s));
      CAP_TRY(cap_comm_allreduce_sum(d->comm, Tmp, nb * w, (void*)s));
      cap_acc_rw(Tmp, nb, nb, w); cap_acc_r(V + i * nb, npad, nb, w);
      hipLaunchKernelGGL(sub_from_kernel, rows_grid(nb), dim3(256), 0, s, Tmp, nb, V + i * nb, npad, nb, w);
      CAP_HIP(hipGetLastError());
      CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, nb, w, nb, 1.0, d->Dall + i * nb2, nb, Tmp, nb, 0.0, V + i * nb, npad, 0, s, 32));
      if (p == owner && i > 0) {
        const double* Rcol = d->R64 + (i / P) * nb * npad;
        CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, i * nb, w, nb, 1.0, Rcol, npad, V + i * nb, npad, 1.0, d->Acc, npad, 0, s));
      }
    }
    return CAP_OK;
  };
  // zero-padded working copies: B, X (= first solve)
  CAP_HIP(hipMemsetAsync(d->Bw, 0, sizeof(double) * npad * w, s));
  CAP_TRY(cap_copy_rect(B, ldb, d->Bw, npad, n, nrhs, s));
  CAP_HIP(hipMemcpyAsync(d->V, d->Bw, sizeof(double) * npad * w, hipMemcpyDeviceToDevice, s));
  CAP_TRY(apply_Ainv(d->V));
  CAP_TRY(cap_sumsq(d->Bw, npad, npad, w, 0, 0, d->norms, stream));
  double h[2] = {0, 0};
  int it = 0; double rr = 0, prev = 1e300;
  for (;;) {
    // r = b - A x: my rows of A x from my block columns (A symmetric), scattered and summed over the ranks
    CAP_HIP(hipMemsetAsync(d->Rw, 0, sizeof(double) * npad * w, s));
    if (d->lc_valid > 0) {
      CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, d->lc_valid, w, n, 1.0, Aloc, lda, d->V, npad, 0.0, d->Q, d->lc_valid, 0, s));
      cap_acc_r(d->Q, d->lc_valid, d->lc_valid, w); cap_acc_w(d->Rw, npad, npad, w);       // (my rows of the replicated buffer: noted as the whole of it)
      hipLaunchKernelGGL(scatter_rows_bc_kernel, rows_grid(d->lc_valid), dim3(256), 0, s, d->Q, d->lc_valid, d->Rw, npad, nb, (int)P, (int)p,
                         d->lc_valid, w);
      CAP_HIP(hipGetLastError());
    }
    CAP_TRY(cap_comm_allreduce_sum(d->comm, d->Rw, npad * w, (void*)s));
    cap_acc_rw(d->Rw, npad, npad, w); cap_acc_r(d->Bw, npad, npad, w);
    hipLaunchKernelGGL(sub_from_kernel, rows_grid(npad), dim3(256), 0, s, d->Rw, npad, d->Bw, npad, npad, w);     // Rw <- Bw - Rw
    CAP_HIP(hipGetLastError());
    CAP_TRY(cap_sumsq(d->Rw, npad, npad, w, 0, 0, d->norms + 1, stream));
    CAP_HIP(hipMemcpyAsync(h, d->norms, sizeof(double) * 2, hipMemcpyDeviceToHost, s));
    CAP_HIP(hipStreamSynchronize(s));
    rr = h[0] > 0 ? std::sqrt(h[1] / h[0]) : 0.0;
    if (rr <= tol || it >= max_iter || !(rr == rr) || (it >= 2 && rr > 0.5 * prev && rr < 1e-10)) break;
    prev = rr;
    CAP_TRY(apply_Ainv(d->Rw));
    cap_acc_rw(d->V, npad, npad, w); cap_acc_r(d->Rw, npad, npad, w);
    hipLaunchKernelGGL(axpy_cols_kernel, rows_grid(npad), dim3(256), 0, s, d->V, npad, d->Rw, npad, npad, w);
    CAP_HIP(hipGetLastError());
    it++;
  }
  CAP_TRY(cap_copy_rect(d->V, npad, X, ldx, n, nrhs, s));
  if (iters) *iters = it;
  if (relres) *relres = rr;
  return CAP_OK;
}

}  // extern "C"
