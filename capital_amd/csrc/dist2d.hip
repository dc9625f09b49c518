// Multi-GPU blocked Cholesky (upper, A = R^T R) on a 2D Pr x Pc BLOCK-CYCLIC process grid, one process per GPU, RCCL over xGMI.
//
// Layout (the "2D block-cyclic matrix descriptor" of the north star; upstream distributes ELEMENT-cyclically over a d x d x c
// grid, matrix.hpp:8-11, topology.h:67-143): block (I, J) of nb x nb elements lives on process (I % Pr, J % Pc) as local block
// (I / Pr, J / Pc); rank = pr * Pc + pc; the local array is column-major (local rows x local columns).  N is padded to a
// multiple of nb with an identity tail (policy.h:196 `span`).  Pr must divide Pc (1 x P, 2 x 2, 2 x 4, 4 x 4 ...).
//
// Right-looking step k (block row k; owner process row prk = k % Pr, owner column pck = k % Pc) - the roles of upstream's
// SUMMA collectives (summa.hpp:163-253: row broadcast, column broadcast, base-case gather policy.h:160-305) with RCCL:
//   1. (prk, pck)            factor + invert the diagonal block (the fused 64-blocked chain of leaf.hip)
//   2. process row prk       ncclBroadcast of Dinv(k) along the row communicator                                [nb^2]
//   3. process row prk       S_k = Dinv^T R[k, my columns J > k]                                               (1 GEMM)
//   4. every process column  ncclBroadcast of S_k's piece down the column communicator (root prk): the B operand
//                            S_k[:, J = pc mod Pc]                                                             [nb x n / Pc]
//   5. every process row     the A operand S_k[:, I = pr mod Pr]: those blocks sit, after 4., on the columns pc' = pr mod Pr,
//                            + Pr, ... of my own process row - each of these Pc / Pr contributors broadcasts its piece along
//                            the row communicator (the "transpose" exchange of a symmetric update, util.hpp:232-247)
//                                                                                                              [nb x n / Pr]
//   6. every process         C[I, J] -= S_k[:, I]^T S_k[:, J] on its local blocks k < I <= J: the staircase MFMA update of
//                            gemm.hip, generalised to row-cyclic C (GemmArgs::rP); HEAD = the block row k + 1 first (panel
//                            stream, feeds the next diagonal block), bulk = the rows below on the caller's stream.
// Per step a process receives nb x (n / Pc + n / Pr) doubles instead of the 1 x P layout's nb x n - but it issues
// 2 + Pc / Pr collectives on two communicator families instead of 1 broadcast + 1/2 all-gather (DESIGN.md section 5 compares the
// measured launch counts).  Look-ahead depth 1: the panel of step k + 1 overlaps with the bulk update of step k.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "common.h"

int cap_rec_cholinv_full(double* R, int64_t ldr, double* Ri, int64_t ldi, int64_t n, double* W, int64_t wcap, int* info,
                         hipStream_t s, int64_t info_base = 0);
int64_t cap_rec_work_size(int64_t n);

struct cap_dist2d_plan {
  int64_t n, npad, nb, nblk;
  int Pr, Pc, pr, pc, P, rank;
  cap_comm *world, *row, *col; bool owns_row, owns_col;
  int64_t nlr, nlc;              // local row / column blocks
  int64_t lr_valid, lc_valid;    // local rows / columns that exist in the n x n matrix
  double* R; int64_t ld;         // (nlr nb) x (nlc nb)
  double* B[2];                  // B operand / my solved piece of block row k: nb x (nlc nb), ld = nb
  double* A[2];                  // A operand: Pc / Pr pieces of nb x apiece_cols
  double* Dinv[2];               // nb x nb
  double* W; int64_t wcap;
  int64_t apiece;                // doubles per A piece (nb x max local columns of any process column)
  int* info_dev; double* info_red;
  hipStream_t s_panel, s_comm;
  std::vector<hipEvent_t> ev_fact, ev_msg, ev_solved, ev_gather, ev_head, ev_bulk;
  hipEvent_t ev_init, ev_join_p, ev_join_c;
  int64_t occ1_m;
  int64_t cnt_gemm, cnt_chain, cnt_copy, cnt_coll;      // launches / collectives of the LAST factor call (this rank)
};

namespace {
__host__ __device__ inline int64_t lbfirst2(int64_t r, int64_t k, int64_t P) { return k >= r ? (k - r) / P + 1 : 0; }   // blocks J <= k owned by r
__host__ __device__ inline int64_t nblocks_of2(int64_t r, int64_t nblk, int64_t P) { return r < nblk ? (nblk - 1 - r) / P + 1 : 0; }

// valid elements (inside the n x n matrix) among the local blocks of process coordinate q of Q
int64_t valid_extent(int64_t n, int64_t nb, int64_t Q, int64_t q) {
  const int64_t nblk = (n + nb - 1) / nb;
  const int64_t nl = nblocks_of2(q, nblk, Q);
  if (nl == 0) return 0;
  int64_t e = nl * nb;
  if ((nblk - 1) % Q == q) e -= nblk * nb - n;
  return e;
}

// upstream's distribute_symmetric (structure.hpp:68-103) for this layout: closed-form drand48 of the global index pair
__global__ void fill_symmetric_bc2d_kernel(double* out, int64_t ld, int64_t n, int64_t nb, int Pr, int Pc, int pr, int pc, int dom,
                                           int64_t lr_valid, int64_t lc_valid) {
  const int64_t lrow = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t lcol = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (lrow >= lr_valid || lcol >= lc_valid) return;
  const int64_t grow = ((lrow / nb) * Pr + pr) * nb + lrow % nb;
  const int64_t gcol = ((lcol / nb) * Pc + pc) * nb + lcol % nb;
  const int64_t hi = gcol > grow ? gcol : grow, lo = gcol > grow ? grow : gcol;
  const uint64_t seed = (uint64_t)(hi + n * lo);
  const uint64_t x0 = ((seed & 0xFFFFFFFFull) << 16) | 0x330Eull;
  const uint64_t x1 = (0x5DEECE66Dull * x0 + 0xBull) & ((1ull << 48) - 1);
  double v = (double)x1 * (1.0 / 281474976710656.0);
  if (dom && gcol == grow) v += (double)n;
  out[lrow + lcol * ld] = v;
}

// local piece of A -> R, with the identity tail of the padded matrix (rows / columns >= n)
__global__ void import_pad_2d_kernel(const double* A, int64_t lda, double* R, int64_t ld, int64_t n, int64_t nb, int Pr, int Pc, int pr, int pc,
                                     int64_t lrows, int64_t lcols) {
  const int64_t lrow = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t lcol = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (lrow >= lrows || lcol >= lcols) return;
  const int64_t grow = ((lrow / nb) * Pr + pr) * nb + lrow % nb;
  const int64_t gcol = ((lcol / nb) * Pc + pc) * nb + lcol % nb;
  R[lrow + lcol * ld] = (grow < n && gcol < n) ? A[lrow + lcol * lda] : (grow == gcol ? 1.0 : 0.0);
}

// construct_R: my valid local piece, zero below the GLOBAL diagonal
__global__ void export_upper_2d_kernel(const double* R, int64_t ld, double* out, int64_t ldo, int64_t nb, int Pr, int Pc, int pr, int pc,
                                       int64_t lr_valid, int64_t lc_valid) {
  const int64_t lrow = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t lcol = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (lrow >= lr_valid || lcol >= lc_valid) return;
  const int64_t grow = ((lrow / nb) * Pr + pr) * nb + lrow % nb;
  const int64_t gcol = ((lcol / nb) * Pc + pc) * nb + lcol % nb;
  out[lrow + lcol * ldo] = grow <= gcol ? R[lrow + lcol * ld] : 0.0;
}

__global__ void info_to_double2(const int* info, double* out) { *out = (double)*info; }

dim3 grid_cols(int64_t rows, int64_t cols) {
  return dim3((unsigned)cap_ceil_div(std::max<int64_t>(rows, 1), 256), (unsigned)std::min<int64_t>(std::max<int64_t>(cols, 1), 65535),
              (unsigned)cap_ceil_div(std::max<int64_t>(cols, 1), 65535));
}

int ensure_events2(cap_dist2d_plan* d) {
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
  return CAP_OK;
}
}  // namespace

extern "C" {

// valid local rows (which = 0) / columns (which = 1) of process (pr, pc)
int64_t cap_bc2d_local_extent(int64_t n, int64_t nb, int Pr, int Pc, int pr, int pc, int which) {
  if (n <= 0 || nb <= 0 || Pr < 1 || Pc < 1 || pr < 0 || pr >= Pr || pc < 0 || pc >= Pc) return -1;
  return which == 0 ? valid_extent(n, nb, Pr, pr) : valid_extent(n, nb, Pc, pc);
}

int cap_fill_symmetric_bc2d(double* local, int64_t ld, int64_t n, int64_t nb, int Pr, int Pc, int pr, int pc, int diagonally_dominant, void* stream) {
  if (!local || n <= 0 || nb <= 0 || Pr < 1 || Pc < 1 || pr < 0 || pr >= Pr || pc < 0 || pc >= Pc) return CAP_ERR_ARG;
  const int64_t lr = valid_extent(n, nb, Pr, pr), lc = valid_extent(n, nb, Pc, pc);
  if (lr == 0 || lc == 0) return CAP_OK;
  if (ld < lr) return CAP_ERR_ARG;
  hipLaunchKernelGGL(fill_symmetric_bc2d_kernel, grid_cols(lr, lc), dim3(256), 0, cap_stream(stream), local, ld, n, nb, Pr, Pc, pr, pc,
                     diagonally_dominant, lr, lc);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

// world: communicator of all P = Pr Pc ranks (rank = pr Pc + pc).  row / col: the communicators of my process row (Pc
// ranks, ordered by pc) and column (Pr ranks, ordered by pr); NULL = split them off `world` here (ncclCommSplit), as
// topo::square does (topology.h:84-94).  nb: block size, multiple of 128 (0 = 512).
int cap_dist2d_plan_create(cap_dist2d_plan** plan, int64_t n, int64_t nb, cap_comm* world, int Pr, cap_comm* row, cap_comm* col) {
  if (!plan || n <= 0 || Pr < 1) return CAP_ERR_ARG;
  if (nb <= 0) nb = 512;
  if (nb % 128) return CAP_ERR_UNSUPPORTED;
  const int P = cap_comm_size(world), rank = cap_comm_rank(world);
  if (P % Pr) return CAP_ERR_ARG;
  const int Pc = P / Pr;
  if (Pc % Pr) return CAP_ERR_UNSUPPORTED;                 // the A-operand contributors are whole process columns only if Pr | Pc
  if (Pc > 8) return CAP_ERR_UNSUPPORTED;                  // the update kernel carries 8 piece offsets
  cap_dist2d_plan* d = new (std::nothrow) cap_dist2d_plan();
  if (!d) return CAP_ERR_ALLOC;
  d->n = n; d->nb = nb; d->nblk = cap_ceil_div(n, nb); d->npad = d->nblk * nb;
  d->Pr = Pr; d->Pc = Pc; d->P = P; d->rank = rank; d->pr = rank / Pc; d->pc = rank % Pc;
  d->world = world; d->row = row; d->col = col; d->owns_row = d->owns_col = false;
  d->R = nullptr; d->W = nullptr; d->info_dev = nullptr; d->info_red = nullptr;
  for (int i = 0; i < 2; i++) d->B[i] = d->A[i] = d->Dinv[i] = nullptr;
  d->s_panel = d->s_comm = nullptr;
  d->occ1_m = getenv("CAP_OCC1_M") ? atoll(getenv("CAP_OCC1_M")) : 16384;
  d->cnt_gemm = d->cnt_chain = d->cnt_copy = d->cnt_coll = 0;
  if (!d->row && P > 1) {
    int st = cap_comm_split(world, d->pr, d->pc, &d->row);
    if (st != CAP_OK) { delete d; return st; }
    d->owns_row = true;
  }
  if (!d->col && P > 1) {
    int st = cap_comm_split(world, d->pc, d->pr, &d->col);
    if (st != CAP_OK) { if (d->owns_row) cap_comm_destroy(d->row); delete d; return st; }
    d->owns_col = true;
  }
  if (P > 1 && (cap_comm_size(d->row) != Pc || cap_comm_rank(d->row) != d->pc || cap_comm_size(d->col) != Pr || cap_comm_rank(d->col) != d->pr)) {
    cap_dist2d_plan_destroy(d);
    return CAP_ERR_ARG;
  }
  d->nlr = nblocks_of2(d->pr, d->nblk, Pr); d->nlc = nblocks_of2(d->pc, d->nblk, Pc);
  d->lr_valid = valid_extent(n, nb, Pr, d->pr); d->lc_valid = valid_extent(n, nb, Pc, d->pc);
  d->ld = std::max<int64_t>(d->nlr * nb, 2);
  const int64_t maxcols = nblocks_of2(0, d->nblk, Pc) * nb;        // process column 0 owns the most blocks
  d->apiece = nb * maxcols;
  d->wcap = cap_rec_work_size(nb);
  hipError_t e = hipMalloc((void**)&d->R, sizeof(double) * std::max<int64_t>(d->ld * d->nlc * nb, 2));
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc((void**)&d->B[i], sizeof(double) * nb * (maxcols + nb));
    if (e == hipSuccess) e = hipMemset(d->B[i], 0, sizeof(double) * nb * (maxcols + nb));
    if (e == hipSuccess) e = hipMalloc((void**)&d->A[i], sizeof(double) * d->apiece * (Pc / Pr));
    if (e == hipSuccess) e = hipMalloc((void**)&d->Dinv[i], sizeof(double) * nb * nb);
    if (e == hipSuccess) e = hipMemset(d->Dinv[i], 0, sizeof(double) * nb * nb);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&d->W, sizeof(double) * d->wcap);
  if (e == hipSuccess) e = hipMalloc((void**)&d->info_dev, sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&d->info_red, sizeof(double) * (P + 1));
  if (e != hipSuccess) { cap_dist2d_plan_destroy(d); return CAP_ERR_ALLOC; }
  *plan = d;
  return CAP_OK;
}

int cap_dist2d_plan_destroy(cap_dist2d_plan* d) {
  if (!d) return CAP_OK;
  if (d->R) (void)hipFree(d->R);
  for (int i = 0; i < 2; i++) { if (d->B[i]) (void)hipFree(d->B[i]); if (d->A[i]) (void)hipFree(d->A[i]); if (d->Dinv[i]) (void)hipFree(d->Dinv[i]); }
  if (d->W) (void)hipFree(d->W);
  if (d->info_dev) (void)hipFree(d->info_dev);
  if (d->info_red) (void)hipFree(d->info_red);
  if (!d->ev_msg.empty()) {
    for (auto* v : {&d->ev_fact, &d->ev_msg, &d->ev_solved, &d->ev_gather, &d->ev_head, &d->ev_bulk})
      for (auto e : *v) (void)hipEventDestroy(e);
    (void)hipEventDestroy(d->ev_init); (void)hipEventDestroy(d->ev_join_p); (void)hipEventDestroy(d->ev_join_c);
    (void)hipStreamDestroy(d->s_panel); (void)hipStreamDestroy(d->s_comm);
  }
  if (d->owns_row && d->row) cap_comm_destroy(d->row);
  if (d->owns_col && d->col) cap_comm_destroy(d->col);
  delete d;
  return CAP_OK;
}

// which: 0 valid local rows, 1 valid local columns, 2 Pr, 3 Pc, 4 pr, 5 pc, 6 nb, 7 padded n,
//        8..11 launches of the last factor call on this rank: MFMA GEMM / update kernels, diagonal-block chains, copy kernels, collectives
int64_t cap_dist2d_get(const cap_dist2d_plan* d, int which) {
  if (!d) return -1;
  switch (which) {
    case 0: return d->lr_valid; case 1: return d->lc_valid; case 2: return d->Pr; case 3: return d->Pc; case 4: return d->pr; case 5: return d->pc;
    case 6: return d->nb; case 7: return d->npad; case 8: return d->cnt_gemm; case 9: return d->cnt_chain; case 10: return d->cnt_copy;
    case 11: return d->cnt_coll;
  }
  return -1;
}

double* cap_dist2d_R_ptr(cap_dist2d_plan* d, int64_t* ld) { if (!d) return nullptr; if (ld) *ld = d->ld; return d->R; }

// Alocal: this process's valid local piece (cap_bc2d_local_extent rows x columns, column-major, lda >= rows), read-only
int cap_dist2d_factor(cap_dist2d_plan* d, const double* Aloc, int64_t lda, void* stream) {
  if (!d) return CAP_ERR_ARG;
  if (d->lr_valid > 0 && d->lc_valid > 0 && (!Aloc || lda < d->lr_valid)) return CAP_ERR_ARG;
  CAP_TRY(ensure_events2(d));
  CapRange frange("CI::factor");
  hipStream_t s0 = cap_stream(stream), s1 = d->s_panel, sc = d->s_comm;
  const int64_t nb = d->nb, nblk = d->nblk, ld = d->ld, nb2 = nb * nb;
  const int Pr = d->Pr, Pc = d->Pc, pr = d->pr, pc = d->pc, ncon = Pc / Pr;
  d->cnt_gemm = d->cnt_chain = d->cnt_copy = d->cnt_coll = 0;
  CAP_HIP(hipMemsetAsync(d->info_dev, 0, sizeof(int), s0));
  if (d->nlr > 0 && d->nlc > 0) {
    hipLaunchKernelGGL(import_pad_2d_kernel, grid_cols(d->nlr * nb, d->nlc * nb), dim3(256), 0, s0, Aloc, lda, d->R, ld, d->n, nb, Pr, Pc, pr, pc,
                       d->nlr * nb, d->nlc * nb);
    CAP_HIP(hipGetLastError());
    d->cnt_copy++;
  }
  CAP_HIP(hipEventRecord(d->ev_init, s0));
  CAP_HIP(hipStreamWaitEvent(s1, d->ev_init, 0));
  CAP_HIP(hipStreamWaitEvent(sc, d->ev_init, 0));

  for (int64_t k = 0; k < nblk; k++) {
    const int par = (int)(k & 1);
    const int prk = (int)(k % Pr), pck = (int)(k % Pc);
    const bool in_row = pr == prk, owner = in_row && pc == pck;
    double* Dinv = d->Dinv[par]; double* Bk = d->B[par]; double* Ak = d->A[par];
    const int64_t lbk = lbfirst2(pc, k, Pc);                    // my local column blocks with J <= k
    const int64_t ncols = (d->nlc - lbk) * nb;                  // my columns J > k
    // ---- 1. diagonal block (needs every earlier update of block row k: HEAD(k-1) is on this stream, bulk(k-2) by event)
    if (k >= 2) CAP_HIP(hipStreamWaitEvent(s1, d->ev_bulk[k - 2], 0));
    if (owner) {
      CapRange range("CI::factor_diag");
      double* D = d->R + (k / Pr) * nb + (k / Pc) * nb * ld;
      CAP_TRY(cap_rec_cholinv_full(D, ld, Dinv, nb, nb, d->W, d->wcap, d->info_dev, s1, k * nb));
      d->cnt_chain++;
      CAP_HIP(hipEventRecord(d->ev_fact[k], s1));
    }
    // ---- 2. Dinv(k) along process row prk
    if (in_row) {
      if (owner) CAP_HIP(hipStreamWaitEvent(sc, d->ev_fact[k], 0));
      else if (k >= 2) CAP_HIP(hipStreamWaitEvent(sc, d->ev_solved[k - 2], 0));     // the broadcast overwrites the inverse step k-2 solved with
      if (Pc > 1) { CAP_TRY(cap_comm_bcast(d->row, Dinv, nb2, pck, (void*)sc)); d->cnt_coll++; }
      CAP_HIP(hipEventRecord(d->ev_msg[k], sc));
      // ---- 3. my part of block row k
      CAP_HIP(hipStreamWaitEvent(s1, d->ev_msg[k], 0));
      if (ncols > 0) {
        CapRange range("CI::trsm");
        double* Rrow = d->R + (k / Pr) * nb + lbk * nb * ld;
        CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, ncols, nb, 1.0, Dinv, nb, Rrow, ld, 0.0, Bk, nb, 0, s1, 2 | 16));
        CAP_TRY(cap_copy_rect(Bk, nb, Rrow, ld, nb, ncols, s1));
        d->cnt_gemm++; d->cnt_copy++;
      }
    }
    CAP_HIP(hipEventRecord(d->ev_solved[k], s1));
    if (k + 1 >= nblk) continue;                                 // last block row: nothing below it

    // ---- 4. B operand: S_k[:, my columns J > k] down my process column (root: process row prk)
    CAP_HIP(hipStreamWaitEvent(sc, d->ev_solved[k], 0));         // root: the piece is solved; others: s1 has passed HEAD(k-2), the last reader of Bk on it
    if (k >= 2) CAP_HIP(hipStreamWaitEvent(sc, d->ev_bulk[k - 2], 0));   // bulk(k-2) read Bk / Ak
    if (Pr > 1 && ncols > 0) { CAP_TRY(cap_comm_bcast(d->col, Bk, nb * ncols, prk, (void*)sc)); d->cnt_coll++; }
    // ---- 5. A operand: blocks I = pr mod Pr of S_k, held (after 4.) by the columns pc' = pr mod Pr + m Pr of my process row
    int gstart[8];
    for (int m = 0; m < 8; m++) gstart[m] = 0;
    for (int m = 0; m < ncon; m++) {
      const int pcs = pr % Pr + m * Pr;                          // contributor's column coordinate
      const int64_t lbs = lbfirst2(pcs, k, Pc), cs = (nblocks_of2(pcs, nblk, Pc) - lbs) * nb;
      gstart[m] = (int)lbs;
      if (cs <= 0) continue;
      double* slot = Ak + (int64_t)m * d->apiece;
      if (pcs == pc) { CAP_TRY(cap_copy_rect(Bk, nb, slot, nb, nb, cs, sc)); d->cnt_copy++; }
      if (Pc > 1) { CAP_TRY(cap_comm_bcast(d->row, slot, nb * cs, pcs, (void*)sc)); d->cnt_coll++; }
    }
    CAP_HIP(hipEventRecord(d->ev_gather[k], sc));

    // ---- 6. updates with step k: local blocks k < I <= J
    const int64_t lbc1 = lbfirst2(pc, k, Pc);                    // first local column block with J >= k + 1
    const int64_t ncols1 = (d->nlc - lbc1) * nb;
    // HEAD: block row k + 1 (on its process row), panel stream - the next diagonal block and block row depend on it
    CAP_HIP(hipStreamWaitEvent(s1, d->ev_gather[k], 0));
    if (k >= 1) CAP_HIP(hipStreamWaitEvent(s1, d->ev_bulk[k - 1], 0));      // bulk(k-1) also updates block row k + 1
    if ((int)((k + 1) % Pr) == pr && ncols1 > 0) {
      CapRange range("CI::tmu");
      const int64_t rl = (k + 1) / Pr;
      CAP_TRY(cap_dist_update_launch(nb, ncols1, nb, Ak, d->apiece, gstart, Bk + (lbc1 - lbk) * nb * nb, d->R + rl * nb + lbc1 * nb * ld, ld, Pc, pc,
                                     (int)nb, (int)(k + 1), (int)lbc1, s1, 0, Pr, pr, (int)rl));
      d->cnt_gemm++;
    }
    CAP_HIP(hipEventRecord(d->ev_head[k], s1));
    // bulk: local block rows I >= k + 2, columns J >= k + 2, caller's stream
    CAP_HIP(hipStreamWaitEvent(s0, d->ev_gather[k], 0));
    if (k + 2 < nblk) {
      const int64_t rl2 = lbfirst2(pr, k + 1, Pr), lbc2 = lbfirst2(pc, k + 1, Pc);
      const int64_t mrows = (d->nlr - rl2) * nb, ncols2 = (d->nlc - lbc2) * nb;
      if (mrows > 0 && ncols2 > 0) {
        CapRange range("CI::tmu");
        const int occ = (d->occ1_m > 0 && (double)mrows * (double)ncols2 <= (double)d->occ1_m * (double)d->occ1_m) ? -1 : 0;
        CAP_TRY(cap_dist_update_launch(mrows, ncols2, nb, Ak, d->apiece, gstart, Bk + (lbc2 - lbk) * nb * nb, d->R + rl2 * nb + lbc2 * nb * ld, ld, Pc,
                                       pc, (int)nb, (int)(k + 2), (int)lbc2, s0, occ, Pr, pr, (int)rl2));
        d->cnt_gemm++;
      }
    }
    CAP_HIP(hipEventRecord(d->ev_bulk[k], s0));
  }
  CAP_HIP(hipEventRecord(d->ev_join_p, s1));
  CAP_HIP(hipEventRecord(d->ev_join_c, sc));
  CAP_HIP(hipStreamWaitEvent(s0, d->ev_join_p, 0));
  CAP_HIP(hipStreamWaitEvent(s0, d->ev_join_c, 0));
  return CAP_OK;
}

// construct_R (cholinv.hpp:30-37) for this layout: my valid local piece of R, zero below the global diagonal
int cap_dist2d_get_R(cap_dist2d_plan* d, double* out, int64_t ldo, void* stream) {
  if (!d) return CAP_ERR_ARG;
  if (d->lr_valid == 0 || d->lc_valid == 0) return CAP_OK;
  if (!out || ldo < d->lr_valid) return CAP_ERR_ARG;
  hipLaunchKernelGGL(export_upper_2d_kernel, grid_cols(d->lr_valid, d->lc_valid), dim3(256), 0, cap_stream(stream), d->R, d->ld, out, ldo, d->nb,
                     d->Pr, d->Pc, d->pr, d->pc, d->lr_valid, d->lc_valid);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

// 0, or the smallest 1-based failing pivot any diagonal owner reported.  Collective over the world communicator.
int cap_dist2d_info(cap_dist2d_plan* d, void* stream, int64_t* info) {
  if (!d || !info) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  double* mine = d->info_red + d->P;
  hipLaunchKernelGGL(info_to_double2, dim3(1), dim3(1), 0, s, d->info_dev, mine);
  CAP_HIP(hipGetLastError());
  CAP_TRY(cap_comm_allgather(d->world, mine, d->info_red, 1, stream));
  std::vector<double> h((size_t)d->P, 0.0);
  CAP_HIP(hipMemcpyAsync(h.data(), d->info_red, sizeof(double) * d->P, hipMemcpyDeviceToHost, s));
  CAP_HIP(hipStreamSynchronize(s));
  int64_t best = 0;
  for (int r = 0; r < d->P; r++) { const int64_t v = (int64_t)h[(size_t)r]; if (v > 0 && (best == 0 || v < best)) best = v; }
  *info = best;
  return best == 0 ? CAP_OK : CAP_ERR_NOT_SPD;
}

int cap_dist2d_set_option(cap_dist2d_plan* d, const char* key, int64_t value) {
  if (!d || !key) return CAP_ERR_ARG;
  if (!strcmp(key, "occ1_m")) { if (value < 0) return CAP_ERR_ARG; d->occ1_m = value; return CAP_OK; }
  return CAP_ERR_ARG;
}

}  // extern "C"
