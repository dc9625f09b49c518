// Multi-GPU blocked Cholesky (upper, A = R^T R) on a 2D Pr x Pc BLOCK-CYCLIC process grid, one process per GPU, RCCL over xGMI.
//
// Layout (the "2D block-cyclic matrix descriptor" of the north star; upstream distributes ELEMENT-cyclically over a d x d x c
// grid, matrix.hpp:8-11, topology.h:67-143): block (I, J) of nb x nb elements lives on process (I % Pr, J % Pc) as local block
// (I / Pr, J / Pc); rank = pr * Pc + pc; the local array is column-major (local rows x local columns).  N is padded to a
// multiple of nb with an identity tail (policy.h:196 `span`).  Pr must divide Pc (1 x P, 2 x 2, 2 x 4, 4 x 4 ...).
//
// Schedule (round 4: the same two-level blocking and look-ahead as the 1 x P plan of dist.hip): strips of `strip` = 2 block rows
// (a, b = a + 1), one K = 2 nb update per strip.  Owner process row prk = k % Pr, owner column pck = k % Pc; the roles of upstream's
// SUMMA collectives (summa.hpp:163-253: row broadcast, column broadcast, base-case gather policy.h:160-305) with RCCL:
//   per block row k of the strip (panel stream + the message stream):
//     1. (prk, pck)            [row b: D(b) -= S_a(:, b)^T S_a(:, b) first]  factor + invert the diagonal block (fused 64-blocked chain)
//     2. process row prk       msg(k) = [ R(a, b) | Dinv(k) ] along the row communicator                            [2 nb^2]
//     3. process row prk       [row b: R[b, J] -= R(a, b)^T S_a(:, J)]   S_k = Dinv^T R[k, my columns J > k]          (1-2 GEMMs)
//     4. every process column  S_k's piece down the column communicator (root prk) into the strip buffer (K-contiguous, ld = 2 nb):
//                              the B operand, and what row b's in-strip update and the inverse read                  [nb x n / Pc]
//   per strip:
//     5. every process row     the A operand S[:, I = pr mod Pr]: those blocks sit, after 4., on the columns pc' = pr mod Pr, + Pr, ... of
//                              my own process row - each of these Pc / Pr contributors broadcasts its strip piece along the row
//                              communicator (the "transpose" exchange of a symmetric update, util.hpp:232-247)    [2 nb x n / Pr]
//     6. every process         C[I, J] -= S[:, I]^T S[:, J] on its local blocks b < I <= J, K = 2 nb: the staircase MFMA update of gemm.hip,
//                              generalised to row-cyclic C (GemmArgs::rP); HEAD = the rows of strip t + 1 (panel stream), bulk = the rows
//                              below on the caller's stream, itself split into the rows of strip t + 2 (signals the panel stream:
//                              look-ahead depth 2) + the rest.
// Per strip a process receives 2 nb x (n / Pc + n / Pr) doubles instead of the 1 x P layout's 2 nb x n, for 4 + Pc / Pr collectives
// (two messages, two column broadcasts, Pc / Pr row broadcasts) against 2 + 1 (DESIGN.md section 5 compares measured launch counts).
// Options: "strip" (1 | 2), "depth2", "occ1_m", "complete_inv" / "split" (R^-1 streamed with the sweep like dist.hip: per block row
// one more nb^2 broadcast down the owner's process column and one row broadcast of the finished block column of R^-1), "ipc" (both
// operand moves as IPC peer copies on SDMA engines instead of RCCL broadcasts).
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
  cap_comm* row2;                // the small messages of a process row travel on their own communicator + stream (owned duplicate)
  int64_t nlr, nlc;              // local row / column blocks
  int64_t lr_valid, lc_valid;    // local rows / columns that exist in the n x n matrix
  double* R; int64_t ld;         // (nlr nb) x (nlc nb)
  double* Bt[2];                 // solved block row k, my columns J > k, contiguous (ld = nb): payload of the column broadcast; ring over block rows
  double* S[3];                  // strip buffer: the strip's solved rows at my columns J > a, K-contiguous (ld = q nb); ring of THREE over
                                 // strips: it is the B operand of the bulk update of its strip, which runs two strips behind the panel
  double* A[2];                  // A operand: Pc / Pr pieces of (q nb) x maxcols; ring over strips
  double* msg[4];                // [ R(a, b) | Dinv(k) ], ring over block rows
  double* W; int64_t wcap;
  int64_t maxcols, apiece;       // widest local column count of any process column; doubles per A piece
  int* info_dev; double* info_red;
  hipStream_t s_panel, s_comm, s_msg;
  std::vector<hipEvent_t> ev_fact, ev_msg, ev_rowdone, ev_colb, ev_inv;      // per block row
  std::vector<hipEvent_t> ev_gather, ev_head, ev_head2, ev_rest;              // per strip
  hipEvent_t ev_init, ev_join_p, ev_join_c, ev_join_m, ev_join_i;
  int strip, depth2, safe;
  int64_t occ1_m;
  // replay aids (single-GPU replay of one rank, tools/replay.py): what a foreign owner's diagonal-block chain / a foreign process row's block-row
  // solve take, spun on the message / communication stream behind my own panel stream's position (0: off)
  int remote_chain_us, remote_solve_us;
  int64_t cnt_gemm, cnt_chain, cnt_copy, cnt_coll;      // launches / collectives of the LAST factor call (this rank)
  // R^-1 streamed with the sweep (complete_inv = 0 / 1): see inverse_step2d
  int complete_inv; int64_t split;
  double* Ri;                    // my piece of R^-1 (same shape as R): the partial product X_k while the sweep runs
  double* Dall;                  // Dinv(k) of the diagonal blocks I own (nblk slots)
  double* Dc[2];                 // Dinv(k) on the owner's process column (nb x nb), ring
  double* Xc[2];                 // the finished block column k of R^-1 at my rows ((nlr nb) x nb), ring
  cap_comm *row3, *col3;         // the inverse's broadcasts (owned duplicates)
  hipStream_t s_inv;
  // option "ipc": both operand moves as IPC peer copies (SDMA engines, one xGMI link per peer) instead of RCCL broadcasts - the column
  // move pushes the solved row into the payload buffers of my process column, the row move pushes my strip piece into slot m of the A
  // buffers of my process row; two 8-byte all-reduces per move carry the synchronisation (buffers free / pushes landed), like dist.hip
  int ipc; bool ipc_ready, ipc_failed;
  double* peerBt[4][2]; double* peerA[8][2];
  hipStream_t s_pcol[4], s_prow[8]; hipEvent_t ev_px, ev_pcol[4], ev_prow[8];
  double* tok;                   // [0] barrier token, [1] agreement flag, then handle scratch
};

namespace {
// busy-wait for about `us` microseconds (wall_clock64 ticks at 100 MHz): the replay aids' stand-in for work another rank does
__global__ void spin_kernel_2d(int us) {
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < (uint64_t)us * 100ull) __builtin_amdgcn_s_sleep(32);
}
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
  CAP_TRY(mk(d->ev_fact, cnt)); CAP_TRY(mk(d->ev_msg, cnt)); CAP_TRY(mk(d->ev_rowdone, cnt)); CAP_TRY(mk(d->ev_colb, cnt)); CAP_TRY(mk(d->ev_inv, cnt));
  CAP_TRY(mk(d->ev_gather, cnt)); CAP_TRY(mk(d->ev_head, cnt)); CAP_TRY(mk(d->ev_head2, cnt)); CAP_TRY(mk(d->ev_rest, cnt));
  for (hipEvent_t* e : {&d->ev_init, &d->ev_join_p, &d->ev_join_c, &d->ev_join_m, &d->ev_join_i}) CAP_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
  int lo = 0, hi = 0;
  CAP_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CAP_HIP(hipStreamCreateWithPriority(&d->s_panel, hipStreamNonBlocking, hi));
  CAP_HIP(hipStreamCreateWithPriority(&d->s_comm, hipStreamNonBlocking, hi));
  CAP_HIP(hipStreamCreateWithPriority(&d->s_msg, hipStreamNonBlocking, hi));
  CAP_HIP(hipStreamCreateWithPriority(&d->s_inv, hipStreamNonBlocking, lo));
  return CAP_OK;
}

// X_{-1} = I on my piece: local column l (global g) carries a 1 at global row g if that row is mine
__global__ void identity_2d_kernel(double* X, int64_t ld, int64_t nb, int Pr, int Pc, int pr, int pc, int64_t lcols) {
  const int64_t l = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (l >= lcols) return;
  const int64_t J = (l / nb) * Pc + pc;
  if ((int)(J % Pr) != pr) return;
  X[(J / Pr) * nb + l % nb + l * ld] = 1.0;
}
// complete_inv == 0 with the root partition inside a block: clear Ri[global rows < n1, global columns >= n1]
__global__ void zero_root_2d_kernel(double* X, int64_t ld, int64_t nb, int Pr, int Pc, int pr, int pc, int64_t lrows, int64_t lcols, int64_t n1) {
  const int64_t lrow = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t lcol = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (lrow >= lrows || lcol >= lcols) return;
  const int64_t grow = ((lrow / nb) * Pr + pr) * nb + lrow % nb, gcol = ((lcol / nb) * Pc + pc) * nb + lcol % nb;
  if (grow < n1 && gcol >= n1) X[lrow + lcol * ld] = 0.0;
}

struct Cut2 { bool cut; int64_t kcut, n1; };
inline Cut2 inv_cut2(const cap_dist2d_plan* d) {
  const int64_t n1 = d->n >> d->split;
  const bool cut = d->complete_inv == 0 && n1 > 0 && n1 < d->n;
  return Cut2{cut, (cut && n1 % d->nb == 0) ? n1 / d->nb : 0, n1};
}

// Step k of the streamed inverse on the Pr x Pc grid (the left-to-right product of dist.hip's inverse_step): block column k of R^-1
// lives on process column pck with its rows spread over the process rows.
//   a. Dinv(k) down process column pck (root prk)                                                              [nb^2]
//   b. process column pck: X[my rows <= k, k] <- X[my rows <= k, k] Dinv(k)        (the column is final)
//   c. that piece along every process row (root pck)                                                           [(k + 1) nb / Pr x nb]
//   d. every process: X[my rows <= k, my J > k] -= X[my rows <= k, k] R[k, my J]   (R[k, my J] = the solved row in my strip buffer)
// Srow: block row k at my columns J > k inside the strip buffer (ld = ldS, column origin = local block lbk).
int inverse_step2d(cap_dist2d_plan* d, int64_t k, const double* Srow, int64_t ldS, hipStream_t s, cap_comm* crow, cap_comm* ccol) {
  CapRange range("CI::inverse");
  const int64_t nb = d->nb, nb2 = nb * nb, ld = d->ld;
  const int Pr = d->Pr, Pc = d->Pc, pr = d->pr, pc = d->pc;
  const int prk = (int)(k % Pr), pck = (int)(k % Pc);
  const Cut2 ic = inv_cut2(d);
  const int64_t r0 = (ic.kcut > 0 && k >= ic.kcut) ? lbfirst2(pr, ic.kcut - 1, Pr) : 0;      // my first active row block
  const int64_t r1 = lbfirst2(pr, k, Pr);                                                    // my row blocks I <= k
  const int64_t rows = (r1 - r0) * nb;
  double* Dk = d->Dc[k & 1]; double* Xk = d->Xc[k & 1];
  if (pc == pck) {
    if (pr == prk) CAP_TRY(cap_copy_rect(d->Dall + k * nb2, nb, Dk, nb, nb, nb, s));
    if (Pr > 1) { CAP_TRY(cap_comm_bcast(ccol, Dk, nb2, prk, (void*)s)); d->cnt_coll++; }
    if (rows > 0) {
      double* Xcol = d->Ri + r0 * nb + (k / Pc) * nb * ld;
      CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, rows, nb, nb, 1.0, Xcol, ld, Dk, nb, 0.0, Xk, rows, 0, s, 8));
      CAP_TRY(cap_copy_rect(Xk, rows, Xcol, ld, rows, nb, s));
      d->cnt_gemm++; d->cnt_copy++;
    }
  }
  if (rows > 0 && Pc > 1) { CAP_TRY(cap_comm_bcast(crow, Xk, rows * nb, pck, (void*)s)); d->cnt_coll++; }
  const int64_t lb0 = lbfirst2(pc, k, Pc);
  int64_t lb1 = d->nlc;
  if (ic.kcut > 0 && k < ic.kcut) lb1 = lbfirst2(pc, ic.kcut - 1, Pc);
  if (rows > 0 && lb1 > lb0) {
    CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, rows, (lb1 - lb0) * nb, nb, -1.0, Xk, rows, Srow, ldS, 1.0, d->Ri + r0 * nb + lb0 * nb * ld, ld, 0, s));
    d->cnt_gemm++;
  }
  return CAP_OK;
}

// map the column peers' payload buffers and the row peers' A buffers (collective over the row and column communicators).  Local
// failures only clear `ok`; every rank runs every collective; one failing rank sends everybody back to the RCCL broadcasts.
int ensure_ipc2d(cap_dist2d_plan* d, hipStream_t s) {
  if (d->ipc_ready || d->ipc_failed || d->P == 1) return CAP_OK;
  auto soft = [](hipError_t e) { if (e != hipSuccess) { (void)hipGetLastError(); return false; } return true; };
  const int Pr = d->Pr, Pc = d->Pc;
  bool ok = true;                              // (d->tok: allocated with the plan)
  auto gather_handles = [&](cap_comm* comm, int np, int me, double* const bufs[2], double* scratch_dev, std::vector<hipIpcMemHandle_t>& all) -> int {
    hipIpcMemHandle_t mine[2];
    for (int b = 0; b < 2; b++) ok = ok && soft(hipIpcGetMemHandle(&mine[b], bufs[b]));
    if (!ok) memset(mine, 0, sizeof(mine));
    if (!soft(hipMemcpyAsync(scratch_dev + 16 * np, mine, 128, hipMemcpyHostToDevice, s))) ok = false;
    CAP_TRY(cap_comm_allgather(comm, scratch_dev + 16 * np, scratch_dev, 16, (void*)s));
    all.assign((size_t)2 * np, hipIpcMemHandle_t());
    memset(all.data(), 0, all.size() * sizeof(hipIpcMemHandle_t));
    if (!soft(hipMemcpyAsync(all.data(), scratch_dev, (size_t)128 * np, hipMemcpyDeviceToHost, s)) || !soft(hipStreamSynchronize(s))) ok = false;
    (void)me;
    return CAP_OK;
  };
  std::vector<hipIpcMemHandle_t> hc, hr;
  double* const bt[2] = {d->Bt[0], d->Bt[1]}; double* const aa[2] = {d->A[0], d->A[1]};
  if (Pr > 1) CAP_TRY(gather_handles(d->col, Pr, d->pr, bt, d->tok + 2, hc));
  if (Pc > 1) CAP_TRY(gather_handles(d->row, Pc, d->pc, aa, d->tok + 2 + 16 * (Pr + 1), hr));
  hipIpcMemHandle_t zero; memset(&zero, 0, sizeof(zero));
  auto open_all = [&](const std::vector<hipIpcMemHandle_t>& all, int np, int me, double* (*peer)[2], hipStream_t* streams, hipEvent_t* events) {
    for (int r = 0; r < np && ok; r++) {
      if (r == me) continue;
      for (int b = 0; b < 2 && ok; b++) {
        if (!memcmp(&all[(size_t)2 * r + b], &zero, sizeof(zero))) { ok = false; break; }
        void* q = nullptr;
        if (!soft(hipIpcOpenMemHandle(&q, all[(size_t)2 * r + b], hipIpcMemLazyEnablePeerAccess))) { ok = false; break; }
        peer[r][b] = (double*)q;
      }
      if (ok) ok = soft(hipStreamCreateWithFlags(&streams[r], hipStreamNonBlocking)) && soft(hipEventCreateWithFlags(&events[r], hipEventDisableTiming));
    }
  };
  if (Pr > 1 && ok) open_all(hc, Pr, d->pr, d->peerBt, d->s_pcol, d->ev_pcol);
  if (Pc > 1 && ok) open_all(hr, Pc, d->pc, d->peerA, d->s_prow, d->ev_prow);
  if (ok) ok = soft(hipEventCreateWithFlags(&d->ev_px, hipEventDisableTiming));
  double flag = ok ? 0.0 : 1.0, tot = 0.0;
  if (!soft(hipMemcpyAsync(d->tok + 1, &flag, sizeof(double), hipMemcpyHostToDevice, s))) (void)hipMemsetAsync(d->tok + 1, 0x7f, sizeof(double), s);
  CAP_TRY(cap_comm_allreduce_sum(d->world, d->tok + 1, 1, (void*)s));
  if (!soft(hipMemcpyAsync(&tot, d->tok + 1, sizeof(double), hipMemcpyDeviceToHost, s)) || !soft(hipStreamSynchronize(s))) tot = 1.0;
  (void)hipMemsetAsync(d->tok + 1, 0, sizeof(double), s);
  if (tot != 0.0 || !ok) {
    for (int r = 0; r < 4; r++) for (int b = 0; b < 2; b++) if (d->peerBt[r][b]) { (void)hipIpcCloseMemHandle(d->peerBt[r][b]); d->peerBt[r][b] = nullptr; }
    for (int r = 0; r < 8; r++) for (int b = 0; b < 2; b++) if (d->peerA[r][b]) { (void)hipIpcCloseMemHandle(d->peerA[r][b]); d->peerA[r][b] = nullptr; }
    // ... and the copy streams / events open_all created before the failure (dist.hip's ensure_ipc does the same)
    for (int r = 0; r < 4; r++) {
      if (d->s_pcol[r]) { cap_stream_destroy(d->s_pcol[r]); d->s_pcol[r] = nullptr; }
      if (d->ev_pcol[r]) { (void)hipEventDestroy(d->ev_pcol[r]); d->ev_pcol[r] = nullptr; }
    }
    for (int r = 0; r < 8; r++) {
      if (d->s_prow[r]) { cap_stream_destroy(d->s_prow[r]); d->s_prow[r] = nullptr; }
      if (d->ev_prow[r]) { (void)hipEventDestroy(d->ev_prow[r]); d->ev_prow[r] = nullptr; }
    }
    if (d->ev_px) { (void)hipEventDestroy(d->ev_px); d->ev_px = nullptr; }
    d->ipc_failed = true;
    fprintf(stderr, "capital_amd: IPC mapping of the 2D plan's peer buffers failed on %s rank; using the RCCL broadcasts\n", ok ? "another" : "this");
    return CAP_OK;
  }
  d->ipc_ready = true;
  return CAP_OK;
}

// one operand move as peer copies: `count` doubles from `src` to offset `off` of buffer `b` of every peer of the communicator (np ranks,
// I am `me`); roots push, everybody runs the two barriers.  sc: the communication stream.
int push_move(cap_dist2d_plan* d, cap_comm* comm, int np, int me, bool i_push, const double* src, double* (*peer)[2], int b, int64_t off, int64_t count,
              hipStream_t* streams, hipEvent_t* events, hipStream_t sc) {
  CAP_TRY(cap_comm_allreduce_sum(comm, d->tok, 1, (void*)sc)); d->cnt_coll++;            // every target buffer is free
  if (i_push && count > 0) {
    CAP_HIP(hipEventRecord(d->ev_px, sc));
    for (int r = 0; r < np; r++) {
      if (r == me) continue;
      CAP_HIP(hipStreamWaitEvent(streams[r], d->ev_px, 0));
      CAP_HIP(hipMemcpyAsync(peer[r][b] + off, src, sizeof(double) * count, hipMemcpyDeviceToDevice, streams[r]));
      CAP_HIP(hipEventRecord(events[r], streams[r]));
      d->cnt_copy++;
    }
    for (int r = 0; r < np; r++) if (r != me) CAP_HIP(hipStreamWaitEvent(sc, events[r], 0));
  }
  CAP_TRY(cap_comm_allreduce_sum(comm, d->tok, 1, (void*)sc)); d->cnt_coll++;            // every push has landed
  return CAP_OK;
}

int ensure_inverse2d(cap_dist2d_plan* d) {
  if (d->Ri) return CAP_OK;
  hipError_t e = hipMalloc((void**)&d->Ri, sizeof(double) * std::max<int64_t>(d->ld * d->nlc * d->nb, 2));
  if (e == hipSuccess) e = hipMalloc((void**)&d->Dall, sizeof(double) * d->nblk * d->nb * d->nb);
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc((void**)&d->Dc[i], sizeof(double) * d->nb * d->nb);
    if (e == hipSuccess) e = hipMalloc((void**)&d->Xc[i], sizeof(double) * std::max<int64_t>(d->nlr, 1) * d->nb * d->nb);
  }
  if (e != hipSuccess) {
    (void)hipGetLastError();
    for (double** q : {&d->Ri, &d->Dall, &d->Dc[0], &d->Dc[1], &d->Xc[0], &d->Xc[1]}) if (*q) { (void)hipFree(*q); *q = nullptr; }
    return CAP_ERR_ALLOC;
  }
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
  cap_acc_w(local, ld, lr, lc);
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
  if (Pc > 8 || 4 % Pr) return CAP_ERR_UNSUPPORTED;       // the update kernel carries 8 piece offsets; the message ring of 4 needs Pr | 4
  cap_dist2d_plan* d = new (std::nothrow) cap_dist2d_plan();
  if (!d) return CAP_ERR_ALLOC;
  d->n = n; d->nb = nb; d->nblk = cap_ceil_div(n, nb); d->npad = d->nblk * nb;
  d->Pr = Pr; d->Pc = Pc; d->P = P; d->rank = rank; d->pr = rank / Pc; d->pc = rank % Pc;
  d->world = world; d->row = row; d->col = col; d->owns_row = d->owns_col = false; d->row2 = d->row3 = d->col3 = nullptr;
  d->R = nullptr; d->W = nullptr; d->info_dev = nullptr; d->info_red = nullptr;
  for (int i = 0; i < 2; i++) d->Bt[i] = d->A[i] = d->Dc[i] = d->Xc[i] = nullptr;
  for (int i = 0; i < 3; i++) d->S[i] = nullptr;
  for (int i = 0; i < 4; i++) d->msg[i] = nullptr;
  d->Ri = d->Dall = nullptr;
  d->s_panel = d->s_comm = d->s_msg = d->s_inv = nullptr;
  d->remote_chain_us = d->remote_solve_us = 0;
  d->occ1_m = 16384 * (int64_t)(P > 1 ? P : 1);        // grows with the rank count like the 1 x P plan's (dist.hip: measured by the single-GPU replay)
  d->strip = d->nblk >= 8 ? 2 : 1; d->depth2 = 1; d->safe = 0; d->complete_inv = -1; d->split = 1;
  d->ipc = 0; d->ipc_ready = d->ipc_failed = false; d->tok = nullptr; d->ev_px = nullptr;
  for (int r = 0; r < 4; r++) { d->peerBt[r][0] = d->peerBt[r][1] = nullptr; d->s_pcol[r] = nullptr; d->ev_pcol[r] = nullptr; }
  for (int r = 0; r < 8; r++) { d->peerA[r][0] = d->peerA[r][1] = nullptr; d->s_prow[r] = nullptr; d->ev_prow[r] = nullptr; }
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
  if (P > 1) {      // the small messages and the inverse's broadcasts run next to the operand moves: their own communicators
    int st = cap_comm_dup(d->row, &d->row2);
    if (st == CAP_OK) st = cap_comm_dup(d->row, &d->row3);
    if (st == CAP_OK) st = cap_comm_dup(d->col, &d->col3);
    if (st != CAP_OK) { cap_dist2d_plan_destroy(d); return st; }
  }
  d->nlr = nblocks_of2(d->pr, d->nblk, Pr); d->nlc = nblocks_of2(d->pc, d->nblk, Pc);
  d->lr_valid = valid_extent(n, nb, Pr, d->pr); d->lc_valid = valid_extent(n, nb, Pc, d->pc);
  d->ld = std::max<int64_t>(d->nlr * nb, 2);
  d->maxcols = nblocks_of2(0, d->nblk, Pc) * nb;           // process column 0 owns the most blocks
  d->apiece = 2 * nb * d->maxcols;
  d->wcap = cap_rec_work_size(nb);
  hipError_t e = hipMalloc((void**)&d->R, sizeof(double) * std::max<int64_t>(d->ld * d->nlc * nb, 2));
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc((void**)&d->Bt[i], sizeof(double) * nb * (d->maxcols + nb));
    if (e == hipSuccess) e = hipMalloc((void**)&d->A[i], sizeof(double) * d->apiece * (Pc / Pr));
  }
  for (int i = 0; i < 3 && e == hipSuccess; i++) {
    e = hipMalloc((void**)&d->S[i], sizeof(double) * 2 * nb * (d->maxcols + nb));
    if (e == hipSuccess) e = hipMemset(d->S[i], 0, sizeof(double) * 2 * nb * (d->maxcols + nb));
  }
  for (int i = 0; i < 4 && e == hipSuccess; i++) {
    e = hipMalloc((void**)&d->msg[i], sizeof(double) * 2 * nb * nb);
    if (e == hipSuccess) e = hipMemset(d->msg[i], 0, sizeof(double) * 2 * nb * nb);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&d->W, sizeof(double) * d->wcap);
  if (e == hipSuccess) e = hipMalloc((void**)&d->info_dev, sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&d->info_red, sizeof(double) * (P + 1));
  // the token / handle scratch of the IPC operand moves lives with the plan: an allocation that fails is reported HERE, by every
  // rank's own create call, and never in the middle of ensure_ipc2d's collectives (where the peers would wait for ever)
  const int64_t tok_elems = 2 + 16 * (Pr + Pc + 2);
  if (e == hipSuccess) e = hipMalloc((void**)&d->tok, sizeof(double) * tok_elems);
  if (e == hipSuccess) e = hipMemset(d->tok, 0, sizeof(double) * tok_elems);
  if (e != hipSuccess) { cap_dist2d_plan_destroy(d); return CAP_ERR_ALLOC; }
  // helper streams and events exist from here on (round 6: never created inside the first factor call)
  { const int st = ensure_events2(d); if (st != CAP_OK) { cap_dist2d_plan_destroy(d); return st; } }
  *plan = d;
  return CAP_OK;
}

int cap_dist2d_plan_destroy(cap_dist2d_plan* d) {
  if (!d) return CAP_OK;
  for (double* q : {d->R, d->Bt[0], d->Bt[1], d->S[0], d->S[1], d->S[2], d->A[0], d->A[1], d->msg[0], d->msg[1], d->msg[2], d->msg[3], d->W, d->info_red, d->Ri,
                    d->Dall, d->Dc[0], d->Dc[1], d->Xc[0], d->Xc[1]})
    if (q) (void)hipFree(q);
  if (d->info_dev) (void)hipFree(d->info_dev);
  if (!d->ev_msg.empty()) {
    for (auto* v : {&d->ev_fact, &d->ev_msg, &d->ev_rowdone, &d->ev_colb, &d->ev_inv, &d->ev_gather, &d->ev_head, &d->ev_head2, &d->ev_rest})
      for (auto e : *v) (void)hipEventDestroy(e);
    for (hipEvent_t e : {d->ev_init, d->ev_join_p, d->ev_join_c, d->ev_join_m, d->ev_join_i}) (void)hipEventDestroy(e);
    for (hipStream_t st : {d->s_panel, d->s_comm, d->s_msg, d->s_inv}) { (void)hipStreamSynchronize(st); cap_stream_destroy(st); }
  }
  for (int r = 0; r < 4; r++) {
    for (int b = 0; b < 2; b++) if (d->peerBt[r][b]) (void)hipIpcCloseMemHandle(d->peerBt[r][b]);
    if (d->s_pcol[r]) { (void)hipStreamSynchronize(d->s_pcol[r]); cap_stream_destroy(d->s_pcol[r]); }
    if (d->ev_pcol[r]) (void)hipEventDestroy(d->ev_pcol[r]);
  }
  for (int r = 0; r < 8; r++) {
    for (int b = 0; b < 2; b++) if (d->peerA[r][b]) (void)hipIpcCloseMemHandle(d->peerA[r][b]);
    if (d->s_prow[r]) { (void)hipStreamSynchronize(d->s_prow[r]); cap_stream_destroy(d->s_prow[r]); }
    if (d->ev_prow[r]) (void)hipEventDestroy(d->ev_prow[r]);
  }
  if (d->ev_px) (void)hipEventDestroy(d->ev_px);
  if (d->tok) (void)hipFree(d->tok);
  for (cap_comm* c : {d->row2, d->row3, d->col3}) if (c) cap_comm_destroy(c);
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
    case 12: return (d->ipc && d->ipc_ready) ? 1 : 0;        // the operand moves really run as IPC peer copies
  }
  return -1;
}

double* cap_dist2d_R_ptr(cap_dist2d_plan* d, int64_t* ld) { if (!d) return nullptr; if (ld) *ld = d->ld; return d->R; }
double* cap_dist2d_Rinv_ptr(cap_dist2d_plan* d, int64_t* ld) { if (!d || !d->Ri) return nullptr; if (ld) *ld = d->ld; return d->Ri; }

// Alocal: this process's valid local piece (cap_bc2d_local_extent rows x columns, column-major, lda >= rows), read-only
int cap_dist2d_factor(cap_dist2d_plan* d, const double* Aloc, int64_t lda, void* stream) {
  if (!d) return CAP_ERR_ARG;
  if (d->lr_valid > 0 && d->lc_valid > 0 && (!Aloc || lda < d->lr_valid)) return CAP_ERR_ARG;
  CAP_TRY(ensure_events2(d));
  const bool inv = d->complete_inv >= 0;
  if (inv) CAP_TRY(ensure_inverse2d(d));
  if (d->ipc) CAP_TRY(ensure_ipc2d(d, cap_stream(stream)));
  const bool ipc = d->ipc && d->ipc_ready;
  CapRange frange("CI::factor");
  hipStream_t s0 = cap_stream(stream), s1 = d->s_panel, sc = d->s_comm, sm = d->safe ? d->s_comm : d->s_msg;
  cap_comm* rmsg = (d->safe || !d->row2) ? d->row : d->row2;
  const bool inv_overlap = inv && !d->safe;
  hipStream_t si = inv_overlap ? d->s_inv : s0;
  cap_comm* rinv = (inv_overlap && d->row3) ? d->row3 : d->row;
  cap_comm* cinv = (inv_overlap && d->col3) ? d->col3 : d->col;
  const int64_t nb = d->nb, nblk = d->nblk, ld = d->ld, nb2 = nb * nb;
  const int Pr = d->Pr, Pc = d->Pc, pr = d->pr, pc = d->pc, ncon = Pc / Pr;
  d->cnt_gemm = d->cnt_chain = d->cnt_copy = d->cnt_coll = 0;
  CAP_HIP(hipMemsetAsync(d->info_dev, 0, sizeof(int), s0));
  if (d->nlr > 0 && d->nlc > 0) {
    cap_acc_r(Aloc, lda, d->lr_valid, d->lc_valid); cap_acc_w(d->R, ld, d->nlr * nb, d->nlc * nb);
    hipLaunchKernelGGL(import_pad_2d_kernel, grid_cols(d->nlr * nb, d->nlc * nb), dim3(256), 0, s0, Aloc, lda, d->R, ld, d->n, nb, Pr, Pc, pr, pc,
                       d->nlr * nb, d->nlc * nb);
    CAP_HIP(hipGetLastError());
    d->cnt_copy++;
  }
  if (inv) {
    CAP_HIP(hipMemsetAsync(d->Ri, 0, sizeof(double) * std::max<int64_t>(ld * d->nlc * nb, 2), s0));
    if (d->nlr > 0 && d->nlc > 0) {
      cap_acc_w(d->Ri, ld, d->nlr * nb, d->nlc * nb);
      hipLaunchKernelGGL(identity_2d_kernel, dim3((unsigned)cap_ceil_div(d->nlc * nb, 256)), dim3(256), 0, s0, d->Ri, ld, nb, Pr, Pc, pr, pc, d->nlc * nb);
      CAP_HIP(hipGetLastError());
    }
  }
  CAP_HIP(hipEventRecord(d->ev_init, s0));
  for (hipStream_t st : {s1, sc, sm}) CAP_HIP(hipStreamWaitEvent(st, d->ev_init, 0));
  if (inv_overlap) CAP_HIP(hipStreamWaitEvent(si, d->ev_init, 0));

  // strips: sb[t] = first block row, sq[t] = block rows (d->strip, the last one may be shorter)
  std::vector<int64_t> sb, sq;
  for (int64_t k = 0; k < nblk; k += d->strip) { sb.push_back(k); sq.push_back(std::min<int64_t>(d->strip, nblk - k)); }
  const int64_t nstrips = (int64_t)sb.size();

  for (int64_t t = 0; t < nstrips; t++) {
    const int par = (int)(t & 1);
    const int64_t a = sb[t], q = sq[t], b = a + q - 1, e = b + 1;
    const int64_t ldS = q * nb;
    const int64_t lbS = lbfirst2(pc, a, Pc);                      // my first local column block with J > a: column origin of S[par]
    double* S = d->S[t % 3]; double* Ak = d->A[par];
    // S[t % 3] was read by the HEAD (panel stream), the bulk update (caller's stream) and the inverse of strip t - 3
    if (t >= 3) {
      CAP_HIP(hipStreamWaitEvent(sc, d->ev_head[t - 3], 0));
      CAP_HIP(hipStreamWaitEvent(sc, d->ev_rest[t - 3], 0));
      if (inv_overlap) CAP_HIP(hipStreamWaitEvent(sc, d->ev_inv[sb[t - 3] + sq[t - 3] - 1], 0));
      if (Pr == 1) {                                             // one process row: the panel stream writes the strip buffer itself
        CAP_HIP(hipStreamWaitEvent(s1, d->ev_rest[t - 3], 0));
        if (inv_overlap) CAP_HIP(hipStreamWaitEvent(s1, d->ev_inv[sb[t - 3] + sq[t - 3] - 1], 0));
      }
    }
    for (int64_t r = 0; r < q; r++) {
      const int64_t k = a + r;
      const int prk = (int)(k % Pr), pck = (int)(k % Pc);
      const bool in_row = pr == prk, owner = in_row && pc == pck;
      double* mb = d->msg[k & 3]; double* Dinv = mb + nb2; double* Bt = d->Bt[k & 1];
      const int64_t lbk = lbfirst2(pc, k, Pc);                    // my local column blocks with J <= k
      const int64_t ncols = (d->nlc - lbk) * nb;                  // my columns J > k
      const int64_t rlk = k / Pr;                                 // local row block of block row k on process row prk
      // ---- 1. diagonal block on its owner (every earlier update of block row k is in: HEAD(t-1) ran on this stream behind the
      //         head of bulk(t-2), which covered the rows of strip t)
      if (owner) {
        double* D = d->R + rlk * nb + (k / Pc) * nb * ld;
        if (r == 1) {
          // in-strip: D(b) -= S_a(:, b)^T S_a(:, b); S_a's column block b came down my process column
          if (Pr > 1) CAP_HIP(hipStreamWaitEvent(s1, d->ev_colb[a], 0));
          const double* Sab = S + (k / Pc - lbS) * nb * ldS;
          CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, nb, nb, -1.0, Sab, ldS, Sab, ldS, 1.0, D, ld, 1, s1, 2)); d->cnt_gemm++;
        }
        {
          CapRange range("CI::factor_diag");
          CAP_TRY(cap_rec_cholinv_full(D, ld, Dinv, nb, nb, d->W, d->wcap, d->info_dev, s1, k * nb)); d->cnt_chain++;
        }
        if (r == 1) { CAP_TRY(cap_copy_rect(S + (k / Pc - lbS) * nb * ldS, ldS, mb, nb, nb, nb, s1)); d->cnt_copy++; }     // R(a, b) rides along
        if (inv) { CAP_TRY(cap_copy_rect(Dinv, nb, d->Dall + k * nb2, nb, nb, nb, s1)); d->cnt_copy++; }
        CAP_HIP(hipEventRecord(d->ev_fact[k], s1));
        CAP_HIP(hipStreamWaitEvent(sm, d->ev_fact[k], 0));
      } else if (in_row) {
        if (k >= 4) CAP_HIP(hipStreamWaitEvent(sm, d->ev_rowdone[k - 4], 0));  // the broadcast overwrites the message block row k - 4 was solved with
        if (d->remote_chain_us > 0) {     // replay: the owner reaches this point when I do and then needs its chain (see dist.hip, remote_chain_us)
          CAP_HIP(hipEventRecord(d->ev_fact[k], s1));
          CAP_HIP(hipStreamWaitEvent(sm, d->ev_fact[k], 0));
          cap_acc_none(); hipLaunchKernelGGL(spin_kernel_2d, dim3(1), dim3(64), 0, sm, d->remote_chain_us); CAP_HIP(hipGetLastError());
        }
      }
      // ---- 2. msg(k) = [ R(a, b) | Dinv(k) ] along process row prk
      if (in_row) {
        if (Pc > 1) { CAP_TRY(cap_comm_bcast(rmsg, mb, 2 * nb2, pck, (void*)sm)); d->cnt_coll++; }
        CAP_HIP(hipEventRecord(d->ev_msg[k], sm));
        // ---- 3. my part of block row k
        CAP_HIP(hipStreamWaitEvent(s1, d->ev_msg[k], 0));
        if (k >= 2) CAP_HIP(hipStreamWaitEvent(s1, d->ev_colb[k - 2], 0));          // Bt[k & 1] was the payload of block row k - 2
        if (ncols > 0) {
          CapRange range("CI::trsm");
          double* Rrow = d->R + rlk * nb + lbk * nb * ld;
          if (r == 1) {   // in-strip update: R[b, mine] -= R(a, b)^T S_a(:, mine)
            if (Pr > 1) CAP_HIP(hipStreamWaitEvent(s1, d->ev_colb[a], 0));
            CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, ncols, nb, -1.0, mb, nb, S + (lbk - lbS) * nb * ldS, ldS, 1.0, Rrow, ld, 0, s1, 2));
            d->cnt_gemm++;
          }
          // one process row: nobody else needs the row - solve straight into the strip buffer (no payload copy, no hop to the
          // communication stream on the strip's critical path), like the 1 x P plan
          double* Sk = S + r * nb + (lbk - lbS) * nb * ldS;
          if (Pr == 1) {
            CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, ncols, nb, 1.0, Dinv, nb, Rrow, ld, 0.0, Sk, ldS, 0, s1, 2 | 16));
            CAP_TRY(cap_copy_rect(Sk, ldS, Rrow, ld, nb, ncols, s1));
          } else {
            CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, ncols, nb, 1.0, Dinv, nb, Rrow, ld, 0.0, Bt, nb, 0, s1, 2 | 16));
            CAP_TRY(cap_copy_rect(Bt, nb, Rrow, ld, nb, ncols, s1));
          }
          d->cnt_gemm++; d->cnt_copy++;
        }
      }
      CAP_HIP(hipEventRecord(d->ev_rowdone[k], s1));
      // ---- 4. S_k's piece down my process column (root: process row prk), then into the strip buffer (rows r nb .., ld = q nb)
      CAP_HIP(hipStreamWaitEvent(sc, d->ev_rowdone[k], 0));
      if (!in_row && ncols > 0 && Pr > 1 && d->remote_chain_us + d->remote_solve_us > 0) {
        // replay: the process row that owns block row k needs its chain, the message and its row solve before the payload can leave
        cap_acc_none(); hipLaunchKernelGGL(spin_kernel_2d, dim3(1), dim3(64), 0, sc, d->remote_chain_us + d->remote_solve_us); CAP_HIP(hipGetLastError());
      }
      if (ncols > 0 && Pr > 1) {
        if (ipc) CAP_TRY(push_move(d, d->col, Pr, pr, pr == prk, Bt, d->peerBt, (int)(k & 1), 0, nb * ncols, d->s_pcol, d->ev_pcol, sc));
        else { CAP_TRY(cap_comm_bcast(d->col, Bt, nb * ncols, prk, (void*)sc)); d->cnt_coll++; }
        CAP_TRY(cap_copy_rect(Bt, nb, S + r * nb + (lbk - lbS) * nb * ldS, ldS, nb, ncols, sc)); d->cnt_copy++;
      }
      CAP_HIP(hipEventRecord(d->ev_colb[k], sc));
      if (inv_overlap) {       // block row k is final and in my strip buffer: step k of the streamed inverse on its own stream
        CAP_HIP(hipStreamWaitEvent(si, d->ev_colb[k], 0));
        CAP_TRY(inverse_step2d(d, k, S + r * nb + (lbk - lbS) * nb * ldS, ldS, si, rinv, cinv));
        CAP_HIP(hipEventRecord(d->ev_inv[k], si));
      }
    }
    if (e >= nblk) continue;                                     // last strip: nothing below it

    // ---- 5. A operand: blocks I = pr mod Pr of the strip, held (after 4.) by the columns pc' = pr mod Pr + m Pr of my process row
    const int64_t lbe = lbfirst2(pc, b, Pc);                      // my first local column block with J >= e
    int gstart[8];
    for (int m = 0; m < 8; m++) gstart[m] = 0;
    const double* G = Ak; int64_t gpiece = d->apiece;
    if (t >= 2) {                                                // A[par] was read by the HEAD and the bulk update of strip t - 2
      CAP_HIP(hipStreamWaitEvent(sc, d->ev_head[t - 2], 0));
      CAP_HIP(hipStreamWaitEvent(sc, d->ev_rest[t - 2], 0));
    }
    if (Pc == 1) {                                               // one process: the strip buffer IS the A operand
      gstart[0] = (int)lbe; G = S + (lbe - lbS) * nb * ldS; gpiece = 0;
    } else {
      for (int m = 0; m < ncon; m++) {
        const int pcs = pr % Pr + m * Pr;                        // contributor's column coordinate
        const int64_t lbs = lbfirst2(pcs, b, Pc), cs = (nblocks_of2(pcs, nblk, Pc) - lbs) * nb;
        gstart[m] = (int)lbs;
        if (cs <= 0) continue;
        double* slot = Ak + (int64_t)m * d->apiece;
        if (pcs == pc) { CAP_TRY(cap_copy_rect(S + (lbe - lbS) * nb * ldS, ldS, slot, ldS, ldS, cs, sc)); d->cnt_copy++; }
        if (ipc) CAP_TRY(push_move(d, d->row, Pc, pc, pcs == pc, slot, d->peerA, par, (int64_t)m * d->apiece, ldS * cs, d->s_prow, d->ev_prow, sc));
        else { CAP_TRY(cap_comm_bcast(d->row, slot, ldS * cs, pcs, (void*)sc)); d->cnt_coll++; }
      }
    }
    CAP_HIP(hipEventRecord(d->ev_gather[t], sc));

    // ---- 6. updates with strip t, K = q nb: local blocks b < I <= J
    const int64_t q1 = sq[t + 1], e2 = e + q1;
    auto upd = [&](int64_t I0, int64_t I1, hipStream_t s, int occ) -> int {     // my row blocks with I0 <= I < I1 (I1 < 0: all below I0)
      const int64_t rl0 = lbfirst2(pr, I0 - 1, Pr), rl1 = I1 < 0 ? d->nlr : lbfirst2(pr, I1 - 1, Pr);
      const int64_t lbc = lbfirst2(pc, I0 - 1, Pc), mrows = (rl1 - rl0) * nb, nc = (d->nlc - lbc) * nb;
      if (mrows <= 0 || nc <= 0) return CAP_OK;
      CapRange range("CI::tmu");
      if (occ < 0) occ = (d->occ1_m > 0 && (double)mrows * (double)nc <= (double)d->occ1_m * (double)d->occ1_m) ? -1 : 0;
      CAP_TRY(cap_dist_update_launch(mrows, nc, ldS, G, gpiece, gstart, S + (lbc - lbS) * nb * ldS, d->R + rl0 * nb + lbc * nb * ld, ld, Pc, pc, (int)nb,
                                     (int)I0, (int)lbc, s, occ, Pr, pr, (int)rl0));
      d->cnt_gemm++;
      return CAP_OK;
    };
    // HEAD: the rows of strip t + 1 (panel stream: its diagonal blocks and block rows depend on it)
    CAP_HIP(hipStreamWaitEvent(s1, d->ev_gather[t], 0));
    if (t >= 1) CAP_HIP(hipStreamWaitEvent(s1, d->ev_head2[t - 1], 0));     // strip t - 1's bulk update has passed these rows
    CAP_TRY(upd(e, e2, s1, 0));
    CAP_HIP(hipEventRecord(d->ev_head[t], s1));
    // bulk: rows below strip t + 1, caller's stream; split so that the rows of strip t + 2 release the panel stream early
    CAP_HIP(hipStreamWaitEvent(s0, d->ev_gather[t], 0));
    if (e2 < nblk) {
      const int64_t q2 = (t + 2 < nstrips) ? sq[t + 2] : 0, e3 = e2 + q2;
      if (d->depth2 && q2 > 0 && e3 < nblk) {
        CAP_TRY(upd(e2, e3, s0, -1));
        CAP_HIP(hipEventRecord(d->ev_head2[t], s0));
        CAP_TRY(upd(e3, -1, s0, -1));
      } else {
        CAP_TRY(upd(e2, -1, s0, -1));
        CAP_HIP(hipEventRecord(d->ev_head2[t], s0));
      }
    } else {
      CAP_HIP(hipEventRecord(d->ev_head2[t], s0));
    }
    CAP_HIP(hipEventRecord(d->ev_rest[t], s0));
  }
  CAP_HIP(hipEventRecord(d->ev_join_p, s1));
  CAP_HIP(hipEventRecord(d->ev_join_c, sc));
  CAP_HIP(hipEventRecord(d->ev_join_m, sm));
  for (hipEvent_t ev : {d->ev_join_p, d->ev_join_c, d->ev_join_m}) CAP_HIP(hipStreamWaitEvent(s0, ev, 0));
  if (inv) {
    if (inv_overlap) {
      CAP_HIP(hipEventRecord(d->ev_join_i, si));
      CAP_HIP(hipStreamWaitEvent(s0, d->ev_join_i, 0));
    } else {
      // safe mode: the steps after the sweep, in program order on the one communicator family; block row k is read back from R
      // (row k lives on process row prk only: one more column broadcast per step into the payload buffer)
      for (int64_t k = 0; k < nblk; k++) {
        const int prk = (int)(k % Pr);
        const int64_t lbk = lbfirst2(pc, k, Pc), ncols = (d->nlc - lbk) * nb;
        double* Bt = d->Bt[k & 1];
        if (ncols > 0) {
          if (pr == prk) CAP_TRY(cap_copy_rect(d->R + (k / Pr) * nb + lbk * nb * ld, ld, Bt, nb, nb, ncols, s0));
          if (Pr > 1) { CAP_TRY(cap_comm_bcast(d->col, Bt, nb * ncols, prk, (void*)s0)); d->cnt_coll++; }
        }
        CAP_TRY(inverse_step2d(d, k, Bt, nb, s0, rinv, cinv));
      }
    }
    const Cut2 ic = inv_cut2(d);
    if (ic.cut && ic.kcut == 0 && d->nlr > 0 && d->nlc > 0) {
      cap_acc_w(d->Ri, ld, d->nlr * nb, d->nlc * nb);
      hipLaunchKernelGGL(zero_root_2d_kernel, grid_cols(d->nlr * nb, d->nlc * nb), dim3(256), 0, s0, d->Ri, ld, nb, Pr, Pc, pr, pc, d->nlr * nb,
                         d->nlc * nb, ic.n1);
      CAP_HIP(hipGetLastError());
    }
  }
  return CAP_OK;
}

// construct_R (cholinv.hpp:30-37) for this layout: my valid local piece of R, zero below the global diagonal
int cap_dist2d_get_R(cap_dist2d_plan* d, double* out, int64_t ldo, void* stream) {
  if (!d) return CAP_ERR_ARG;
  if (d->lr_valid == 0 || d->lc_valid == 0) return CAP_OK;
  if (!out || ldo < d->lr_valid) return CAP_ERR_ARG;
  cap_acc_r(d->R, d->ld, d->lr_valid, d->lc_valid); cap_acc_w(out, ldo, d->lr_valid, d->lc_valid);
  hipLaunchKernelGGL(export_upper_2d_kernel, grid_cols(d->lr_valid, d->lc_valid), dim3(256), 0, cap_stream(stream), d->R, d->ld, out, ldo, d->nb,
                     d->Pr, d->Pc, d->pr, d->pc, d->lr_valid, d->lc_valid);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

// 0, or the smallest 1-based failing pivot any diagonal owner reported.  Collective over the world communicator.
int cap_dist2d_info(cap_dist2d_plan* d, void* stream, int64_t* info) {
  if (!d || !info) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  CAP_TRY(cap_drain_streams({d->s_panel, d->s_comm, d->s_msg, d->s_inv}));
  double* mine = d->info_red + d->P;
  cap_acc_r(d->info_dev, 1, 1, 1, 0, 4); cap_acc_w(mine, 1, 1, 1);
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

// construct_Rinv (cholinv.hpp:39-46) for this layout (options complete_inv = 0 / 1)
int cap_dist2d_get_Rinv(cap_dist2d_plan* d, double* out, int64_t ldo, void* stream) {
  if (!d) return CAP_ERR_ARG;
  if (d->complete_inv < 0 || !d->Ri) return CAP_ERR_UNSUPPORTED;
  if (d->lr_valid == 0 || d->lc_valid == 0) return CAP_OK;
  if (!out || ldo < d->lr_valid) return CAP_ERR_ARG;
  cap_acc_r(d->Ri, d->ld, d->lr_valid, d->lc_valid); cap_acc_w(out, ldo, d->lr_valid, d->lc_valid);
  hipLaunchKernelGGL(export_upper_2d_kernel, grid_cols(d->lr_valid, d->lc_valid), dim3(256), 0, cap_stream(stream), d->Ri, d->ld, out, ldo, d->nb,
                     d->Pr, d->Pc, d->pr, d->pc, d->lr_valid, d->lc_valid);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

// ---- descriptor forms (block-cyclic kind of cap_desc: nb, Pr x Pc, my position) - the layout is checked, not assumed
static int desc_matches_plan2d(const cap_dist2d_plan* d, const cap_desc* m) {
  if (!d || !m) return CAP_ERR_ARG;
  if (cap_desc_get(m, 9) != 1 || cap_desc_get(m, 0) != d->n || cap_desc_get(m, 1) != d->n || cap_desc_get(m, 10) != d->nb) return CAP_ERR_ARG;
  if (cap_desc_get(m, 7) != d->Pr || cap_desc_get(m, 6) != d->Pc || cap_desc_get(m, 12) != d->pr || cap_desc_get(m, 11) != d->pc) return CAP_ERR_ARG;
  if (cap_desc_get(m, 3) != d->lr_valid || cap_desc_get(m, 2) != d->lc_valid) return CAP_ERR_ARG;
  return CAP_OK;
}
int cap_dist2d_factor_desc(cap_dist2d_plan* d, const cap_desc* A, void* stream) {
  CAP_TRY(desc_matches_plan2d(d, A));
  return cap_dist2d_factor(d, cap_desc_data(const_cast<cap_desc*>(A)), cap_desc_get(A, 4), stream);
}
int cap_dist2d_get_R_desc(cap_dist2d_plan* d, cap_desc* R, void* stream) {
  CAP_TRY(desc_matches_plan2d(d, R));
  return cap_dist2d_get_R(d, cap_desc_data(R), cap_desc_get(R, 4), stream);
}
int cap_dist2d_get_Rinv_desc(cap_dist2d_plan* d, cap_desc* Rinv, void* stream) {
  CAP_TRY(desc_matches_plan2d(d, Rinv));
  return cap_dist2d_get_Rinv(d, cap_desc_data(Rinv), cap_desc_get(Rinv, 4), stream);
}

int cap_dist2d_set_option(cap_dist2d_plan* d, const char* key, int64_t value) {
  if (!d || !key) return CAP_ERR_ARG;
  if (!strcmp(key, "occ1_m")) { if (value < 0) return CAP_ERR_ARG; d->occ1_m = value; return CAP_OK; }
  if (!strcmp(key, "strip")) { if (value < 1 || value > 2) return CAP_ERR_ARG; d->strip = (int)value; return CAP_OK; }
  if (!strcmp(key, "depth2")) { d->depth2 = value != 0; return CAP_OK; }
  if (!strcmp(key, "safe")) { d->safe = value != 0; return CAP_OK; }
  if (!strcmp(key, "ipc")) { d->ipc = value != 0; return CAP_OK; }
  if (!strcmp(key, "complete_inv")) {
    if (value < -1 || value > 1) return CAP_ERR_ARG;
    if (value >= 0) CAP_TRY(ensure_inverse2d(d));
    d->complete_inv = (int)value;
    return CAP_OK;
  }
  if (!strcmp(key, "split")) { if (value <= 0) return CAP_ERR_ARG; d->split = value; return CAP_OK; }
  if (!strcmp(key, "remote_chain_us")) { if (value < 0 || value > 1000000) return CAP_ERR_ARG; d->remote_chain_us = (int)value; return CAP_OK; }
  if (!strcmp(key, "remote_solve_us")) { if (value < 0 || value > 1000000) return CAP_ERR_ARG; d->remote_solve_us = (int)value; return CAP_OK; }
  return CAP_ERR_ARG;
}

}  // extern "C"
