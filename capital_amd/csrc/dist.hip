// Multi-GPU blocked Cholesky (upper, A = R^T R), one process per GPU, RCCL over xGMI.
//
// Layout: 1 x P block-column-cyclic (the 2D block-cyclic descriptor with Pr = 1): global block
// column J (nb wide) lives on rank J % P as local block J / P; every rank stores all rows of its
// columns (column-major).  N need not be a multiple of nb: the plan works on the matrix padded to
// npad = ceil(N / nb) nb with an identity tail ([[A, 0], [0, I]] -> [[R, 0], [0, I]]), the way the
// reference pads its base case (`span`, policy.h:196).  The reference distributes element-cyclically
// over a d x d x c grid and moves operands with MPI_Bcast / MPI_Allreduce / MPI_Allgather
// (summa.hpp:163-253, policy.h:160-305); here the same roles are played by
//   * one small ncclBroadcast per block row k:  msg(k) = [ R(k-1,k) | Dinv(k) ]  (2 nb^2 doubles) on its
//     own communicator + stream, so it never queues behind the big collective,
//   * one ncclAllGather per STRIP (two block rows): the solved strip right of itself, 2nb x cols.
//
// Schedule = the single-GPU schedule of cholinv.hip (two-level blocking, look-ahead depth 2), per strip
// t = block rows (a, b = a + 1), on four HIP streams tied together by events only (no host sync):
//   panel stream (high priority)   owner(a): factor + invert D(a)            | msg(a) |
//                                  all: S_a = Dinv(a)^T R[a, mine]           (1 GEMM)
//                                  owner(b): D(b) -= S_a^T S_a, factor + invert | msg(b) = [R(a,b) | Dinv(b)] |
//                                  all: R[b, mine] -= R(a,b)^T S_a;  S_b = Dinv(b)^T R[b, mine]
//                                  after the strip's all-gather: HEAD = K = 2nb update of the rows of strip t+1
//   msg stream / comm2             the small broadcasts
//   comm stream / comm             all-gather of the solved strip (feeds HEAD and the bulk)
//   main stream                    bulk K = 2nb update of my columns, rows below strip t+1, upper staircase only,
//                                  A operand read straight out of the gathered pieces (no repacking); split into
//                                  the rows of strip t+2 (signals the panel stream early: look-ahead depth 2) + rest
// so the bulk update of strip t overlaps with factorization + communication of strips t+1 and t+2.
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

struct cap_dist_plan {
  int64_t n, npad, nb, nblk;
  int P, p;
  cap_comm* comm;                // all-gathers (not owned)
  cap_comm* comm2;               // small broadcasts (owned duplicate)
  int64_t nloc_blocks, lc;       // local blocks / padded local columns
  int64_t lc_valid;              // local columns that exist in the n x n matrix
  int64_t nmax0;                 // widest gathered piece (columns)
  double* R; int64_t ld;         // npad x lc
  double* S[2];                  // my solved strip (q nb x cols, ld = q nb)
  double* G[2];                  // gathered strip: P pieces
  double* msg[4];                // [R(k-1,k) | Dinv(k)], 2 nb^2 each, ring over block rows
  double* W; int64_t wcap;       // recursion scratch
  int* info_dev; double* info_red;
  hipStream_t s_panel, s_comm, s_msg;
  std::vector<hipEvent_t> ev_fact, ev_msg, ev_rowdone;             // per block row
  std::vector<hipEvent_t> ev_solved, ev_gather, ev_head2, ev_rest; // per strip
  hipEvent_t ev_init, ev_join_p, ev_join_c, ev_join_m;
  // schedule knobs
  int strip;                     // block rows per strip (1 or 2)
  int depth2;                    // split the bulk update (look-ahead depth 2)
  int64_t occ1_m;                // bulk updates with rows x local columns <= occ1_m^2 run one workgroup per CU (cholinv.hip, occ1_m)
  // stress testing: random spin kernels in front of every launch group (exposes missing event edges)
  uint64_t jitter_state; int jitter_max_us;
  int remote_chain_us;           // replay aid: a foreign owner's message arrives this long after my panel stream reached the block row (0: off)
  // live profile of the bulk update (HIP events on its stream)
  int profile; std::vector<hipEvent_t> prof_ev; std::vector<double> prof_flops; int prof_used;
  // profile mode also brackets the other launch groups: busy ms per stream role (cap_dist_profile_streams)
  //   0 panel: diagonal-block chains   1 panel: block-row solves + in-strip updates   2 panel: HEAD updates
  //   3 msg broadcasts                 4 strip exchange (all-gather / peer copies)    5 main: bulk updates
  std::vector<hipEvent_t> bk_ev; std::vector<int> bk_kind; int bk_used;
  // safe mode: ONE communicator and ONE communication stream - broadcasts and strip exchanges are issued in program order
  // (the data dependencies order them anyway: msg(a), msg(b), exchange(t), msg(a'), ...); the default keeps the small
  // messages on their own communicator + stream
  int safe;
  // launches / collectives issued by the LAST factor call on this rank (options count_gemm / count_chain / count_copy / count_coll)
  int64_t cnt_gemm, cnt_chain, cnt_copy, cnt_coll;
  // Strip exchange off the CUs (option "ipc"): every rank maps its peers' gathered-strip buffers (hipIpcGetMemHandle /
  // hipIpcOpenMemHandle, exchanged once through the communicator) and PUSHES its piece into each of them with plain
  // device-to-device copies on one copy stream per peer (SDMA engines, one xGMI link each) - no RCCL kernel competes with
  // the bulk update for CU slots (the reference pipelines the same move with MPI_Ibcast, summa.hpp:185-214).  Two 8-byte
  // all-reduces per strip carry the synchronisation: "every rank's buffer is free" before the pushes, "every push has
  // landed" after them.  RCCL stays the default and the fallback (option off, or if mapping a peer fails).
  // R^-1 on the distributed factor (complete_inv = 0 / 1, the reference's semantics on P > 1): see inverse_step
  int complete_inv; int64_t split;
  int64_t root_n1;               // > 0: the root partition of R^-1 (rows / columns [0, root_n1) | [root_n1, n)) given by the caller instead of n >> split
  double* Dall;                  // the diagonal-block inverses of MY block columns' steps (nblk x nb x nb slots), saved from the msg broadcasts
  double* Ri;                    // my block columns of R^-1 (npad x lc, ld = npad): the partial product X_k while the sweep runs
  double* Cb[2];                 // ring of two broadcast buffers for a finished block column of R^-1 (npad x nb each)
  cap_comm* comm3;               // the inverse's broadcasts (owned duplicate): they run on their own stream next to the sweep's collectives
  hipStream_t s_inv; hipEvent_t ev_join_i;
  hipEvent_t ev_t0, ev_sweep_end, ev_inv_end;    // profile mode (timing): start of the call, the sweep's join, end of the inverse
  int ipc; bool ipc_ready; bool ipc_failed;
  double* peerG[8][2]; hipStream_t s_peer[8]; hipEvent_t ev_x0, ev_xr[8]; double* token; int ipc_nocu;
};

namespace {
__host__ __device__ inline int64_t lbfirst(int64_t r, int64_t k, int64_t P) { return k >= r ? (k - r) / P + 1 : 0; }  // blocks J <= k owned by r
__host__ __device__ inline int64_t nblocks_of(int64_t r, int64_t nblk, int64_t P) { return r < nblk ? (nblk - 1 - r) / P + 1 : 0; }

__global__ void fill_symmetric_bc_kernel(double* out, int64_t ld, int64_t n, int64_t nb, int P, int p, int dom, int64_t lc_valid) {
  // same closed form as fill_symmetric_kernel (aux.hip), local column -> global column through the block-cyclic map
  int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t lcol = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (row >= n || lcol >= lc_valid) return;
  int64_t gcol = ((lcol / nb) * P + p) * nb + lcol % nb;
  int64_t hi = gcol > row ? gcol : row, lo = gcol > row ? row : gcol;
  uint64_t seed = (uint64_t)(hi + n * lo);
  uint64_t x0 = ((seed & 0xFFFFFFFFull) << 16) | 0x330Eull;
  uint64_t x1 = (0x5DEECE66Dull * x0 + 0xBull) & ((1ull << 48) - 1);
  double v = (double)x1 * (1.0 / 281474976710656.0);
  if (dom && gcol == row) v += (double)n;
  out[row + lcol * ld] = v;
}

// identity tail of the padded matrix: entries with row >= n or global column >= n become (row == gcol)
__global__ void pad_identity_kernel(double* R, int64_t ld, int64_t n, int64_t npad, int64_t nb, int P, int p, int64_t row0,
                                    int64_t col0, int64_t rows, int64_t cols) {
  int64_t row = row0 + blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t lcol = col0 + blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (row >= row0 + rows || lcol >= col0 + cols) return;
  int64_t gcol = ((lcol / nb) * P + p) * nb + lcol % nb;
  if (row >= n || gcol >= n) R[row + lcol * ld] = (row == gcol) ? 1.0 : 0.0;
}

// construct_R for the block-cyclic layout: copy my valid columns, zero below the GLOBAL diagonal
__global__ void export_upper_bc_kernel(const double* R, int64_t ld, double* out, int64_t ldo, int64_t n, int64_t nb, int P, int p,
                                       int64_t lc_valid) {
  int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  int64_t lcol = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (row >= n || lcol >= lc_valid) return;
  int64_t gcol = ((lcol / nb) * P + p) * nb + lcol % nb;
  out[row + lcol * ldo] = row <= gcol ? R[row + lcol * ld] : 0.0;
}

__global__ void info_to_double(const int* info, double* out) { *out = (double)*info; }

// busy-wait for about `us` microseconds (wall_clock64 ticks at 100 MHz)
__global__ void spin_kernel(int us) {
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < (uint64_t)us * 100ull) __builtin_amdgcn_s_sleep(32);
}

int jitter(cap_dist_plan* d, hipStream_t s) {
  if (d->jitter_max_us <= 0) return CAP_OK;
  d->jitter_state = d->jitter_state * 6364136223846793005ull + 1442695040888963407ull;
  const int us = (int)((d->jitter_state >> 33) % (uint64_t)(d->jitter_max_us + 1));
  if (us > 0) { cap_acc_none(); hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, s, us); CAP_HIP(hipGetLastError()); }
  return CAP_OK;
}

// profile mode: bracket a launch group on stream s (kind: see cap_dist_plan::bk_kind)
struct Bucket {
  cap_dist_plan* d; hipStream_t s; hipEvent_t e1; bool on;
  Bucket(cap_dist_plan* d_, int kind, hipStream_t s_) : d(d_), s(s_), e1(nullptr), on(false) {
    if (!d->profile) return;
    if ((size_t)d->bk_used + 2 > d->bk_ev.size())
      for (int i = 0; i < 64; i++) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) return; d->bk_ev.push_back(e); d->bk_kind.push_back(0); }
    hipEvent_t e0 = d->bk_ev[(size_t)d->bk_used]; e1 = d->bk_ev[(size_t)d->bk_used + 1];
    d->bk_kind[(size_t)d->bk_used] = kind;
    if (hipEventRecord(e0, s) != hipSuccess) return;
    on = true;
  }
  ~Bucket() { if (on && hipEventRecord(e1, s) == hipSuccess) d->bk_used += 2; }
};

// map the peers' G buffers (collective over d->comm; synchronises).  On any failure the plan stays on the RCCL exchange.
// Every LOCAL failure (allocation, handle export, copies, mapping) only clears `ok`: a rank never leaves before the two collectives
// below, so a failing rank cannot strand its peers inside them; RCCL errors themselves are fatal for the communicator anyway.
int ensure_ipc(cap_dist_plan* d, hipStream_t s) {
  if (d->ipc_ready || d->ipc_failed || d->P == 1) return CAP_OK;
  const int P = (int)d->P, p = (int)d->p;
  static_assert(sizeof(hipIpcMemHandle_t) == 64, "hipIpcMemHandle_t is 64 bytes");
  auto soft = [](hipError_t e) { if (e != hipSuccess) { (void)hipGetLastError(); return false; } return true; };
  bool ok = true;
  // device scratch: [P][2] handles (8 doubles each) + my two + the agreement flag; part of the plan so that no path leaks it
  if (!d->token) {
    ok = soft(hipMalloc((void**)&d->token, sizeof(double) * (2 + 16 * (P + 1))));
    if (ok) ok = soft(hipMemset(d->token, 0, sizeof(double) * (2 + 16 * (P + 1))));
    if (!ok) { d->token = nullptr; d->ipc_failed = true; return CAP_ERR_ALLOC; }     // nothing collective has happened yet - but every rank
  }                                                                                    // allocates the same few KB: treated as fatal
  double* hx = d->token + 2;
  hipIpcMemHandle_t mine[2];
  for (int b = 0; b < 2; b++) ok = ok && soft(hipIpcGetMemHandle(&mine[b], d->G[b]));
  if (!ok) memset(mine, 0, sizeof(mine));
  if (!soft(hipMemcpyAsync(hx + 16 * P, mine, 128, hipMemcpyHostToDevice, s))) ok = false;
  CAP_TRY(cap_comm_allgather(d->comm, hx + 16 * P, hx, 16, (void*)s));
  std::vector<hipIpcMemHandle_t> all((size_t)2 * P);
  memset(all.data(), 0, all.size() * sizeof(hipIpcMemHandle_t));
  if (!soft(hipMemcpyAsync(all.data(), hx, (size_t)128 * P, hipMemcpyDeviceToHost, s))) ok = false;
  if (!soft(hipStreamSynchronize(s))) ok = false;
  hipIpcMemHandle_t zero; memset(&zero, 0, sizeof(zero));
  for (int r = 0; r < P && ok; r++) {
    if (r == p) continue;
    for (int b = 0; b < 2 && ok; b++) {
      if (!memcmp(&all[(size_t)2 * r + b], &zero, sizeof(zero))) { ok = false; break; }
      void* q = nullptr;
      if (!soft(hipIpcOpenMemHandle(&q, all[(size_t)2 * r + b], hipIpcMemLazyEnablePeerAccess))) { ok = false; break; }
      d->peerG[r][b] = (double*)q;
    }
  }
  if (ok) {
    for (int r = 0; r < P && ok; r++) {
      if (r == p) continue;
      ok = soft(hipStreamCreateWithFlags(&d->s_peer[r], hipStreamNonBlocking)) && soft(hipEventCreateWithFlags(&d->ev_xr[r], hipEventDisableTiming));
    }
    if (ok) ok = soft(hipEventCreateWithFlags(&d->ev_x0, hipEventDisableTiming));
  }
  // agree: one failing rank sends everybody back to RCCL (the exchange is collective)
  double flag = ok ? 0.0 : 1.0, tot = 0.0;
  const bool up = soft(hipMemcpyAsync(d->token + 1, &flag, sizeof(double), hipMemcpyHostToDevice, s));
  if (!up) (void)hipMemsetAsync(d->token + 1, 0x7f, sizeof(double), s);        // any non-zero pattern: "failed" (a huge positive double)
  CAP_TRY(cap_comm_allreduce_sum(d->comm, d->token + 1, 1, (void*)s));
  if (!soft(hipMemcpyAsync(&tot, d->token + 1, sizeof(double), hipMemcpyDeviceToHost, s)) || !soft(hipStreamSynchronize(s))) tot = 1.0;
  (void)hipMemsetAsync(d->token + 1, 0, sizeof(double), s);                    // the strip exchange uses token[0] only; keep [1] clean
  if (tot != 0.0 || !ok) {
    for (int r = 0; r < P; r++) {
      for (int b = 0; b < 2; b++) if (d->peerG[r][b]) { (void)hipIpcCloseMemHandle(d->peerG[r][b]); d->peerG[r][b] = nullptr; }
      if (d->s_peer[r]) { cap_stream_destroy(d->s_peer[r]); d->s_peer[r] = nullptr; }
      if (d->ev_xr[r]) { (void)hipEventDestroy(d->ev_xr[r]); d->ev_xr[r] = nullptr; }
    }
    if (d->ev_x0) { (void)hipEventDestroy(d->ev_x0); d->ev_x0 = nullptr; }
    d->ipc_failed = true;
    fprintf(stderr, "capital_amd: IPC mapping of the peers' strip buffers failed on %s rank; using the RCCL all-gather\n", ok ? "another" : "this");
    return CAP_OK;
  }
  d->ipc_ready = true;
  return CAP_OK;
}

// the strip exchange: `piece` doubles from `src` into slot p of EVERY rank's gathered buffer G[par]
int strip_exchange(cap_dist_plan* d, const double* src, int par, int64_t piece, hipStream_t sc) {
  double* G = d->G[par];
  if (!(d->ipc && d->ipc_ready)) return cap_comm_allgather(d->comm, src, G, piece, (void*)sc);
  const int P = (int)d->P, p = (int)d->p;
  const hipMemcpyKind kind = d->ipc_nocu ? hipMemcpyDeviceToDeviceNoCU : hipMemcpyDeviceToDevice;
  CAP_TRY(cap_comm_allreduce_sum(d->comm, d->token, 1, (void*)sc));        // every rank's G[par] is free (each waited for its own readers)
  CAP_HIP(hipEventRecord(d->ev_x0, sc));
  for (int r = 0; r < P; r++) {
    if (r == p) continue;
    CAP_HIP(hipStreamWaitEvent(d->s_peer[r], d->ev_x0, 0));
    CAP_HIP(hipMemcpyAsync(d->peerG[r][par] + (int64_t)p * piece, src, sizeof(double) * piece, kind, d->s_peer[r]));
    CAP_HIP(hipEventRecord(d->ev_xr[r], d->s_peer[r]));
  }
  CAP_HIP(hipMemcpyAsync(G + (int64_t)p * piece, src, sizeof(double) * piece, hipMemcpyDeviceToDevice, sc));
  for (int r = 0; r < P; r++) if (r != p) CAP_HIP(hipStreamWaitEvent(sc, d->ev_xr[r], 0));
  return cap_comm_allreduce_sum(d->comm, d->token, 1, (void*)sc);          // every push into MY G[par] has landed
}

int ensure_events(cap_dist_plan* d) {
  if (!d->ev_msg.empty()) return CAP_OK;
  auto mk = [&](std::vector<hipEvent_t>& v, size_t cnt) -> int {
    v.resize(cnt);
    for (auto& e : v) CAP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    return CAP_OK;
  };
  size_t cnt = (size_t)d->nblk + 2;
  CAP_TRY(mk(d->ev_fact, cnt)); CAP_TRY(mk(d->ev_msg, cnt)); CAP_TRY(mk(d->ev_rowdone, cnt));
  CAP_TRY(mk(d->ev_solved, cnt)); CAP_TRY(mk(d->ev_gather, cnt)); CAP_TRY(mk(d->ev_head2, cnt)); CAP_TRY(mk(d->ev_rest, cnt));
  CAP_HIP(hipEventCreateWithFlags(&d->ev_init, hipEventDisableTiming));
  CAP_HIP(hipEventCreateWithFlags(&d->ev_join_p, hipEventDisableTiming));
  CAP_HIP(hipEventCreateWithFlags(&d->ev_join_c, hipEventDisableTiming));
  CAP_HIP(hipEventCreateWithFlags(&d->ev_join_m, hipEventDisableTiming));
  int lo = 0, hi = 0;
  CAP_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CAP_HIP(hipStreamCreateWithPriority(&d->s_panel, hipStreamNonBlocking, hi));
  CAP_HIP(hipStreamCreateWithPriority(&d->s_comm, hipStreamNonBlocking, hi));
  CAP_HIP(hipStreamCreateWithPriority(&d->s_msg, hipStreamNonBlocking, hi));
  return CAP_OK;
}

// bulk / head update launch, optionally bracketed by profiling events
int update(cap_dist_plan* d, int64_t m, int64_t nloc, int64_t K, const double* G, int64_t piece, const int* gstart, const double* B,
           double* C, int64_t J0, int64_t lb0, hipStream_t s, bool prof) {
  if (m <= 0 || nloc <= 0) return CAP_OK;
  CapRange range("CI::tmu");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (prof && d->profile) {
    if ((size_t)d->prof_used + 2 > d->prof_ev.size())
      for (int i = 0; i < 64; i++) { hipEvent_t e; CAP_HIP(hipEventCreate(&e)); d->prof_ev.push_back(e); }
    e0 = d->prof_ev[d->prof_used]; e1 = d->prof_ev[d->prof_used + 1];
    CAP_HIP(hipEventRecord(e0, s));
  }
  // chain-bound steps (my share of the update is short: on P ranks that is most of them): one bulk workgroup per CU, so the
  // diagonal-block chain of the owner always finds a free slot and shares its SIMDs with one fp64-MFMA wave instead of two
  const int occ = (prof && d->occ1_m > 0 && (double)m * (double)nloc <= (double)d->occ1_m * (double)d->occ1_m) ? -1 : 0;
  CAP_TRY(cap_dist_update_launch(m, nloc, K, G, piece, gstart, B, C, d->ld, d->P, d->p, (int)d->nb, (int)J0, (int)lb0, s, occ));
  d->cnt_gemm++;
  if (e0) {
    CAP_HIP(hipEventRecord(e1, s));
    d->prof_used += 2;
    // algorithmic flops: 2K per element of my part of the upper staircase
    double elems = 0;
    for (int64_t lb = lb0; lb < d->nloc_blocks; lb++) {
      const int64_t J = lb * d->P + d->p;
      const int64_t rows_above = std::min<int64_t>(m, (J - J0) * d->nb);   // full rows above the diagonal block
      elems += (double)rows_above * d->nb;
      if ((J - J0) * d->nb < m) elems += 0.5 * (double)d->nb * (d->nb + 1);
    }
    d->prof_flops.push_back(2.0 * (double)K * elems);
  }
  return CAP_OK;
}
// R^-1 of the distributed factor (complete_inv = 0 / 1 with P > 1: the reference's `factor` leaves both R and R^-1 on its grid,
// cholinv.hpp:85-165 + the distributed TRMM of summa.hpp:46-83) - STREAMED with the sweep, no replicated R.
//   R = E_{nblk-1} ... E_1 E_0,  E_k = identity with block row k replaced by block row k of R, so
//   R^-1 = E_0^-1 E_1^-1 ... E_{nblk-1}^-1,  E_k^-1 = identity with block row k replaced by [ Dinv(k) | -Dinv(k) R[k, k+1:] ],
// and the partial products X_k = E_0^-1 ... E_k^-1 can be built LEFT TO RIGHT, i.e. in the order the sweep finishes block rows:
//   step k   owner(k):  X[0:k+1, k] <- X[0:k+1, k] Dinv(k)            (block column k of R^-1 is final: rows 0 .. k)
//            broadcast of that column ((k + 1) nb x nb doubles, the volume of HALF an all-gather of R over the whole run)
//            every rank: X[0:k+1, Js] -= X[0:k+1, k] R[k, Js]          (Js = my block columns J > k; R[k, Js] is my OWN piece of the
//                                                                      solved block row: no other rank's part of R is ever needed)
// Each rank holds its block columns of R and of R^-1 plus two n x nb broadcast buffers: per-rank memory O(n^2 / P + nb n) where the
// previous form replicated R (32 GiB per rank at N = 65536).  Exactly (n^3 / 3) / P flops per rank, one MFMA GEMM per step, enqueued on
// its own stream behind "block row k is solved" - the inverse runs inside the factorization wherever that is chain- or
// communication-bound (cap_dist_profile_inverse reports how much of it was left after the sweep's join).
// complete_inv == 0: the root block Ri[0:n1, n1:n] (n1 = n >> split, cholinv.hpp:107,147) stays empty - columns J >= n1 / nb take no
// updates from block rows k < n1 / nb and are only finished from row n1 on (n1 on a block boundary), or the block is cleared
// afterwards (n1 inside a block).
__global__ void identity_blocks_bc_kernel(double* X, int64_t ld, int64_t nb, int P, int p, int64_t lc) {
  const int64_t l = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (l >= lc) return;
  const int64_t g = ((l / nb) * P + p) * nb + l % nb;        // global (= row) index of local column l
  X[g + l * ld] = 1.0;
}

struct InvCut { bool cut; int64_t kcut, n1; };
inline InvCut inv_cut(const cap_dist_plan* d) {
  const int64_t n1 = d->root_n1 > 0 ? d->root_n1 : d->n >> d->split;
  const bool cut = d->complete_inv == 0 && n1 > 0 && n1 < d->n;
  return InvCut{cut, (cut && n1 % d->nb == 0) ? n1 / d->nb : 0, n1};
}

// step k of the product above on stream s (broadcast on communicator c); see Bucket kind 6
int inverse_step(cap_dist_plan* d, int64_t k, hipStream_t s, cap_comm* c) {
  CapRange range("CI::inverse");
  Bucket bk(d, 6, s);
  const int64_t nb = d->nb, npad = d->npad, P = d->P, p = d->p, nb2 = nb * nb;
  const InvCut ic = inv_cut(d);
  const int64_t rows0 = (ic.kcut > 0 && k >= ic.kcut) ? ic.kcut * nb : 0;     // columns right of the root partition start at its row
  const int64_t rows = (k + 1) * nb - rows0;
  const int owner = (int)(k % P);
  double* Cbuf = d->Cb[k & 1];
  CAP_TRY(jitter(d, s));
  if (p == owner) {
    double* Xk = d->Ri + rows0 + (k / P) * nb * npad;
    // (X_{k-1}[:, k]) Dinv(k): Dinv upper triangular -> K range of a column tile stops at its diagonal (tag 8)
    CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, rows, nb, nb, 1.0, Xk, npad, d->Dall + k * nb2, nb, 0.0, Cbuf, rows, 0, s, 8));
    CAP_TRY(cap_copy_rect(Cbuf, rows, Xk, npad, rows, nb, s));
    d->cnt_gemm++; d->cnt_copy++;
  }
  CAP_TRY(cap_comm_bcast(c, Cbuf, rows * nb, owner, (void*)s)); d->cnt_coll++;
  const int64_t lb0 = lbfirst(p, k, P);
  int64_t lb1 = d->nloc_blocks;
  if (ic.kcut > 0 && k < ic.kcut) lb1 = lbfirst(p, ic.kcut - 1, P);          // left of the root partition only
  if (lb1 > lb0) {
    CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, rows, (lb1 - lb0) * nb, nb, -1.0, Cbuf, rows, d->R + k * nb + lb0 * nb * d->ld, d->ld, 1.0,
                            d->Ri + rows0 + lb0 * nb * npad, npad, 0, s));
    d->cnt_gemm++;
  }
  return CAP_OK;
}

int inverse_finish(cap_dist_plan* d, hipStream_t s) {
  const InvCut ic = inv_cut(d);
  if (ic.cut && ic.kcut == 0) {
    // root partition inside a block: clear Ri[0:n1, columns >= n1] of my columns
    for (int64_t lb = 0; lb < d->nloc_blocks; lb++) {
      const int64_t c0 = (lb * d->P + d->p) * d->nb;
      const int64_t from = std::max<int64_t>(c0, ic.n1), to = c0 + d->nb;
      if (from < to) CAP_TRY(cap_zero_rect(d->Ri + (lb * d->nb + (from - c0)) * d->npad, d->npad, ic.n1, to - from, s));
    }
  }
  return CAP_OK;
}

// buffers of the streamed inverse; called when the option is set, so an allocation failure is reported BEFORE any rank enters a collective
int ensure_inverse_buffers(cap_dist_plan* d) {
  if (d->Ri) return CAP_OK;
  const int64_t cols = std::max<int64_t>(d->lc, d->nb);
  hipError_t e = hipMalloc((void**)&d->Ri, sizeof(double) * d->npad * cols);
  if (e == hipSuccess) e = hipMalloc((void**)&d->Dall, sizeof(double) * d->nblk * d->nb * d->nb);
  for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMalloc((void**)&d->Cb[i], sizeof(double) * d->npad * d->nb);
  if (e != hipSuccess) {
    (void)hipGetLastError();
    for (double** q : {&d->Ri, &d->Dall, &d->Cb[0], &d->Cb[1]}) if (*q) { (void)hipFree(*q); *q = nullptr; }
    return CAP_ERR_ALLOC;
  }
  return CAP_OK;
}

}  // namespace

extern "C" {

int cap_bc_owner(int64_t J, int P) { return (int)(J % P); }
int64_t cap_bc_local_block(int64_t J, int P) { return J / P; }
// columns of the n x n matrix stored on rank p (the last block may be partial)
int64_t cap_bc_num_local_cols(int64_t n, int64_t nb, int P, int p) {
  int64_t nblk = (n + nb - 1) / nb;
  int64_t nl = nblocks_of(p, nblk, P);
  if (nl == 0) return 0;
  int64_t cols = nl * nb;
  const int64_t last = nblk - 1;
  if (last % P == p) cols -= nblk * nb - n;
  return cols;
}

int cap_fill_symmetric_bc(double* local, int64_t ld, int64_t n, int64_t nb, int P, int p, int diagonally_dominant, void* stream) {
  if (!local || n <= 0 || nb <= 0 || P < 1 || p < 0 || p >= P || ld < n) return CAP_ERR_ARG;
  int64_t lc = cap_bc_num_local_cols(n, nb, P, p);
  if (lc == 0) return CAP_OK;
  dim3 grid((unsigned)cap_ceil_div(n, 256), (unsigned)std::min<int64_t>(lc, 65535), (unsigned)cap_ceil_div(lc, 65535));
  cap_acc_w(local, ld, n, lc);
  hipLaunchKernelGGL(fill_symmetric_bc_kernel, grid, dim3(256), 0, cap_stream(stream), local, ld, n, nb, P, p, diagonally_dominant, lc);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

int cap_dist_plan_create(cap_dist_plan** plan, int64_t n, int64_t nb, cap_comm* comm) {
  if (!plan || n <= 0) return CAP_ERR_ARG;
  if (nb <= 0) nb = 512;
  if (nb % 128) return CAP_ERR_UNSUPPORTED;                // block width: multiple of the 128 MFMA tile
  cap_dist_plan* d = new (std::nothrow) cap_dist_plan();
  if (!d) return CAP_ERR_ALLOC;
  d->n = n; d->nb = nb; d->nblk = cap_ceil_div(n, nb); d->npad = d->nblk * nb; d->comm = comm; d->comm2 = nullptr;
  d->P = cap_comm_size(comm); d->p = cap_comm_rank(comm);
  if (d->P > 8) { delete d; return CAP_ERR_UNSUPPORTED; }  // one xGMI node; the update kernel carries 8 piece offsets
  d->nloc_blocks = nblocks_of(d->p, d->nblk, d->P); d->lc = d->nloc_blocks * nb;
  d->lc_valid = cap_bc_num_local_cols(n, nb, d->P, d->p);
  d->nmax0 = nblocks_of(0, d->nblk, d->P) * nb;            // rank 0 owns the most blocks
  d->ld = d->npad;
  d->R = nullptr; d->W = nullptr; d->info_dev = nullptr; d->info_red = nullptr;
  for (int i = 0; i < 2; i++) d->S[i] = d->G[i] = nullptr;
  for (int i = 0; i < 4; i++) d->msg[i] = nullptr;
  d->s_panel = d->s_comm = d->s_msg = nullptr;
  d->strip = d->nblk >= 8 ? 2 : 1; d->depth2 = 1;
  d->jitter_state = 0x9E3779B97F4A7C15ull * (uint64_t)(d->p + 1); d->jitter_max_us = 0; d->remote_chain_us = 0;
  // one bulk workgroup per CU while rows x local columns <= occ1_m^2.  On P ranks a rank waits for EVERY diagonal-block chain (its own and, through
  // the broadcasts, the peers') while its bulk work per strip is 1 / P of the single-GPU plan's: the threshold grows with P (single-GPU replay of
  // one rank at N = 65536, tools/replay.py: P = 8 slowest rank 286 -> 258 ms, P = 4 445 -> 418, P = 2 755 -> 752; always-on LOSES at P = 2: 771)
  d->occ1_m = 16384 * (int64_t)(d->P > 1 ? d->P : 1);
  d->profile = 0; d->prof_used = 0; d->bk_used = 0; d->safe = 0;
  d->cnt_gemm = d->cnt_chain = d->cnt_copy = d->cnt_coll = 0;
  d->complete_inv = -1; d->split = 1; d->root_n1 = 0; d->Dall = d->Ri = nullptr; d->Cb[0] = d->Cb[1] = nullptr; d->comm3 = nullptr;
  d->s_inv = nullptr; d->ev_join_i = nullptr; d->ev_t0 = d->ev_sweep_end = d->ev_inv_end = nullptr;
  d->ipc = CAP_ENV("CAP_DIST_IPC") ? atoi(CAP_ENV("CAP_DIST_IPC")) : 0; d->ipc_ready = false; d->ipc_failed = false; d->token = nullptr;
  d->ipc_nocu = CAP_ENV("CAP_DIST_IPC_NOCU") ? atoi(CAP_ENV("CAP_DIST_IPC_NOCU")) : 0;
  for (int r = 0; r < 8; r++) { d->peerG[r][0] = d->peerG[r][1] = nullptr; d->s_peer[r] = nullptr; d->ev_xr[r] = nullptr; }
  d->ev_x0 = nullptr;
  d->wcap = cap_rec_work_size(nb);
  if (cap_comm_size(comm) > 1 || cap_comm_backend(comm) != 0) {
    int st = cap_comm_dup(comm, &d->comm2);
    if (st != CAP_OK) { delete d; return st; }
  }
  hipError_t e = hipMalloc((void**)&d->R, sizeof(double) * std::max<int64_t>(d->npad * d->lc, 2));
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    // S: one extra block column - a rank that owns the strip's last block sends from one block further in, and the
    // equal-sized padded pieces are as wide as the widest rank's remainder
    e = hipMalloc((void**)&d->S[i], sizeof(double) * 2 * nb * (d->nmax0 + nb));
    if (e == hipSuccess) e = hipMalloc((void**)&d->G[i], sizeof(double) * 2 * nb * d->nmax0 * d->P);
    if (e == hipSuccess) e = hipMemset(d->S[i], 0, sizeof(double) * 2 * nb * (d->nmax0 + nb));
  }
  for (int i = 0; i < 4 && e == hipSuccess; i++) {
    e = hipMalloc((void**)&d->msg[i], sizeof(double) * 2 * nb * nb);
    if (e == hipSuccess) e = hipMemset(d->msg[i], 0, sizeof(double) * 2 * nb * nb);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&d->W, sizeof(double) * d->wcap);
  if (e == hipSuccess) e = hipMalloc((void**)&d->info_dev, sizeof(int));
  if (e == hipSuccess) e = hipMalloc((void**)&d->info_red, sizeof(double) * (d->P + 1));
  if (e != hipSuccess) { cap_dist_plan_destroy(d); return CAP_ERR_ALLOC; }
  // helper streams and events exist from here on (round 6: never created inside the first factor call)
  { const int st = ensure_events(d); if (st != CAP_OK) { cap_dist_plan_destroy(d); return st; } }
  *plan = d;
  return CAP_OK;
}

int cap_dist_plan_destroy(cap_dist_plan* d) {
  if (!d) return CAP_OK;
  if (d->R) (void)hipFree(d->R);
  for (int i = 0; i < 2; i++) { if (d->S[i]) (void)hipFree(d->S[i]); if (d->G[i]) (void)hipFree(d->G[i]); }
  for (int i = 0; i < 4; i++) if (d->msg[i]) (void)hipFree(d->msg[i]);
  if (d->W) (void)hipFree(d->W);
  if (d->info_dev) (void)hipFree(d->info_dev);
  if (d->info_red) (void)hipFree(d->info_red);
  for (double* q : {d->Dall, d->Ri, d->Cb[0], d->Cb[1]}) if (q) (void)hipFree(q);
  if (d->s_inv) { (void)hipStreamSynchronize(d->s_inv); cap_stream_destroy(d->s_inv); }
  for (hipEvent_t e : {d->ev_join_i, d->ev_t0, d->ev_sweep_end, d->ev_inv_end}) if (e) (void)hipEventDestroy(e);
  if (d->comm3) cap_comm_destroy(d->comm3);
  if (!d->ev_msg.empty()) {
    for (auto* v : {&d->ev_fact, &d->ev_msg, &d->ev_rowdone, &d->ev_solved, &d->ev_gather, &d->ev_head2, &d->ev_rest})
      for (auto e : *v) (void)hipEventDestroy(e);
    (void)hipEventDestroy(d->ev_init); (void)hipEventDestroy(d->ev_join_p); (void)hipEventDestroy(d->ev_join_c);
    (void)hipEventDestroy(d->ev_join_m);
    cap_stream_destroy(d->s_panel); cap_stream_destroy(d->s_comm); cap_stream_destroy(d->s_msg);
  }
  for (hipEvent_t e : d->prof_ev) (void)hipEventDestroy(e);
  for (hipEvent_t e : d->bk_ev) (void)hipEventDestroy(e);
  for (int r = 0; r < 8; r++) {
    for (int b = 0; b < 2; b++) if (d->peerG[r][b]) (void)hipIpcCloseMemHandle(d->peerG[r][b]);
    if (d->s_peer[r]) { (void)hipStreamSynchronize(d->s_peer[r]); cap_stream_destroy(d->s_peer[r]); }
    if (d->ev_xr[r]) (void)hipEventDestroy(d->ev_xr[r]);
  }
  if (d->ev_x0) (void)hipEventDestroy(d->ev_x0);
  if (d->token) (void)hipFree(d->token);
  if (d->comm2) cap_comm_destroy(d->comm2);
  delete d;
  return CAP_OK;
}

int cap_dist_set_option(cap_dist_plan* d, const char* key, int64_t value) {
  if (!d || !key) return CAP_ERR_ARG;
  std::string k(key);
  if (k == "strip") { if (value < 1 || value > 2) return CAP_ERR_ARG; d->strip = (int)value; return CAP_OK; }
  if (k == "depth2") { d->depth2 = value != 0; return CAP_OK; }
  if (k == "occ1_m") { if (value < 0) return CAP_ERR_ARG; d->occ1_m = value; return CAP_OK; }
  if (k == "jitter_us") { if (value < 0 || value > 100000) return CAP_ERR_ARG; d->jitter_max_us = (int)value; return CAP_OK; }
  if (k == "remote_chain_us") { if (value < 0 || value > 1000000) return CAP_ERR_ARG; d->remote_chain_us = (int)value; return CAP_OK; }
  if (k == "jitter_seed") { d->jitter_state = 0x9E3779B97F4A7C15ull * (uint64_t)(value + 1) + (uint64_t)d->p; return CAP_OK; }
  if (k == "profile") { d->profile = value != 0; return CAP_OK; }
  if (k == "safe") { d->safe = value != 0; return CAP_OK; }
  if (k == "ipc") { d->ipc = value != 0; return CAP_OK; }
  if (k == "complete_inv") {
    if (value < -1 || value > 1) return CAP_ERR_ARG;
    if (value >= 0) CAP_TRY(ensure_inverse_buffers(d));       // O(n^2 / P + nb n) doubles; a failure is reported here, not inside factor
    d->complete_inv = (int)value;
    return CAP_OK;
  }
  if (k == "split") { if (value <= 0) return CAP_ERR_ARG; d->split = value; return CAP_OK; }
  // the root partition as the caller's layout defines it (upstream cuts its LOCAL dimension: (ceil(n / d) >> split) d global rows,
  // cholinv.hpp:107 - the same as n >> split unless n is ragged); 0 = n >> split
  if (k == "root_n1") { if (value < 0 || value > d->n) return CAP_ERR_ARG; d->root_n1 = value; return CAP_OK; }
  if (k == "ipc_nocu") { d->ipc_nocu = value != 0; return CAP_OK; }
  return CAP_ERR_ARG;
}

int64_t cap_dist_get_option(const cap_dist_plan* d, const char* key) {
  if (!d || !key) return -1;
  std::string k(key);
  if (k == "strip") return d->strip;
  if (k == "depth2") return d->depth2;
  if (k == "occ1_m") return d->occ1_m;
  if (k == "jitter_us") return d->jitter_max_us;
  if (k == "remote_chain_us") return d->remote_chain_us;
  if (k == "safe") return d->safe;
  if (k == "ipc") return d->ipc;
  if (k == "count_gemm") return d->cnt_gemm;
  if (k == "count_chain") return d->cnt_chain;
  if (k == "count_copy") return d->cnt_copy;
  if (k == "count_coll") return d->cnt_coll;
  if (k == "complete_inv") return d->complete_inv;
  if (k == "split") return d->split;
  if (k == "root_n1") return d->root_n1;
  if (k == "ipc_active") return (d->ipc && d->ipc_ready) ? 1 : 0;
  if (k == "nb") return d->nb;
  if (k == "n") return d->n;
  if (k == "npad") return d->npad;
  if (k == "P") return d->P;
  if (k == "p") return d->p;
  return -1;
}

int64_t cap_dist_local_cols(const cap_dist_plan* d) { return d ? d->lc_valid : 0; }
double* cap_dist_R_ptr(cap_dist_plan* d, int64_t* ld) { if (!d) return nullptr; if (ld) *ld = d->ld; return d->R; }

int cap_dist_factor(cap_dist_plan* d, const double* Aloc, int64_t lda, void* stream) {
  if (!d || (d->lc_valid > 0 && (!Aloc || lda < d->n))) return CAP_ERR_ARG;
  CAP_TRY(ensure_events(d));
  if (d->ipc) CAP_TRY(ensure_ipc(d, cap_stream(stream)));
  const bool inv = d->complete_inv >= 0;
  if (inv) {
    CAP_TRY(ensure_inverse_buffers(d));
    if (!d->s_inv) {
      int lo = 0, hi = 0;
      CAP_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
      CAP_HIP(hipStreamCreateWithPriority(&d->s_inv, hipStreamNonBlocking, lo));      // below the panel stream: it fills the gaps
      CAP_HIP(hipEventCreateWithFlags(&d->ev_join_i, hipEventDisableTiming));
      CAP_HIP(hipEventCreate(&d->ev_t0)); CAP_HIP(hipEventCreate(&d->ev_sweep_end)); CAP_HIP(hipEventCreate(&d->ev_inv_end));
      if (d->comm2) CAP_TRY(cap_comm_dup(d->comm, &d->comm3));
    }
  }
  hipStream_t s0 = cap_stream(stream), s1 = d->s_panel, sc = d->s_comm, sm = d->safe ? d->s_comm : d->s_msg;
  cap_comm* cmsg = (d->safe || !d->comm2) ? d->comm : d->comm2;
  // the inverse's steps: overlapped (own stream + own communicator, enqueued as the block rows finish) or - safe mode: ONE
  // communicator, collectives in program order - after the sweep on the caller's stream
  const bool inv_overlap = inv && !d->safe;
  hipStream_t si = inv_overlap ? d->s_inv : s0;
  cap_comm* cinv = (inv_overlap && d->comm3) ? d->comm3 : d->comm;
  const int64_t n = d->n, npad = d->npad, nb = d->nb, nblk = d->nblk, P = d->P, p = d->p, ld = d->ld;
  const int64_t nb2 = nb * nb;
  d->prof_used = 0; d->prof_flops.clear(); d->bk_used = 0;
  d->cnt_gemm = d->cnt_chain = d->cnt_copy = d->cnt_coll = 0;
  CAP_HIP(hipMemsetAsync(d->info_dev, 0, sizeof(int), s0));
  if (d->lc_valid > 0) CAP_TRY(cap_copy_rect(Aloc, lda, d->R, ld, n, d->lc_valid, s0));
  if (npad != n && d->lc > 0) {
    // rows [n, npad) of every local column, and the padding columns of the last block on its owner
    cap_acc_w(d->R + n, ld, npad - n, d->lc);
    hipLaunchKernelGGL(pad_identity_kernel, dim3((unsigned)cap_ceil_div(npad - n, 256), (unsigned)std::min<int64_t>(d->lc, 65535),
                                                 (unsigned)cap_ceil_div(d->lc, 65535)), dim3(256), 0, s0, d->R, ld, n,
                       npad, nb, (int)P, (int)p, n, (int64_t)0, npad - n, d->lc);
    if (d->lc > d->lc_valid) cap_acc_w(d->R + d->lc_valid * ld, ld, npad, d->lc - d->lc_valid);
    if (d->lc > d->lc_valid)      // (at most nb columns: no grid.z folding needed, the kernel tolerates it anyway)
      hipLaunchKernelGGL(pad_identity_kernel, dim3((unsigned)cap_ceil_div(npad, 256), (unsigned)(d->lc - d->lc_valid)), dim3(256), 0, s0,
                         d->R, ld, n, npad, nb, (int)P, (int)p, (int64_t)0, d->lc_valid, npad, d->lc - d->lc_valid);
    CAP_HIP(hipGetLastError());
  }
  if (inv) {
    // X_{-1} = I on my block columns
    if (d->profile) CAP_HIP(hipEventRecord(d->ev_t0, s0));
    CAP_HIP(hipMemsetAsync(d->Ri, 0, sizeof(double) * d->npad * std::max<int64_t>(d->lc, 1), s0));
    if (d->lc > 0) {
      cap_acc_w(d->Ri, npad, npad, d->lc);
      hipLaunchKernelGGL(identity_blocks_bc_kernel, dim3((unsigned)cap_ceil_div(d->lc, 256)), dim3(256), 0, s0, d->Ri, npad, nb, (int)P, (int)p, d->lc);
      CAP_HIP(hipGetLastError());
    }
  }
  CAP_HIP(hipEventRecord(d->ev_init, s0));
  CAP_HIP(hipStreamWaitEvent(s1, d->ev_init, 0));
  CAP_HIP(hipStreamWaitEvent(sc, d->ev_init, 0));
  CAP_HIP(hipStreamWaitEvent(sm, d->ev_init, 0));
  if (inv_overlap) CAP_HIP(hipStreamWaitEvent(si, d->ev_init, 0));

  // strips: sb[t] = first block row, sq[t] = block rows (d->strip, the last one may be shorter)
  std::vector<int64_t> sb, sq;
  for (int64_t k = 0; k < nblk; k += d->strip) { sb.push_back(k); sq.push_back(std::min<int64_t>(d->strip, nblk - k)); }
  const int64_t nstrips = (int64_t)sb.size();

  for (int64_t t = 0; t < nstrips; t++) {
    const int par = (int)(t & 1);
    const int64_t a = sb[t], q = sq[t], b = a + q - 1, e = b + 1;
    const int64_t ldS = q * nb;
    const int64_t lbS = lbfirst(p, a, P);                        // my first local block with J > a: column origin of S[par]
    double* S = d->S[par];

    for (int64_t r = 0; r < q; r++) {
      const int64_t k = a + r;
      const int owner = (int)(k % P);
      double* mb = d->msg[k & 3];
      double* Dinv = mb + nb2;
      // ---- panel, owner of block column k: finish its diagonal block, factor + invert it
      if (p == owner) {
        double* D = d->R + k * nb + (k / P) * nb * ld;
        CAP_TRY(jitter(d, s1));
        if (r == 1) {
          // in-strip: D(b) -= S_a(:, blk b)^T S_a(:, blk b); block b is my first column block of S
          const double* Sab = S + (k / P - lbS) * nb * ldS;
          CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, nb, nb, -1.0, Sab, ldS, Sab, ldS, 1.0, D, ld, 1, s1, 2)); d->cnt_gemm++;
        }
        {
          CapRange range("CI::factor_diag"); Bucket bk(d, 0, s1);
          CAP_TRY(cap_rec_cholinv_full(D, ld, Dinv, nb, nb, d->W, d->wcap, d->info_dev, s1, k * nb)); d->cnt_chain++;
        }
        if (r == 1) { CAP_TRY(cap_copy_rect(S + (k / P - lbS) * nb * ldS, ldS, mb, nb, nb, nb, s1)); d->cnt_copy++; }   // R(a,b) rides along
        CAP_HIP(hipEventRecord(d->ev_fact[k], s1));
        CAP_HIP(hipStreamWaitEvent(sm, d->ev_fact[k], 0));
      } else {
        if (k >= 4) CAP_HIP(hipStreamWaitEvent(sm, d->ev_rowdone[k - 4], 0));   // the broadcast overwrites the buffer block row k-4 read
        if (d->remote_chain_us > 0) {
          // single-GPU replay of one rank (tools/replay.py): the owner's diagonal-block chain is not run here, so the message would
          // arrive the moment it is asked for.  Model: the owner reaches this point when I do (symmetric ranks) and then needs
          // remote_chain_us for its chain - a spin kernel on the message stream behind my own panel stream's position
          CAP_HIP(hipEventRecord(d->ev_fact[k], s1));
          CAP_HIP(hipStreamWaitEvent(sm, d->ev_fact[k], 0));
          cap_acc_none(); hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, sm, d->remote_chain_us); CAP_HIP(hipGetLastError());
        }
      }
      // ---- msg: msg(k) = [ R(k-1,k) | Dinv(k) ] from the owner, on the small-message communicator
      CAP_TRY(jitter(d, sm));
      { Bucket bk(d, 3, sm); CAP_TRY(cap_comm_bcast(cmsg, mb, 2 * nb2, owner, (void*)sm)); d->cnt_coll++; }
      CAP_HIP(hipEventRecord(d->ev_msg[k], sm));

      // ---- panel, every rank: block row k of my columns J > k
      CAP_HIP(hipStreamWaitEvent(s1, d->ev_msg[k], 0));
      if (inv && p == owner) { CAP_TRY(cap_copy_rect(Dinv, nb, d->Dall + k * nb2, nb, nb, nb, s1)); d->cnt_copy++; }   // kept for step k of the inverse
      if (r == 0 && t >= 2) CAP_HIP(hipStreamWaitEvent(s1, d->ev_gather[t - 2], 0));   // S[par] was the all-gather source of strip t-2
      const int64_t lbk = lbfirst(p, k, P);
      const int64_t ncols = (d->nloc_blocks - lbk) * nb;
      if (ncols > 0) {
        CapRange range("CI::trsm"); Bucket bk(d, 1, s1);
        CAP_TRY(jitter(d, s1));
        double* Rrow = d->R + k * nb + lbk * nb * ld;
        double* Scol = S + (lbk - lbS) * nb * ldS;                // my columns J > k inside the strip buffer
        if (r == 1) {   // in-strip update: R[b, mine] -= R(a,b)^T S_a(:, mine)
          CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, ncols, nb, -1.0, mb, nb, Scol, ldS, 1.0, Rrow, ld, 0, s1, 2));
          d->cnt_gemm++;
        }
        // S_k = Dinv(k)^T R[k, mine]  (TRSM by the inverse, cholinv.hpp:118-121), then back into R
        CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, nb, ncols, nb, 1.0, Dinv, nb, Rrow, ld, 0.0, Scol + r * nb, ldS, 0, s1, 2 | 16));
        CAP_TRY(cap_copy_rect(Scol + r * nb, ldS, Rrow, ld, nb, ncols, s1));
        d->cnt_gemm++; d->cnt_copy++;
      }
      CAP_HIP(hipEventRecord(d->ev_rowdone[k], s1));
      if (inv_overlap) {       // block row k is final: step k of the streamed inverse (its own stream, next to everything below)
        CAP_HIP(hipStreamWaitEvent(si, d->ev_rowdone[k], 0));
        CAP_TRY(inverse_step(d, k, si, cinv));
      }
    }
    CAP_HIP(hipEventRecord(d->ev_solved[t], s1));
    if (e >= nblk) continue;

    // ---- comm: all-gather the strip's columns J > b (equal-sized padded pieces)
    int gstart[8];
    int64_t nmax = 0;
    for (int64_t r = 0; r < 8; r++) gstart[r] = r < P ? (int)lbfirst(r, b, P) : 0;
    for (int64_t r = 0; r < P; r++) nmax = std::max(nmax, (nblocks_of(r, nblk, P) - lbfirst(r, b, P)) * nb);
    const int64_t piece = ldS * nmax;
    const int64_t lbe = lbfirst(p, b, P);                        // my first local block with J >= e
    double* G = d->G[par];
    CAP_HIP(hipStreamWaitEvent(sc, d->ev_solved[t], 0));
    if (t >= 2) CAP_HIP(hipStreamWaitEvent(sc, d->ev_rest[t - 2], 0));   // G[par] was read by the bulk update of strip t-2
    CAP_TRY(jitter(d, sc));
    { Bucket bk(d, 4, sc); CAP_TRY(strip_exchange(d, S + (lbe - lbS) * nb * ldS, par, piece, sc)); }
    if (d->ipc && d->ipc_ready) { d->cnt_coll += 2; d->cnt_copy += P; } else d->cnt_coll++;
    CAP_HIP(hipEventRecord(d->ev_gather[t], sc));

    // ---- panel: HEAD - bring the rows of strip t+1 up to date with strip t (my columns J >= e)
    const int64_t q1 = sq[t + 1], e2 = e + q1;
    const int64_t ncols_e = (d->nloc_blocks - lbe) * nb;
    CAP_HIP(hipStreamWaitEvent(s1, d->ev_gather[t], 0));
    if (t >= 1) CAP_HIP(hipStreamWaitEvent(s1, d->ev_head2[t - 1], 0));   // strip t-1's bulk update has passed these rows
    CAP_TRY(jitter(d, s1));
    { Bucket bk(d, 2, s1); CAP_TRY(update(d, q1 * nb, ncols_e, ldS, G, piece, gstart, G + p * piece, d->R + e * nb + lbe * nb * ld, e, lbe, s1, false)); }

    // ---- main: bulk update of the rows below strip t+1 (my columns J >= e2), upper staircase
    CAP_HIP(hipStreamWaitEvent(s0, d->ev_gather[t], 0));
    if (e2 < nblk) {
      const int64_t lbe2 = lbfirst(p, e2 - 1, P);
      const int64_t ncols_e2 = (d->nloc_blocks - lbe2) * nb;
      const double* B2 = G + p * piece + (lbe2 - lbe) * nb * ldS;
      const int64_t q2 = (t + 2 < nstrips) ? sq[t + 2] : 0, e3 = e2 + q2;
      CAP_TRY(jitter(d, s0));
      if (d->depth2 && q2 > 0 && e3 < nblk) {
        // head of the bulk: the rows of strip t+2 first, then release the panel stream (look-ahead depth 2)
        CAP_TRY(update(d, q2 * nb, ncols_e2, ldS, G, piece, gstart, B2, d->R + e2 * nb + lbe2 * nb * ld, e2, lbe2, s0, true));
        CAP_HIP(hipEventRecord(d->ev_head2[t], s0));
        const int64_t lbe3 = lbfirst(p, e3 - 1, P);
        const int64_t ncols_e3 = (d->nloc_blocks - lbe3) * nb;
        CAP_TRY(jitter(d, s0));
        CAP_TRY(update(d, npad - e3 * nb, ncols_e3, ldS, G, piece, gstart, G + p * piece + (lbe3 - lbe) * nb * ldS,
                       d->R + e3 * nb + lbe3 * nb * ld, e3, lbe3, s0, true));
      } else {
        CAP_TRY(update(d, npad - e2 * nb, ncols_e2, ldS, G, piece, gstart, B2, d->R + e2 * nb + lbe2 * nb * ld, e2, lbe2, s0, true));
        CAP_HIP(hipEventRecord(d->ev_head2[t], s0));
      }
    } else {
      CAP_HIP(hipEventRecord(d->ev_head2[t], s0));
    }
    CAP_HIP(hipEventRecord(d->ev_rest[t], s0));
  }
  // join the helper streams back into the caller's stream
  CAP_HIP(hipEventRecord(d->ev_join_p, s1));
  CAP_HIP(hipEventRecord(d->ev_join_c, sc));
  CAP_HIP(hipEventRecord(d->ev_join_m, sm));
  CAP_HIP(hipStreamWaitEvent(s0, d->ev_join_p, 0));
  CAP_HIP(hipStreamWaitEvent(s0, d->ev_join_c, 0));
  CAP_HIP(hipStreamWaitEvent(s0, d->ev_join_m, 0));
  if (inv) {
    if (d->profile) CAP_HIP(hipEventRecord(d->ev_sweep_end, s0));
    if (inv_overlap) {
      CAP_HIP(hipEventRecord(d->ev_join_i, si));
      CAP_HIP(hipStreamWaitEvent(s0, d->ev_join_i, 0));
    } else {
      for (int64_t k = 0; k < nblk; k++) CAP_TRY(inverse_step(d, k, s0, cinv));
    }
    CAP_TRY(inverse_finish(d, s0));
    if (d->profile) CAP_HIP(hipEventRecord(d->ev_inv_end, s0));
  }
  return CAP_OK;
}

// construct_R (cholinv.hpp:30-37) for this layout: my columns of R (n x local_cols), zero below the global diagonal
int cap_dist_get_R(cap_dist_plan* d, double* out, int64_t ldo, void* stream) {
  if (!d || (d->lc_valid > 0 && (!out || ldo < d->n))) return CAP_ERR_ARG;
  if (d->lc_valid == 0) return CAP_OK;
  dim3 grid((unsigned)cap_ceil_div(d->n, 256), (unsigned)std::min<int64_t>(d->lc_valid, 65535), (unsigned)cap_ceil_div(d->lc_valid, 65535));
  cap_acc_r(d->R, d->ld, d->n, d->lc_valid); cap_acc_w(out, ldo, d->n, d->lc_valid);
  hipLaunchKernelGGL(export_upper_bc_kernel, grid, dim3(256), 0, cap_stream(stream), d->R, d->ld, out, ldo, d->n, d->nb, d->P, d->p,
                     d->lc_valid);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

// construct_Rinv (cholinv.hpp:39-46) for this layout: my columns of R^-1 (n x local_cols), zero below the global diagonal
int cap_dist_get_Rinv(cap_dist_plan* d, double* out, int64_t ldo, void* stream) {
  if (!d || (d->lc_valid > 0 && (!out || ldo < d->n))) return CAP_ERR_ARG;
  if (d->complete_inv < 0 || !d->Ri) return CAP_ERR_UNSUPPORTED;
  if (d->lc_valid == 0) return CAP_OK;
  dim3 grid((unsigned)cap_ceil_div(d->n, 256), (unsigned)std::min<int64_t>(d->lc_valid, 65535), (unsigned)cap_ceil_div(d->lc_valid, 65535));
  cap_acc_r(d->Ri, d->ld, d->n, d->lc_valid); cap_acc_w(out, ldo, d->n, d->lc_valid);
  hipLaunchKernelGGL(export_upper_bc_kernel, grid, dim3(256), 0, cap_stream(stream), d->Ri, d->ld, out, ldo, d->n, d->nb, d->P, d->p,
                     d->lc_valid);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}
double* cap_dist_Rinv_ptr(cap_dist_plan* d, int64_t* ld) { if (!d || !d->Ri) return nullptr; if (ld) *ld = d->ld; return d->Ri; }

// Agreed on all ranks: 0, or the smallest 1-based failing pivot index any owner reported.  Collective on the
// plan's communicator: call it on the stream cap_dist_factor was given (or after synchronising it).
int cap_dist_info(cap_dist_plan* d, void* stream, int64_t* info) {
  if (!d || !info) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  CAP_TRY(cap_drain_streams({d->s_panel, d->s_comm, d->s_msg, d->s_inv}));
  for (int r = 0; r < 8; r++) CAP_TRY(cap_drain_streams({d->s_peer[r]}));
  double* mine = d->info_red + d->P;
  cap_acc_r(d->info_dev, 1, 1, 1, 0, 4); cap_acc_w(mine, 1, 1, 1);
  hipLaunchKernelGGL(info_to_double, dim3(1), dim3(1), 0, s, d->info_dev, mine);
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

// live profile of the bulk updates of the LAST factor call (see cap_cholinv_profile)
int cap_dist_profile(cap_dist_plan* d, int64_t* launches, double* ms_total, double* flops_total) {
  if (!d || !launches || !ms_total || !flops_total) return CAP_ERR_ARG;
  *launches = 0; *ms_total = 0; *flops_total = 0;
  for (int i = 0; i + 1 < d->prof_used; i += 2) {
    CAP_HIP(hipEventSynchronize(d->prof_ev[i + 1]));
    float ms = 0;
    CAP_HIP(hipEventElapsedTime(&ms, d->prof_ev[i], d->prof_ev[i + 1]));
    *ms_total += ms; *flops_total += d->prof_flops[i / 2]; (*launches)++;
  }
  return CAP_OK;
}

// the same per launch: up to `cap` entries of (ms, flops) in launch order; returns the number of launches through *count
int cap_dist_profile_launches(cap_dist_plan* d, double* ms_out, double* flops_out, int64_t cap, int64_t* count) {
  if (!d || !count || (cap > 0 && (!ms_out || !flops_out))) return CAP_ERR_ARG;
  *count = d->prof_used / 2;
  for (int i = 0; i + 1 < d->prof_used && i / 2 < cap; i += 2) {
    CAP_HIP(hipEventSynchronize(d->prof_ev[(size_t)i + 1]));
    float ms = 0;
    CAP_HIP(hipEventElapsedTime(&ms, d->prof_ev[(size_t)i], d->prof_ev[(size_t)i + 1]));
    ms_out[i / 2] = ms; flops_out[i / 2] = d->prof_flops[(size_t)i / 2];
  }
  return CAP_OK;
}

// busy milliseconds per stream role of the LAST factor call in profile mode (see cap_dist_plan::bk_kind): out[0..5] =
// chains, row solves, HEAD updates, message broadcasts, strip exchanges, bulk updates.  Synchronises on the recorded events.
int cap_dist_profile_streams(cap_dist_plan* d, double* out6) {
  if (!d || !out6) return CAP_ERR_ARG;
  for (int i = 0; i < 6; i++) out6[i] = 0.0;
  for (int i = 0; i + 1 < d->bk_used; i += 2) {
    CAP_HIP(hipEventSynchronize(d->bk_ev[(size_t)i + 1]));
    float ms = 0;
    CAP_HIP(hipEventElapsedTime(&ms, d->bk_ev[(size_t)i], d->bk_ev[(size_t)i + 1]));
    const int kind = d->bk_kind[(size_t)i];
    if (kind >= 0 && kind < 5) out6[kind] += ms;
  }
  for (int i = 0; i + 1 < d->prof_used; i += 2) {
    CAP_HIP(hipEventSynchronize(d->prof_ev[(size_t)i + 1]));
    float ms = 0;
    CAP_HIP(hipEventElapsedTime(&ms, d->prof_ev[(size_t)i], d->prof_ev[(size_t)i + 1]));
    out6[5] += ms;
  }
  return CAP_OK;
}

// profile mode, complete_inv >= 0: out3 = busy ms of the inverse's launch groups (summed), ms between the sweep's join and the end of
// the inverse (what the overlap did NOT hide; the whole inverse in safe mode), ms of the whole factor call.  Synchronises.
int cap_dist_profile_inverse(cap_dist_plan* d, double* out3) {
  if (!d || !out3) return CAP_ERR_ARG;
  out3[0] = out3[1] = out3[2] = 0.0;
  if (d->complete_inv < 0 || !d->ev_inv_end || !d->profile) return CAP_OK;
  for (int i = 0; i + 1 < d->bk_used; i += 2) {
    if (d->bk_kind[(size_t)i] != 6) continue;
    CAP_HIP(hipEventSynchronize(d->bk_ev[(size_t)i + 1]));
    float ms = 0;
    CAP_HIP(hipEventElapsedTime(&ms, d->bk_ev[(size_t)i], d->bk_ev[(size_t)i + 1]));
    out3[0] += ms;
  }
  CAP_HIP(hipEventSynchronize(d->ev_inv_end));
  float a = 0, b = 0;
  CAP_HIP(hipEventElapsedTime(&a, d->ev_sweep_end, d->ev_inv_end));
  CAP_HIP(hipEventElapsedTime(&b, d->ev_t0, d->ev_inv_end));
  out3[1] = a; out3[2] = b;
  return CAP_OK;
}

// Watchdog support: how far the event chain of the factor call in flight has got, WITHOUT blocking.  out[0..6] = number of
// leading events already complete among  fact (owner's diagonal block), msg (broadcast), rowdone (block row solved) - per
// block row - and solved, gather (strip exchanged), head2, rest (bulk update) - per strip; out[7] = block rows, out[8] = strips.
int cap_dist_progress(cap_dist_plan* d, int64_t* out9) {
  if (!d || !out9) return CAP_ERR_ARG;
  for (int i = 0; i < 9; i++) out9[i] = 0;
  if (d->ev_msg.empty()) return CAP_OK;
  const int64_t nstrips = cap_ceil_div(d->nblk, d->strip);
  std::vector<hipEvent_t>* vs[7] = {&d->ev_fact, &d->ev_msg, &d->ev_rowdone, &d->ev_solved, &d->ev_gather, &d->ev_head2, &d->ev_rest};
  for (int v = 0; v < 7; v++) {
    const int64_t cnt = v < 3 ? d->nblk : nstrips;
    int64_t done = 0;
    for (int64_t i = 0; i < cnt; i++) {
      // ev_fact is only recorded by the owner of a block row and ev_gather / head2 / rest not for the last strip: an event
      // that was never recorded reports success - so count "complete" entries, the first gap shows where the chain stopped
      if (hipEventQuery((*vs[v])[(size_t)i]) == hipSuccess) done++;
      else break;
    }
    (void)hipGetLastError();
    out9[v] = done;
  }
  out9[7] = d->nblk; out9[8] = nstrips;
  return CAP_OK;
}

}  // extern "C"
