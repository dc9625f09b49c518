// Host-side factorization schedule (C++ behind the C ABI) for the Cholesky hot path.
//
// Replaces cholesky::cholinv::factor / invoke / base_case (reference
// src/alg/cholesky/cholinv/cholinv.hpp:6-183) and the SUMMA carrier it calls
// (src/alg/matmult/summa/summa.hpp) for a GPU-resident matrix.
//
// Two schedules share the same kernels:
//  * reference-shaped recursion (complete_inv = 0 / 1): cholinv.hpp:85-165 restated for
//    one device - recurse on A11, R12 = Ri11^T A12 ("TRSM by inverse", :118-121),
//    A22 -= R12^T R12 (:128-137), recurse on A22, Ri12 = -Ri11 R12 Ri22 (:150-154, skipped
//    at the root when complete_inv == 0, :147).  Produces R and R^-1 like upstream.
//  * blocked right-looking Cholesky with look-ahead (complete_inv = -1):
//    panel width nb; the nb x nb diagonal block is factored AND inverted by the recursion
//    above (exactly upstream's "invert then multiply" at tile granularity), the block row
//    is solved with one GEMM against the inverse, the trailing matrix is updated by a
//    SYRK on upper tiles only.  The update of the NEXT block row runs first on a
//    high-priority stream followed by the next panel factorization, overlapping with the
//    bulk of the trailing update on the main stream (HIP events, no host sync).
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <vector>

#include "common.h"

namespace {

// ---- reference-shaped recursion on the window [off, off+n) of R (in place) and Ri ----------
// Ri's strictly-lower part must be zero (it is a GEMM operand as a full square).
struct RecCtx {
  double* R; int64_t ldr;
  double* Ri; int64_t ldi;
  double* W; int64_t wcap;   // scratch >= max n1*n2 doubles
  int* info;
  int64_t leaf;
  int64_t split;             // root split shift (cholinv.hpp:107)
  int complete_inv;
  hipStream_t s;
  int ctag = 0;              // CAP_TAG_NO_ATOMIC when R is unverified caller memory (cap_dpotrf)
};

int64_t pick_split(int64_t n, int64_t leaf) {
  // below the root the partition is free: keep blocks multiples of 128 so the fast GEMM path applies
  int64_t h = n / 2;
  int64_t q = n >= 512 ? 128 : leaf;
  h = (h / q) * q;
  if (h <= 0) h = std::min<int64_t>(leaf, n - 1);
  return h;
}

int rec_cholinv(const RecCtx& c, int64_t off, int64_t n, bool is_root, int64_t info_base) {
  double* R = c.R + off + off * c.ldr;
  double* Ri = c.Ri + off + off * c.ldi;
  // is_root is only set when upstream itself would partition the root (see cap_cholinv_factor):
  // then the n >> split partition is honoured even for tiny matrices, because with
  // complete_inv == 0 it decides which block of R^-1 stays empty (cholinv.hpp:107,147)
  if (n <= c.leaf && !is_root) return cap_leaf_cholinv(R, c.ldr, Ri, c.ldi, (int)n, 1, c.info, (int)(info_base + off), c.s);
  int64_t n1 = is_root ? (n >> c.split) : pick_split(n, c.leaf);
  if (n1 <= 0 || n1 >= n) n1 = pick_split(n, c.leaf);
  int64_t n2 = n - n1;
  CAP_TRY(rec_cholinv(c, off, n1, false, info_base));
  double* R12 = R + n1 * c.ldr;
  double* R22 = R + n1 + n1 * c.ldr;
  double* Ri12 = Ri + n1 * c.ldi;
  double* Ri22 = Ri + n1 + n1 * c.ldi;
  if (n1 * n2 > c.wcap) return CAP_ERR_ALLOC;
  // R12 = Ri11^T * A12  (out of place into W, then back: the reference serializes through
  // rect_table1 the same way, cholinv.hpp:122-125)
  // tags: 2 panel-chain priority, 8 / 16 / 32 triangular operand hints (K ranges that only see zeros are skipped)
  CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, n1, n2, n1, 1.0, Ri, c.ldi, R12, c.ldr, 0.0, c.W, n1, 0, c.s, 2 | 16));
  CAP_TRY(cap_copy_rect(c.W, n1, R12, c.ldr, n1, n2, c.s));
  // A22 -= R12^T R12 on upper tiles
  CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, n2, n2, n1, -1.0, R12, c.ldr, R12, c.ldr, 1.0, R22, c.ldr, 1, c.s, 2 | c.ctag));
  CAP_TRY(rec_cholinv(c, off + n1, n2, false, info_base));
  if (!(is_root && c.complete_inv == 0)) {
    // Ri12 = -Ri11 * (R12 * Ri22)
    CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, n1, n2, n2, 1.0, R12, c.ldr, Ri22, c.ldi, 0.0, c.W, n1, 0, c.s, 2 | 8));
    CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, n1, n2, n1, -1.0, Ri, c.ldi, c.W, n1, 0.0, Ri12, c.ldi, 0, c.s, 2 | 32));
  }
  return CAP_OK;
}

// ---- one-launch diagonal-block chain: workgroup count (process default, per-thread override) + the per-stream state of its launches ---
// Resident workgroups of a launch (0 / 1: one launch per step, the round-3 form): the process default comes from CAP_CHAIN_COOP; a plan
// that was given its own value (option "chain_coop") installs it for the duration of ITS factor call on the calling host thread
// (CapChainScope), and a caller that launches on a CU-masked stream bounds the count the same way (cap_chain_coop_cap) - both are
// thread_local: two host threads / two plans never see each other's settings (they were process globals in round 4).
const int g_coop_default = CAP_ENV("CAP_CHAIN_COOP") ? atoi(CAP_ENV("CAP_CHAIN_COOP")) : 32;
thread_local int tl_coop_wgs = -1;        // -1: the process default
thread_local int tl_coop_cap = 0;
// Per (device, stream): four counter words (end-of-step meetings, exit count, mid-step meetings, state: 0 / 1 = a workgroup gave up
// waiting / 2 = recovery under way) and a backup of the diagonal block's upper 64 x 64 blocks (n <= 1024: 136 blocks = 4.25 MiB).
// Launches on one stream are ordered, so they can share the words (the kernel leaves them zero) and the backup; launches on
// different streams never share (round 4's ring of 64 slots could hand one slot to two streams' launches that were enqueued far
// apart but ran together).
constexpr int COOP_MAX_NBLK = 16;
constexpr size_t COOP_BACKUP_DOUBLES = (size_t)COOP_MAX_NBLK * (COOP_MAX_NBLK + 1) / 2 * 64 * 64;
struct CoopSlot { int dev; hipStream_t s; int* ctr; double* backup; };
std::mutex g_coop_mu;
std::vector<CoopSlot>* g_coop_slots = nullptr;
int* g_coop_fallbacks[16] = {};           // per device: how many chains were re-run after a workgroup gave up waiting (device word)
hipEvent_t g_coop_fb_zeroed[16] = {};     // recorded behind the one memset that zeroes those words
hipStream_t g_coop_init_stream[16] = {};  // ... on a stream that lives as long as the process (never the caller's: plans come and go)
long long* g_coop_trace = nullptr;
int64_t g_coop_trace_at = -1;
// (no host synchronisation in here: the first chain of a rank is enqueued in the middle of a multi-stream, multi-rank schedule, and a
// host that waits for its device there waits for collectives whose partners may not have been enqueued yet.)  The fallback words of a
// device are zeroed ONCE, on a stream of this module; every stream that is handed them
// waits for that memset's event when its slot is created (chains on other streams read fallbacks[1] with no ordering
// against the memset; hipMalloc does not zero).  A failing memset leaves nothing published.
int coop_slot(int** ctr, double** backup, int** fallbacks, hipStream_t s) {
  int dev = 0;
  CAP_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return CAP_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lk(g_coop_mu);
  if (!g_coop_slots) g_coop_slots = new std::vector<CoopSlot>();
  if (!g_coop_fallbacks[dev]) {
    // The memset and its event live on a stream of THIS module that is never destroyed: the event used to be recorded on the asking plan's
    // stream, and querying it after that plan (and its streams) had been destroyed failed once in a full suite run with "operation not
    // permitted on an event last recorded in a capturing stream" (round 6, profiles/r06_experiments.md section 9).
    int* fb = nullptr; hipEvent_t ev = nullptr;
    if (!g_coop_init_stream[dev]) CAP_HIP(hipStreamCreateWithFlags(&g_coop_init_stream[dev], hipStreamNonBlocking));
    CAP_HIP(hipMalloc((void**)&fb, 4 * sizeof(int)));
    if (hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess || hipMemsetAsync(fb, 0, 4 * sizeof(int), g_coop_init_stream[dev]) != hipSuccess ||
        hipEventRecord(ev, g_coop_init_stream[dev]) != hipSuccess) {
      (void)hipGetLastError();
      if (ev) (void)hipEventDestroy(ev);
      (void)hipFree(fb);
      return CAP_ERR_HIP;
    }
    g_coop_fallbacks[dev] = fb; g_coop_fb_zeroed[dev] = ev;
  }
  *fallbacks = g_coop_fallbacks[dev];
  for (const CoopSlot& c : *g_coop_slots)
    if (c.dev == dev && c.s == s) { *ctr = c.ctr; *backup = c.backup; return CAP_OK; }
  // first chain of this stream: behind the zeroing of the device's fallback words (no event: cap_chain_inject_timeouts zeroed them synchronously)
  if (g_coop_fb_zeroed[dev]) {
    const hipError_t q = hipEventQuery(g_coop_fb_zeroed[dev]);          // long complete for every stream but the first few: then there is nothing to wait for
    if (q == hipErrorNotReady) { (void)hipGetLastError(); CAP_HIP(hipStreamWaitEvent(s, g_coop_fb_zeroed[dev], 0)); }
    else if (q != hipSuccess) CAP_HIP(q);
  }
  CoopSlot c{dev, s, nullptr, nullptr};
  CAP_HIP(hipMalloc((void**)&c.ctr, 4 * sizeof(int)));
  if (hipMalloc((void**)&c.backup, COOP_BACKUP_DOUBLES * sizeof(double)) != hipSuccess) { (void)hipGetLastError(); c.backup = nullptr; }   // no backup: no recovery
  if (hipMemsetAsync(c.ctr, 0, 4 * sizeof(int), s) != hipSuccess) {
    (void)hipGetLastError(); (void)hipFree(c.ctr); if (c.backup) (void)hipFree(c.backup);
    return CAP_ERR_HIP;
  }
  g_coop_slots->push_back(c);
  *ctr = c.ctr; *backup = c.backup;
  return CAP_OK;
}

}  // namespace

// the stream is about to be destroyed: give its counter words and backup buffer back.  The slot is found by the stream HANDLE alone (a
// plan may be destroyed while another device is current - ADVICE round 5: the per-device lookup leaked it then) and the buffers are freed
// after the stream itself has drained, so that nothing of this stream can still be using them.
void cap_coop_slot_release(hipStream_t s) {
  if (!s) return;
  int* ctr = nullptr; double* backup = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_coop_mu);
    if (!g_coop_slots) return;
    for (size_t i = 0; i < g_coop_slots->size(); i++)
      if ((*g_coop_slots)[i].s == s) {
        ctr = (*g_coop_slots)[i].ctr; backup = (*g_coop_slots)[i].backup;
        g_coop_slots->erase(g_coop_slots->begin() + (std::ptrdiff_t)i);
        break;
      }
  }
  if (!ctr && !backup) return;
  (void)hipStreamSynchronize(s);
  if (ctr) (void)hipFree(ctr);
  if (backup) (void)hipFree(backup);
}

namespace {
int64_t rec_work_size(int64_t n) { return cap_round_up(std::max<int64_t>((n / 2 + 1) * (n / 2 + 1), 128 * n), 2); }

// Diagonal-block fast path (n = 64 * nblk <= 1024): 64-blocked right-looking potrf with ONE fused launch per step
// (cap_panel64_solve_update; its diagonal workgroup also runs the next leaf) + the inverse assembled level by level, one fused
// launch per level (cap_trinv_merge): nblk + log2(nblk) dependent launches instead of the recursion's ~5.4 nblk (11 instead of
// 43 at n = 512; 21 before the leaf was folded in and the two products of a level were fused).  Same arithmetic,
// different association order.  Ri's strictly-lower blocks must be zero on entry (they are never written).
int blocked_cholinv(double* R, int64_t ldr, double* Ri, int64_t ldi, int64_t n, double* W, int64_t wcap, int* info,
                    int64_t info_base, hipStream_t s) {
  const int nblk = (int)(n / 64);
  if (128 * n > wcap) return CAP_ERR_ALLOC;
  static const bool fold = CAP_ENV("CAP_FOLD_LEAF") ? atoi(CAP_ENV("CAP_FOLD_LEAF")) != 0 : true;
  int coop = cap_chain_coop_get();
  if (tl_coop_cap > 0) coop = std::min(coop, tl_coop_cap);
  coop = std::min(coop, cap_chain64_coop_max_resident());      // never more workgroups than the device can hold at once
  int64_t merged = 0;             // the launch above has already assembled the inverse up to pairs of this size
  if (coop >= 2 && nblk >= 4) {
    // the whole factor phase in one launch of `coop` resident workgroups (chain64_coop_kernel, leaf.hip)
    int* ctr = nullptr; double* backup = nullptr; int* fallbacks = nullptr;
    CAP_TRY(coop_slot(&ctr, &backup, &fallbacks, s));
    if (nblk > COOP_MAX_NBLK) backup = nullptr;
    // release / acquire fences around the meeting counter: on by default since round 5 (the relaxed protocol relies on behaviour the
    // AMDGPU memory model does not promise; measured cost in profiles/r05_experiments.log), CAP_CHAIN_FENCE=0 for the A/B run
    static const int fence = CAP_ENV("CAP_CHAIN_FENCE") ? atoi(CAP_ENV("CAP_CHAIN_FENCE")) : 1;
    long long* trace = nullptr;
    if (g_coop_trace && g_coop_trace_at-- == 0) trace = g_coop_trace;       // instrumentation of ONE launch (cap_chain_trace_arm)
    static const int merge_env = CAP_ENV("CAP_CHAIN_MERGE") ? atoi(CAP_ENV("CAP_CHAIN_MERGE")) : 256;   // inverse levels done in the same launch
    merged = std::min<int64_t>(merge_env, n / 2);
    CAP_TRY(cap_chain64_coop(R, ldr, Ri, ldi, nblk, info, (int)info_base, ctr, coop, fence, (int)merged, s, trace, backup, fallbacks));
  } else if (fold) {
    // one launch per step: the fused solve + update of step i also runs the leaf of step i + 1 (leaf.hip).  The solved block
    // row of step i sits in half (i & 1) of W until the launch of step i + 1 moves it into R (the other workgroups of step i
    // still read the unsolved blocks), the last step - a single workgroup - writes its piece in place.
    CAP_TRY(cap_leaf_cholinv(R, ldr, Ri, ldi, 64, 1, info, (int)info_base, s));
    for (int i = 0; i + 1 < nblk; i++) {
      double* Dii = Ri + (int64_t)i * 64 * (ldi + 1);
      double* Xw = W + (int64_t)(i & 1) * 64 * n;
      const double* cj_src = i ? W + (int64_t)((i - 1) & 1) * 64 * n : nullptr;
      double* cj_dst = i ? R + (int64_t)(i - 1) * 64 + (int64_t)i * 64 * ldr : nullptr;
      CAP_TRY(cap_panel64_solve_update(R, ldr, Dii, ldi, i, nblk, Xw, s, Ri + (int64_t)(i + 1) * 64 * (ldi + 1), ldi, info,
                                       (int)(info_base + (i + 1) * 64), cj_src, cj_dst, ldr, i ? (nblk - i) * 64 : 0,
                                       nblk - 1 - i == 1));
    }
  } else {
  double* Xs = W;      // block row solved by the fused step, moved into R by the NEXT leaf launch (W is free until the inverse phase)
  for (int i = 0; i < nblk; i++) {
    double* Rii = R + (int64_t)i * 64 * (ldr + 1);
    double* Dii = Ri + (int64_t)i * 64 * (ldi + 1);
    if (i == 0) CAP_TRY(cap_leaf_cholinv(Rii, ldr, Dii, ldi, 64, 1, info, (int)(info_base + i * 64), s));
    else CAP_TRY(cap_leaf_cholinv(Rii, ldr, Dii, ldi, 64, 1, info, (int)(info_base + i * 64), s, Xs,
                                  R + (int64_t)(i - 1) * 64 + (int64_t)i * 64 * ldr, ldr, (nblk - i) * 64));
    CAP_TRY(cap_panel64_solve_update(R, ldr, Dii, ldi, i, nblk, Xs, s));
  }
  }
  for (int64_t h = 64; h < n; h *= 2) {
    const int npairs = (int)(n / (2 * h));
    if (h <= merged) continue;
    if (h * h * npairs > wcap) return CAP_ERR_ALLOC;
    static const bool merge1 = CAP_ENV("CAP_TRINV_MERGE") ? atoi(CAP_ENV("CAP_TRINV_MERGE")) != 0 : true;
    if (merge1 && h <= 256) {
      CAP_TRY(cap_trinv_merge(R, ldr, Ri, ldi, h, npairs, s));     // both products of the level in one launch (leaf.hip)
    } else if (h <= 256) {
      // W_z = R12_z * Ri22_z ;  Ri12_z = -Ri11_z * W_z   for every aligned pair z at once
      CAP_TRY(cap_gemm_small_batched(CAP_NOTRANS, CAP_NOTRANS, h, h, h, 1.0, R + h * ldr, ldr, 2 * h * (ldr + 1),
                                     Ri + h + h * ldi, ldi, 2 * h * (ldi + 1), 0.0, W, h, h * h, npairs, s));
      CAP_TRY(cap_gemm_small_batched(CAP_NOTRANS, CAP_NOTRANS, h, h, h, -1.0, Ri, ldi, 2 * h * (ldi + 1), W, h, h * h, 0.0,
                                     Ri + h * ldi, ldi, 2 * h * (ldi + 1), npairs, s));
    } else {
      for (int z = 0; z < npairs; z++) {
        double* Rz = R + (int64_t)z * 2 * h * (ldr + 1); double* Iz = Ri + (int64_t)z * 2 * h * (ldi + 1);
        CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, h, h, h, 1.0, Rz + h * ldr, ldr, Iz + h + h * ldi, ldi, 0.0, W, h, 0, s, 2));
        CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, h, h, h, -1.0, Iz, ldi, W, h, 0.0, Iz + h * ldi, ldi, 0, s, 2));
      }
    }
  }
  return CAP_OK;
}

// ---- in-place triangular inverse (upper): R <- R^-1, recursion + leaf kernel -----------------
int rec_trtri(double* R, int64_t ldr, int64_t n, double* W, int64_t wcap, int64_t leaf, hipStream_t s) {
  if (n <= leaf) return cap_leaf_trtri(R, ldr, R, ldr, (int)n, s);
  int64_t n1 = pick_split(n, leaf), n2 = n - n1;
  double* R12 = R + n1 * ldr;
  double* R22 = R + n1 + n1 * ldr;
  CAP_TRY(rec_trtri(R, ldr, n1, W, wcap, leaf, s));
  CAP_TRY(rec_trtri(R22, ldr, n2, W, wcap, leaf, s));
  if (n1 * n2 > wcap) return CAP_ERR_ALLOC;
  // R12 <- -Ri11 * (R12 * Ri22); the strictly-lower parts of R11/R22 must be zero
  CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, n1, n2, n2, 1.0, R12, ldr, R22, ldr, 0.0, W, n1, 0, s));
  CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, n1, n2, n1, -1.0, R, ldr, W, n1, 0.0, R12, ldr, 0, s));
  return CAP_OK;
}

// ---- R^-1 of the blocked factorization (complete_inv = 0 / 1 on matrices of several panels) -------------------------------
// The panel chain of the right-looking sweep already inverts every nb x nb diagonal block; the rest of R^-1 is assembled
// by a binary tree over the panels: node (a, mid, b) fills  Ri[a:mid, mid:b] = -Ri[a:mid, a:mid] R[a:mid, mid:b] Ri[mid:b, mid:b]
// (the reference's inverse completion, cholinv.hpp:144-159, at every level instead of only along the recursion).  Each node is
// split so that as little as possible is left for the end of the factorization:
//   phase 1   W = -Ri11 R12     ready once panel mid/nb - 1 is factored and solved (Ri11 and the rows of R12 are final)
//   phase 2   Ri12 = W Ri22     ready once the diagonal block of panel b/nb - 1 is inverted
// One node per tree depth is pending at a time, so W lives in one scratch slot per depth.
struct InvNode { int64_t a, mid, b; int depth; int64_t woff; };
struct InvTree {
  std::vector<InvNode> nodes;                 // post-order: children before parents, non-decreasing b
  std::vector<std::vector<int>> ph1, ph2;     // per panel k: nodes whose phase 1 / phase 2 becomes ready after panel k
  int64_t scratch = 0;                        // doubles
};

void inv_tree_rec(InvTree& t, const std::vector<int64_t>& pb, int ia, int ib, int root_mid, int depth) {
  if (ib - ia < 2) return;
  int im = root_mid > ia && root_mid < ib ? root_mid : ia + (ib - ia + 1) / 2;
  inv_tree_rec(t, pb, ia, im, -1, depth + 1);
  inv_tree_rec(t, pb, im, ib, -1, depth + 1);
  t.nodes.push_back(InvNode{pb[(size_t)ia], pb[(size_t)im], pb[(size_t)ib], depth, 0});
  t.ph1[(size_t)im - 1].push_back((int)t.nodes.size() - 1);
  t.ph2[(size_t)ib - 1].push_back((int)t.nodes.size() - 1);
}

// pb: panel boundaries (0 = pb[0] < ... < pb[np] = n); root_mid: forced panel index of the root split (or -1);
// skip_root: the root node is left out (complete_inv == 0 with the root partition on a panel boundary)
InvTree inv_tree_build(const std::vector<int64_t>& pb, int root_mid, bool skip_root) {
  InvTree t;
  const int np = (int)pb.size() - 1;
  t.ph1.assign((size_t)std::max(np, 1), {}); t.ph2.assign((size_t)std::max(np, 1), {});
  inv_tree_rec(t, pb, 0, np, root_mid, 0);
  if (skip_root && !t.nodes.empty()) {
    const int r = (int)t.nodes.size() - 1;            // post-order: the root is last
    for (auto* v : {&t.ph1, &t.ph2}) for (auto& l : *v) l.erase(std::remove(l.begin(), l.end(), r), l.end());
    t.nodes.pop_back();
  }
  int maxd = -1;
  for (const InvNode& nd : t.nodes) maxd = std::max(maxd, nd.depth);
  std::vector<int64_t> need((size_t)(maxd + 1), 0);
  for (const InvNode& nd : t.nodes) need[(size_t)nd.depth] = std::max(need[(size_t)nd.depth], cap_round_up((nd.mid - nd.a) * (nd.b - nd.mid), 2));
  std::vector<int64_t> off((size_t)(maxd + 2), 0);
  for (int d = 0; d <= maxd; d++) off[(size_t)d + 1] = off[(size_t)d] + need[(size_t)d];
  for (InvNode& nd : t.nodes) nd.woff = off[(size_t)nd.depth];
  t.scratch = off[(size_t)(maxd + 1)];
  return t;
}

std::vector<int64_t> panel_bounds(int64_t n, int64_t nb) {
  std::vector<int64_t> pb;
  for (int64_t j = 0; j < n; j += nb) pb.push_back(j);
  pb.push_back(n);
  return pb;
}

}  // namespace

// =================================================================================================
// plan
// =================================================================================================
struct cap_cholinv_plan {
  int64_t n; int complete_inv; int64_t split, bc_mult_dim; char dir;
  cap_comm* comm;
  // schedule knobs
  int64_t nb, leaf; int lookahead;
  int64_t outer;   // outer strip height NB (multiple of nb): K of the big trailing SYRK
  int64_t tail;    // trailing sizes <= tail fall back to nb-wide strips
  int64_t serial_m; // remaining rows below which the chain is no longer overlapped with the bulk update (see right_looking)
  int fastdiag;     // diagonal blocks by the 64-blocked fused path (default) instead of the recursion
  int ctag;         // CAP_TAG_NO_ATOMIC when the factor lives in caller memory that is not a plain device allocation (cap_dpotrf)
  int depth2;       // split each bulk update into head (next-next strip's rows) + rest: look-ahead depth 2
  int chain_coop;   // resident workgroups of the one-launch diagonal-block chain for THIS plan (-1: the process default)
  // device state
  double* R; int64_t ldr;
  double* Rinv; int64_t ldi;      // n x n (complete_inv >= 0) or nb x nb diagonal-block inverse
  double* work; int64_t work_elems;
  int* info_dev;
  hipStream_t s_panel; hipEvent_t ev_panel[2], ev_update[2], ev_fork, ev_join;
  // bulk updates run on an internal stream whose CU mask leaves `reserve` CUs (one per XCD per 8) to the
  // panel stream, so the latency-bound diagonal-block chain is not time-sliced against 512 resident bulk
  // workgroups (hipExtStreamCreateWithCUMask: bit i -> XCD i % 8, tools/cumask_probe.hip)
  hipStream_t s_bulk; hipEvent_t ev_join_b; int64_t reserve; bool bulk_ready;
  // reserve_m > 0: the masks only apply to the chain-bound tail (at most reserve_m columns left) - the bulk update has slack there,
  // so the 10 % a masked launch loses (tools/gemm_bench.bin MASK_OFF) costs nothing while the chain runs at its isolated speed;
  // reserve_m == 0: masks for the whole factorization (measured slower, kept for A/B runs)
  int64_t reserve_m; bool chain_masked; hipEvent_t ev_bulk_sw;
  // factor(A): the A -> R copy is left to right_looking (srcA != nullptr on entry).  fuse_copy: only the first strip's rows are
  // copied, the updates of step 0 - which together write the whole trailing triangle - read their C input from A instead
  const double* srcA; int64_t src_lda; int fuse_copy;
  // ... and the diagonal-block chain itself (leaf / fused-step / assembly kernels: a handful of workgroups each) runs on a stream
  // masked to exactly those reserved CUs, so none of its waves ever shares a SIMD with fp64-MFMA bulk waves
  hipStream_t s_chain; hipEvent_t ev_chain[2];
  // column-split look-ahead (right_looking_split): the panel stream only touches the columns the NEXT chain needs; the wide
  // remainder of every block-row solve / in-strip update runs on s_rest
  int inner_la; bool split_ready;
  int64_t occ1_m;       // bulk updates of trailing matrices this wide or narrower run with one workgroup per CU
  hipStream_t s_rest; hipEvent_t ev_crit[2], ev_near[2], ev_pchain[2][8], ev_psolve[2][8], ev_join_r;
  double* Dring;        // 2 x 8 diagonal-block inverses (nb x nb each, strictly-lower parts stay zero)
  double* Wpan2;        // s_rest's nb x n solve scratch
  bool streams_ready;
  // optional live profile of the dominant kernel (trailing-update SYRK): HIP events on the stream
  // it is launched on, algorithmic flops m(m+1)k per launch
  int profile;
  std::vector<hipEvent_t>* prof_ev; std::vector<double>* prof_flops; int prof_used;
  // multi-GPU plans (comm size > 1): the 1 x P block-cyclic schedule of dist.hip behind the same handle
  cap_dist_plan* dist;
  // option "cyclic_c" = c: the caller speaks the REFERENCE's layout (element-cyclic pieces on topo::square's d x d x c grid,
  // matrix.hpp:8-11) end to end - factor() redistributes the piece into this rank's block columns (redist.hip), get_R / get_Rinv
  // redistribute the result back into a piece on every rank of every layer (construct_R / construct_Rinv, cholinv.hpp:30-46)
  cap_redist_plan* redist; int cyc_c; double* bcA; double* bcOut; int64_t bc_cols;
  // Strip buffers (use_sb): the solved block rows of a strip are written K-contiguously into one of three NB x n buffers
  // (ld = NB) and every update reads its operands from there; the copy into R - R is only the OUTPUT after that - runs on
  // its own stream off the panel stream's critical path (it was 64 x 134 MB of copies on it at N = 32768)
  int use_sb; double* SB; int64_t sb_ld, sb_cols; hipStream_t s_copy; hipEvent_t ev_sbg, ev_copy[4], ev_join_cp; bool sb_ready;
  int pair_rest;    // the far part of the trailing update takes two strips at a time (one K = 2 NB product per pair of steps), see right_looking
  int cnt_paired;   // K = 2 NB far updates of the last factor call (get_option "count_paired")
  // R^-1 on top of the blocked factorization (complete_inv >= 0, see InvTree): its own stream, one event per panel
  int inv_fast;         // 1: blocked factorization + inverse tree (default for n >= 2 nb), 0: the plain recursion of cholinv.hpp:85-165
  int inv_overlap;      // 1: tree nodes are enqueued as their inputs become final (overlapped with the sweep), 0: after it
  int64_t inv_start_m;  // overlapped mode: nothing of the tree is enqueued while more than this many columns are left to factor
  InvTree* itree; bool inv_active; int inv_tree_skip; int64_t inv_tree_mid;
  double* inv_W;        // tree scratch (inside `work`)
  hipStream_t s_inv; std::vector<hipEvent_t>* ev_pf; hipEvent_t ev_inv_done; bool inv_ready; int inv_pending_from;
};

namespace {

// upstream's leaf test at the root (cholinv.hpp:93): a root that is itself a base case gets the full inverse whatever complete_inv
// says (policy.h:199-201 always runs trtri).  The base-case dimension depends on the process grid (cholinv.hpp:15-18: bcDimLocal
// starts at c d): c = d = 1 for the single-GPU plan and the block-column layouts, the caller's c and d when the plan speaks the
// reference's element-cyclic layout (option cyclic_c) - with bc_mult_dim = 0 on the 2 x 2 x 2 grid upstream partitions the root
// (base case = n / 4) where one process would not (found by the CPU compute harness against the oracle, round 5).
bool root_is_base_case(int64_t n, int64_t split, int64_t bc_mult_dim, int64_t c = 1, int64_t d = 1) {
  const int64_t nloc = cap_ceil_div(n, d);
  int64_t bc = c * d;
  if (bc_mult_dim < 0) { for (int64_t i = 0; i < -bc_mult_dim && bc < nloc; i++) bc *= 2; }
  else { for (int64_t i = 0; i < bc_mult_dim; i++) bc /= 2; }
  bc = std::max<int64_t>(1, std::min<int64_t>(nloc, bc));
  const int64_t bc_dim = d * (nloc / bc);
  return (nloc * d <= bc_dim) || ((nloc >> split) < split);
}
// complete_inv and root partition of the inner multi-GPU plan as upstream defines them on the caller's grid (c = 0: block-column layout)
int apply_root_semantics(cap_cholinv_plan* p, int c) {
  int64_t cc = 1, dd = 1;
  if (c > 0) { cc = c; dd = 1; while (dd * dd * cc < cap_comm_size(p->comm)) dd++; }
  const bool base = p->complete_inv == 0 && root_is_base_case(p->n, p->split, p->bc_mult_dim, cc, dd);
  CAP_TRY(cap_dist_set_option(p->dist, "complete_inv", base ? 1 : p->complete_inv));
  CAP_TRY(cap_dist_set_option(p->dist, "split", p->split));
  // upstream cuts the LOCAL dimension (cholinv.hpp:107): global rows [0, (ceil(n / d) >> split) d)
  const int64_t n1 = std::min<int64_t>(p->n, (cap_ceil_div(p->n, dd) >> p->split) * dd);
  return cap_dist_set_option(p->dist, "root_n1", c > 0 ? n1 : 0);
}

// (re)build the reference-layout front end of a multi-GPU plan: redistribution plan + the two block-column staging arrays
int setup_cyclic(cap_cholinv_plan* p, int c) {
  if (p->redist) { (void)cap_redist_plan_destroy(p->redist); p->redist = nullptr; }
  if (p->bcA) { (void)hipFree(p->bcA); p->bcA = nullptr; }
  if (p->bcOut) { (void)hipFree(p->bcOut); p->bcOut = nullptr; }
  p->cyc_c = 0;
  if (c <= 0) return apply_root_semantics(p, 0);
  CAP_TRY(cap_redist_plan_create(&p->redist, p->n, cap_dist_get_option(p->dist, "nb"), p->comm, c, 1));
  p->bc_cols = cap_dist_local_cols(p->dist);
  const size_t bytes = sizeof(double) * (size_t)p->n * (size_t)std::max<int64_t>(p->bc_cols, 1);
  bool ok = hipMalloc((void**)&p->bcA, bytes) == hipSuccess && hipMalloc((void**)&p->bcOut, bytes) == hipSuccess;
  if (!ok) (void)hipGetLastError();
  // every rank must come out of this call with the SAME answer: a rank that alone reported CAP_ERR_ALLOC would stay out of the
  // redistribution all-to-all of the next factor call while its peers wait in it.  One 8-byte sum over the plan's communicator.
  double* flag = nullptr;
  if (hipMalloc((void**)&flag, sizeof(double)) != hipSuccess) { (void)hipGetLastError(); return CAP_ERR_ALLOC; }
  double h = ok ? 0.0 : 1.0;
  int st = hipMemcpy(flag, &h, sizeof(double), hipMemcpyHostToDevice) == hipSuccess ? CAP_OK : CAP_ERR_HIP;
  if (st == CAP_OK) st = cap_comm_allreduce_sum(p->comm, flag, 1, nullptr);
  if (st == CAP_OK && (hipStreamSynchronize(nullptr) != hipSuccess || hipMemcpy(&h, flag, sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)) st = CAP_ERR_HIP;
  (void)hipFree(flag);
  if (st != CAP_OK || h != 0.0) {
    // (the caller's plan stays usable on the block-column layout)
    if (p->bcA) { (void)hipFree(p->bcA); p->bcA = nullptr; }
    if (p->bcOut) { (void)hipFree(p->bcOut); p->bcOut = nullptr; }
    (void)cap_redist_plan_destroy(p->redist); p->redist = nullptr;
    (void)apply_root_semantics(p, 0);
    return st != CAP_OK ? st : CAP_ERR_ALLOC;
  }
  p->cyc_c = c;
  return apply_root_semantics(p, c);
}

int64_t default_nb(int64_t n, int64_t bc_mult_dim) {
  // Panel width of the GPU schedule.  The sweet spot on MI355X is 512 (256 for small matrices): wider panels
  // lengthen the latency-bound diagonal-block chain, narrower ones starve the MFMA update of K.  upstream's
  // base-case knob (bcDimension of cholinv.hpp:15-18 with c = d = 1) can only narrow it.
  int64_t nb = n >= 8192 ? 512 : 256;
  if (bc_mult_dim < 0) {
    int64_t bc = 1;
    for (int64_t i = 0; i < -bc_mult_dim && bc < n; i++) bc *= 2;
    bc = std::max<int64_t>(1, std::min<int64_t>(n, bc));
    const int64_t hint = ((n / bc) / 128) * 128;
    if (hint >= 128) nb = std::min(nb, hint);
    else nb = std::min<int64_t>(nb, 128);
  }
  return nb;
}

int plan_alloc(cap_cholinv_plan* p) {
  const int64_t n = p->n;
  p->ldr = cap_round_up(n, 2);   // power-of-two column strides were measured harmless (HBM address hashing)
  CAP_HIP(hipMalloc((void**)&p->R, sizeof(double) * p->ldr * n));
  // R's strictly-lower triangle is zeroed ONCE here: no kernel of either schedule ever writes below the
  // diagonal (upper-tile masks everywhere), so factor() only has to copy A's upper triangle in
  CAP_HIP(hipMemset(p->R, 0, sizeof(double) * p->ldr * n));
  if (p->complete_inv >= 0) {
    p->ldi = cap_round_up(n, 2);
    CAP_HIP(hipMalloc((void**)&p->Rinv, sizeof(double) * p->ldi * n));
    CAP_HIP(hipMemset(p->Rinv, 0, sizeof(double) * p->ldi * n));
    // blocked factorization + inverse tree (the path cap_cholinv_factor takes for n >= 2 nb): chain scratch + panel scratch + the
    // tree's scratch for THIS plan's panels and root partition; otherwise the plain recursion's (n/2 + 1)^2.  Both paths check
    // the size again at factor time and grow the buffer if an option (nb, inv_fast, leaf) moved the plan to the other one.
    if (p->inv_fast && n >= 2 * p->nb && p->leaf == CAP_LEAF_MAX) {
      const int64_t n1 = n >> p->split;
      const bool on_panel = n1 > 0 && n1 < n && n1 % p->nb == 0;
      const InvTree t = inv_tree_build(panel_bounds(n, p->nb), on_panel ? (int)(n1 / p->nb) : -1, on_panel && p->complete_inv == 0);
      p->work_elems = rec_work_size(p->nb) + cap_round_up(p->nb * n, 2) + t.scratch;
    } else {
      p->work_elems = rec_work_size(n);
    }
  } else {
    p->ldi = p->nb;
    CAP_HIP(hipMalloc((void**)&p->Rinv, sizeof(double) * p->nb * p->nb));
    CAP_HIP(hipMemset(p->Rinv, 0, sizeof(double) * p->nb * p->nb));
    p->work_elems = rec_work_size(p->nb) + cap_round_up(p->nb * n, 2);
  }
  CAP_HIP(hipMalloc((void**)&p->work, sizeof(double) * p->work_elems));
  CAP_HIP(hipMalloc((void**)&p->info_dev, sizeof(int)));
  CAP_HIP(hipMemset(p->info_dev, 0, sizeof(int)));
  return CAP_OK;
}

int ensure_streams(cap_cholinv_plan* p) {
  if (p->streams_ready) return CAP_OK;
  int lo = 0, hi = 0;
  CAP_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CAP_HIP(hipStreamCreateWithPriority(&p->s_panel, hipStreamNonBlocking, hi));
  for (int i = 0; i < 2; i++) {
    CAP_HIP(hipEventCreateWithFlags(&p->ev_panel[i], hipEventDisableTiming));
    CAP_HIP(hipEventCreateWithFlags(&p->ev_update[i], hipEventDisableTiming));
  }
  CAP_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
  CAP_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
  CAP_HIP(hipEventCreateWithFlags(&p->ev_join_b, hipEventDisableTiming));
  p->streams_ready = true;
  return CAP_OK;
}

int ensure_bulk_stream(cap_cholinv_plan* p) {
  if (p->bulk_ready || p->reserve <= 0) return CAP_OK;
  uint32_t mask[8];
  for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu;
  for (int64_t b = 0; b < p->reserve && b < 128; b++) mask[b / 32] &= ~(1u << (b % 32));
  CAP_HIP(hipExtStreamCreateWithCUMask(&p->s_bulk, 8, mask));
  for (int i = 0; i < 8; i++) mask[i] = ~mask[i];
  CAP_HIP(hipExtStreamCreateWithCUMask(&p->s_chain, 8, mask));
  for (int i = 0; i < 2; i++) CAP_HIP(hipEventCreateWithFlags(&p->ev_chain[i], hipEventDisableTiming));
  CAP_HIP(hipEventCreateWithFlags(&p->ev_bulk_sw, hipEventDisableTiming));
  p->bulk_ready = true;
  return CAP_OK;
}

// diagonal block of the nb-wide panel starting at j0: R_jj and Dinv = R_jj^-1 (jb x jb, ld = nb, strictly-lower part stays zero)
int panel_chain(cap_cholinv_plan* p, double* R, int64_t ldr, int64_t j0, int64_t jb, double* Dinv, hipStream_t s) {
  CapRange range("CI::factor_diag");            // cholinv.hpp:95-99
  double* Wrec = p->work;                       // rec scratch
  if (p->fastdiag && jb % 64 == 0 && jb >= 128 && jb <= 1024 && (jb & (jb - 1)) == 0 && p->leaf == CAP_LEAF_MAX) {
    hipStream_t sc = s;
    if (p->bulk_ready && p->chain_masked && s != p->s_bulk) {          // (s == s_bulk: the non-overlapped tail, nothing to hide from)
      sc = p->s_chain;
      CAP_HIP(hipEventRecord(p->ev_chain[0], s));
      CAP_HIP(hipStreamWaitEvent(sc, p->ev_chain[0], 0));
    }
    if (sc != s) cap_chain_coop_cap((int)std::max<int64_t>(p->reserve, 1));   // masked stream: no more resident workgroups than its CUs
    const int st_chain = blocked_cholinv(R + j0 + j0 * ldr, ldr, Dinv, p->ldi, jb, Wrec, rec_work_size(p->nb), p->info_dev, j0, sc);
    if (sc != s) cap_chain_coop_cap(0);
    CAP_TRY(st_chain);
    if (sc != s) {
      CAP_HIP(hipEventRecord(p->ev_chain[1], sc));
      CAP_HIP(hipStreamWaitEvent(s, p->ev_chain[1], 0));
    }
  } else {
    RecCtx c{R + j0 + j0 * ldr, ldr, Dinv, p->ldi, Wrec, rec_work_size(p->nb), p->info_dev, p->leaf, 1, 1, s, p->ctag};
    CAP_TRY(rec_cholinv(c, 0, jb, false, j0));
  }
  return CAP_OK;
}

// block-row solve of panel j0 on the columns [c0, c1): R[j0:j0+jb, c0:c1] <- Dinv^T R[j0:j0+jb, c0:c1]  (Wp: jb x n scratch,
// addressed by global column so that concurrent solves of disjoint column ranges can share it)
int panel_solve(cap_cholinv_plan* p, double* R, int64_t ldr, int64_t j0, int64_t jb, const double* Dinv, int64_t c0, int64_t c1,
                double* Wp, hipStream_t s) {
  const int64_t m = c1 - c0;
  if (m <= 0) return CAP_OK;
  CapRange range("CI::trsm");                   // cholinv.hpp:114-125
  double* Rpan = R + j0 + c0 * ldr;
  double* W = Wp + c0 * jb;
  CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, jb, m, jb, 1.0, Dinv, p->ldi, Rpan, ldr, 0.0, W, jb, 0, s, 2 | 16));
  CAP_TRY(cap_copy_rect(W, jb, Rpan, ldr, jb, m, s));
  return CAP_OK;
}

// factor the nb-wide panel starting at j0: diagonal block (R, Dinv), then the block row solve
int inverse_after_panel(cap_cholinv_plan* p, int64_t k, int64_t cols_left, hipStream_t s);

// where the solved rows of the strip being factored go (strip-buffer mode): element (Js + r, j) at SB[r + j * ld]
struct StripCtx { double* SB; int64_t ld, Js; };

int panel_factor(cap_cholinv_plan* p, double* R, int64_t ldr, int64_t n, int64_t j0, int64_t jb, hipStream_t s, const StripCtx* sc = nullptr) {
  // complete_inv >= 0 on the blocked path: the diagonal-block inverse IS the diagonal block of R^-1 (written in place, ld = ldi)
  double* Dinv = p->inv_active ? p->Rinv + j0 + j0 * p->ldi : p->Rinv;
  CAP_TRY(panel_chain(p, R, ldr, j0, jb, Dinv, s));
  const int64_t j1 = j0 + jb, m = n - j1;
  if (!sc) {
    CAP_TRY(panel_solve(p, R, ldr, j0, jb, Dinv, j1, n, p->work + rec_work_size(p->nb), s));
    if (p->inv_active) CAP_TRY(inverse_after_panel(p, j0 / p->nb, m, s));
    return CAP_OK;
  }
  // strip-buffer mode: the solve writes straight into the strip buffer; R gets its copy on the copy stream
  hipStream_t scp = p->s_copy;
  if (m > 0) {
    CapRange range("CI::trsm");
    double* Rpan = R + j0 + j1 * ldr;
    double* Srow = sc->SB + (j0 - sc->Js) + j1 * sc->ld;
    CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, jb, m, jb, 1.0, Dinv, p->ldi, Rpan, ldr, 0.0, Srow, sc->ld, 0, s, 2 | 16));
    CAP_HIP(hipEventRecord(p->ev_sbg, s));
    CAP_HIP(hipStreamWaitEvent(scp, p->ev_sbg, 0));
    CAP_TRY(cap_copy_rect(Srow, sc->ld, Rpan, ldr, jb, m, scp));
  } else if (p->inv_active) {
    CAP_HIP(hipEventRecord(p->ev_sbg, s));
    CAP_HIP(hipStreamWaitEvent(scp, p->ev_sbg, 0));
  }
  if (p->inv_active) CAP_TRY(inverse_after_panel(p, j0 / p->nb, m, scp));     // the tree reads R12 from R: behind the copy
  return CAP_OK;
}

// trailing update C[M x N] -= A^T B on upper tiles (the dominant kernel), optionally bracketed by events.
// M == N, A == B: the SYRK on the bulk of the trailing matrix; M < N: the strip of rows updated first so
// that the panel chain of the step after next can start early (look-ahead depth 2).
int trailing_update(cap_cholinv_plan* p, int64_t M, int64_t N, int64_t k, const double* A, const double* B, double* C,
                    int64_t ldr, hipStream_t s, int64_t ldab = 0, const double* Cin = nullptr, int64_t ldcin = 0) {
  if (ldab == 0) ldab = ldr;              // operands inside R itself (no strip buffer)
  CapRange range("CI::tmu");                    // cholinv.hpp:129-136
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (p->profile && p->prof_ev) {
    if ((size_t)p->prof_used + 2 > p->prof_ev->size()) {
      for (int i = 0; i < 64; i++) { hipEvent_t e; CAP_HIP(hipEventCreate(&e)); p->prof_ev->push_back(e); }
    }
    e0 = (*p->prof_ev)[p->prof_used]; e1 = (*p->prof_ev)[p->prof_used + 1];
    CAP_HIP(hipEventRecord(e0, s));
  }
  // chain-bound tail (N columns left <= occ1_m): one bulk workgroup per CU instead of two, see launch_tn_dma
  const int occ = (p->occ1_m > 0 && N <= p->occ1_m) ? -1 : 0;
  CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, M, N, k, -1.0, A, ldab, B, ldab, 1.0, C, ldr, 1, s, 1 | p->ctag, occ, Cin, ldcin));
  if (e0) {
    CAP_HIP(hipEventRecord(e1, s));
    p->prof_used += 2;
    // algorithmic flops: 2k per element of the upper staircase  sum_{i<M} (N - i)
    p->prof_flops->push_back(2.0 * (double)k * ((double)M * (double)N - 0.5 * (double)M * (double)(M - 1)));
  }
  return CAP_OK;
}

// Factor the strip R[J0 : J0+rows, J0 : n] (its leading rows x rows block is diagonal), assuming every
// update from earlier strips has been applied: inner right-looking sweep with nb-wide panels whose
// trailing update is confined to the strip's own rows.  One stream.
int factor_strip(cap_cholinv_plan* p, double* R, int64_t ldr, int64_t n, int64_t J0, int64_t rows, hipStream_t s, const StripCtx* sc = nullptr) {
  const int64_t nb = p->nb, Jend = J0 + rows;
  for (int64_t j0 = J0; j0 < Jend; j0 += nb) {
    const int64_t jb = std::min(nb, Jend - j0);
    CAP_TRY(panel_factor(p, R, ldr, n, j0, jb, s, sc));
    const int64_t j1 = j0 + jb, rows_left = Jend - j1, cols = n - j1;
    if (rows_left > 0 && cols > 0) {
      // jb x cols block row just solved: inside R, or in the strip buffer
      const double* Rpan = sc ? sc->SB + (j0 - sc->Js) + j1 * sc->ld : R + j0 + j1 * ldr;
      const int64_t ldp = sc ? sc->ld : ldr;
      CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, rows_left, cols, jb, -1.0, Rpan, ldp, Rpan, ldp, 1.0,
                              R + j1 + j1 * ldr, ldr, 1, s, p->ctag));
    }
  }
  return CAP_OK;
}

// Strip buffers: FOUR strips in two pair buffers of 2 NB x n (ld = 2 NB): strip t lives in pair (t / 2) % 2 at row offset (t % 2) NB, so the
// solved rows of strips 2 j and 2 j + 1 are K-contiguous - what the paired far update (K = 2 NB) reads as ONE operand.
int ensure_sb(cap_cholinv_plan* p, int64_t NB, int64_t n) {
  if (p->sb_ready && p->sb_ld == 2 * NB && p->sb_cols == n) return CAP_OK;
  if (p->sb_ready) { CAP_HIP(hipDeviceSynchronize()); (void)hipFree(p->SB); p->SB = nullptr; }
  else {
    CAP_HIP(hipStreamCreateWithFlags(&p->s_copy, hipStreamNonBlocking));
    CAP_HIP(hipEventCreateWithFlags(&p->ev_sbg, hipEventDisableTiming));
    CAP_HIP(hipEventCreateWithFlags(&p->ev_join_cp, hipEventDisableTiming));
    for (int i = 0; i < 4; i++) CAP_HIP(hipEventCreateWithFlags(&p->ev_copy[i], hipEventDisableTiming));
  }
  p->sb_ready = true; p->sb_ld = 0;
  CAP_HIP(hipMalloc((void**)&p->SB, sizeof(double) * 4 * NB * n));
  p->sb_ld = 2 * NB; p->sb_cols = n;
  return CAP_OK;
}

void release_sb(cap_cholinv_plan* p) {
  if (!p->sb_ready) return;
  (void)hipStreamSynchronize(p->s_copy); cap_stream_destroy(p->s_copy);
  (void)hipEventDestroy(p->ev_sbg); (void)hipEventDestroy(p->ev_join_cp);
  for (int i = 0; i < 4; i++) (void)hipEventDestroy(p->ev_copy[i]);
  if (p->SB) (void)hipFree(p->SB);
  p->SB = nullptr; p->sb_ready = false;
}

// ---- inverse tree on its own stream (see InvTree) ------------------------------------------------------------------------
int inv_node_phase(cap_cholinv_plan* p, const InvNode& nd, int phase, hipStream_t s) {
  CapRange range(phase == 1 ? "CI::inverse_ph1" : "CI::inverse_ph2");   // cholinv.hpp:145-158 (CI::tmu upstream)
  const int64_t h1 = nd.mid - nd.a, h2 = nd.b - nd.mid;
  double* W = p->inv_W + nd.woff;
  double* Ri = p->Rinv;
  if (phase == 1)   // W = -Ri11 R12   (Ri11 upper triangular: K range of a row tile starts at its diagonal)
    return cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, h1, h2, h1, -1.0, Ri + nd.a + nd.a * p->ldi, p->ldi, p->R + nd.a + nd.mid * p->ldr, p->ldr,
                           0.0, W, h1, 0, s, 32);
  // Ri12 = W Ri22   (Ri22 upper triangular: K range of a column tile stops at its diagonal)
  return cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, h1, h2, h2, 1.0, W, h1, Ri + nd.mid + nd.mid * p->ldi, p->ldi, 0.0,
                         Ri + nd.a + nd.mid * p->ldi, p->ldi, 0, s, 8);
}

// everything of the tree that became ready with the panels [inv_pending_from, k]
int inverse_flush(cap_cholinv_plan* p, int64_t k) {
  const InvTree& t = *p->itree;
  for (int64_t kk = p->inv_pending_from; kk <= k; kk++) {
    if (t.ph1[(size_t)kk].empty() && t.ph2[(size_t)kk].empty()) continue;
    CAP_HIP(hipStreamWaitEvent(p->s_inv, (*p->ev_pf)[(size_t)kk], 0));
    for (int i : t.ph2[(size_t)kk]) CAP_TRY(inv_node_phase(p, t.nodes[(size_t)i], 2, p->s_inv));   // inner nodes first (post-order)
    for (int i : t.ph1[(size_t)kk]) CAP_TRY(inv_node_phase(p, t.nodes[(size_t)i], 1, p->s_inv));
  }
  p->inv_pending_from = (int)(k + 1);
  return CAP_OK;
}

int inverse_after_panel(cap_cholinv_plan* p, int64_t k, int64_t cols_left, hipStream_t s) {
  CAP_HIP(hipEventRecord((*p->ev_pf)[(size_t)k], s));
  if (p->inv_overlap && cols_left <= p->inv_start_m) CAP_TRY(inverse_flush(p, k));
  return CAP_OK;
}

void release_inverse(cap_cholinv_plan* p) {
  if (p->inv_ready) {
    (void)hipStreamSynchronize(p->s_inv); cap_stream_destroy(p->s_inv);
    (void)hipEventDestroy(p->ev_inv_done);
    p->inv_ready = false;
  }
  if (p->ev_pf) { for (hipEvent_t e : *p->ev_pf) (void)hipEventDestroy(e); delete p->ev_pf; p->ev_pf = nullptr; }
  delete p->itree; p->itree = nullptr;
}

void release_split(cap_cholinv_plan* p) {
  if (!p->split_ready) return;
  (void)hipStreamSynchronize(p->s_rest); cap_stream_destroy(p->s_rest);
  for (int i = 0; i < 2; i++) {
    (void)hipEventDestroy(p->ev_crit[i]); (void)hipEventDestroy(p->ev_near[i]);
    for (int q = 0; q < 8; q++) { (void)hipEventDestroy(p->ev_pchain[i][q]); (void)hipEventDestroy(p->ev_psolve[i][q]); }
  }
  (void)hipEventDestroy(p->ev_join_r);
  (void)hipFree(p->Dring); (void)hipFree(p->Wpan2);
  p->Dring = nullptr; p->Wpan2 = nullptr; p->split_ready = false;
}

int ensure_split(cap_cholinv_plan* p) {
  if (p->split_ready) return CAP_OK;
  int lo = 0, hi = 0;
  CAP_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CAP_HIP(hipStreamCreateWithPriority(&p->s_rest, hipStreamNonBlocking, hi));
  for (int i = 0; i < 2; i++) {
    CAP_HIP(hipEventCreateWithFlags(&p->ev_crit[i], hipEventDisableTiming));
    CAP_HIP(hipEventCreateWithFlags(&p->ev_near[i], hipEventDisableTiming));
    for (int q = 0; q < 8; q++) {
      CAP_HIP(hipEventCreateWithFlags(&p->ev_pchain[i][q], hipEventDisableTiming));
      CAP_HIP(hipEventCreateWithFlags(&p->ev_psolve[i][q], hipEventDisableTiming));
    }
  }
  CAP_HIP(hipEventCreateWithFlags(&p->ev_join_r, hipEventDisableTiming));
  CAP_HIP(hipMalloc((void**)&p->Dring, sizeof(double) * 16 * p->nb * p->nb));
  CAP_HIP(hipMemset(p->Dring, 0, sizeof(double) * 16 * p->nb * p->nb));
  CAP_HIP(hipMalloc((void**)&p->Wpan2, sizeof(double) * cap_round_up(p->nb * p->n, 2)));
  p->split_ready = true;
  return CAP_OK;
}

// R[r0:r1, c0:c1] -= R[k0:k0+kb, r0:r1]^T R[k0:k0+kb, c0:c1]; upper mask when the block starts on the diagonal (c0 == r0),
// blocks right of it (c0 >= r1) are full
int strip_update(double* R, int64_t ldr, int64_t r0, int64_t r1, int64_t c0, int64_t c1, int64_t k0, int64_t kb, hipStream_t s) {
  if (r1 <= r0 || c1 <= c0) return CAP_OK;
  return cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, r1 - r0, c1 - c0, kb, -1.0, R + k0 + r0 * ldr, ldr, R + k0 + c0 * ldr, ldr, 1.0,
                         R + r0 + c0 * ldr, ldr, c0 == r0 ? 1 : 0, s);
}

// Column-split look-ahead.  Measured at N = 32768 (profiles/r02_bench_n32768_kernel_stats.csv): the panel stream's serial
// time (diagonal-block chains + block-row solves + in-strip updates, all slowed by the co-resident bulk update) equals the
// whole factorization - the bulk stream hides beneath it.  But the chain of the next panel only needs the columns of the
// next diagonal blocks.  So while strip k+1 = [J1, J2) is factored (outer step k):
//   panel stream s1 : everything restricted to the columns [J1, Jc), Jc = end of strip k+2 ("crit": what the chains of this
//                     and of the next outer step read)
//   s_rest          : the same solves / updates on [Jc, Jn) ("near" = strip k+3's columns, which s1 reads one step later;
//                     its last piece is recorded as ev_near) and on [Jn, n) ("far", only the bulk update reads it)
//   main stream     : bulk update of step k (unchanged), waits for both halves of strip k.
// Diagonal-block inverses live in a ring (2 parities x 8 panels): s_rest of step k still reads them while s1 runs step k+1.
int right_looking_split(cap_cholinv_plan* p, double* R, int64_t ldr, int64_t n, hipStream_t s0, const std::vector<int64_t>& bnd) {
  const int64_t nb = p->nb;
  const int64_t nstrip = (int64_t)bnd.size() - 1;
  auto B = [&](int64_t i) { return bnd[(size_t)std::min<int64_t>(i, nstrip)]; };
  CAP_TRY(ensure_streams(p));
  CAP_TRY(ensure_split(p));
  hipStream_t s1 = p->s_panel, sr = p->s_rest;
  double* Wpan = p->work + rec_work_size(nb);
  CAP_HIP(hipEventRecord(p->ev_fork, s0));
  CAP_HIP(hipStreamWaitEvent(s1, p->ev_fork, 0));
  CAP_HIP(hipStreamWaitEvent(sr, p->ev_fork, 0));
  // strip 0: nothing to overlap with yet - whole width on the panel stream
  CAP_TRY(factor_strip(p, R, ldr, n, 0, bnd[1], s1));
  CAP_HIP(hipEventRecord(p->ev_crit[0], s1));
  CAP_HIP(hipEventRecord(p->ev_panel[0], s1));
  CAP_HIP(hipEventRecord(p->ev_near[1], s1));            // "step -1": parity (k - 1) & 1 of k = 0
  for (int64_t k = 0; k + 1 < nstrip; k++) {
    const int64_t J0 = bnd[k], J1 = bnd[k + 1], J2 = bnd[k + 2], rows = J1 - J0, m = n - J1;
    const int64_t Jc = B(k + 3), Jn = B(k + 4);
    const int par = (int)(k & 1);
    // ---- (a) panel stream: strip k+1 on the columns [J1, Jc)
    if (k > 0) CAP_HIP(hipStreamWaitEvent(s1, p->ev_update[(k - 1) & 1], 0));     // head of the bulk update of step k-1
    CAP_HIP(hipStreamWaitEvent(s1, p->ev_near[(k - 1) & 1], 0));                  // strip k's rows on [J2, Jc), from s_rest
    CAP_TRY(strip_update(R, ldr, J1, J2, J1, Jc, J0, rows, s1));
    int q = 0;
    for (int64_t j0 = J1; j0 < J2; j0 += nb, q++) {
      const int64_t jb = std::min(nb, J2 - j0), j1 = j0 + jb;
      double* D = p->Dring + (int64_t)(par * 8 + q) * nb * nb;
      CAP_TRY(panel_chain(p, R, ldr, j0, jb, D, s1));
      CAP_HIP(hipEventRecord(p->ev_pchain[par][q], s1));
      CAP_TRY(panel_solve(p, R, ldr, j0, jb, D, j1, Jc, Wpan, s1));
      CAP_HIP(hipEventRecord(p->ev_psolve[par][q], s1));
      CAP_TRY(strip_update(R, ldr, j1, J2, j1, Jc, j0, jb, s1));
    }
    CAP_HIP(hipEventRecord(p->ev_crit[(k + 1) & 1], s1));
    // ---- (b) s_rest: the same on [Jc, Jn) and [Jn, n)
    if (Jc < n) {
      if (k > 0) CAP_HIP(hipStreamWaitEvent(sr, p->ev_update[(k - 1) & 1], 0));
      CAP_HIP(hipStreamWaitEvent(sr, p->ev_crit[par], 0));     // strip k's rows on [J1, J2): the A operand of the strip update
      CAP_TRY(strip_update(R, ldr, J1, J2, Jc, Jn, J0, rows, sr));
      CAP_TRY(strip_update(R, ldr, J1, J2, Jn, n, J0, rows, sr));
      q = 0;
      for (int64_t j0 = J1; j0 < J2; j0 += nb, q++) {
        const int64_t jb = std::min(nb, J2 - j0), j1 = j0 + jb;
        const double* D = p->Dring + (int64_t)(par * 8 + q) * nb * nb;
        const bool last = j1 >= J2;
        CAP_HIP(hipStreamWaitEvent(sr, p->ev_pchain[par][q], 0));
        CAP_TRY(panel_solve(p, R, ldr, j0, jb, D, Jc, Jn, p->Wpan2, sr));
        if (last) CAP_HIP(hipEventRecord(p->ev_near[par], sr));
        CAP_TRY(panel_solve(p, R, ldr, j0, jb, D, Jn, n, p->Wpan2, sr));
        if (!last) {
          CAP_HIP(hipStreamWaitEvent(sr, p->ev_psolve[par][q], 0));    // X[:, j1:J2) (crit columns) is the A operand
          CAP_TRY(strip_update(R, ldr, j1, J2, Jc, Jn, j0, jb, sr));
          CAP_TRY(strip_update(R, ldr, j1, J2, Jn, n, j0, jb, sr));
        }
      }
    } else {
      CAP_HIP(hipStreamWaitEvent(sr, p->ev_crit[(k + 1) & 1], 0));
      CAP_HIP(hipEventRecord(p->ev_near[par], sr));
    }
    CAP_HIP(hipEventRecord(p->ev_panel[(k + 1) & 1], sr));
    // ---- (c) main stream: bulk of the trailing update of step k (rows below strip k+1)
    CAP_HIP(hipStreamWaitEvent(s0, p->ev_crit[par], 0));
    CAP_HIP(hipStreamWaitEvent(s0, p->ev_panel[par], 0));
    const int64_t rows1 = J2 - J1, m2 = m - rows1;
    double* S = R + J0 + J1 * ldr;
    if (m2 > 0) {
      double* S2 = S + rows1 * ldr;
      const int64_t rows2 = (k + 3 <= nstrip) ? bnd[k + 3] - bnd[k + 2] : m2;   // height of strip k+2
      if (p->depth2 && rows2 < m2) {
        CAP_TRY(trailing_update(p, rows2, m2, rows, S2, S2, R + J2 + J2 * ldr, ldr, s0));
        CAP_HIP(hipEventRecord(p->ev_update[par], s0));
        const int64_t m3 = m2 - rows2;
        double* S3 = S2 + rows2 * ldr;
        CAP_TRY(trailing_update(p, m3, m3, rows, S3, S3, R + (J2 + rows2) + (J2 + rows2) * ldr, ldr, s0));
      } else {
        CAP_TRY(trailing_update(p, m2, m2, rows, S2, S2, R + J2 + J2 * ldr, ldr, s0));
        CAP_HIP(hipEventRecord(p->ev_update[par], s0));
      }
    } else {
      CAP_HIP(hipEventRecord(p->ev_update[par], s0));
    }
  }
  CAP_HIP(hipEventRecord(p->ev_join, s1));
  CAP_HIP(hipStreamWaitEvent(s0, p->ev_join, 0));
  CAP_HIP(hipEventRecord(p->ev_join_r, sr));
  CAP_HIP(hipStreamWaitEvent(s0, p->ev_join_r, 0));
  return CAP_OK;
}

// Blocked right-looking Cholesky (upper) in place on R (n x n, ldr), two-level blocking:
// outer strips of NB rows (factored by factor_strip with nb-wide panels), one K = NB trailing
// SYRK per strip, look-ahead of one strip on the high-priority stream.
int right_looking(cap_cholinv_plan* p, double* R, int64_t ldr, int64_t n, hipStream_t s0) {
  const int64_t nb = p->nb;
  int64_t NB = std::max(nb, (p->outer / nb) * nb);
  // strip boundaries: wide strips while the trailing matrix is large, nb-wide ones in the tail where the
  // strip chain could no longer hide behind the trailing update
  std::vector<int64_t> bnd;
  bnd.push_back(0);
  while (bnd.back() < n) {
    const int64_t j = bnd.back(), rest = n - j;
    int64_t w = (rest > p->tail) ? NB : nb;
    bnd.push_back(std::min(n, j + w));
  }
  const int64_t nstrip = (int64_t)bnd.size() - 1;
  const bool la = p->lookahead && nstrip > 2;
  p->prof_used = 0;
  if (p->prof_flops) p->prof_flops->clear();
  // serialize<uppertri,uppertri>(A -> R), cholinv.hpp:13: only A's upper triangle is consumed.  Fused form: see cap_cholinv_plan::srcA
  const double* srcA = p->srcA; const int64_t src_lda = p->src_lda;
  p->srcA = nullptr;
  const bool split_path = p->inner_la && p->serial_m == 0 && p->reserve == 0 && NB / nb <= 8 && !p->inv_active;
  const bool fuse = srcA && la && !split_path && p->fuse_copy && n % 128 == 0 && nb % 128 == 0 && !(src_lda & 1) &&
                    (((uintptr_t)srcA) & 15) == 0 && n - bnd[1] > p->serial_m;
  if (srcA) CAP_TRY(cap_copy_window(srcA, 0, src_lda, 0, 0, R, 0, ldr, 0, 0, fuse ? bnd[1] : n, n, 1, 0, (void*)s0));
  if (!la) {
    for (int64_t k = 0; k < nstrip; k++) {
      const int64_t J0 = bnd[k], rows = bnd[k + 1] - J0, m = n - bnd[k + 1];
      CAP_TRY(factor_strip(p, R, ldr, n, J0, rows, s0));
      if (m > 0) {
        double* S = R + J0 + bnd[k + 1] * ldr;
        CAP_TRY(trailing_update(p, m, m, rows, S, S, R + bnd[k + 1] + bnd[k + 1] * ldr, ldr, s0));
      }
    }
    return CAP_OK;
  }
  if (split_path) return right_looking_split(p, R, ldr, n, s0, bnd);
  CAP_TRY(ensure_streams(p));
  CAP_TRY(ensure_bulk_stream(p));
  hipStream_t s1 = p->s_panel;
  hipStream_t s_user = s0;
  // strip-buffer mode (plan-owned factor, overlapped schedule): see cap_cholinv_plan::use_sb
  const bool sbm = p->use_sb && p->serial_m == 0 && R == p->R;
  if (sbm) CAP_TRY(ensure_sb(p, NB, n));
  auto sctx = [&](int64_t t) { return StripCtx{p->SB + ((t / 2) % 2) * p->sb_ld * n + (t % 2) * NB, p->sb_ld, bnd[(size_t)t]}; };
  // strip t is factored into slot t % 4: its previous tenant (strip t - 4) has been read by the bulk updates of the steps t - 4 and
  // (paired far update) t - 3, both enqueued on the bulk stream before the head of step t - 2 that the panel stream has waited for,
  // and by the copy stream (ev_copy)
  auto strip = [&](int64_t t, hipStream_t s) -> int {
    if (!sbm) return factor_strip(p, R, ldr, n, bnd[(size_t)t], bnd[(size_t)t + 1] - bnd[(size_t)t], s);
    if (t >= 4) CAP_HIP(hipStreamWaitEvent(s, p->ev_copy[t % 4], 0));
    const StripCtx c = sctx(t);
    CAP_TRY(factor_strip(p, R, ldr, n, bnd[(size_t)t], bnd[(size_t)t + 1] - bnd[(size_t)t], s, &c));
    CAP_HIP(hipEventRecord(p->ev_copy[t % 4], p->s_copy));
    return CAP_OK;
  };
  // fork: the panel stream (and the CU-masked bulk stream) join the caller's stream
  CAP_HIP(hipEventRecord(p->ev_fork, s0));
  CAP_HIP(hipStreamWaitEvent(s1, p->ev_fork, 0));
  if (sbm) CAP_HIP(hipStreamWaitEvent(p->s_copy, p->ev_fork, 0));
  // CU masks (option reserve): whole run (reserve_m == 0), or from the step before the first one with <= reserve_m columns left
  const bool tail_res = p->bulk_ready && p->reserve_m > 0;
  int64_t km = nstrip;                       // first step whose trailing matrix is at most reserve_m wide
  if (tail_res) for (int64_t k = 0; k < nstrip; k++) if (n - bnd[k + 1] <= p->reserve_m) { km = k; break; }
  p->chain_masked = p->bulk_ready && !tail_res;
  if (p->bulk_ready && !tail_res) { s0 = p->s_bulk; CAP_HIP(hipStreamWaitEvent(s0, p->ev_fork, 0)); }
  CAP_TRY(strip(0, s1));
  CAP_HIP(hipEventRecord(p->ev_panel[0], s1));
  // Optional (serial_m > 0, default off): from strip ksw on the chain and the bulk update are NOT overlapped any more.
  // Measured (tools/contend.cpp): the chain's kernels run 8-15x slower next to the bulk kernel - a co-resident wave issuing
  // 64-cycle fp64 MFMAs back to back leaves the chain's dependent VALU / MFMA instructions one issue slot per MFMA, and
  // every launch first waits for a slot of a retiring bulk workgroup (stream priorities change nothing).  Letting the
  // chain run alone in the tail was measured too (tools/sweep_serial.sh): N = 32768: 209.0 ms overlapped vs 212.3 ms with
  // serial_m = 16384, N = 16384: 49.2 vs 50.7 - the overlapped schedule already costs about the serial sum, no gain.
  int64_t ksw = nstrip;
  for (int64_t k = 1; k < nstrip; k++) if (n - bnd[k + 1] <= p->serial_m) { ksw = k; break; }
  bool deferred = false, deferred_cin = false;     // paired far update: an even step left the region below strip k+3 to the next step
  p->cnt_paired = 0;
  for (int64_t k = 0; k < nstrip; k++) {
    const int64_t J0 = bnd[k], rows = bnd[k + 1] - J0, m = n - bnd[k + 1];
    if (m <= 0) break;
    if (k >= ksw) {
      // serial phase: strip k is factored (panel stream, ev_panel[k & 1]); everything from here on runs on the bulk stream
      CAP_HIP(hipStreamWaitEvent(s0, p->ev_panel[k & 1], 0));
      for (int64_t q = k; q < nstrip; q++) {
        const int64_t Jq = bnd[q], rq = bnd[q + 1] - Jq, mq = n - bnd[q + 1];
        if (q > k) CAP_TRY(factor_strip(p, R, ldr, n, Jq, rq, s0));
        if (mq > 0) {
          double* Sq = R + Jq + bnd[q + 1] * ldr;
          CAP_TRY(trailing_update(p, mq, mq, rq, Sq, Sq, R + bnd[q + 1] + bnd[q + 1] * ldr, ldr, s0));
        }
      }
      break;
    }
    if (tail_res && k + 1 >= km && s0 != p->s_bulk) {   // the bulk moves to the masked stream one step before the chain does
      CAP_HIP(hipEventRecord(p->ev_bulk_sw, s0));
      s0 = p->s_bulk;
      CAP_HIP(hipStreamWaitEvent(s0, p->ev_bulk_sw, 0));
    }
    if (tail_res) p->chain_masked = k >= km;
    const int64_t J1 = bnd[k + 1], rows1 = bnd[k + 2] - J1, m2 = m - rows1;
    // strip k right of its diagonal block (rows x m): inside R, or in its strip buffer (K-contiguous, ld = NB)
    const StripCtx ck = sbm ? sctx(k) : StripCtx{nullptr, 0, 0};
    const double* S = sbm ? ck.SB + J1 * ck.ld : R + J0 + J1 * ldr;
    const int64_t lds_ = sbm ? ck.ld : ldr;
    // (a) panel stream: bring strip k+1 up to date (K = rows, upper part), then factor it.
    //     needs strip k (same stream) and the HEAD of the bulk update of step k-1 (main stream): the head is the
    //     part of that update that touches strip k+1's rows, so the chain runs one more step ahead of the bulk
    if (k > 0) CAP_HIP(hipStreamWaitEvent(s1, p->ev_update[(k - 1) & 1], 0));
    // step 0 of the fused form: C input from A (the three updates of this step cover the trailing triangle exactly once)
    const bool fz = fuse && k == 0;
    auto cin = [&](int64_t j) -> const double* { return fz ? srcA + j + j * src_lda : nullptr; };
    CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, rows1, m, rows, -1.0, S, lds_, S, lds_, 1.0, R + J1 + J1 * ldr, ldr, 1, s1, p->ctag, 0,
                            cin(J1), src_lda));
    CAP_TRY(strip(k + 1, s1));
    CAP_HIP(hipEventRecord(p->ev_panel[(k + 1) & 1], s1));
    // (b) main stream: bulk of the trailing update (rows below strip k+1)
    CAP_HIP(hipStreamWaitEvent(s0, p->ev_panel[k & 1], 0));
    if (m2 > 0) {
      const double* S2 = S + rows1 * lds_;
      const int64_t J2 = J1 + rows1;
      const int64_t rows2 = (k + 3 <= nstrip) ? bnd[k + 3] - bnd[k + 2] : m2;   // height of strip k+2
      if (p->depth2 && rows2 < m2) {
        // head: strip k+2's rows first, then signal the panel stream; rest: everything below
        CAP_TRY(trailing_update(p, rows2, m2, rows, S2, S2, R + J2 + J2 * ldr, ldr, s0, lds_, cin(J2), src_lda));
        CAP_HIP(hipEventRecord(p->ev_update[k & 1], s0));
        const int64_t m3 = m2 - rows2, J3 = J2 + rows2;
        const double* S3 = S2 + rows2 * lds_;
        // Paired far update (round 5, option "pair_rest").  The rows below strip k+3 are not read before step k+2, so an EVEN step k
        // only brings strip k+3's rows up to date (one more head) and leaves the region below to the ODD step k+1, whose own rest is
        // exactly that region: it then takes the rows of BOTH strips in ONE product with K = 2 NB (the two strips are K-contiguous:
        // adjacent rows of R, or the two halves of a pair buffer).  Half the launches, half the C traffic and half the per-tile
        // prologue / epilogue for the bulk of the flops: the update kernel runs at a + b K per tile with b at 99 % of peak.
        if (deferred) {
          const int64_t Jp = bnd[k - 1];                              // first row of the pair (strip k-1)
          const double* P = sbm ? sctx(k - 1).SB + J3 * p->sb_ld : R + Jp + J3 * ldr;
          CAP_TRY(trailing_update(p, m3, m3, (J0 - Jp) + rows, P, P, R + J3 + J3 * ldr, ldr, s0, sbm ? p->sb_ld : ldr,
                                  deferred_cin ? srcA + J3 + J3 * src_lda : nullptr, src_lda));
          deferred = false; p->cnt_paired++;
        } else if (p->pair_rest && (k & 1) == 0 && k + 4 < nstrip && k + 1 < ksw && rows == NB && rows1 == NB && !tail_res) {
          const int64_t rows3 = bnd[k + 4] - bnd[k + 3];
          CAP_TRY(trailing_update(p, rows3, m3, rows, S3, S3, R + J3 + J3 * ldr, ldr, s0, lds_, cin(J3), src_lda));
          deferred = true; deferred_cin = fz;
        } else {
          CAP_TRY(trailing_update(p, m3, m3, rows, S3, S3, R + J3 + J3 * ldr, ldr, s0, lds_, cin(J3), src_lda));
        }
      } else {
        CAP_TRY(trailing_update(p, m2, m2, rows, S2, S2, R + J2 + J2 * ldr, ldr, s0, lds_, cin(J2), src_lda));
        CAP_HIP(hipEventRecord(p->ev_update[k & 1], s0));
      }
    } else {
      CAP_HIP(hipEventRecord(p->ev_update[k & 1], s0));
    }
  }
  // join
  p->chain_masked = false;
  if (sbm) { CAP_HIP(hipEventRecord(p->ev_join_cp, p->s_copy)); CAP_HIP(hipStreamWaitEvent(s_user, p->ev_join_cp, 0)); }
  CAP_HIP(hipEventRecord(p->ev_join, s1));
  CAP_HIP(hipStreamWaitEvent(s_user, p->ev_join, 0));
  if (s0 != s_user) { CAP_HIP(hipEventRecord(p->ev_join_b, s0)); CAP_HIP(hipStreamWaitEvent(s_user, p->ev_join_b, 0)); }
  return CAP_OK;
}

// complete_inv = 0 / 1 on a matrix of several panels: R by the blocked right-looking sweep (look-ahead, fused diagonal-block
// chain - the same schedule as complete_inv = -1), R^-1 by the inverse tree, overlapped with the sweep's chain-bound tail.
// Same results as the recursion of cholinv.hpp:85-165 up to the association order: R^-1's diagonal blocks [0, n1) and
// [n1, n) (n1 = n >> split) are complete, the root block Ri[0:n1, n1:n] stays empty when complete_inv == 0 (cholinv.hpp:147).
int factor_with_inverse(cap_cholinv_plan* p, bool root_is_base, hipStream_t s) {
  const int64_t n = p->n, nb = p->nb;
  const int eff_ci = root_is_base ? 1 : p->complete_inv;
  const int64_t n1 = n >> p->split;
  int root_mid = -1; bool skip_root = false; int64_t zero_n1 = 0;
  if (!root_is_base && n1 > 0 && n1 < n) {
    if (n1 % nb == 0) { root_mid = (int)(n1 / nb); skip_root = eff_ci == 0; }
    else if (eff_ci == 0) zero_n1 = n1;      // root partition inside a panel: build the full inverse, then empty the root block
  }
  if (!p->itree || p->inv_tree_skip != (skip_root ? 0 : 1) || p->inv_tree_mid != root_mid) {
    delete p->itree;
    p->itree = new (std::nothrow) InvTree(inv_tree_build(panel_bounds(n, nb), root_mid, skip_root));
    if (!p->itree) return CAP_ERR_ALLOC;
    p->inv_tree_skip = skip_root ? 0 : 1; p->inv_tree_mid = root_mid;
  }
  const int64_t np = cap_ceil_div(n, nb);
  // workspace: [chain scratch | panel scratch nb x n | tree scratch]
  const int64_t base = rec_work_size(nb) + cap_round_up(nb * n, 2);
  if (base + p->itree->scratch > p->work_elems) {
    CAP_HIP(hipDeviceSynchronize());
    (void)hipFree(p->work); p->work = nullptr;
    p->work_elems = base + p->itree->scratch;
    CAP_HIP(hipMalloc((void**)&p->work, sizeof(double) * p->work_elems));
  }
  p->inv_W = p->work + base;
  if (!p->inv_ready) {
    CAP_HIP(hipStreamCreateWithFlags(&p->s_inv, hipStreamNonBlocking));
    CAP_HIP(hipEventCreateWithFlags(&p->ev_inv_done, hipEventDisableTiming));
    p->inv_ready = true;
  }
  if (!p->ev_pf) p->ev_pf = new (std::nothrow) std::vector<hipEvent_t>();
  if (!p->ev_pf) return CAP_ERR_ALLOC;
  while ((int64_t)p->ev_pf->size() < np) { hipEvent_t e; CAP_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming)); p->ev_pf->push_back(e); }
  p->inv_pending_from = 0;
  p->inv_active = true;
  int st = right_looking(p, p->R, p->ldr, n, s);
  p->inv_active = false;
  CAP_TRY(st);
  // whatever of the tree has not been enqueued yet (everything when inv_overlap == 0: then it also waits for the sweep's join)
  if (!p->inv_overlap) { CAP_HIP(hipEventRecord(p->ev_inv_done, s)); CAP_HIP(hipStreamWaitEvent(p->s_inv, p->ev_inv_done, 0)); }
  CAP_TRY(inverse_flush(p, np - 1));
  CAP_HIP(hipEventRecord(p->ev_inv_done, p->s_inv));
  CAP_HIP(hipStreamWaitEvent(s, p->ev_inv_done, 0));
  if (zero_n1) CAP_TRY(cap_zero_rect(p->Rinv + zero_n1 * p->ldi, p->ldi, zero_n1, n - zero_n1, s));
  return CAP_OK;
}

}  // namespace

extern "C" {

int cap_cholinv_plan_create(cap_cholinv_plan** plan, int64_t n, int complete_inv, int64_t split, int64_t bc_mult_dim,
                            char dir, cap_comm* comm) {
  if (!plan || n <= 0 || split <= 0) return CAP_ERR_ARG;          // assert(args.split>0), cholinv.hpp:9
  if (dir != 'U') return CAP_ERR_UNSUPPORTED;                      // assert(args.dir == 'U'), cholinv.hpp:9
  if (complete_inv < -1 || complete_inv > 1) return CAP_ERR_ARG;
  const bool multi = comm && cap_comm_size(comm) > 1;
  // multi-GPU: the blocked Cholesky of dist.hip; complete_inv = 0 / 1 add R^-1 by the streamed column-broadcast product of
  // dist.hip (inverse_step) - upstream's R + R^-1 semantics on P > 1
  cap_cholinv_plan* p = new (std::nothrow) cap_cholinv_plan();
  if (!p) return CAP_ERR_ALLOC;
  memset(p, 0, sizeof(*p));
  p->n = n; p->complete_inv = complete_inv; p->split = split; p->bc_mult_dim = bc_mult_dim; p->dir = dir; p->comm = comm;
  p->nb = default_nb(n, bc_mult_dim); p->leaf = CAP_LEAF_MAX; p->lookahead = 1;
  if (multi) {
    // A is this rank's block-cyclic column set (cap_bc_num_local_cols columns, all n rows); see dist.hip
    int st = cap_dist_plan_create(&p->dist, n, std::max<int64_t>(128, (p->nb / 128) * 128), comm);
    if (st != CAP_OK) { delete p; return st; }
    // the single-GPU rule applies on P > 1 too: a root that is itself a base case gets the full inverse (cholinv.hpp:93)
    st = cap_dist_set_option(p->dist, "complete_inv", (complete_inv == 0 && root_is_base_case(n, split, bc_mult_dim)) ? 1 : complete_inv);
    if (st == CAP_OK) st = cap_dist_set_option(p->dist, "split", split);
    if (st != CAP_OK) { (void)cap_dist_plan_destroy(p->dist); delete p; return st; }
    *plan = p;
    return CAP_OK;
  }
  // two-level blocking defaults (tools/sweep.sh on MI355X): K = 2 nb bulk updates while the trailing matrix is
  // large, nb-wide strips for the last n/8 columns where the strip chain could no longer hide
  p->outer = n >= 8192 ? 2 * p->nb : p->nb; p->tail = n >= 8192 ? n / 8 : 0;
  p->reserve = CAP_ENV("CAP_RESERVE") ? atoll(CAP_ENV("CAP_RESERVE")) : 0;
  p->reserve_m = CAP_ENV("CAP_RESERVE_M") ? atoll(CAP_ENV("CAP_RESERVE_M")) : 0; p->chain_masked = false;
  p->srcA = nullptr; p->src_lda = 0; p->fuse_copy = CAP_ENV("CAP_FUSE_COPY") ? atoi(CAP_ENV("CAP_FUSE_COPY")) : 1;
  // fused 64-blocked diagonal-block path (11 dependent launches per 512 panel instead of 43; N = 8192 alone, first version:
  // 20.8 -> 15.5 ms).  Its workgroups use 84 KiB of LDS so that they fit into ONE slot vacated by a bulk
  // workgroup - a first 135 KiB version needed a fully idle CU and lost 4 % under a concurrent bulk update.
  p->fastdiag = CAP_ENV("CAP_FASTDIAG") ? atoi(CAP_ENV("CAP_FASTDIAG")) : 1;
  p->serial_m = 0;
  p->inner_la = CAP_ENV("CAP_INNER_LA") ? atoi(CAP_ENV("CAP_INNER_LA")) : 0;
  // below ~16K remaining columns a step's chain (2 x 512 diagonal blocks, ~4 ms next to the bulk update) outlasts its bulk
  // update (m^2 x 1024 flops): from there on the bulk runs one workgroup per CU (N = 32768: 59.3 -> 61.2 TF, 16384: 34.4 -> 37.1)
  p->occ1_m = CAP_ENV("CAP_OCC1_M") ? atoll(CAP_ENV("CAP_OCC1_M")) : 16384;
  p->chain_coop = -1;
  p->pair_rest = CAP_ENV("CAP_PAIR_REST") ? atoi(CAP_ENV("CAP_PAIR_REST")) : 1;
  p->cnt_paired = 0;
  p->depth2 = n >= 24576;     // look-ahead depth 2 pays once a bulk update is long enough to split (+2 % at N = 32768)
  // reference semantics (R and R^-1): blocked factorization + inverse tree; the tree starts once the sweep is chain-bound
  p->use_sb = CAP_ENV("CAP_USE_SB") ? atoi(CAP_ENV("CAP_USE_SB")) : 1;
  p->inv_fast = CAP_ENV("CAP_INV_FAST") ? atoi(CAP_ENV("CAP_INV_FAST")) : 1;
  p->inv_overlap = CAP_ENV("CAP_INV_OVERLAP") ? atoi(CAP_ENV("CAP_INV_OVERLAP")) : 1;
  p->inv_start_m = CAP_ENV("CAP_INV_START_M") ? atoll(CAP_ENV("CAP_INV_START_M")) : std::max<int64_t>(16384, n / 2);
  int st = plan_alloc(p);
  if (st != CAP_OK) { cap_cholinv_plan_destroy(p); return st; }
  *plan = p;
  return CAP_OK;
}

int cap_cholinv_plan_destroy(cap_cholinv_plan* p) {
  if (!p) return CAP_OK;
  if (p->redist) (void)cap_redist_plan_destroy(p->redist);
  if (p->bcA) (void)hipFree(p->bcA);
  if (p->bcOut) (void)hipFree(p->bcOut);
  if (p->dist) (void)cap_dist_plan_destroy(p->dist);
  if (p->R) (void)hipFree(p->R);
  if (p->Rinv) (void)hipFree(p->Rinv);
  if (p->work) (void)hipFree(p->work);
  if (p->info_dev) (void)hipFree(p->info_dev);
  if (p->streams_ready) {
    cap_stream_destroy(p->s_panel);
    for (int i = 0; i < 2; i++) { (void)hipEventDestroy(p->ev_panel[i]); (void)hipEventDestroy(p->ev_update[i]); }
    (void)hipEventDestroy(p->ev_fork); (void)hipEventDestroy(p->ev_join); (void)hipEventDestroy(p->ev_join_b);
  }
  if (p->bulk_ready) {
    cap_stream_destroy(p->s_bulk); cap_stream_destroy(p->s_chain);
    for (int i = 0; i < 2; i++) (void)hipEventDestroy(p->ev_chain[i]);
    (void)hipEventDestroy(p->ev_bulk_sw);
  }
  release_split(p);
  release_inverse(p);
  release_sb(p);
  if (p->prof_ev) { for (hipEvent_t e : *p->prof_ev) (void)hipEventDestroy(e); delete p->prof_ev; delete p->prof_flops; }
  delete p;
  return CAP_OK;
}

int cap_cholinv_set_option(cap_cholinv_plan* p, const char* key, int64_t value) {
  if (!p || !key) return CAP_ERR_ARG;
  std::string k(key);
  // resident workgroups of the one-launch diagonal-block chain (leaf.hip chain64_coop_kernel) for THIS plan's factor calls (also a
  // multi-rank plan's: the schedule behind it runs on the calling thread), 0 = one launch per step; -1 = back to the process default
  if (k == "chain_coop") { if (value < -1 || value > 256) return CAP_ERR_ARG; p->chain_coop = (int)value; return CAP_OK; }
  if (p->dist) {
    if (k == "nb") {        // block width of the distribution: rebuild the inner plan
      if (value < 128 || value % 128) return CAP_ERR_ARG;
      if (value == cap_dist_get_option(p->dist, "nb")) return CAP_OK;
      cap_dist_plan* nd = nullptr;
      CAP_TRY(cap_dist_plan_create(&nd, p->n, value, p->comm));
      (void)cap_dist_plan_destroy(p->dist);
      p->dist = nd; p->nb = value;
      CAP_TRY(apply_root_semantics(p, 0));
      if (p->cyc_c) CAP_TRY(setup_cyclic(p, p->cyc_c));       // the redistribution follows the block width (and re-applies the grid's root rule)
      return CAP_OK;
    }
    if (k == "cyclic_c") {     // 0: block-column pieces (default); c >= 1: element-cyclic pieces on the d x d x c grid
      if (value < 0) return CAP_ERR_ARG;
      if (value == p->cyc_c) return CAP_OK;
      return setup_cyclic(p, (int)value);
    }
    return cap_dist_set_option(p->dist, key, value);
  }
  if (k == "nb") {
    if (value < 64 || value % 64) return CAP_ERR_ARG;
    if (p->complete_inv >= 0) {          // workspace is re-checked at factor time (factor_with_inverse); the tree follows the panels
      if (value > 1024) return CAP_ERR_ARG;
      if (value != p->nb) { delete p->itree; p->itree = nullptr; }
      p->nb = value; return CAP_OK;
    }
    if (value == p->nb) return CAP_OK;
    // workspace depends on nb: reallocate
    release_split(p);
    (void)hipFree(p->Rinv); (void)hipFree(p->work);
    p->Rinv = nullptr; p->work = nullptr;
    p->nb = value; p->ldi = value;
    CAP_HIP(hipMalloc((void**)&p->Rinv, sizeof(double) * p->nb * p->nb));
    CAP_HIP(hipMemset(p->Rinv, 0, sizeof(double) * p->nb * p->nb));
    p->work_elems = rec_work_size(p->nb) + cap_round_up(p->nb * p->n, 2);
    CAP_HIP(hipMalloc((void**)&p->work, sizeof(double) * p->work_elems));
    return CAP_OK;
  }
  if (k == "leaf") { if (value < 1 || value > CAP_LEAF_MAX) return CAP_ERR_ARG; p->leaf = value; return CAP_OK; }
  if (k == "lookahead") { p->lookahead = value != 0; return CAP_OK; }
  if (k == "outer") { if (value < 64) return CAP_ERR_ARG; p->outer = value; return CAP_OK; }
  if (k == "tail") { if (value < 0) return CAP_ERR_ARG; p->tail = value; return CAP_OK; }
  if (k == "serial_m") { if (value < 0) return CAP_ERR_ARG; p->serial_m = value; return CAP_OK; }
  if (k == "depth2") { p->depth2 = value != 0; return CAP_OK; }
  if (k == "pair_rest") { p->pair_rest = value != 0; return CAP_OK; }
  if (k == "inner_la") { p->inner_la = value != 0; return CAP_OK; }
  if (k == "occ1_m") { if (value < 0) return CAP_ERR_ARG; p->occ1_m = value; return CAP_OK; }
  if (k == "fastdiag") { p->fastdiag = value != 0; return CAP_OK; }
  if (k == "use_sb") { p->use_sb = value != 0; return CAP_OK; }
  if (k == "inv_fast") { p->inv_fast = value != 0; return CAP_OK; }
  if (k == "inv_overlap") { p->inv_overlap = value != 0; return CAP_OK; }
  if (k == "inv_start_m") { if (value < 0) return CAP_ERR_ARG; p->inv_start_m = value; return CAP_OK; }
  if (k == "reserve") {   // CUs kept free for the panel chain (multiple of 8 keeps the XCDs balanced); 0 = off
    if (value < 0 || value > 64) return CAP_ERR_ARG;
    if (p->bulk_ready) {
      (void)hipStreamSynchronize(p->s_bulk); cap_stream_destroy(p->s_bulk);
      (void)hipStreamSynchronize(p->s_chain); cap_stream_destroy(p->s_chain);
      for (int i = 0; i < 2; i++) (void)hipEventDestroy(p->ev_chain[i]);
      (void)hipEventDestroy(p->ev_bulk_sw);
      p->bulk_ready = false;
    }
    p->reserve = value; return CAP_OK;
  }
  if (k == "reserve_m") { if (value < 0) return CAP_ERR_ARG; p->reserve_m = value; return CAP_OK; }
  if (k == "fuse_copy") { p->fuse_copy = value != 0; return CAP_OK; }
  if (k == "profile") {
    p->profile = value != 0;
    if (p->profile && !p->prof_ev) { p->prof_ev = new std::vector<hipEvent_t>(); p->prof_flops = new std::vector<double>(); }
    return CAP_OK;
  }
  return CAP_ERR_ARG;
}

int64_t cap_cholinv_get_option(cap_cholinv_plan* p, const char* key) {
  if (!p || !key) return -1;
  std::string k(key);
  if (k == "chain_coop") { if (p->chain_coop >= 0) return p->chain_coop; return cap_chain_coop_get(); }
  if (k == "chain_fallbacks") return cap_chain_fallbacks();
  if (p->dist) {
    if (k == "complete_inv") return p->complete_inv;
    if (k == "split") return p->split;
    if (k == "bc_mult_dim") return p->bc_mult_dim;
    if (k == "local_cols") return cap_dist_local_cols(p->dist);
    if (k == "cyclic_c") return p->cyc_c;
    if (k == "piece") return p->redist ? cap_redist_get(p->redist, 0) : 0;        // edge of the element-cyclic piece
    return cap_dist_get_option(p->dist, key);
  }
  if (k == "local_cols") return p->n;
  if (k == "nb") return p->nb;
  if (k == "leaf") return p->leaf;
  if (k == "lookahead") return p->lookahead;
  if (k == "outer") return p->outer;
  if (k == "tail") return p->tail;
  if (k == "serial_m") return p->serial_m;
  if (k == "use_sb") return p->use_sb;
  if (k == "inv_fast") return p->inv_fast;
  if (k == "inv_overlap") return p->inv_overlap;
  if (k == "inv_start_m") return p->inv_start_m;
  if (k == "depth2") return p->depth2;
  if (k == "pair_rest") return p->pair_rest;
  if (k == "count_paired") return p->cnt_paired;
  if (k == "fastdiag") return p->fastdiag;
  if (k == "reserve") return p->reserve;
  if (k == "reserve_m") return p->reserve_m;
  if (k == "fuse_copy") return p->fuse_copy;
  if (k == "inner_la") return p->inner_la;
  if (k == "occ1_m") return p->occ1_m;
  if (k == "n") return p->n;
  if (k == "complete_inv") return p->complete_inv;
  if (k == "split") return p->split;
  if (k == "bc_mult_dim") return p->bc_mult_dim;
  return -1;
}

int cap_cholinv_factor(cap_cholinv_plan* p, const double* A, int64_t lda, void* stream) {
  if (!p) return CAP_ERR_ARG;
  CapChainScope chain_scope(p->chain_coop);     // this plan's own "chain_coop", visible to every launch made from this call
  if (p->dist && p->redist) {       // A = my element-cyclic piece (ceil(n/d) x ceil(n/d), lda)
    CAP_TRY(cap_redistribute_cyclic_to_bc(p->redist, A, lda, p->bcA, p->n, stream));
    return cap_dist_factor(p->dist, p->bcA, p->n, stream);
  }
  if (p->dist) return cap_dist_factor(p->dist, A, lda, stream);
  if (!A || lda < p->n) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  const int64_t n = p->n;
  CAP_HIP(hipMemsetAsync(p->info_dev, 0, sizeof(int), s));
  // serialize<uppertri,uppertri>(A -> R), cholinv.hpp:13: only A's upper triangle is consumed
  p->srcA = A; p->src_lda = lda;             // right_looking does the A -> R copy (all of it, or the first strip's rows)
  if (p->complete_inv < 0) return right_looking(p, p->R, p->ldr, n, s);
  const bool root_is_base = root_is_base_case(n, p->split, p->bc_mult_dim);
  if (p->inv_fast && n >= 2 * p->nb && p->leaf == CAP_LEAF_MAX) return factor_with_inverse(p, root_is_base, s);
  p->srcA = nullptr;
  if (p->work_elems < rec_work_size(n)) {       // the plan was sized for the blocked path (plan_alloc) and an option left it
    CAP_HIP(hipDeviceSynchronize());
    (void)hipFree(p->work); p->work = nullptr;
    p->work_elems = rec_work_size(n);
    CAP_HIP(hipMalloc((void**)&p->work, sizeof(double) * p->work_elems));
  }
  CAP_TRY(cap_copy_window(A, 0, lda, 0, 0, p->R, 0, p->ldr, 0, 0, n, n, 1, 0, stream));
  RecCtx c{p->R, p->ldr, p->Rinv, p->ldi, p->work, p->work_elems, p->info_dev, p->leaf, p->split, p->complete_inv, s};
  return rec_cholinv(c, 0, n, !root_is_base, 0);
}

int cap_cholinv_get_R(cap_cholinv_plan* p, double* out, int64_t ld, void* stream) {
  if (p && p->dist && p->redist) {                                     // my element-cyclic piece (construct_R, cholinv.hpp:30-37)
    CAP_TRY(cap_dist_get_R(p->dist, p->bcOut, p->n, stream));
    return cap_redistribute_bc_to_cyclic(p->redist, p->bcOut, p->n, out, ld, stream);
  }
  if (p && p->dist) return cap_dist_get_R(p->dist, out, ld, stream);   // my block-cyclic columns
  if (!p || !out || ld < p->n) return CAP_ERR_ARG;
  return cap_copy_window(p->R, 0, p->ldr, 0, 0, out, 0, ld, 0, 0, p->n, p->n, 1, 1, stream);
}

int cap_cholinv_get_Rinv(cap_cholinv_plan* p, double* out, int64_t ld, void* stream) {
  if (p && p->dist && p->redist) {
    if (p->complete_inv < 0) return CAP_ERR_UNSUPPORTED;
    CAP_TRY(cap_dist_get_Rinv(p->dist, p->bcOut, p->n, stream));
    return cap_redistribute_bc_to_cyclic(p->redist, p->bcOut, p->n, out, ld, stream);
  }
  if (p && p->dist) return p->complete_inv < 0 ? CAP_ERR_UNSUPPORTED : cap_dist_get_Rinv(p->dist, out, ld, stream);   // my block-cyclic columns
  if (!p || !out || ld < p->n) return CAP_ERR_ARG;
  if (p->complete_inv < 0) return CAP_ERR_UNSUPPORTED;   // this mode never builds R^-1
  return cap_copy_window(p->Rinv, 0, p->ldi, 0, 0, out, 0, ld, 0, 0, p->n, p->n, 1, 1, stream);
}

double* cap_cholinv_R_ptr(cap_cholinv_plan* p, int64_t* ld) {
  if (!p) return nullptr;
  if (p->dist) return cap_dist_R_ptr(p->dist, ld);
  if (ld) *ld = p->ldr;
  return p->R;
}
double* cap_cholinv_Rinv_ptr(cap_cholinv_plan* p, int64_t* ld) {
  if (!p || p->complete_inv < 0) return nullptr;
  if (p->dist) return cap_dist_Rinv_ptr(p->dist, ld);
  if (ld) *ld = p->ldi;
  return p->Rinv;
}

// ---- the same three calls on descriptors: cholinv::factor(const Matrix& A, ...), construct_R / construct_Rinv -> matrix (cholinv.h:46-53)
// The descriptor must describe exactly the piece the plan works on; anything else is CAP_ERR_ARG (never a silent reinterpretation):
//   single-GPU plan              any descriptor whose local piece is the whole n x n matrix (1 x 1 grid of either kind)
//   multi-rank plan (1 x P)      block-cyclic kind, Pr = 1, Pc = P, pc = my rank, nb = the plan's block width
//   multi-rank plan, "cyclic_c"  element-cyclic kind on the d x d grid (upstream's own layout)
static int desc_matches_plan(cap_cholinv_plan* p, const cap_desc* d) {
  if (!p || !d) return CAP_ERR_ARG;
  if (cap_desc_get(d, 0) != p->n || cap_desc_get(d, 1) != p->n) return CAP_ERR_ARG;
  const int64_t kind = cap_desc_get(d, 9), px = cap_desc_get(d, 6), py = cap_desc_get(d, 7);
  if (!p->dist) return (px == 1 && py == 1) ? CAP_OK : CAP_ERR_ARG;
  if (p->redist) return (kind == 0 && px == py && cap_desc_get(d, 3) == cap_redist_get(p->redist, 0)) ? CAP_OK : CAP_ERR_ARG;
  if (kind != 1 || py != 1 || px != cap_comm_size(p->comm) || cap_desc_get(d, 11) != cap_comm_rank(p->comm)) return CAP_ERR_ARG;
  if (cap_desc_get(d, 10) != cap_dist_get_option(p->dist, "nb") || cap_desc_get(d, 2) != cap_dist_local_cols(p->dist)) return CAP_ERR_ARG;
  return CAP_OK;
}
int cap_cholinv_factor_desc(cap_cholinv_plan* p, const cap_desc* A, void* stream) {
  CAP_TRY(desc_matches_plan(p, A));
  return cap_cholinv_factor(p, cap_desc_data(const_cast<cap_desc*>(A)), cap_desc_get(A, 4), stream);
}
int cap_cholinv_get_R_desc(cap_cholinv_plan* p, cap_desc* R, void* stream) {
  CAP_TRY(desc_matches_plan(p, R));
  return cap_cholinv_get_R(p, cap_desc_data(R), cap_desc_get(R, 4), stream);
}
int cap_cholinv_get_Rinv_desc(cap_cholinv_plan* p, cap_desc* Rinv, void* stream) {
  CAP_TRY(desc_matches_plan(p, Rinv));
  return cap_cholinv_get_Rinv(p, cap_desc_data(Rinv), cap_desc_get(Rinv, 4), stream);
}

int cap_cholinv_info(cap_cholinv_plan* p, void* stream, int64_t* info) {
  if (!p || !info) return CAP_ERR_ARG;
  if (p->dist) return cap_dist_info(p->dist, stream, info);
  CAP_TRY(cap_drain_streams({p->s_panel, p->s_bulk, p->s_chain, p->s_rest, p->s_copy, p->s_inv}));
  int h = 0;
  CAP_HIP(hipMemcpyAsync(&h, p->info_dev, sizeof(int), hipMemcpyDeviceToHost, cap_stream(stream)));
  CAP_HIP(hipStreamSynchronize(cap_stream(stream)));
  *info = h;
  return h == 0 ? CAP_OK : CAP_ERR_NOT_SPD;
}

// live profile of the last factor call: launches of the trailing-update kernel, their summed
// duration (ms, HIP events on the launch stream) and summed algorithmic flops.  Synchronises.
int cap_cholinv_profile(cap_cholinv_plan* p, int64_t* launches, double* ms_total, double* flops_total) {
  if (!p || !launches || !ms_total || !flops_total) return CAP_ERR_ARG;
  if (p->dist) return cap_dist_profile(p->dist, launches, ms_total, flops_total);
  *launches = 0; *ms_total = 0; *flops_total = 0;
  if (!p->prof_ev) return CAP_OK;
  for (int i = 0; i + 1 < p->prof_used; i += 2) {
    CAP_HIP(hipEventSynchronize((*p->prof_ev)[i + 1]));
    float ms = 0;
    CAP_HIP(hipEventElapsedTime(&ms, (*p->prof_ev)[i], (*p->prof_ev)[i + 1]));
    *ms_total += ms; *flops_total += (*p->prof_flops)[i / 2]; (*launches)++;
  }
  return CAP_OK;
}

// the trailing-update launches of the LAST factor call in profile mode, one by one: up to cap entries of (ms, algorithmic flops); *count = launches
int cap_cholinv_profile_launches(cap_cholinv_plan* p, double* ms_out, double* flops_out, int64_t cap, int64_t* count) {
  if (!p || !ms_out || !flops_out || !count || cap < 0) return CAP_ERR_ARG;
  if (p->dist) return cap_dist_profile_launches(p->dist, ms_out, flops_out, cap, count);
  *count = 0;
  if (!p->prof_ev) return CAP_OK;
  for (int i = 0; i + 1 < p->prof_used; i += 2) {
    CAP_HIP(hipEventSynchronize((*p->prof_ev)[i + 1]));
    float ms = 0;
    CAP_HIP(hipEventElapsedTime(&ms, (*p->prof_ev)[i], (*p->prof_ev)[i + 1]));
    if (*count < cap) { ms_out[*count] = ms; flops_out[*count] = (*p->prof_flops)[i / 2]; }
    (*count)++;
  }
  return CAP_OK;
}

// -------------------------------------------------------------------------------------------------
// operator seam built from the same blocks (lapack::engine::_potrf/_trtri, blas::engine::_trmm, DTRSM)
// -------------------------------------------------------------------------------------------------
// schedule knobs of the LAPACK-shaped entry point: the same choices cap_cholinv_plan_create makes
static void potrf_knobs(int64_t n, int64_t* nb, int64_t* outer, int64_t* tail, int* depth2) {
  int64_t b = std::min<int64_t>(default_nb(n, 0), cap_round_up(std::max<int64_t>(n, 1), 64));
  *nb = b; *outer = n >= 8192 ? 2 * b : b; *tail = n >= 8192 ? n / 8 : 0; *depth2 = n >= 24576;
}

int64_t cap_dpotrf_work_size(int64_t n) {
  int64_t nb, outer, tail; int d2;
  potrf_knobs(n, &nb, &outer, &tail, &d2);
  return nb * nb + rec_work_size(nb) + cap_round_up(nb * n, 2) + 8;
}

// helper stream + events of cap_dpotrf's look-ahead (process-wide, created once; the mutex serialises the
// ENQUEUE of concurrent calls - the work itself is ordered by events on the callers' streams)
namespace {
struct PotrfCtx { cap_cholinv_plan plan; bool ready; };
std::mutex g_potrf_mu;
PotrfCtx g_potrf_ctx[8];     // one per device
}

int cap_dpotrf(int uplo, int64_t n, double* A, int64_t lda, int* info, double* work, void* stream) {
  if (n < 0 || (n > 0 && (!A || lda < n || !work))) return CAP_ERR_ARG;
  if (uplo != CAP_UPPER) return CAP_ERR_UNSUPPORTED;    // upstream removed 'L' as well (cholinv.hpp:9)
  if (n == 0) return CAP_OK;
  hipStream_t s = cap_stream(stream);
  int dev = 0;
  CAP_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_potrf_mu);
  // a plan view over caller memory; streams/events are the only state that survives the call
  PotrfCtx& ctx = g_potrf_ctx[dev & 7];
  if (!ctx.ready) { memset(&ctx.plan, 0, sizeof(ctx.plan)); ctx.ready = true; }
  cap_cholinv_plan& p = ctx.plan;
  p.n = n; p.complete_inv = -1; p.leaf = CAP_LEAF_MAX; p.fastdiag = 1;
  potrf_knobs(n, &p.nb, &p.outer, &p.tail, &p.depth2);
  p.serial_m = 0; p.inner_la = 0;
  p.ctag = cap_plain_device_ptr(A) ? 0 : CAP_TAG_NO_ATOMIC;
  p.lookahead = n >= 4096;
  p.ldi = p.nb;
  p.Rinv = work;                                // nb x nb
  p.work = work + p.nb * p.nb;                  // rec scratch + panel scratch
  p.info_dev = info;                            // may be NULL: the leaf kernels then drop the pivot report
  CAP_HIP(hipMemsetAsync(p.Rinv, 0, sizeof(double) * p.nb * p.nb, s));
  if (info) CAP_HIP(hipMemsetAsync(info, 0, sizeof(int), s));
  return right_looking(&p, A, lda, n, s);
}

int64_t cap_dtrtri_work_size(int64_t n) { return n * n + rec_work_size(n); }

int cap_dtrtri(int uplo, int64_t n, double* A, int64_t lda, double* work, void* stream) {
  if (n < 0 || (n > 0 && (!A || lda < n || !work))) return CAP_ERR_ARG;
  if (uplo != CAP_UPPER) return CAP_ERR_UNSUPPORTED;
  if (n == 0) return CAP_OK;
  hipStream_t s = cap_stream(stream);
  // work on a zero-lower copy (LAPACK does not reference the other triangle), then copy the upper part back
  double* T = work; double* W = work + n * n;
  CAP_TRY(cap_copy_window(A, 0, lda, 0, 0, T, 0, n, 0, 0, n, n, 1, 1, stream));
  CAP_TRY(rec_trtri(T, n, n, W, rec_work_size(n), CAP_LEAF_MAX, s));
  return cap_copy_window(T, 0, n, 0, 0, A, 0, lda, 0, 0, n, n, 1, 0, stream);
}

// B = alpha op(T) B  or  alpha B op(T)  (blas::engine::_trmm, blas/interface.hpp:61-79): the upper triangle is copied
// into a zero-filled square (BLAS does not reference the other triangle, the MFMA kernel reads full tiles), then ONE
// out-of-place GEMM whose K ranges stop at the triangle (tags 8 / 16 / 32: half the flops of a full product).
// work layout: [td*td triangle copy][m*n result]
static int trmm_impl(int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha, const double* T,
                     int64_t ldt, double* B, int64_t ldb, double* work, hipStream_t s) {
  if (m < 0 || n < 0) return CAP_ERR_ARG;
  if (m == 0 || n == 0) return CAP_OK;
  if (!T || !B || !work || ldb < m) return CAP_ERR_ARG;
  if (uplo != CAP_UPPER || diag != CAP_NONUNIT) return CAP_ERR_UNSUPPORTED;
  const int64_t td = side == CAP_LEFT ? m : n;
  if (ldt < td) return CAP_ERR_ARG;
  double* Tc = work; double* Out = work + cap_round_up(td * td, 2);
  CAP_TRY(cap_copy_window(T, 0, ldt, 0, 0, Tc, 0, td, 0, 0, td, td, 1, 1, (void*)s));
  if (side == CAP_LEFT)
    CAP_TRY(cap_gemm_launch(trans, CAP_NOTRANS, m, n, m, alpha, Tc, td, B, ldb, 0.0, Out, m, 0, s, trans == CAP_TRANS ? 16 : 32));
  else
    CAP_TRY(cap_gemm_launch(CAP_NOTRANS, trans, m, n, n, alpha, B, ldb, Tc, td, 0.0, Out, m, 0, s, trans == CAP_TRANS ? 0 : 8));
  return cap_copy_rect(Out, m, B, ldb, m, n, s);
}

int64_t cap_dtrmm_work_size(int side, int64_t m, int64_t n) {
  const int64_t td = side == CAP_LEFT ? m : n;
  return cap_round_up(td * td, 2) + cap_round_up(m * n, 2);
}

int cap_dtrmm(int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha, const double* T, int64_t ldt,
              double* B, int64_t ldb, double* work, void* stream) {
  return trmm_impl(side, uplo, trans, diag, m, n, alpha, T, ldt, B, ldb, work, cap_stream(stream));
}

// ---- DTRSM: blocked substitution.  Only the diagonal blocks (width tb <= 512) are inverted (recursive TRTRI on zero-lower
// copies, all blocks independent); every block step is one GEMM with the block inverse plus one GEMM update of the
// remaining right-hand sides - td^2 n flops like a substitution, no td^3 inverse of the whole triangle.
//   LEFT  NOTRANS  T X = aB    blocks bottom-up:  X_i = Tii^-1 B_i ;           B_0..i-1   -= T(0..i-1, i) X_i
//   LEFT  TRANS    T^T X = aB  blocks top-down:   X_i = Tii^-T B_i ;           B_i+1..    -= T(i, i+1..)^T X_i
//   RIGHT NOTRANS  X T = aB    block columns left-right: X_j = B_j Tjj^-1 ;    B_j+1..    -= X_j T(j, j+1..)
//   RIGHT TRANS    X T^T = aB  right-left:        X_j = B_j Tjj^-T ;           B_0..j-1   -= X_j T(0..j-1, j)^T
// work layout: [nblk * tb * tb block inverses][tb * max(m, n) block temp][rec_trtri scratch]
static int64_t trsm_block(int64_t td) { return td >= 2048 ? 512 : (td >= 512 ? 256 : 128); }

int64_t cap_dtrsm_work_size(int side, int64_t m, int64_t n) {
  const int64_t td = side == CAP_LEFT ? m : n, tb = std::min(trsm_block(td), cap_round_up(std::max<int64_t>(td, 1), 2));
  const int64_t nblk = cap_ceil_div(td, tb), other = side == CAP_LEFT ? n : m;
  return nblk * tb * tb + cap_round_up(tb * other, 2) + rec_work_size(tb) + 8;
}

int cap_dtrsm(int side, int uplo, int trans, int64_t m, int64_t n, double alpha, const double* T, int64_t ldt, double* B,
              int64_t ldb, double* work, void* stream) {
  if (m < 0 || n < 0) return CAP_ERR_ARG;
  if (m == 0 || n == 0) return CAP_OK;
  if (!T || !B || !work || ldb < m) return CAP_ERR_ARG;
  if (uplo != CAP_UPPER) return CAP_ERR_UNSUPPORTED;
  hipStream_t s = cap_stream(stream);
  const bool left = side == CAP_LEFT;
  const int64_t td = left ? m : n, other = left ? n : m;
  if (ldt < td) return CAP_ERR_ARG;
  const int64_t tb = std::min(trsm_block(td), cap_round_up(td, 2));
  const int64_t nblk = cap_ceil_div(td, tb);
  double* Inv = work; double* X = work + nblk * tb * tb; double* W = X + cap_round_up(tb * other, 2);
  CAP_TRY(cap_trsm_prepare(T, ldt, td, tb, Inv, W, s));
  // alpha once, up front (B <- alpha B through the GEMM launcher's scaling path); the sweep then solves op(T) X = B
  if (alpha != 1.0) CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, m, n, 0, 0.0, B, ldb, B, ldb, alpha, B, ldb, 0, s));
  // B is caller memory: atomic-add epilogue of the block updates only on plain device allocations
  return cap_trsm_apply(side, trans, m, n, T, ldt, Inv, tb, B, ldb, X, s, cap_plain_device_ptr(B) ? 0 : CAP_TAG_NO_ATOMIC);
}

}  // extern "C"

// ---- the two halves of cap_dtrsm, also used by the mixed-precision solver (which keeps the block inverses between
// refinement sweeps).  Inv: nblk * tb * tb doubles, W: rec_work_size(tb) scratch, X: tb * max(m, n) block temp.
int64_t cap_trsm_block(int64_t td) { return std::min(trsm_block(td), cap_round_up(std::max<int64_t>(td, 1), 2)); }
int64_t cap_trsm_prepare_work(int64_t tb) { return rec_work_size(tb); }

int cap_trsm_prepare(const double* T, int64_t ldt, int64_t td, int64_t tb, double* Inv, double* W, hipStream_t s) {
  const int64_t nblk = cap_ceil_div(td, tb);
  for (int64_t i = 0; i < nblk; i++) {
    const int64_t o = i * tb, w = std::min(tb, td - o);
    double* Ii = Inv + i * tb * tb;
    // zero-lower copy, so the inverse can be a GEMM operand as a full square
    CAP_TRY(cap_copy_window(T, 0, ldt, o, o, Ii, 0, tb, 0, 0, w, w, 1, 1, (void*)s));
    CAP_TRY(rec_trtri(Ii, tb, w, W, rec_work_size(tb), CAP_LEAF_MAX, s));
  }
  return CAP_OK;
}

int cap_trsm_apply(int side, int trans, int64_t m, int64_t n, const double* T, int64_t ldt, const double* Inv, int64_t tb, double* B,
                   int64_t ldb, double* X, hipStream_t s, int ctag, const float* T32, int64_t ldt32) {
  const bool left = side == CAP_LEFT, tr = trans == CAP_TRANS;
  const int64_t td = left ? m : n;
  const int64_t nblk = cap_ceil_div(td, tb);
  const bool forward = left ? tr : !tr;      // order of the block sweep
  for (int64_t step = 0; step < nblk; step++) {
    const int64_t i = forward ? step : nblk - 1 - step;
    const int64_t o = i * tb, w = std::min(tb, td - o);
    const double* Ii = Inv + i * tb * tb;
    if (left) {
      // X_i = op(Tii^-1) * B_i   (w x n)
      CAP_TRY(cap_gemm_launch(trans, CAP_NOTRANS, w, n, w, 1.0, Ii, tb, B + o, ldb, 0.0, X, w, 0, s, tr ? 16 : 32));
      CAP_TRY(cap_copy_rect(X, w, B + o, ldb, w, n, s));
      // (with an fp32 copy of T and a few right-hand sides the update streams THAT: same sums, half the bytes; shapes it does not take fall through)
      if (tr) {          // rows below: B_r -= T(i, r)^T X_i
        const int64_t r0 = o + w, rows = td - r0;
        if (rows > 0) {
          int st = T32 ? cap_skinny_f32a_launch(CAP_TRANS, rows, n, w, -1.0, T32 + o + r0 * ldt32, ldt32, X, w, 1.0, B + r0, ldb, s) : CAP_ERR_UNSUPPORTED;
          if (st == CAP_ERR_UNSUPPORTED) st = cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, rows, n, w, -1.0, T + o + r0 * ldt, ldt, X, w, 1.0, B + r0, ldb, 0, s, ctag);
          CAP_TRY(st);
        }
      } else {           // rows above: B_r -= T(r, i) X_i
        if (o > 0) {
          int st = T32 ? cap_skinny_f32a_launch(CAP_NOTRANS, o, n, w, -1.0, T32 + o * ldt32, ldt32, X, w, 1.0, B, ldb, s) : CAP_ERR_UNSUPPORTED;
          if (st == CAP_ERR_UNSUPPORTED) st = cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, o, n, w, -1.0, T + o * ldt, ldt, X, w, 1.0, B, ldb, 0, s, ctag);
          CAP_TRY(st);
        }
      }
    } else {
      // X_j = B_j * op(Tjj^-1)   (m x w)
      CAP_TRY(cap_gemm_launch(CAP_NOTRANS, trans, m, w, w, 1.0, B + o * ldb, ldb, Ii, tb, 0.0, X, m, 0, s, tr ? 0 : 8));
      CAP_TRY(cap_copy_rect(X, m, B + o * ldb, ldb, m, w, s));
      if (!tr) {         // columns to the right: B_c -= X_j T(j, c)
        const int64_t c0 = o + w, cols = td - c0;
        if (cols > 0) CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_NOTRANS, m, cols, w, -1.0, X, m, T + o + c0 * ldt, ldt, 1.0, B + c0 * ldb, ldb, 0, s, ctag));
      } else {           // columns to the left: B_c -= X_j T(c, j)^T
        if (o > 0) CAP_TRY(cap_gemm_launch(CAP_NOTRANS, CAP_TRANS, m, o, w, -1.0, X, m, T + o * ldt, ldt, 1.0, B, ldb, 0, s, ctag));
      }
    }
  }
  return CAP_OK;
}

int cap_chain_coop_swap(int wgs) { const int old = tl_coop_wgs; tl_coop_wgs = wgs < 0 ? -1 : wgs; return old; }
int cap_chain_coop_get() { return tl_coop_wgs >= 0 ? tl_coop_wgs : g_coop_default; }
void cap_chain_coop_cap(int cap) { tl_coop_cap = cap < 0 ? 0 : cap; }
// chains that were re-run by the recovery launch on the current device (a workgroup gave up waiting for its peers); synchronises
extern "C" int64_t cap_chain_fallbacks() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return -1;
  int* w = nullptr;
  { std::lock_guard<std::mutex> lk(g_coop_mu); w = g_coop_fallbacks[dev]; }
  if (!w) return 0;
  int h[4] = {0, 0, 0, 0};
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(h, w, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return -1;
  return h[0];
}
// test hook: the next `count` chain launches on this device behave as if a workgroup had given up at their first meeting
extern "C" int cap_chain_inject_timeouts(int count) {
  int dev = 0;
  CAP_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16) return CAP_ERR_UNSUPPORTED;
  CAP_HIP(hipDeviceSynchronize());
  int* w = nullptr;
  {
    std::lock_guard<std::mutex> lk(g_coop_mu);
    if (!g_coop_fallbacks[dev]) {
      CAP_HIP(hipMalloc((void**)&g_coop_fallbacks[dev], 4 * sizeof(int)));
      CAP_HIP(hipMemset(g_coop_fallbacks[dev], 0, 4 * sizeof(int)));
    }
    w = g_coop_fallbacks[dev];
  }
  CAP_HIP(hipMemcpy(w + 1, &count, sizeof(int), hipMemcpyHostToDevice));
  return CAP_OK;
}

// Instrumentation (tools/chain_trace.py): the `which`-th one-launch chain from now on records, per workgroup (first 64) and step,
// four 100 MHz time stamps (step start / own blocks done / arrival at the counter / release); cap_chain_trace_read copies them out.
extern "C" int cap_chain_trace_arm(int64_t which) {
  if (!g_coop_trace) CAP_HIP(hipMalloc((void**)&g_coop_trace, 64 * 32 * 8 * sizeof(long long)));
  CAP_HIP(hipMemset(g_coop_trace, 0, 64 * 32 * 8 * sizeof(long long)));
  CAP_HIP(hipDeviceSynchronize());
  g_coop_trace_at = which;
  return CAP_OK;
}
extern "C" int cap_chain_trace_read(int64_t* out) {
  if (!g_coop_trace || !out) return CAP_ERR_ARG;
  CAP_HIP(hipDeviceSynchronize());
  CAP_HIP(hipMemcpy(out, g_coop_trace, 64 * 32 * 8 * sizeof(long long), hipMemcpyDeviceToHost));
  return CAP_OK;
}

// used by cacqr.hip: full cholinv (R in place, Ri = R^-1) of an n x n block on one stream
int cap_rec_cholinv_full(double* R, int64_t ldr, double* Ri, int64_t ldi, int64_t n, double* W, int64_t wcap, int* info,
                         hipStream_t s, int64_t info_base) {
  static const bool fast = CAP_ENV("CAP_FASTDIAG") ? atoi(CAP_ENV("CAP_FASTDIAG")) != 0 : true;   // see cap_cholinv_plan::fastdiag
  if (fast && n % 64 == 0 && n >= 128 && n <= 1024 && (n & (n - 1)) == 0) return blocked_cholinv(R, ldr, Ri, ldi, n, W, wcap, info, info_base, s);
  RecCtx c{R, ldr, Ri, ldi, W, wcap, info, CAP_LEAF_MAX, 1, 1, s};
  return rec_cholinv(c, 0, n, false, info_base);
}
int64_t cap_rec_work_size(int64_t n) { return rec_work_size(n); }
