// Shared internals of libcapital_amd.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <initializer_list>

#include "../../include/capital_amd.h"

typedef double d4 __attribute__((ext_vector_type(4)));
typedef double d2 __attribute__((ext_vector_type(2)));

#define CAP_HIP(x)                                                                      \
  do {                                                                                  \
    hipError_t e_ = (x);                                                                \
    if (e_ != hipSuccess) {                                                             \
      fprintf(stderr, "capital_amd: HIP error '%s' at %s:%d\n", hipGetErrorString(e_),  \
              __FILE__, __LINE__);                                                      \
      return CAP_ERR_HIP;                                                               \
    }                                                                                   \
  } while (0)

#define CAP_TRY(x)             \
  do {                         \
    int s_ = (x);              \
    if (s_ != CAP_OK) return s_; \
  } while (0)

static inline hipStream_t cap_stream(void* s) { return (hipStream_t)s; }
// Timing-surgery kernel variants (CAP_DIAG, CAP_CQR_DIAG: results are WRONG by construction) are only compiled into
// experiment builds: set this to true, rebuild, measure, set it back.  Release libraries cannot be switched into them.
constexpr bool CAP_EXPERIMENTS = false;
// The A/B switches of the experiment logs (CAP_USE_SB, CAP_CHAIN_COOP, CAP_BF16_V2, ... - profiles/HISTORY.md lists them) are read from the
// environment ONLY in experiment builds.  A release library ignores every CAP_* variable: a stray one in a caller's environment cannot change
// the product's code path (round 5 review).  What a caller may choose is a per-plan option (cap_*_set_option).  The expression folds to a
// null pointer at compile time, so the variable names are not even in the release binary (tests/test_abi.py checks that).
static inline const char* cap_env_(const char* name) {
  if constexpr (CAP_EXPERIMENTS) return getenv(name);
  else return nullptr;
}
#define CAP_ENV(name) cap_env_(name)
// tag bit of cap_gemm_launch: C is caller-supplied memory that has not been verified to be plain device memory -
// the fire-and-forget atomic epilogue (hardware fp64 atomics are dropped on fine-grained / host-mapped memory) is off
constexpr int CAP_TAG_NO_ATOMIC = 64;
// true when `p` is plain hipMalloc'ed device memory (not managed, not host-mapped): the operator seam's check
static inline bool cap_plain_device_ptr(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeDevice && !a.isManaged;
}
// ---- access notes: what the NEXT launch / collective this host thread enqueues reads and writes, declared where the launch is made.
// The hook is null in the product (one load + branch per note).  The recording HIP stand-in of tests/hipshim installs one, attaches
// the notes to the launch that follows them and replays every schedule with vector clocks: two operations that touch the same bytes,
// at least one of them writing, must be ordered by stream order, an event edge or a host synchronisation.
// A note is a column-major window: `cols` columns of `rows` elements, ld elements apart; tri = 1 / 2 restricts it to the elements
// with row <= col / row >= col (window-relative), the element masks of the SYRK-shaped updates and of the triangular copies.
typedef void (*cap_access_hook_fn)(int mode, const void* base, int64_t pitch_bytes, int64_t row_bytes, int64_t cols, int tri, int elem_bytes);
extern "C" cap_access_hook_fn cap_access_hook;
enum { CAP_ACC_R = 1, CAP_ACC_W = 2, CAP_ACC_RW = 3, CAP_ACC_ATOMIC = 4 };      // ATOMIC: device-scope atomic read-modify-write (conflicts with R / W, not with itself)
static inline bool cap_acc_on() { return __builtin_expect(cap_access_hook != nullptr, 0); }
static inline void cap_acc(int mode, const void* p, int64_t ld, int64_t rows, int64_t cols, int tri = 0, int elem = 8) {
  if (cap_acc_on() && p && rows > 0 && cols > 0) cap_access_hook(mode, p, ld * elem, rows * elem, cols, tri, elem);
}
static inline void cap_acc_r(const void* p, int64_t ld, int64_t rows, int64_t cols, int tri = 0, int elem = 8) { cap_acc(CAP_ACC_R, p, ld, rows, cols, tri, elem); }
static inline void cap_acc_w(const void* p, int64_t ld, int64_t rows, int64_t cols, int tri = 0, int elem = 8) { cap_acc(CAP_ACC_W, p, ld, rows, cols, tri, elem); }
static inline void cap_acc_rw(const void* p, int64_t ld, int64_t rows, int64_t cols, int tri = 0, int elem = 8) { cap_acc(CAP_ACC_RW, p, ld, rows, cols, tri, elem); }
static inline void cap_acc_atomic(const void* p, int64_t count, int elem) { cap_acc(CAP_ACC_ATOMIC, p, count, count, 1, 0, elem); }
// the HOST itself reads / writes a window now (pinned staging buffers): checked against the copies that are still in flight
enum { CAP_ACC_HOST = 16 };
static inline void cap_acc_host(int mode, const void* p, int64_t ld, int64_t rows, int64_t cols) { cap_acc(mode | CAP_ACC_HOST, p, ld, rows, cols); }
// a launch whose accesses are all on memory no other stream can name (per-stream scratch) still says so: mode 0 = "nothing shared"
static inline void cap_acc_none() { if (cap_acc_on()) cap_access_hook(0, nullptr, 0, 0, 0, 0, 0); }

// Every status / result query of a plan drains the plan's OWN helper streams before it reads (round 6, after round 5's one late read of a
// factor that cap_dmp_info + a device synchronisation had both declared finished): the join of the helper streams into the caller's stream
// by events is what the schedule relies on, this is the belt to those braces - a query is a host synchronisation point anyway.
static inline int cap_drain_streams(std::initializer_list<hipStream_t> streams) {
  for (hipStream_t s : streams)
    if (s && hipStreamSynchronize(s) != hipSuccess) { (void)hipGetLastError(); return CAP_ERR_HIP; }
  return CAP_OK;
}

static inline int64_t cap_ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t cap_round_up(int64_t a, int64_t b) { return cap_ceil_div(a, b) * b; }

// Named profiling regions (the reference brackets CI::factor_diag / CI::trsm / CI::tmu / CQR::gram with critter's
// CRITTER_START/STOP, cholinv.hpp:94-96,113-137, shared.h:26-35): roctx ranges around the ENQUEUE of each group of launches,
// visible in rocprofv3 --marker-trace.  The roctx library is resolved lazily with dlopen (aux.hip); absent -> no-ops.
void cap_range_push(const char* name);
void cap_range_pop();
struct CapRange {
  explicit CapRange(const char* name) { cap_range_push(name); }
  ~CapRange() { cap_range_pop(); }
  CapRange(const CapRange&) = delete;
  CapRange& operator=(const CapRange&) = delete;
};

// internal launchers shared between translation units ---------------------------------------
// C = alpha*op(A)*op(B) + beta*C with optional "upper tiles only" (SYRK-style) masking:
// tri = 0 full, 1 = only tiles/elements with row <= col (upper), 2 = row >= col (lower).
int cap_gemm_launch(int transa, int transb, int64_t m, int64_t n, int64_t k, double alpha, const double* A,
                    int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int tri,
                    hipStream_t stream, int tag = 0, int persist_wgs = 0, const double* Cin = nullptr, int64_t ldcin = 0);

// bf16_tn3.hip: third-generation bf16 trailing update (256 x 256 C-stationary tile, read - add - store epilogue); nst = 3 / 4: LDS ring of
// 3 / 4 stages of 32 k, 5: two stages of 64 k
bool cap_bf16_tn3_applies(int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int tri, int nst);
int cap_bf16_tn3_launch(int64_t m, int64_t n, int64_t k, float alpha, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C,
                        int64_t ldc, int tri, int nst, int st, hipStream_t s, int dbg = 0);

// leaf.hip: in-LDS cholinv (potrf + trtri) / trtri of one n <= 64 block
constexpr int CAP_LEAF_MAX = 64;
int cap_leaf_cholinv(double* A, int64_t lda, double* Rinv, int64_t ldr, int n, int zero_lower, int* info,
                     int info_base, hipStream_t stream, const double* cjob_src = nullptr, double* cjob_dst = nullptr,
                     int64_t cjob_ld = 0, int cjob_cols = 0);
int cap_leaf_trtri(const double* R, int64_t ldr, double* Rinv, int64_t ldi, int n, hipStream_t stream);

// aux.hip
int cap_copy_rect(const double* src, int64_t lds_, double* dst, int64_t ldd, int64_t rows, int64_t cols, hipStream_t s);
int cap_zero_rect(double* dst, int64_t ldd, int64_t rows, int64_t cols, hipStream_t s);
double* cap_scratch(int64_t elems, hipStream_t stream);  // gemm.hip: per-stream device scratch (split-K partials)

// gemm.hip: distributed (1 x P block-column-cyclic) trailing update with staircase mask + gathered A operand
// Pr, pr, rlb0: rows of C block-cyclic over Pr process rows too (local row block b = global block pr + Pr (rlb0 + b));
// the defaults are the 1 x P layout (rows global, origin at block J0)
int cap_dist_update_launch(int64_t m, int64_t nloc, int64_t k, const double* G, int64_t piece, const int* gstart,
                           const double* B, double* C, int64_t ldc, int P, int p, int nb, int J0, int lb0,
                           hipStream_t stream, int persist_wgs = 0, int Pr = 1, int pr = 0, int rlb0 = 0);

// leaf.hip: fused block-row solve + rank-64 trailing update of the 64-blocked diagonal-block factorization
// (Dnext != nullptr: the workgroup of block (i + 1, i + 1) also runs the leaf of step i + 1 and writes R_{i+1,i+1}, Dinv_{i+1};
//  cj_*: move the previous step's solved block row from scratch into R; direct: single-workgroup launch writes its piece in place)
int cap_panel64_solve_update(double* R, int64_t ldr, const double* Dinv, int64_t ldi, int i, int nblk, double* Xs,
                             hipStream_t stream, double* Dnext = nullptr, int64_t ldn = 0, int* info = nullptr, int info_base = 0,
                             const double* cj_src = nullptr, double* cj_dst = nullptr, int64_t cj_ld = 0, int cj_cols = 0,
                             int direct = 0);
// leaf.hip: one level of the triangular-inverse assembly, Ri12 = -Ri11 (R12 Ri22) for npairs aligned pairs (h = 64 / 128 / 256)
int cap_trinv_merge(const double* R, int64_t ldr, double* Ri, int64_t ldi, int64_t h, int npairs, hipStream_t stream);
// backup != nullptr (room for the upper 64 x 64 blocks of the diagonal block): the launch saves the block before it touches it and a
// second, normally empty launch re-runs the chain on two workgroups if a workgroup of the first gave up waiting (fallbacks[0]++)
int cap_chain64_coop(double* R, int64_t ldr, double* Ri, int64_t ldi, int nblk, int* info, int info_base, int* ctr, int wgs, int fence,
                     int hmax, hipStream_t stream, long long* trace = nullptr, double* backup = nullptr, int* fallbacks = nullptr);
int cap_chain64_coop_max_resident();      // workgroups of that kernel the current device holds at once (occupancy x CUs)
// The one-launch chain keeps four counter words and a 4.25 MiB backup buffer per (device, stream) it has run on (cholinv.hip).  A plan
// that destroys a helper stream hands them back first - they were kept for the life of the process before, one set per stream handle
// the runtime ever returned (found by the recording stand-in of tests/hipshim: 282 live allocations after 286 plans).
void cap_coop_slot_release(hipStream_t s);
void cap_scratch_release(hipStream_t s);       // the split-K scratch buffer gemm.hip keeps per (device, stream): same story
static inline void cap_stream_destroy(hipStream_t s) {
  if (!s) return;
  cap_coop_slot_release(s);
  cap_scratch_release(s);
  (void)hipStreamDestroy(s);
}
// resident workgroups of the one-launch diagonal-block chain (0: one launch per step) as the CALLING THREAD sees it: the process
// default unless a CapChainScope is active; cap > 0 bounds it (CU-masked streams), also per thread
int cap_chain_coop_swap(int wgs);         // installs wgs (-1: process default) for this thread, returns the previous value
int cap_chain_coop_get();
void cap_chain_coop_cap(int cap);
struct CapChainScope {
  int prev; bool on;
  explicit CapChainScope(int wgs) : prev(-1), on(wgs >= 0) { if (on) prev = cap_chain_coop_swap(wgs); }
  ~CapChainScope() { if (on) cap_chain_coop_swap(prev); }
  CapChainScope(const CapChainScope&) = delete; CapChainScope& operator=(const CapChainScope&) = delete;
};
// gemm.hip: batched 64x64-tile products (batch = blockIdx.z, affine strides)
int cap_gemm_small_batched(int transa, int transb, int64_t m, int64_t n, int64_t k, double alpha, const double* A, int64_t lda,
                           int64_t sa, const double* B, int64_t ldb, int64_t sb, double beta, double* C, int64_t ldc, int64_t sc,
                           int nbatch, hipStream_t stream);

// cholinv.hip: the two halves of cap_dtrsm (diagonal-block inverses once, block substitution per right-hand side set)
int64_t cap_trsm_block(int64_t td);
int64_t cap_trsm_prepare_work(int64_t tb);
int cap_trsm_prepare(const double* T, int64_t ldt, int64_t td, int64_t tb, double* Inv, double* W, hipStream_t s);
// T32 (optional, LEFT side, n <= 8): the same triangle in fp32 (ld ldt32) - the block updates then stream it instead of T (identical sums when T is its promotion)
int cap_trsm_apply(int side, int trans, int64_t m, int64_t n, const double* T, int64_t ldt, const double* Inv, int64_t tb, double* B,
                   int64_t ldb, double* X, hipStream_t s, int ctag = 0, const float* T32 = nullptr, int64_t ldt32 = 0);
// gemm.hip: the streaming product with an fp32 operand (n <= 8, m >= 1024)
int cap_skinny_f32a_launch(int transa, int64_t m, int64_t n, int64_t k, double alpha, const float* A32, int64_t lda, const double* B, int64_t ldb,
                           double beta, double* C, int64_t ldc, hipStream_t stream);
int cap_rec_cholinv_full(double* R, int64_t ldr, double* Ri, int64_t ldi, int64_t n, double* W, int64_t wcap, int* info,
                         hipStream_t s, int64_t info_base);
int64_t cap_rec_work_size(int64_t n);
