// libcapital_amd_cblas.so - see include/capital_amd_cblas.h.  Host code only: HIP runtime calls for the staging, the operators of
// libcapital_amd.so (capital_amd.h) for the arithmetic.  No CPU arithmetic here: without the library and a device every call fails loudly.
#include "capital_amd_cblas.h"
#include "capital_amd.h"
#include <hip/hip_runtime_api.h>
#include <atomic>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

namespace {

std::atomic<long long> g_calls{0}, g_in{0}, g_out{0}, g_us{0};
// host microseconds spent inside the entry points (staging + operator + copy back): what the seam costs a program, next to the program's own time
struct Tick { std::chrono::steady_clock::time_point t0 = std::chrono::steady_clock::now();
  ~Tick() { g_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count(); } };

// CAPCB_REPORT=1: one line on stderr when the process ends - how many calls were served, how many bytes crossed
void report() {
  int dev = -1;
  (void)hipGetDevice(&dev);
  fprintf(stderr, "capital_amd_cblas: %lld calls served, %lld bytes host -> device, %lld bytes device -> host, device %d, %.1f ms inside the entry points\n",
          g_calls.load(), g_in.load(), g_out.load(), dev, g_us.load() / 1e3);
}
struct Reporter { Reporter() { const char* e = getenv("CAPCB_REPORT"); if (e && *e && *e != '0') atexit(report); } } reporter;

[[noreturn]] void die(const char* fn, const char* what) {
  fprintf(stderr, "capital_amd_cblas: %s: %s\n", fn, what);
  abort();
}
// an argument BLAS itself rejects (or a form this library does not take): like a CPU BLAS's xerbla - say so, do nothing, return.
// (Upstream does issue such calls: a 0-column split piece reaches cblas_dtrmm with ldb = 0, and MKL answers "Parameter 10 was incorrect".)
bool bad_arg(const char* fn, const char* what) {
  fprintf(stderr, "capital_amd_cblas: %s: %s - call ignored\n", fn, what);
  return true;
}
// a LEGAL BLAS form this library does not implement (row-major, ConjTrans, Lower / Unit triangular operands: none has a call site in the
// reference): returning would leave the caller's output buffer as it was and the program computing on with wrong numbers - a library that
// stands in for MKL must not do that silently.  Abort with the reason.
[[noreturn]] bool unsupported(const char* fn, const char* what) {
  fprintf(stderr, "capital_amd_cblas: %s: %s - a legal BLAS form this offload library does not implement; aborting instead of returning wrong numbers\n", fn, what);
  abort();
}
bool valid_trans(int t) { return t == CAPCB_NOTRANS || t == CAPCB_TRANS || t == CAPCB_CONJTRANS; }
void hip_ok(hipError_t e, const char* fn) { if (e != hipSuccess) die(fn, hipGetErrorString(e)); }
void cap_ok(int st, const char* fn) { if (st != CAP_OK) die(fn, cap_status_string(st)); }

// Which GPU: one process per GPU is the layout this library is built for, and an MPI program linked with it has no line of its own
// that could say so.  CAPCB_DEVICE=<index> names the device; otherwise the launcher's local rank (MPICH / Open MPI / Slurm) modulo the
// number of devices.  With one visible device (a launcher that sets ROCR_VISIBLE_DEVICES per rank, or a one-GPU box) nothing is touched.
void choose_device_once() {
  static thread_local bool done = false;
  if (done) return;
  done = true;
  int count = 0;
  if (hipGetDeviceCount(&count) != hipSuccess || count <= 1) return;
  const char* v = getenv("CAPCB_DEVICE");
  for (const char* name : {"MPI_LOCALRANKID", "OMPI_COMM_WORLD_LOCAL_RANK", "SLURM_LOCALID", "PMI_LOCAL_RANK"})
    if (!v || !*v) v = getenv(name);
  if (!v || !*v) return;
  const long idx = strtol(v, nullptr, 10);
  if (idx >= 0 && hipSetDevice((int)(idx % count)) != hipSuccess) die("device selection", "hipSetDevice failed");
}

// grow-only device buffers of this thread: three operand windows, operator scratch, the info word
struct Pool {
  void* p[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  size_t cap[5] = {0, 0, 0, 0, 0};
  double* get(int i, size_t doubles, const char* fn) {
    choose_device_once();
    const size_t need = (doubles ? doubles : 1) * sizeof(double);
    if (need > cap[i]) {
      if (p[i]) hip_ok(hipFree(p[i]), fn);
      p[i] = nullptr; cap[i] = 0;
      const size_t want = need + need / 4;
      hip_ok(hipMalloc(&p[i], want), fn);
      cap[i] = want;
    }
    return (double*)p[i];
  }
  // no destructor on purpose: a thread's buffers would be freed while the process (and possibly the HIP runtime) is being torn down;
  // like a CPU BLAS's per-thread buffers they live until the process ends (capcb_release() returns the calling thread's early)
  void release() { for (int i = 0; i < 5; i++) { if (p[i]) (void)hipFree(p[i]); p[i] = nullptr; cap[i] = 0; } }
};
thread_local Pool pool;

// a column-major window: rows x cols, leading dimension ld on the host; on the device the columns start 16 bytes aligned
// (dev_ld: rows rounded up to even, on hipMalloc's 256-byte aligned base - the geometry the kernels' aligned paths take)
int64_t dev_ld(int64_t rows) { return rows > 0 ? rows + (rows & 1) : 2; }
int64_t ld_of(int64_t rows) { return rows > 0 ? rows : 1; }      // the smallest leading dimension BLAS accepts from the caller
void up(double* dev, const double* host, int64_t ld, int64_t rows, int64_t cols, const char* fn) {
  if (rows <= 0 || cols <= 0) return;
  hip_ok(hipMemcpy2D(dev, (size_t)dev_ld(rows) * 8, host, (size_t)ld * 8, (size_t)rows * 8, (size_t)cols, hipMemcpyHostToDevice), fn);
  g_in += rows * cols * 8;
}
void down(double* host, int64_t ld, const double* dev, int64_t rows, int64_t cols, const char* fn) {
  if (rows <= 0 || cols <= 0) return;
  hip_ok(hipMemcpy2D(host, (size_t)ld * 8, dev, (size_t)dev_ld(rows) * 8, (size_t)rows * 8, (size_t)cols, hipMemcpyDeviceToHost), fn);
  g_out += rows * cols * 8;
}

}  // namespace

extern "C" {

void cblas_dgemm(int layout, int transa, int transb, int m, int n, int k, double alpha, const double* A, int lda, const double* B, int ldb,
                 double beta, double* C, int ldc) {
  const char* fn = "cblas_dgemm";
  Tick whole;
  if (layout != CAPCB_COL_MAJOR && layout != CAPCB_ROW_MAJOR && bad_arg(fn, "layout")) return;
  if (layout != CAPCB_COL_MAJOR) unsupported(fn, "row-major (the reference passes AblasColumnMajor everywhere)");
  if ((!valid_trans(transa) || !valid_trans(transb)) && bad_arg(fn, "transpose flag")) return;
  if (transa == CAPCB_CONJTRANS) transa = CAPCB_TRANS;              // real arithmetic: the conjugate transpose IS the transpose
  if (transb == CAPCB_CONJTRANS) transb = CAPCB_TRANS;
  if ((m < 0 || n < 0 || k < 0) && bad_arg(fn, "negative dimension")) return;
  const int64_t ar = transa == CAPCB_NOTRANS ? m : k, ac = transa == CAPCB_NOTRANS ? k : m;
  const int64_t br = transb == CAPCB_NOTRANS ? k : n, bc = transb == CAPCB_NOTRANS ? n : k;
  if ((lda < ld_of(ar) || ldb < ld_of(br) || ldc < ld_of(m)) && bad_arg(fn, "leading dimension smaller than the window")) return;
  g_calls++;
  if (m == 0 || n == 0) return;
  double* dA = pool.get(0, (size_t)dev_ld(ar) * ac, fn); double* dB = pool.get(1, (size_t)dev_ld(br) * bc, fn); double* dC = pool.get(2, (size_t)dev_ld(m) * n, fn);
  up(dA, A, lda, ar, ac, fn); up(dB, B, ldb, br, bc, fn);
  if (beta != 0.0) up(dC, C, ldc, m, n, fn);                          // BLAS: C is not read when beta == 0
  cap_ok(cap_dgemm(transa == CAPCB_TRANS, transb == CAPCB_TRANS, m, n, k, alpha, dA, dev_ld(ar), dB, dev_ld(br), beta, dC, dev_ld(m), nullptr), fn);
  down(C, ldc, dC, m, n, fn);
}

void cblas_dtrmm(int layout, int side, int uplo, int transa, int diag, int m, int n, double alpha, const double* A, int lda, double* B,
                 int ldb) {
  const char* fn = "cblas_dtrmm";
  Tick whole;
  if (layout != CAPCB_COL_MAJOR && layout != CAPCB_ROW_MAJOR && bad_arg(fn, "layout")) return;
  if (side != CAPCB_LEFT && side != CAPCB_RIGHT && bad_arg(fn, "side")) return;
  if (uplo != CAPCB_UPPER && uplo != CAPCB_LOWER && bad_arg(fn, "uplo")) return;
  if (diag != CAPCB_NONUNIT && diag != CAPCB_UNIT && bad_arg(fn, "diag")) return;
  if (!valid_trans(transa) && bad_arg(fn, "transpose flag")) return;
  if (layout != CAPCB_COL_MAJOR) unsupported(fn, "row-major");
  if (uplo != CAPCB_UPPER) unsupported(fn, "lower-triangular operand (every call site of the reference is Upper)");
  if (diag != CAPCB_NONUNIT) unsupported(fn, "unit-diagonal operand (every call site of the reference is NonUnit)");
  if (transa == CAPCB_CONJTRANS) transa = CAPCB_TRANS;
  if ((m < 0 || n < 0) && bad_arg(fn, "negative dimension")) return;
  const int64_t t = side == CAPCB_LEFT ? m : n;
  if ((lda < ld_of(t) || ldb < ld_of(m)) && bad_arg(fn, "leading dimension smaller than the window")) return;
  g_calls++;
  if (m == 0 || n == 0) return;
  const int cside = side == CAPCB_LEFT ? CAP_LEFT : CAP_RIGHT;
  double* dT = pool.get(0, (size_t)dev_ld(t) * t, fn); double* dB = pool.get(2, (size_t)dev_ld(m) * n, fn);
  double* work = pool.get(3, (size_t)cap_dtrmm_work_size(cside, m, n), fn);
  up(dT, A, lda, t, t, fn); up(dB, B, ldb, m, n, fn);               // (the strictly lower part of T travels and is never referenced)
  cap_ok(cap_dtrmm(cside, CAP_UPPER, transa == CAPCB_TRANS, CAP_NONUNIT, m, n, alpha, dT, dev_ld(t), dB, dev_ld(m), work, nullptr), fn);
  down(B, ldb, dB, m, n, fn);
}

void cblas_dsyrk(int layout, int uplo, int trans, int n, int k, double alpha, const double* A, int lda, double beta, double* C, int ldc) {
  const char* fn = "cblas_dsyrk";
  Tick whole;
  if (layout != CAPCB_COL_MAJOR && layout != CAPCB_ROW_MAJOR && bad_arg(fn, "layout")) return;
  if (uplo != CAPCB_UPPER && uplo != CAPCB_LOWER && bad_arg(fn, "uplo")) return;
  if (!valid_trans(trans) && bad_arg(fn, "transpose flag")) return;
  if (layout != CAPCB_COL_MAJOR) unsupported(fn, "row-major");
  if (trans == CAPCB_CONJTRANS) trans = CAPCB_TRANS;
  if ((n < 0 || k < 0) && bad_arg(fn, "negative dimension")) return;
  const int64_t ar = trans == CAPCB_NOTRANS ? n : k, ac = trans == CAPCB_NOTRANS ? k : n;
  if ((lda < ld_of(ar) || ldc < ld_of(n)) && bad_arg(fn, "leading dimension smaller than the window")) return;
  g_calls++;
  if (n == 0) return;
  double* dA = pool.get(0, (size_t)dev_ld(ar) * ac, fn); double* dC = pool.get(2, (size_t)dev_ld(n) * n, fn);
  up(dA, A, lda, ar, ac, fn);
  up(dC, C, ldc, n, n, fn);                                          // always: the other triangle of the window must come back as it was
  cap_ok(cap_dsyrk(uplo == CAPCB_UPPER ? CAP_UPPER : CAP_LOWER, trans == CAPCB_TRANS, n, k, alpha, dA, dev_ld(ar), beta, dC, dev_ld(n), nullptr), fn);
  down(C, ldc, dC, n, n, fn);
}

int LAPACKE_dpotrf(int layout, char uplo, int n, double* a, int lda) {
  const char* fn = "LAPACKE_dpotrf";
  Tick whole;
  if (layout != CAPCB_COL_MAJOR) { fprintf(stderr, "capital_amd_cblas: %s: row-major is not taken\n", fn); return -1; }
  if (uplo != 'U' && uplo != 'u') { fprintf(stderr, "capital_amd_cblas: %s: uplo '%c' is not taken (the reference removed 'L', cholinv.hpp:9)\n", fn, uplo); return -2; }
  if (n < 0) return -3;
  if (lda < ld_of(n)) return -5;
  g_calls++;
  if (n == 0) return 0;
  double* dA = pool.get(0, (size_t)dev_ld(n) * n, fn); double* work = pool.get(3, (size_t)cap_dpotrf_work_size(n), fn);
  int* dinfo = (int*)pool.get(4, 1, fn);
  up(dA, a, lda, n, n, fn);
  const int st = cap_dpotrf(CAP_UPPER, n, dA, dev_ld(n), dinfo, work, nullptr);
  if (st != CAP_OK && st != CAP_ERR_NOT_SPD) cap_ok(st, fn);
  int info = 0;
  hip_ok(hipMemcpy(&info, dinfo, sizeof(int), hipMemcpyDeviceToHost), fn);
  down(a, lda, dA, n, n, fn);                                        // like LAPACK: on info > 0 the leading part is factored, the rest is not
  return info;
}

int LAPACKE_dtrtri(int layout, char uplo, char diag, int n, double* a, int lda) {
  const char* fn = "LAPACKE_dtrtri";
  Tick whole;
  if (layout != CAPCB_COL_MAJOR) { fprintf(stderr, "capital_amd_cblas: %s: row-major is not taken\n", fn); return -1; }
  if (uplo != 'U' && uplo != 'u') { fprintf(stderr, "capital_amd_cblas: %s: uplo '%c' is not taken\n", fn, uplo); return -2; }
  if (diag != 'N' && diag != 'n') { fprintf(stderr, "capital_amd_cblas: %s: diag '%c' is not taken\n", fn, diag); return -3; }
  if (n < 0) return -4;
  if (lda < ld_of(n)) return -6;
  g_calls++;
  if (n == 0) return 0;
  for (int j = 0; j < n; j++)                                        // LAPACK's singularity check (info = j: the j-th diagonal entry is zero)
    if (a[(size_t)j * lda + j] == 0.0) return j + 1;
  double* dA = pool.get(0, (size_t)dev_ld(n) * n, fn); double* work = pool.get(3, (size_t)cap_dtrtri_work_size(n), fn);
  up(dA, a, lda, n, n, fn);
  cap_ok(cap_dtrtri(CAP_UPPER, n, dA, dev_ld(n), work, nullptr), fn);
  down(a, lda, dA, n, n, fn);
  return 0;
}

int LAPACKE_dgeqrf(int, int, int, double*, int, double*) {
  fprintf(stderr, "capital_amd_cblas: LAPACKE_dgeqrf has no call site in the reference and is not offloaded\n");
  return -1010;
}
int LAPACKE_dorgqr(int, int, int, int, double*, int, double*) {
  fprintf(stderr, "capital_amd_cblas: LAPACKE_dorgqr has no call site in the reference and is not offloaded\n");
  return -1010;
}

void capcb_release(void) { pool.release(); }

void capcb_counters(long long* calls, long long* bytes_in, long long* bytes_out) {
  if (calls) *calls = g_calls.load();
  if (bytes_in) *bytes_in = g_in.load();
  if (bytes_out) *bytes_out = g_out.load();
}

}  // extern "C"
