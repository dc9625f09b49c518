/* capital_amd_cblas.h - libcapital_amd_cblas.so: the reference's operator seam with ZERO source changes.
 *
 * tbennun/capital reaches BLAS / LAPACK through seven C entry points and nothing else (blas::engine -> cblas_dgemm / dtrmm / dsyrk,
 * src/blas/interface.hpp:54,74,92; lapack::engine -> LAPACKE_dpotrf / dtrtri / dgeqrf / dorgqr, src/lapack/interface.hpp:39,54,69,84),
 * all on HOST pointers, column-major, LP64 ints.  This library exports exactly those seven symbols: linked (or LD_PRELOADed) in the
 * place of MKL it turns every call into "stage the operands into HBM, run the MI355X operator of libcapital_amd.so (cap_dgemm, cap_dtrmm,
 * cap_dsyrk, cap_dpotrf, cap_dtrtri - capital_amd.h), copy the result window back" - the "offload each GEMM" policy the reference
 * reserves a tag for and never defines (OffloadEachGemm, src/alg/alg.h:9-11).  The reference's own bench programs then run, unmodified,
 * every flop of cholinv / cacqr / summa on the GPU (tests: the real reference linked this way passes its own validators).
 *
 * It is the zero-change seam, not the fast path: every call pays two PCIe crossings.  The resident-matrix entry points of
 * capital_amd.h (cap_cholinv_*, cap_cacqr_*, cap_summa_*) are the ones the headline numbers are measured on.
 *
 * Semantics kept from BLAS / LAPACK: only the `uplo` triangle of a triangular / symmetric operand is referenced or written (the other
 * triangle of the caller's window comes back untouched); C is not read when beta == 0; LAPACKE_dpotrf returns info > 0 for a
 * non-positive pivot; a negative info for an argument this library does not take (row-major, 'L', unit diagonal - none of which the
 * reference uses); the BLAS calls answer an illegal argument the way a CPU BLAS's xerbla does - a line on stderr, the call ignored (CBLAS
 * has no status to return; upstream does issue such calls on degenerate splits and MKL lets them pass) - and abort only when the device
 * or the library fails underneath them.  Staging buffers are per thread
 * and grow only; everything runs on the NULL stream of the thread's device and has completed on return.  The device: with more than one
 * visible, CAPCB_DEVICE=<index>, else the launcher's local rank (MPI_LOCALRANKID, OMPI_COMM_WORLD_LOCAL_RANK, SLURM_LOCALID) modulo the
 * device count - one process per GPU without a line of code in the MPI program; with one visible device nothing is selected.
 * LAPACKE_dgeqrf / LAPACKE_dorgqr have no call site upstream (ArgPack_geqrf / _orgqr are never instantiated); they are exported so that
 * the reference links, and return -1010 (LAPACK_WORK_MEMORY_ERROR's slot) after a message.                                            */
#ifndef CAPITAL_AMD_CBLAS_H
#define CAPITAL_AMD_CBLAS_H
#ifdef __cplusplus
extern "C" {
#endif

/* CBLAS / LAPACKE constants as MKL's mkl.h has them (values are what travels; the reference passes the enums). */
enum { CAPCB_ROW_MAJOR = 101, CAPCB_COL_MAJOR = 102, CAPCB_NOTRANS = 111, CAPCB_TRANS = 112, CAPCB_CONJTRANS = 113, CAPCB_UPPER = 121,
       CAPCB_LOWER = 122, CAPCB_NONUNIT = 131, CAPCB_UNIT = 132, CAPCB_LEFT = 141, CAPCB_RIGHT = 142 };

/* blas/interface.hpp:54  blas::engine::_gemm */
void cblas_dgemm(int layout, int transa, int transb, int m, int n, int k, double alpha, const double* A, int lda, const double* B, int ldb,
                 double beta, double* C, int ldc);
/* blas/interface.hpp:74  blas::engine::_trmm */
void cblas_dtrmm(int layout, int side, int uplo, int transa, int diag, int m, int n, double alpha, const double* A, int lda, double* B,
                 int ldb);
/* blas/interface.hpp:92  blas::engine::_syrk */
void cblas_dsyrk(int layout, int uplo, int trans, int n, int k, double alpha, const double* A, int lda, double beta, double* C, int ldc);
/* lapack/interface.hpp:39  lapack::engine::_potrf */
int LAPACKE_dpotrf(int layout, char uplo, int n, double* a, int lda);
/* lapack/interface.hpp:54  lapack::engine::_trtri */
int LAPACKE_dtrtri(int layout, char uplo, char diag, int n, double* a, int lda);
/* lapack/interface.hpp:69,84  lapack::engine::_geqrf / _orgqr - no call site upstream */
int LAPACKE_dgeqrf(int layout, int m, int n, double* a, int lda, double* tau);
int LAPACKE_dorgqr(int layout, int m, int n, int k, double* a, int lda, double* tau);

/* calls served and bytes staged (host -> HBM, HBM -> host) by this process so far - for tests and for sizing the PCIe cost.
 * Environment CAPCB_REPORT=1 prints the same three numbers on stderr when the process ends.                                   */
void capcb_counters(long long* calls, long long* bytes_in, long long* bytes_out);
/* the calling thread's staging buffers back to the device allocator (they are otherwise kept, grow-only, until the process ends) */
void capcb_release(void);

#ifdef __cplusplus
}
#endif
#endif
