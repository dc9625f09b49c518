/* include/for_upstream/mkl.h - what the reference needs from "mkl.h" when libcapital_amd_cblas.so stands in for MKL.
 *
 * tbennun/capital includes "mkl.h" (src/util/shared.h:24) for exactly seven entry points and the five CBLAS enums they take
 * (blas/interface.hpp:7-41,54,74,92; lapack/interface.hpp:4-27,39,54,69,84).  This declarations-only header provides those and nothing
 * else, with MKL's LP64 types and values, so that the reference compiles on a machine without MKL:
 *     CFLAGS += -I<repo>/include/for_upstream          LIB_PATH = -L<repo>/capital_amd/lib          LIBS = -lcapital_amd_cblas
 * (INTEGRATION.md section 0).  The functions are the ones include/capital_amd_cblas.h documents; there they are declared with plain ints
 * (the enums travel as ints), so include one header or the other in a translation unit, not both.                              */
#ifndef CAPITAL_AMD_FOR_UPSTREAM_MKL_H
#define CAPITAL_AMD_FOR_UPSTREAM_MKL_H
#ifdef __cplusplus
extern "C" {
#endif

typedef enum { CblasRowMajor = 101, CblasColMajor = 102 } CBLAS_LAYOUT;
typedef CBLAS_LAYOUT CBLAS_ORDER;
typedef enum { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 } CBLAS_TRANSPOSE;
typedef enum { CblasUpper = 121, CblasLower = 122 } CBLAS_UPLO;
typedef enum { CblasNonUnit = 131, CblasUnit = 132 } CBLAS_DIAG;
typedef enum { CblasLeft = 141, CblasRight = 142 } CBLAS_SIDE;
#define MKL_INT int
#define LAPACK_ROW_MAJOR 101
#define LAPACK_COL_MAJOR 102

void cblas_dgemm(CBLAS_LAYOUT layout, CBLAS_TRANSPOSE transa, CBLAS_TRANSPOSE transb, MKL_INT m, MKL_INT n, MKL_INT k, double alpha,
                 const double* a, MKL_INT lda, const double* b, MKL_INT ldb, double beta, double* c, MKL_INT ldc);
void cblas_dtrmm(CBLAS_LAYOUT layout, CBLAS_SIDE side, CBLAS_UPLO uplo, CBLAS_TRANSPOSE transa, CBLAS_DIAG diag, MKL_INT m, MKL_INT n,
                 double alpha, const double* a, MKL_INT lda, double* b, MKL_INT ldb);
void cblas_dsyrk(CBLAS_LAYOUT layout, CBLAS_UPLO uplo, CBLAS_TRANSPOSE trans, MKL_INT n, MKL_INT k, double alpha, const double* a,
                 MKL_INT lda, double beta, double* c, MKL_INT ldc);
int LAPACKE_dpotrf(int matrix_layout, char uplo, int n, double* a, int lda);
int LAPACKE_dtrtri(int matrix_layout, char uplo, char diag, int n, double* a, int lda);
int LAPACKE_dgeqrf(int matrix_layout, int m, int n, double* a, int lda, double* tau);
int LAPACKE_dorgqr(int matrix_layout, int m, int n, int k, double* a, int lda, double* tau);

#ifdef __cplusplus
}
#endif
#endif
