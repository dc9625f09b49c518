/* Declarations-only stand-in for Intel MKL's mkl.h (LP64 interface of libmkl_rt).
 * TEST INFRASTRUCTURE (oracle/): only used to compile the upstream reference as a
 * CPU oracle (oracle/ref/build_ref.py).  Prototypes cover exactly the entry points
 * the reference calls: blas/interface.hpp:54,74,92; lapack/interface.hpp:39,54,69,84. */
#pragma once
#ifdef __cplusplus
extern "C" {
#endif
typedef enum { CblasRowMajor = 101, CblasColMajor = 102 } CBLAS_LAYOUT;
typedef CBLAS_LAYOUT CBLAS_ORDER;
typedef enum { CblasNoTrans = 111, CblasTrans = 112, CblasConjTrans = 113 } CBLAS_TRANSPOSE;
typedef enum { CblasUpper = 121, CblasLower = 122 } CBLAS_UPLO;
typedef enum { CblasNonUnit = 131, CblasUnit = 132 } CBLAS_DIAG;
typedef enum { CblasLeft = 141, CblasRight = 142 } CBLAS_SIDE;
void cblas_dgemm(CBLAS_LAYOUT, CBLAS_TRANSPOSE, CBLAS_TRANSPOSE, int, int, int, double,
                 const double*, int, const double*, int, double, double*, int);
void cblas_dtrmm(CBLAS_LAYOUT, CBLAS_SIDE, CBLAS_UPLO, CBLAS_TRANSPOSE, CBLAS_DIAG, int, int,
                 double, const double*, int, double*, int);
void cblas_dsyrk(CBLAS_LAYOUT, CBLAS_UPLO, CBLAS_TRANSPOSE, int, int, double, const double*, int,
                 double, double*, int);
#define LAPACK_ROW_MAJOR 101
#define LAPACK_COL_MAJOR 102
int LAPACKE_dpotrf(int, char, int, double*, int);
int LAPACKE_dtrtri(int, char, char, int, double*, int);
int LAPACKE_dgeqrf(int, int, int, double*, int, double*);
int LAPACKE_dorgqr(int, int, int, int, double*, int, double*);
#ifdef __cplusplus
}
#endif
