// TEST INFRASTRUCTURE (oracle/): driver that runs the REAL upstream matmult::summa::invoke - the GEMM, TRMM and SYRK overloads
// (ref/src/alg/matmult/summa/summa.hpp:6-161) - on operands this driver fills from a closed formula of the GLOBAL indices (so every
// layer holds the same pieces and the zero padding of ragged sizes is upstream's, structure.hpp:96-101), prepared the way upstream's own
// call sites prepare them (cholinv.hpp:113-154: the triangular operand a packed `uppertri` piece, util::transpose before a Trans TRMM),
// and dumps every rank's pieces of the inputs and of the result.  Upstream's SUMMA is one step per layer: the grid must be the cube
// (c == d: 1, 8, 27 ranks).
//
// argv: op M N K c layout num_chunks alpha beta dumpfile
//   op 0 GEMM   C (M x N) <- alpha A (M x K) B (K x N) + beta C            (bench/matmult/summa_gemm.cpp:31-47)
//      1 TRMM   B (M x N) <- alpha T B,    T M x M upper (packed)          (cholinv.hpp:149-150)
//      2 TRMM   B <- alpha T^T B, T's pieces swapped by util::transpose    (cholinv.hpp:114-119)
//      3 TRMM   B <- alpha B T,    T N x N upper (packed)                   (cholinv.hpp:151-153)
//      4 TRMM   B <- alpha B T^T (util::transpose first)
//      5 SYRK   C (N x N upper, packed) <- alpha A^T A + beta C, A K x N   (cholinv.hpp:128-131; the one-operand overload, summa.hpp:85-96)
//      6 SYRK   C <- alpha A A^T + beta C, A N x K
//      7 SYRK   like 5 with a rect C
// <dumpfile>.<rank>: 8 int64 (rank, x, y, z, d, c, number of arrays, 0), then per array 3 int64 (local rows, local columns, packed) and
//   the doubles (column-major rows x columns, or the packed upper triangle: column j at j (j + 1) / 2, structure.h:39).
//   Arrays: GEMM A, B, C_in, C_out; TRMM T (before the transpose), B_in, B_out; SYRK A, C_in, C_out.
#include "ref/src/alg/matmult/summa/summa.h"
#include <cmath>
#include <cstring>
#include <vector>

using T = double; using U = int64_t;

static double value(int64_t gi, int64_t gj, int seed) {
  double v = std::sin(12.9898 * (double)(gi + 1) + 78.233 * (double)(gj + 1) + 37.719 * seed) * 43758.5453;
  return v - std::floor(v) - 0.5;
}

template <typename M>
static void fill_rect(M& a, U rows, U cols, int x, int y, int d, int seed) {   // local (r, q) = global (y + d r, x + d q); padding = 0
  const U rl = a.num_rows_local(), cl = a.num_columns_local();
  for (U q = 0; q < cl; q++)
    for (U r = 0; r < rl; r++) {
      const U gi = y + d * r, gj = x + d * q;
      a.data()[q * rl + r] = (gi < rows && gj < cols) ? value(gi, gj, seed) : 0.;
    }
}

template <typename M>
static void fill_upper(M& a, U n, int x, int y, int d, int seed, bool sym) {   // packed local upper triangle of a globally upper (or symmetric) matrix
  const U cl = a.num_columns_local();
  for (U q = 0; q < cl; q++)
    for (U r = 0; r <= q; r++) {
      const U gi = y + d * r, gj = x + d * q;
      double v = 0.;
      if (gi < n && gj < n) {
        if (gi <= gj) v = value(gi, gj, seed) + (gi == gj ? 2. : 0.);
        else if (sym) v = value(gj, gi, seed);
      }
      a.data()[q * (q + 1) / 2 + r] = v;
    }
}

struct Dump {
  FILE* f;
  void header(int rank, int x, int y, int z, int d, int c, int narr) { int64_t h[8] = {rank, x, y, z, d, c, narr, 0}; fwrite(h, 8, 8, f); }
  void array(const double* p, U rows, U cols, bool packed) {
    int64_t h[3] = {rows, cols, packed ? 1 : 0}; fwrite(h, 8, 3, f);
    fwrite(p, 8, packed ? (size_t)(cols * (cols + 1) / 2) : (size_t)(rows * cols), f);
  }
};

int main(int argc, char** argv) {
  using Rect = matrix<T, U, rect>; using Upper = matrix<T, U, uppertri>;
  int rank, size, prov;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_SINGLE, &prov);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank); MPI_Comm_size(MPI_COMM_WORLD, &size);
  if (argc < 11) { if (!rank) fprintf(stderr, "usage: op M N K c layout chunks alpha beta dump\n"); MPI_Finalize(); return 2; }
  const int op = atoi(argv[1]); const U m = atol(argv[2]), n = atol(argv[3]), k = atol(argv[4]);
  const size_t c = atoi(argv[5]), layout = atoi(argv[6]), chunks = atoi(argv[7]);
  const double alpha = atof(argv[8]), beta = atof(argv[9]);
  char name[4096]; snprintf(name, sizeof(name), "%s.%d", argv[10], rank);
  {
    auto topo = topo::square(MPI_COMM_WORLD, c, layout, chunks);
    const int x = topo.x, y = topo.y, z = topo.z, d = topo.d;
    Dump out{fopen(name, "wb")};
    if (op == 0) {
      Rect A(k, m, d, d), B(n, k, d, d), C(n, m, d, d);
      fill_rect(A, m, k, x, y, d, 1); fill_rect(B, k, n, x, y, d, 2); fill_rect(C, m, n, x, y, d, 3);
      out.header(rank, x, y, z, d, topo.c, 4);
      out.array(A.data(), A.num_rows_local(), A.num_columns_local(), false); out.array(B.data(), B.num_rows_local(), B.num_columns_local(), false);
      out.array(C.data(), C.num_rows_local(), C.num_columns_local(), false);
      blas::ArgPack_gemm<T> pack(blas::Order::AblasColumnMajor, blas::Transpose::AblasNoTrans, blas::Transpose::AblasNoTrans, alpha, beta);
      matmult::summa::invoke(A, B, C, topo, pack);
      out.array(C.data(), C.num_rows_local(), C.num_columns_local(), false);
    } else if (op >= 1 && op <= 4) {
      const bool left = op <= 2, trans = (op == 2 || op == 4);
      const U tn = left ? m : n;
      Upper Tm(tn, tn, d, d); Rect B(n, m, d, d);
      fill_upper(Tm, tn, x, y, d, 4, false); fill_rect(B, m, n, x, y, d, 5);
      out.header(rank, x, y, z, d, topo.c, 3);
      out.array(Tm.data(), Tm.num_rows_local(), Tm.num_columns_local(), true); out.array(B.data(), B.num_rows_local(), B.num_columns_local(), false);
      if (trans) util::transpose(Tm, topo);
      blas::ArgPack_trmm<T> pack(blas::Order::AblasColumnMajor, left ? blas::Side::AblasLeft : blas::Side::AblasRight, blas::UpLo::AblasUpper,
                                 trans ? blas::Transpose::AblasTrans : blas::Transpose::AblasNoTrans, blas::Diag::AblasNonUnit, alpha);
      matmult::summa::invoke(Tm, B, topo, pack);
      out.array(B.data(), B.num_rows_local(), B.num_columns_local(), false);
    } else if (op == 5 || op == 6) {
      const bool trans = op == 5;
      Rect A(trans ? n : k, trans ? k : n, d, d); Upper C(n, n, d, d);
      fill_rect(A, trans ? k : n, trans ? n : k, x, y, d, 6); fill_upper(C, n, x, y, d, 7, true);
      out.header(rank, x, y, z, d, topo.c, 3);
      out.array(A.data(), A.num_rows_local(), A.num_columns_local(), false); out.array(C.data(), C.num_rows_local(), C.num_columns_local(), true);
      blas::ArgPack_syrk<T> pack(blas::Order::AblasColumnMajor, blas::UpLo::AblasUpper, trans ? blas::Transpose::AblasTrans : blas::Transpose::AblasNoTrans, alpha, beta);
      matmult::summa::invoke(A, C, topo, pack);
      out.array(C.data(), C.num_rows_local(), C.num_columns_local(), true);
    } else if (op == 7) {
      Rect A(n, k, d, d); Rect C(n, n, d, d);
      fill_rect(A, k, n, x, y, d, 6); fill_rect(C, n, n, x, y, d, 8);
      out.header(rank, x, y, z, d, topo.c, 3);
      out.array(A.data(), A.num_rows_local(), A.num_columns_local(), false); out.array(C.data(), C.num_rows_local(), C.num_columns_local(), false);
      blas::ArgPack_syrk<T> pack(blas::Order::AblasColumnMajor, blas::UpLo::AblasUpper, blas::Transpose::AblasTrans, alpha, beta);
      matmult::summa::invoke(A, C, topo, pack);
      out.array(C.data(), C.num_rows_local(), C.num_columns_local(), false);
    }
    fclose(out.f);
  }
  MPI_Finalize();
  return 0;
}
