// TEST INFRASTRUCTURE (oracle/): driver that runs the REAL upstream qr::cacqr
// (ref/src/alg/qr/cacqr/cacqr.hpp:217-248) with its validators
// (ref/test/qr/validate.hpp:7-52) on the upstream random generator
// (ref/src/matrix/structure.hpp:105-129).  Protocol: ref/bench/qr/cacqr.cpp:34-53.
//
// argv: variant(1|2) M N c complete_inv split bcMult [dumpfile] [num_iter]
//   c = depth of the c x d x c grid (1 => the 1D path, cacqr.hpp:229).
// dumpfile ("-" = none): 1 rank: A (M*N), Q (M*N), R (N*N) col-major doubles.  More ranks: every rank writes <dumpfile>.<rank> =
//   10 int64 (rank, x, y, z, d, c, local rows / columns of A, local rows / columns of R) followed by its local pieces of A, Q
//   (element-cyclic: rows y, y+d, ..., columns x, x+c, ...) and of R (what construct_R returns) - tests/golden/make_golden.py
//   reassembles the global matrices.
#include "ref/src/alg/qr/cacqr/cacqr.h"
#include "ref/test/qr/validate.h"
#include <algorithm>
#include <cstring>
#include <vector>

int main(int argc, char** argv) {
  using T = double; using U = int64_t; using MatrixType = matrix<T, U, rect>;
  int rank, size, prov;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_SINGLE, &prov);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank); MPI_Comm_size(MPI_COMM_WORLD, &size);
  if (argc < 8) { if (!rank) fprintf(stderr, "usage: variant M N c ci split bcMult [dump] [iters]\n"); MPI_Finalize(); return 2; }
  size_t variant = atoi(argv[1]); U m = atol(argv[2]); U n = atol(argv[3]); size_t c = atoi(argv[4]);
  bool ci = atoi(argv[5]); U split = atoi(argv[6]); U bc = atoi(argv[7]);
  const char* dump = (argc > 8 && strcmp(argv[8], "-")) ? argv[8] : nullptr;
  int iters = argc > 9 ? atoi(argv[9]) : 1;
  using CI = cholesky::cholinv<cholesky::policy::cholinv::NoSerialize, cholesky::policy::cholinv::SaveIntermediates,
                               cholesky::policy::cholinv::ReplicateCommComp>;
  using QT = qr::cacqr<qr::policy::cacqr::NoSerialize, qr::policy::cacqr::SaveIntermediates>;
  {
    auto topo = topo::rect(MPI_COMM_WORLD, c, 0, 0);
    MatrixType A(n, m, topo.c, topo.d);
    A.distribute_random(topo.x, topo.y, topo.c, topo.d, rank / topo.c);
    CI::info<T, U> cip(ci, split, bc, 'U');
    QT::info<T, U, CI> pack(variant, cip);
    QT::factor(A, pack, topo);  // warm-up
    std::vector<double> ts;
    for (int it = 0; it < iters; it++) {
      MPI_Barrier(MPI_COMM_WORLD);
      double t0 = MPI_Wtime();
      QT::factor(A, pack, topo);
      double dt = MPI_Wtime() - t0;
      MPI_Allreduce(MPI_IN_PLACE, &dt, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
      ts.push_back(dt);
    }
    std::sort(ts.begin(), ts.end()); double t = ts[ts.size() / 2];
    double res = qr::validate<QT>::residual(A, pack, topo);
    double orth = qr::validate<QT>::orthogonality(A, pack, topo);
    if (size == 1 && dump) {
      auto Q = QT::construct_Q(pack, topo); auto R = QT::construct_R(pack, topo);
      FILE* f = fopen(dump, "wb");
      fwrite(A.data(), 8, m * n, f); fwrite(Q.data(), 8, m * n, f); fwrite(R.data(), 8, n * n, f);
      fclose(f);
    } else if (dump) {
      auto Q = QT::construct_Q(pack, topo); auto R = QT::construct_R(pack, topo);
      char name[4096]; snprintf(name, sizeof(name), "%s.%d", dump, rank);
      FILE* f = fopen(name, "wb");
      int64_t hdr[10] = {rank, (int64_t)topo.x, (int64_t)topo.y, (int64_t)topo.z, (int64_t)topo.d, (int64_t)topo.c,
                         (int64_t)A.num_rows_local(), (int64_t)A.num_columns_local(), (int64_t)R.num_rows_local(),
                         (int64_t)R.num_columns_local()};
      fwrite(hdr, 8, 10, f);
      const size_t nl = (size_t)A.num_rows_local() * (size_t)A.num_columns_local();
      fwrite(A.data(), 8, nl, f); fwrite(Q.data(), 8, nl, f);
      fwrite(R.data(), 8, (size_t)R.num_rows_local() * (size_t)R.num_columns_local(), f);
      fclose(f);
    }
    double g1, g2; MPI_Reduce(&res, &g1, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
    MPI_Reduce(&orth, &g2, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
    if (rank == 0)
      printf("ranks=%d c=%zu d=%zu variant=%zu m=%ld n=%ld time=%.6f residual=%.6e orthogonality=%.6e\n",
             size, topo.c, topo.d, variant, (long)m, (long)n, t, g1, g2);
  }
  MPI_Finalize();
  return 0;
}
