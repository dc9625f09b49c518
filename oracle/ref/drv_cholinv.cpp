// TEST INFRASTRUCTURE (oracle/): driver that runs the REAL upstream
// cholesky::cholinv (ref/src/alg/cholesky/cholinv/cholinv.hpp:6-28) and its own
// validator (ref/test/cholesky/validate.hpp:7-49) on the upstream generator
// (ref/src/matrix/structure.hpp:68-103).  Protocol follows
// ref/bench/cholesky/cholinv.cpp:44-60 (warm-up call, barrier, timed calls).
//
// argv: N complete_inv split bcMult layout num_chunks policy[0..3] [dumpfile] [num_iter]
//   policy 0 Serialize+NoReplication (bench default; NaN for d>1)
//          1 Serialize+ReplicateCommComp   2 Serialize+ReplicateComp
//          3 NoSerialize+NoReplication
// dumpfile ("-" = none): 1 rank: A, R, Rinv as col-major N*N doubles each.  More ranks: every rank writes
//   <dumpfile>.<rank> = 8 int64 (rank, x, y, z, d, c, local rows, local columns) followed by its local pieces of
//   A, R, Rinv (col-major, local rows x local columns; element-cyclic: piece (x, y) holds global rows y, y+d, ...
//   and columns x, x+d, ..., matrix.hpp:8-11) - tests/golden/make_golden.py reassembles the global matrices.
// stdout (rank 0): one line `ranks=.. c=.. d=.. n=.. ci=.. split=.. bc=.. pol=.. time=<median s> residual=..`
#include "ref/src/alg/cholesky/cholinv/cholinv.h"
#include "ref/test/cholesky/validate.h"
#include <algorithm>
#include <cstring>
#include <vector>

int main(int argc, char** argv) {
  using T = double; using U = int64_t; using MatrixType = matrix<T, U, rect>;
  using namespace cholesky;
  int rank, size, prov;
  MPI_Init_thread(&argc, &argv, MPI_THREAD_SINGLE, &prov);
  MPI_Comm_rank(MPI_COMM_WORLD, &rank); MPI_Comm_size(MPI_COMM_WORLD, &size);
  if (argc < 8) { if (!rank) fprintf(stderr, "usage: N ci split bcMult layout chunks policy [dump] [iters]\n"); MPI_Finalize(); return 2; }
  U n = atol(argv[1]); bool ci = atoi(argv[2]); U split = atoi(argv[3]); U bc = atoi(argv[4]);
  size_t layout = atoi(argv[5]); size_t chunks = atoi(argv[6]); int pol = atoi(argv[7]);
  const char* dump = (argc > 8 && strcmp(argv[8], "-")) ? argv[8] : nullptr;
  int iters = argc > 9 ? atoi(argv[9]) : 1;
  size_t c = std::nearbyint(std::ceil(pow(size, 1. / 3.)));
  {
    auto topo = topo::square(MPI_COMM_WORLD, c, layout, chunks);
    MatrixType A(n, n, topo.d, topo.d);
    A.distribute_symmetric(topo.x, topo.y, topo.d, topo.d, rank / topo.c, true);
    double res = -1, t = 0;
    auto run = [&](auto tag) {
      using CT = decltype(tag);
      typename CT::template info<T, U> pack(ci, split, bc, 'U');
      CT::factor(A, pack, topo);  // warm-up: allocates the plan
      std::vector<double> ts;
      for (int it = 0; it < iters; it++) {
        MPI_Barrier(MPI_COMM_WORLD);
        double t0 = MPI_Wtime();
        CT::factor(A, pack, topo);
        double dt = MPI_Wtime() - t0;
        MPI_Allreduce(MPI_IN_PLACE, &dt, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
        ts.push_back(dt);
      }
      std::sort(ts.begin(), ts.end()); t = ts[ts.size() / 2];
      res = validate<CT>::residual(A, pack, topo);
      if (size == 1 && dump) {
        auto R = CT::construct_R(pack, topo); auto Ri = CT::construct_Rinv(pack, topo);
        FILE* f = fopen(dump, "wb");
        fwrite(A.data(), 8, n * n, f); fwrite(R.data(), 8, n * n, f); fwrite(Ri.data(), 8, n * n, f);
        fclose(f);
      } else if (dump) {
        auto R = CT::construct_R(pack, topo); auto Ri = CT::construct_Rinv(pack, topo);
        char name[4096]; snprintf(name, sizeof(name), "%s.%d", dump, rank);
        FILE* f = fopen(name, "wb");
        int64_t hdr[8] = {rank, (int64_t)topo.x, (int64_t)topo.y, (int64_t)topo.z, (int64_t)topo.d, (int64_t)topo.c,
                          (int64_t)A.num_rows_local(), (int64_t)A.num_columns_local()};
        fwrite(hdr, 8, 8, f);
        const size_t nl = (size_t)A.num_rows_local() * (size_t)A.num_columns_local();
        fwrite(A.data(), 8, nl, f); fwrite(R.data(), 8, nl, f); fwrite(Ri.data(), 8, nl, f);
        fclose(f);
      }
    };
    if (pol == 0) run(cholinv<policy::cholinv::Serialize, policy::cholinv::SaveIntermediates, policy::cholinv::NoReplication>());
    if (pol == 1) run(cholinv<policy::cholinv::Serialize, policy::cholinv::SaveIntermediates, policy::cholinv::ReplicateCommComp>());
    if (pol == 2) run(cholinv<policy::cholinv::Serialize, policy::cholinv::SaveIntermediates, policy::cholinv::ReplicateComp>());
    if (pol == 3) run(cholinv<policy::cholinv::NoSerialize, policy::cholinv::SaveIntermediates, policy::cholinv::NoReplication>());
    double g; MPI_Reduce(&res, &g, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);
    if (rank == 0)
      printf("ranks=%d c=%zu d=%zu n=%ld ci=%d split=%ld bc=%ld pol=%d time=%.6f residual=%.6e\n",
             size, topo.c, topo.d, (long)n, (int)ci, (long)split, (long)bc, pol, t, g);
  }
  MPI_Finalize();
  return 0;
}
