// Distributed redistribution between the reference's element-cyclic d x d x c pieces and the block-cyclic layouts of the
// multi-GPU plans - the way a caller holding upstream-style pieces on P ranks gets into (and out of) cap_dist_* / cap_dist2d_*.
//
// Reference side (matrix.hpp:8-11, topology.h:67-143): rank = z + c x + c d y (layout 0) of the d x d x c grid
// holds the piece (x, y): global rows y, y + d, ... and columns x, x + d, ... (ceil(n / d) each, zero padded), the SAME piece on
// every layer z.  construct_R / construct_Rinv (cholinv.hpp:30-46) return pieces of that shape on every rank.
// GPU side: block (I, J) of nb x nb elements on process (I mod Pr, J mod Pc), rank = pr Pc + pc, local block (I div Pr,
// J div Pc); Pr = 1 is the block-column-cyclic layout of dist.hip (all rows local).
//
// The move is ONE all-to-all over the world communicator (grouped ncclSend / ncclRecv - the role util::block_to_cyclic_* /
// cyclic_to_local play after upstream's MPI_Allgather, util.hpp:56-230): the elements rank s = (xs, ys) shares with rank
// t = (pr, pc) form the cross product  Rows(ys, pr) x Cols(xs, pc)  of two sorted index sets
//     Rows(y, pr) = { g < n : g mod d == y, (g div nb) mod Pr == pr },   Cols(x, pc) likewise with Pc,
// so a message is a dense |Rows| x |Cols| column-major matrix: the sender gathers it with one launch per peer, the receiver
// scatters it with one launch per peer; both sides derive the same sets from (n, nb, d, Pr, Pc) - no index traffic.
//   cyclic -> block-cyclic: the replicas share the work - destination t takes its data from layer z = t mod c only;
//   block-cyclic -> cyclic: every rank of every layer receives its piece (what construct_R returns upstream).
// HBM-bound gathers / scatters (8 B read + 8 B written per element, one side contiguous); not on the timed path.
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"

struct cap_redist_plan {
  int64_t n, nb;
  int P, rank, c, d, x, y, z, Pr, Pc, pr, pc;
  cap_comm* world;
  int64_t pl;                                    // cyclic piece edge ceil(n / d)
  int64_t bc_rows, bc_cols;                      // valid local extent of my block-cyclic piece
  // index lists (device, int64): cyc_rows[pr'] / cyc_cols[pc'] hold CYCLIC-local indices of Rows(y, pr') / Cols(x, pc'),
  // bc_rows_l[y'] / bc_cols_l[x'] hold BLOCK-CYCLIC-local indices of Rows(y', pr) / Cols(x', pc)
  int64_t* idx; int64_t idx_elems;
  std::vector<int64_t> cyc_row_off, cyc_row_cnt, cyc_col_off, cyc_col_cnt;     // per pr' / pc'
  std::vector<int64_t> bc_row_off, bc_row_cnt, bc_col_off, bc_col_cnt;         // per y' / x'
  // all-to-all descriptors per direction (0: cyclic -> bc, 1: bc -> cyclic), per peer
  std::vector<int64_t> scnt[2], sdsp[2], rcnt[2], rdsp[2];
  int64_t stot[2], rtot[2];
  double* sendbuf; double* recvbuf; int64_t buf_elems;
};

namespace {

// out[i + j nr] = src[ri[i] + ci[j] ld]
__global__ void redist_gather_kernel(const double* src, int64_t ld, const int64_t* ri, int64_t nr, const int64_t* ci, int64_t nc, double* out) {
  const int64_t j = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (j >= nc) return;
  const int64_t cj = ci[j] * ld;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nr; i += (int64_t)gridDim.x * blockDim.x)
    out[i + j * nr] = src[ri[i] + cj];
}
// dst[ri[i] + ci[j] ld] = in[i + j nr]
__global__ void redist_scatter_kernel(double* dst, int64_t ld, const int64_t* ri, int64_t nr, const int64_t* ci, int64_t nc, const double* in) {
  const int64_t j = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (j >= nc) return;
  const int64_t cj = ci[j] * ld;
  for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nr; i += (int64_t)gridDim.x * blockDim.x)
    dst[ri[i] + cj] = in[i + j * nr];
}
inline dim3 rgrid(int64_t nr, int64_t nc) {
  return dim3((unsigned)std::min<int64_t>(cap_ceil_div(nr, 256), 1024), (unsigned)std::min<int64_t>(nc, 65535), (unsigned)cap_ceil_div(nc, 65535));
}

// the rank -> coordinates maps of the two layouts
inline void cyc_coords(int r, int c, int d, int* x, int* y, int* z) { *z = r % c; *y = r / (d * c); *x = (r % (d * c)) / c; }

int ensure_bufs(cap_redist_plan* r) {
  const int64_t need = std::max<int64_t>(2, std::max(std::max(r->stot[0], r->stot[1]), std::max(r->rtot[0], r->rtot[1])));
  if (r->buf_elems >= need) return CAP_OK;
  if (r->sendbuf) (void)hipFree(r->sendbuf);
  if (r->recvbuf) (void)hipFree(r->recvbuf);
  r->sendbuf = r->recvbuf = nullptr; r->buf_elems = 0;
  CAP_HIP(hipMalloc((void**)&r->sendbuf, sizeof(double) * need));
  CAP_HIP(hipMalloc((void**)&r->recvbuf, sizeof(double) * need));
  r->buf_elems = need;
  return CAP_OK;
}

}  // namespace

extern "C" {

// c: depth of the reference's d x d x c grid (world size = c d d); Pr: process rows of the block-cyclic target (1 = block
// columns, the layout of cap_dist_* / cap_dmp_*; Pr > 1: cap_dist2d_*, Pc = size / Pr)
int cap_redist_plan_create(cap_redist_plan** plan, int64_t n, int64_t nb, cap_comm* world, int c, int Pr) {
  if (!plan || n <= 0 || nb <= 0 || c < 1 || Pr < 1) return CAP_ERR_ARG;
  const int P = cap_comm_size(world), rank = cap_comm_rank(world);
  if (P % Pr) return CAP_ERR_ARG;
  int d = 0, x = 0, y = 0, z = 0;
  CAP_TRY(cap_topo_coords(0, rank, P, c, &d, &x, &y, &z));       // rejects sizes that are no d x d x c grid
  cap_redist_plan* r = new (std::nothrow) cap_redist_plan();
  if (!r) return CAP_ERR_ALLOC;
  r->n = n; r->nb = nb; r->P = P; r->rank = rank; r->c = c; r->d = d; r->x = x; r->y = y; r->z = z;
  r->Pr = Pr; r->Pc = P / Pr; r->pr = rank / r->Pc; r->pc = rank % r->Pc; r->world = world;
  r->pl = cap_ceil_div(n, d);
  r->idx = nullptr; r->idx_elems = 0; r->sendbuf = r->recvbuf = nullptr; r->buf_elems = 0;
  const int Pc = r->Pc;
  // ---- index sets, as local indices of the side that uses them
  std::vector<int64_t> host;
  auto build = [&](int mod_d, int res_d, int Q, int q, bool cyc_local, std::vector<int64_t>& off, std::vector<int64_t>& cnt) {
    off.push_back((int64_t)host.size());
    int64_t k = 0;
    for (int64_t g = res_d; g < n; g += mod_d) {
      const int64_t blk = g / nb;
      if ((int)(blk % Q) != q) continue;
      host.push_back(cyc_local ? g / mod_d : (blk / Q) * nb + g % nb);
      k++;
    }
    cnt.push_back(k);
  };
  for (int q = 0; q < Pr; q++) build(d, y, Pr, q, true, r->cyc_row_off, r->cyc_row_cnt);
  for (int q = 0; q < Pc; q++) build(d, x, Pc, q, true, r->cyc_col_off, r->cyc_col_cnt);
  for (int yy = 0; yy < d; yy++) build(d, yy, Pr, r->pr, false, r->bc_row_off, r->bc_row_cnt);
  for (int xx = 0; xx < d; xx++) build(d, xx, Pc, r->pc, false, r->bc_col_off, r->bc_col_cnt);
  r->bc_rows = 0; r->bc_cols = 0;
  for (int yy = 0; yy < d; yy++) r->bc_rows += r->bc_row_cnt[(size_t)yy];
  for (int xx = 0; xx < d; xx++) r->bc_cols += r->bc_col_cnt[(size_t)xx];
  r->idx_elems = (int64_t)host.size();
  if (hipMalloc((void**)&r->idx, sizeof(int64_t) * std::max<int64_t>(r->idx_elems, 1)) != hipSuccess) { delete r; return CAP_ERR_ALLOC; }
  if (r->idx_elems && hipMemcpy(r->idx, host.data(), sizeof(int64_t) * r->idx_elems, hipMemcpyHostToDevice) != hipSuccess) {
    (void)hipFree(r->idx); delete r; return CAP_ERR_HIP;
  }
  // ---- message sizes.  As a cyclic rank (x, y, z) my partner t = (pr', pc') shares Rows(y, pr') x Cols(x, pc');
  // as a block-cyclic rank (pr, pc) my partner s = (x', y', z') shares Rows(y', pr) x Cols(x', pc).
  for (int dir = 0; dir < 2; dir++) {
    r->scnt[dir].assign((size_t)P, 0); r->sdsp[dir].assign((size_t)P, 0); r->rcnt[dir].assign((size_t)P, 0); r->rdsp[dir].assign((size_t)P, 0);
  }
  for (int t = 0; t < P; t++) {
    const int tpr = t / Pc, tpc = t % Pc;
    const int64_t as_cyc = r->cyc_row_cnt[(size_t)tpr] * r->cyc_col_cnt[(size_t)tpc];        // me cyclic <-> t block-cyclic
    int sx, sy, sz; cyc_coords(t, c, d, &sx, &sy, &sz);
    const int64_t as_bc = r->bc_row_cnt[(size_t)sy] * r->bc_col_cnt[(size_t)sx];             // me block-cyclic <-> t cyclic
    r->scnt[0][(size_t)t] = (z == t % c) ? as_cyc : 0;            // cyclic -> bc: only the layer t mod c supplies t
    r->rcnt[0][(size_t)t] = (sz == rank % c) ? as_bc : 0;
    r->scnt[1][(size_t)t] = as_bc;                                // bc -> cyclic: every layer receives
    r->rcnt[1][(size_t)t] = as_cyc;
  }
  for (int dir = 0; dir < 2; dir++) {
    int64_t so = 0, ro = 0;
    for (int t = 0; t < P; t++) {
      r->sdsp[dir][(size_t)t] = so; so += cap_round_up(r->scnt[dir][(size_t)t], 2);      // 16-byte aligned pieces
      r->rdsp[dir][(size_t)t] = ro; ro += cap_round_up(r->rcnt[dir][(size_t)t], 2);
    }
    r->stot[dir] = so; r->rtot[dir] = ro;
  }
  *plan = r;
  return CAP_OK;
}

// pure index helper (no GPU, no communicator): doubles rank `from` sends to rank `to` in direction dir (0: cyclic -> block-cyclic, the
// sender is the cyclic rank; 1: block-cyclic -> cyclic, the sender is the block-cyclic rank) - the same counts the plan derives; -1 on
// a size that is no d x d x c grid / Pr does not divide it
int64_t cap_redist_message_elems(int64_t n, int64_t nb, int P, int c, int Pr, int from, int to, int dir) {
  if (n <= 0 || nb <= 0 || P < 1 || c < 1 || Pr < 1 || P % Pr || from < 0 || to < 0 || from >= P || to >= P || dir < 0 || dir > 1) return -1;
  int d = 0, x = 0, y = 0, z = 0;
  const int cyc = dir == 0 ? from : to, bc = dir == 0 ? to : from;
  if (cap_topo_coords(0, cyc, P, c, &d, &x, &y, &z) != CAP_OK) return -1;
  const int Pc = P / Pr, bpr = bc / Pc, bpc = bc % Pc;
  if (dir == 0 && z != bc % c) return 0;                      // only the layer bc mod c supplies bc
  int64_t rows = 0, cols = 0;
  for (int64_t g = y; g < n; g += d) rows += (int)((g / nb) % Pr) == bpr;
  for (int64_t g = x; g < n; g += d) cols += (int)((g / nb) % Pc) == bpc;
  return rows * cols;
}

int cap_redist_plan_destroy(cap_redist_plan* r) {
  if (!r) return CAP_OK;
  if (r->idx) (void)hipFree(r->idx);
  if (r->sendbuf) (void)hipFree(r->sendbuf);
  if (r->recvbuf) (void)hipFree(r->recvbuf);
  delete r;
  return CAP_OK;
}

// which: 0 cyclic piece edge ceil(n / d), 1 valid local rows of my block-cyclic piece, 2 its valid local columns, 3 d, 4 c,
// 5 x, 6 y, 7 z, 8 Pr, 9 Pc, 10 pr, 11 pc, 12 / 13 elements I send in direction 0 / 1, 14 / 15 elements I receive
int64_t cap_redist_get(const cap_redist_plan* r, int which) {
  if (!r) return -1;
  switch (which) {
    case 0: return r->pl; case 1: return r->bc_rows; case 2: return r->bc_cols; case 3: return r->d; case 4: return r->c;
    case 5: return r->x; case 6: return r->y; case 7: return r->z; case 8: return r->Pr; case 9: return r->Pc; case 10: return r->pr;
    case 11: return r->pc; case 12: return r->stot[0]; case 13: return r->stot[1]; case 14: return r->rtot[0]; case 15: return r->rtot[1];
  }
  return -1;
}

// piece: my element-cyclic piece (ceil(n/d) x ceil(n/d), column-major, ldp) - read; bc: my block-cyclic piece (valid local
// rows x columns, ldb) - written.  Collective over the world communicator, asynchronous on `stream`.
int cap_redistribute_cyclic_to_bc(cap_redist_plan* r, const double* piece, int64_t ldp, double* bc, int64_t ldb, void* stream) {
  if (!r || !piece || ldp < r->pl || (r->bc_rows > 0 && r->bc_cols > 0 && (!bc || ldb < r->bc_rows))) return CAP_ERR_ARG;
  CapRange range("redistribute::cyclic_to_bc");
  CAP_TRY(ensure_bufs(r));
  hipStream_t s = cap_stream(stream);
  const int P = r->P, Pc = r->Pc;
  for (int t = 0; t < P; t++) {
    if (r->scnt[0][(size_t)t] == 0) continue;
    const int tpr = t / Pc, tpc = t % Pc;
    const int64_t nr = r->cyc_row_cnt[(size_t)tpr], nc = r->cyc_col_cnt[(size_t)tpc];
    cap_acc_r(piece, ldp, r->pl, r->pl); cap_acc_r(r->idx, 0, r->idx_elems, 1); cap_acc_w(r->sendbuf + r->sdsp[0][(size_t)t], 0, nr * nc, 1);
    hipLaunchKernelGGL(redist_gather_kernel, rgrid(nr, nc), dim3(256), 0, s, piece, ldp, r->idx + r->cyc_row_off[(size_t)tpr], nr,
                       r->idx + r->cyc_col_off[(size_t)tpc], nc, r->sendbuf + r->sdsp[0][(size_t)t]);
  }
  CAP_HIP(hipGetLastError());
  CAP_TRY(cap_comm_alltoallv(r->world, r->sendbuf, r->scnt[0].data(), r->sdsp[0].data(), r->recvbuf, r->rcnt[0].data(), r->rdsp[0].data(), stream));
  for (int sr = 0; sr < P; sr++) {
    if (r->rcnt[0][(size_t)sr] == 0) continue;
    int sx, sy, sz; cyc_coords(sr, r->c, r->d, &sx, &sy, &sz);
    const int64_t nr = r->bc_row_cnt[(size_t)sy], nc = r->bc_col_cnt[(size_t)sx];
    cap_acc_w(bc, ldb, r->bc_rows, r->bc_cols); cap_acc_r(r->idx, 0, r->idx_elems, 1); cap_acc_r(r->recvbuf + r->rdsp[0][(size_t)sr], 0, nr * nc, 1);
    hipLaunchKernelGGL(redist_scatter_kernel, rgrid(nr, nc), dim3(256), 0, s, bc, ldb, r->idx + r->bc_row_off[(size_t)sy], nr,
                       r->idx + r->bc_col_off[(size_t)sx], nc, r->recvbuf + r->rdsp[0][(size_t)sr]);
  }
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

// bc: my block-cyclic piece - read; piece: my element-cyclic piece (ceil(n/d) x ceil(n/d), ldp) - written on EVERY rank of every
// layer, padding row / column zero-filled (matrix.hpp:8-11).
int cap_redistribute_bc_to_cyclic(cap_redist_plan* r, const double* bc, int64_t ldb, double* piece, int64_t ldp, void* stream) {
  if (!r || !piece || ldp < r->pl || (r->bc_rows > 0 && r->bc_cols > 0 && (!bc || ldb < r->bc_rows))) return CAP_ERR_ARG;
  CapRange range("redistribute::bc_to_cyclic");
  CAP_TRY(ensure_bufs(r));
  hipStream_t s = cap_stream(stream);
  const int P = r->P, Pc = r->Pc;
  for (int t = 0; t < P; t++) {
    if (r->scnt[1][(size_t)t] == 0) continue;
    int sx, sy, sz; cyc_coords(t, r->c, r->d, &sx, &sy, &sz);
    const int64_t nr = r->bc_row_cnt[(size_t)sy], nc = r->bc_col_cnt[(size_t)sx];
    cap_acc_r(bc, ldb, r->bc_rows, r->bc_cols); cap_acc_r(r->idx, 0, r->idx_elems, 1); cap_acc_w(r->sendbuf + r->sdsp[1][(size_t)t], 0, nr * nc, 1);
    hipLaunchKernelGGL(redist_gather_kernel, rgrid(nr, nc), dim3(256), 0, s, bc, ldb, r->idx + r->bc_row_off[(size_t)sy], nr,
                       r->idx + r->bc_col_off[(size_t)sx], nc, r->sendbuf + r->sdsp[1][(size_t)t]);
  }
  CAP_HIP(hipGetLastError());
  CAP_TRY(cap_comm_alltoallv(r->world, r->sendbuf, r->scnt[1].data(), r->sdsp[1].data(), r->recvbuf, r->rcnt[1].data(), r->rdsp[1].data(), stream));
  if (r->pl * r->d != r->n) CAP_TRY(cap_zero_rect(piece, ldp, r->pl, r->pl, s));      // ragged n: the padding stays zero
  for (int t = 0; t < P; t++) {
    if (r->rcnt[1][(size_t)t] == 0) continue;
    const int tpr = t / Pc, tpc = t % Pc;
    const int64_t nr = r->cyc_row_cnt[(size_t)tpr], nc = r->cyc_col_cnt[(size_t)tpc];
    cap_acc_w(piece, ldp, r->pl, r->pl); cap_acc_r(r->idx, 0, r->idx_elems, 1); cap_acc_r(r->recvbuf + r->rdsp[1][(size_t)t], 0, nr * nc, 1);
    hipLaunchKernelGGL(redist_scatter_kernel, rgrid(nr, nc), dim3(256), 0, s, piece, ldp, r->idx + r->cyc_row_off[(size_t)tpr], nr,
                       r->idx + r->cyc_col_off[(size_t)tpc], nc, r->recvbuf + r->rdsp[1][(size_t)t]);
  }
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

}  // extern "C"
