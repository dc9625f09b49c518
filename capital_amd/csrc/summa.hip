// Distributed DGEMM: SUMMA on the reference's d x d x c process grid, RCCL over xGMI.
//
// Replaces matmult::summa::invoke (GEMM overload) + distribute + collect (reference
// src/alg/matmult/summa/summa.hpp:6-44, 163-253; driver bench/matmult/summa_gemm.cpp:7-55) for GPU-resident
// element-cyclic pieces (matrix.hpp:8-11): process (x, y, z) owns A[rows = y mod d, cols = x mod d] etc., the
// matrices are replicated over the c layers.
//
//   upstream (c == d only):  layer z broadcasts A's piece of process column z along `row` and B's piece of process
//                            row z along `column`, multiplies locally, MPI_Allreduce over `depth`.
//   here (any c dividing d): layer z walks the inner process indices k' = z, z + c, z + 2c, ... < d - one broadcast
//                            pair + one local MFMA GEMM per step (c == d: upstream's single step, c == 1: plain 2D
//                            SUMMA with d steps) - then ncclAllReduce over `depth` when c > 1.
//
// Overlap (what upstream's num_chunks pipelining is after, summa.hpp:195-215): the row and the column broadcast run on
// two HIP streams with their own communicators; B travels in `num_chunks` column chunks and the GEMM of chunk j runs
// on the caller's stream while chunk j + 1 (and, for c < d, the next step's A piece) is still in flight.  Events only,
// no host synchronisation.
#include <algorithm>
#include <cstring>
#include <new>
#include <vector>

#include "common.h"

struct cap_summa_plan {
  cap_topo* topo;
  int c, d, x, y, z;
  cap_comm *row, *column, *depth;
  int64_t m, n, k, ml, nl, kl;          // global / local (ceil) dimensions
  int num_chunks;
  double* bufA[2]; double* bufB[2]; double* acc;
  double* triA[2]; double* triB[2];     // unpack targets of packed-upper triangular slots (TRMM overload), allocated on first use
  double* xch[2];                       // the transpose-partner copy of the SYRK overload + its exchange scratch
  hipStream_t s_row, s_col;
  std::vector<hipEvent_t> ev_a, ev_b, ev_free;   // [buf], [buf * num_chunks + chunk], [buf]
  std::vector<hipEvent_t> ev_d;                  // collect: [chunk] = "column chunk of acc is final", [num_chunks] = "all depth sums done"
  hipEvent_t ev_start, ev_join_r, ev_join_c;
};

namespace {
__global__ void combine_kernel(double* C, int64_t ldc, const double* acc, int64_t lda, int64_t m, int64_t n, double beta) {
  const int64_t col = blockIdx.y;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < m; row += (int64_t)gridDim.x * blockDim.x) {
    double* p = C + row + col * ldc;
    const double a = acc[row + col * lda];
    *p = beta == 0.0 ? a : beta * (*p) + a;      // beta == 0: C's old content (possibly NaN) is not read into the result
  }
}
}  // namespace

extern "C" {

int cap_summa_plan_create(cap_summa_plan** plan, cap_topo* topo, int64_t m, int64_t n, int64_t k, int num_chunks) {
  if (!plan || !topo || m <= 0 || n <= 0 || k <= 0 || num_chunks < 0) return CAP_ERR_ARG;
  if (cap_topo_get(topo, 9) != 0) return CAP_ERR_ARG;                 // topo::square only
  cap_summa_plan* p = new (std::nothrow) cap_summa_plan();
  if (!p) return CAP_ERR_ALLOC;
  p->topo = topo;
  p->c = cap_topo_get(topo, 2); p->d = cap_topo_get(topo, 3);
  p->x = cap_topo_get(topo, 4); p->y = cap_topo_get(topo, 5); p->z = cap_topo_get(topo, 6);
  if (p->d % p->c) { delete p; return CAP_ERR_UNSUPPORTED; }          // every inner index needs exactly one layer
  p->row = cap_topo_comm(topo, 1); p->column = cap_topo_comm(topo, 2); p->depth = cap_topo_comm(topo, 3);
  p->m = m; p->n = n; p->k = k;
  p->ml = cap_ceil_div(m, p->d); p->nl = cap_ceil_div(n, p->d); p->kl = cap_ceil_div(k, p->d);
  p->num_chunks = std::max(1, num_chunks);
  if (p->num_chunks > p->nl) p->num_chunks = (int)p->nl;
  for (int i = 0; i < 2; i++) p->bufA[i] = p->bufB[i] = p->triA[i] = p->triB[i] = p->xch[i] = nullptr;
  p->acc = nullptr; p->s_row = p->s_col = nullptr;
  hipError_t e = hipSuccess;
  for (int i = 0; i < 2 && e == hipSuccess; i++) {
    e = hipMalloc((void**)&p->bufA[i], sizeof(double) * p->ml * p->kl);
    if (e == hipSuccess) e = hipMalloc((void**)&p->bufB[i], sizeof(double) * p->kl * p->nl);
  }
  if (e == hipSuccess) e = hipMalloc((void**)&p->acc, sizeof(double) * p->ml * p->nl);
  int lo = 0, hi = 0;
  if (e == hipSuccess) e = hipDeviceGetStreamPriorityRange(&lo, &hi);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&p->s_row, hipStreamNonBlocking, hi);
  if (e == hipSuccess) e = hipStreamCreateWithPriority(&p->s_col, hipStreamNonBlocking, hi);
  auto mk = [&](std::vector<hipEvent_t>& v, size_t cnt) {
    v.resize(cnt);
    for (auto& ev : v) if (e == hipSuccess) e = hipEventCreateWithFlags(&ev, hipEventDisableTiming);
  };
  mk(p->ev_a, 2); mk(p->ev_b, 2 * (size_t)p->num_chunks); mk(p->ev_free, 2); mk(p->ev_d, (size_t)p->num_chunks + 1);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_start, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_join_r, hipEventDisableTiming);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_join_c, hipEventDisableTiming);
  if (e != hipSuccess) { cap_summa_plan_destroy(p); return CAP_ERR_ALLOC; }
  *plan = p;
  return CAP_OK;
}

int cap_summa_plan_destroy(cap_summa_plan* p) {
  if (!p) return CAP_OK;
  for (int i = 0; i < 2; i++)
    for (double* q : {p->bufA[i], p->bufB[i], p->triA[i], p->triB[i], p->xch[i]}) if (q) (void)hipFree(q);
  if (p->acc) (void)hipFree(p->acc);
  for (auto* v : {&p->ev_a, &p->ev_b, &p->ev_free, &p->ev_d}) for (auto ev : *v) if (ev) (void)hipEventDestroy(ev);
  if (p->ev_start) { (void)hipEventDestroy(p->ev_start); (void)hipEventDestroy(p->ev_join_r); (void)hipEventDestroy(p->ev_join_c); }
  if (p->s_row) cap_stream_destroy(p->s_row);
  if (p->s_col) cap_stream_destroy(p->s_col);
  delete p;
  return CAP_OK;
}

void cap_summa_local_dims(const cap_summa_plan* p, int64_t* ml, int64_t* nl, int64_t* kl) {
  if (!p) return;
  if (ml) *ml = p->ml; if (nl) *nl = p->nl; if (kl) *kl = p->kl;
}

// ---- the SUMMA carrier shared by the three overloads (summa.hpp:6-161) ------------------------------------------------------
// One slot pair per inner process index kp: the A slot travels along my process row (root x == kp), the B slot down my
// process column (root y == kp); acc (+)= alpha op(Aslot) op(Bslot).  How a slot's piece is staged on its root:
//   SLOT_RECT    the ra x ca piece as it is
//   SLOT_TRI     a square piece holding an upper triangle (rect storage): upper part copied, strictly-lower part zero-filled
//   SLOT_PACKED  the same triangle in upstream's packed-upper storage (structure.h:39): the PACKED image is broadcast
//                (n (n + 1) / 2 doubles, as upstream does, summa.hpp:185) and unpacked behind the broadcast on every rank
// A triangular piece (kp, y) of a globally upper-triangular T has zeros exactly where local row > local column (and on the
// local diagonal when y > kp - stored zeros), so the local product is a GEMM whose K ranges stop at the triangle (tags
// 8 / 16 / 32 of cap_gemm_launch) - what upstream gets from calling cblas_dtrmm on the piece (summa.hpp:66,73).
enum { SLOT_RECT = 0, SLOT_TRI = 1, SLOT_PACKED = 2 };
struct Slot { const double* src; int64_t ld; int mode; int64_t rows, cols; };   // stored dims of the piece

static int summa_core(cap_summa_plan* p, int opa, int opb, const Slot& A, const Slot& B, int64_t M, int64_t N, int64_t K, double alpha,
                      int tag, int chunked, hipStream_t s0) {
  hipStream_t sr = p->s_row, sc = p->s_col;
  const int nch = chunked ? p->num_chunks : 1;
  CAP_HIP(hipEventRecord(p->ev_start, s0));
  CAP_HIP(hipStreamWaitEvent(sr, p->ev_start, 0));
  CAP_HIP(hipStreamWaitEvent(sc, p->ev_start, 0));
  std::vector<int> steps;
  for (int kp = p->z; kp < p->d; kp += p->c) steps.push_back(kp);
  // stage + broadcast one slot; returns the buffer the GEMM reads (the unpack target for packed triangles)
  auto move = [&](const Slot& S, bool is_root, double* buf, double* tribuf, cap_comm* comm, int root, hipStream_t s, int chunks, hipEvent_t* evs,
                  const double** out) -> int {
    if (S.mode == SLOT_PACKED) {
      const int64_t np = S.rows * (S.rows + 1) / 2;
      if (is_root) CAP_HIP(hipMemcpyAsync(buf, S.src, sizeof(double) * np, hipMemcpyDeviceToDevice, s));
      CAP_TRY(cap_comm_bcast(comm, buf, np, root, (void*)s));
      CAP_TRY(cap_copy_window(buf, 1, 0, 0, 0, tribuf, 0, S.rows, 0, 0, S.rows, S.rows, 1, 1, (void*)s));
      CAP_HIP(hipEventRecord(evs[0], s));
      *out = tribuf;
      return CAP_OK;
    }
    if (is_root) {
      if (S.mode == SLOT_TRI) CAP_TRY(cap_copy_window(S.src, 0, S.ld, 0, 0, buf, 0, S.rows, 0, 0, S.rows, S.rows, 1, 1, (void*)s));
      else CAP_TRY(cap_copy_rect(S.src, S.ld, buf, S.rows, S.rows, S.cols, s));
    }
    for (int ch = 0; ch < chunks; ch++) {
      const int64_t c0 = S.cols * ch / chunks, c1 = S.cols * (ch + 1) / chunks;
      CAP_TRY(cap_comm_bcast(comm, buf + c0 * S.rows, S.rows * (c1 - c0), root, (void*)s));
      CAP_HIP(hipEventRecord(evs[ch], s));
    }
    *out = buf;
    return CAP_OK;
  };
  for (size_t si = 0; si < steps.size(); si++) {
    const int kp = steps[si], buf = (int)(si & 1);
    if (si >= 2) { CAP_HIP(hipStreamWaitEvent(sr, p->ev_free[buf], 0)); CAP_HIP(hipStreamWaitEvent(sc, p->ev_free[buf], 0)); }
    const double *Ab = nullptr, *Bb = nullptr;
    CAP_TRY(move(A, p->x == kp, p->bufA[buf], p->triA[buf], p->row, kp, sr, 1, &p->ev_a[buf], &Ab));
    CAP_TRY(move(B, p->y == kp, p->bufB[buf], p->triB[buf], p->column, kp, sc, nch, &p->ev_b[(size_t)buf * p->num_chunks], &Bb));
    CAP_HIP(hipStreamWaitEvent(s0, p->ev_a[buf], 0));
    for (int ch = 0; ch < nch; ch++) {
      const int64_t c0 = N * ch / nch, c1 = N * (ch + 1) / nch;          // chunked: opb == NoTrans, B slot K x N (ld = K)
      CAP_HIP(hipStreamWaitEvent(s0, p->ev_b[(size_t)buf * p->num_chunks + ch], 0));
      CAP_TRY(cap_gemm_launch(opa, opb, M, c1 - c0, K, alpha, Ab, A.rows, Bb + c0 * B.rows, B.rows, si == 0 ? 0.0 : 1.0, p->acc + c0 * M, M, 0, s0, tag));
      if (p->c > 1 && nch > 1 && si + 1 == steps.size()) CAP_HIP(hipEventRecord(p->ev_d[ch], s0));   // this column chunk of acc is final
    }
    CAP_HIP(hipEventRecord(p->ev_free[buf], s0));
  }
  // ---- depth: sum the layers' partial products (collect, summa.hpp:223-253).  num_chunks <= 1: one all-reduce (upstream's
  // num_chunks == 0 branch, :229-237).  Otherwise num_chunks all-reduces like upstream's MPI_Iallreduce sequence (:238-249) - and where
  // the local products ran chunk by chunk, the sum of column chunk j starts on the row stream (idle by now, its own communicator)
  // as soon as the LAST step's product of that chunk is done, i.e. it overlaps the products of the chunks behind it; the caller's
  // stream only waits for the last sum.
  if (p->c > 1) {
    const int dch = p->num_chunks;
    if (dch <= 1) {
      CAP_TRY(cap_comm_allreduce_sum(p->depth, p->acc, M * N, (void*)s0));
    } else if (nch == dch) {
      for (int ch = 0; ch < dch; ch++) {
        const int64_t c0 = N * ch / dch, c1 = N * (ch + 1) / dch;
        CAP_HIP(hipStreamWaitEvent(sr, p->ev_d[ch], 0));
        if (c1 > c0) CAP_TRY(cap_comm_allreduce_sum(p->depth, p->acc + c0 * M, M * (c1 - c0), (void*)sr));
      }
      CAP_HIP(hipEventRecord(p->ev_d[dch], sr));
      CAP_HIP(hipStreamWaitEvent(s0, p->ev_d[dch], 0));
    } else {
      // products that ran as one launch (right-sided TRMM, NoTrans SYRK): upstream's equal pieces, the remainder with the last one
      const int64_t tot = M * N, per = tot / dch;
      for (int ch = 0; ch < dch; ch++) {
        const int64_t cnt = ch == dch - 1 ? tot - per * (dch - 1) : per;
        if (cnt > 0) CAP_TRY(cap_comm_allreduce_sum(p->depth, p->acc + per * ch, cnt, (void*)s0));
      }
    }
  }
  return CAP_OK;
}

static int summa_join(cap_summa_plan* p, hipStream_t s0) {
  CAP_HIP(hipEventRecord(p->ev_join_r, p->s_row));
  CAP_HIP(hipEventRecord(p->ev_join_c, p->s_col));
  CAP_HIP(hipStreamWaitEvent(s0, p->ev_join_r, 0));
  CAP_HIP(hipStreamWaitEvent(s0, p->ev_join_c, 0));
  return CAP_OK;
}

static int ensure_tri(cap_summa_plan* p, int64_t na, int64_t nb_) {
  for (int i = 0; i < 2; i++) {
    if (na > 0 && !p->triA[i]) CAP_HIP(hipMalloc((void**)&p->triA[i], sizeof(double) * na * na));
    if (nb_ > 0 && !p->triB[i]) CAP_HIP(hipMalloc((void**)&p->triB[i], sizeof(double) * nb_ * nb_));
  }
  return CAP_OK;
}

// C_local = alpha * (A B)_local + beta * C_local.  A_local: ml x kl (lda), B_local: kl x nl (ldb), C_local: ml x nl (ldc),
// all column-major element-cyclic pieces with zero padding (matrix.hpp:8-11).  NoTrans x NoTrans, like the reference's
// driver (bench/matmult/summa_gemm.cpp:38); the transposed forms need a partner exchange upstream does outside SUMMA.
int cap_summa_dgemm(cap_summa_plan* p, double alpha, const double* A, int64_t lda, const double* B, int64_t ldb, double beta,
                    double* C, int64_t ldc, void* stream) {
  if (!p || !A || !B || !C || lda < p->ml || ldb < p->kl || ldc < p->ml) return CAP_ERR_ARG;
  hipStream_t s0 = cap_stream(stream);
  const int64_t ml = p->ml, nl = p->nl, kl = p->kl;
  if (nl > 65535) return CAP_ERR_UNSUPPORTED;
  CAP_TRY(summa_core(p, CAP_NOTRANS, CAP_NOTRANS, Slot{A, lda, SLOT_RECT, ml, kl}, Slot{B, ldb, SLOT_RECT, kl, nl}, ml, nl, kl, alpha, 0, 1, s0));
  // C = beta C + acc (summa.hpp:32-35)
  cap_acc(beta != 0.0 ? CAP_ACC_RW : CAP_ACC_W, C, ldc, ml, nl); cap_acc_r(p->acc, ml, ml, nl);
  hipLaunchKernelGGL(combine_kernel, dim3((unsigned)std::min<int64_t>(cap_ceil_div(ml, 256), 1024), (unsigned)nl), dim3(256), 0, s0, C, ldc,
                     p->acc, ml, ml, nl, beta);
  CAP_HIP(hipGetLastError());
  return summa_join(p, s0);
}

// util::transpose (util.hpp:232-247): swap my piece with the transpose partner (x, y, z) <-> (y, x, z) of topo::square - the
// received piece is NOT transposed (it is the partner's piece as stored).  count doubles of `buf`; tmp: scratch of `count`.
int cap_util_transpose(cap_topo* topo, double* buf, double* tmp, int64_t count, void* stream) {
  if (!topo || cap_topo_get(topo, 9) != 0) return CAP_ERR_ARG;
  const int c = cap_topo_get(topo, 2), d = cap_topo_get(topo, 3), x = cap_topo_get(topo, 4), y = cap_topo_get(topo, 5), z = cap_topo_get(topo, 6);
  const int partner = x * c * d + y * c + z;             // layout 0: rank = z + c x + c d y with x and y swapped
  return cap_comm_exchange(cap_topo_comm(topo, 0), buf, tmp, count, partner, stream);
}

// matmult::summa::invoke, TRMM overload (summa.hpp:46-83): B <- alpha op(T) B (side LEFT, T m x m) or alpha B op(T) (RIGHT,
// T n x n) on element-cyclic pieces, in place in B like upstream.  The plan is created with (m, n, k = m) for LEFT and
// (m, n, k = n) for RIGHT.  T_local: my piece of the globally upper-triangular T - rect storage (t_packed = 0: only its upper
// triangle is referenced) or upstream's packed-upper storage (t_packed = 1, ldt ignored).  trans == CAP_TRANS: T_local must be
// the piece AFTER util::transpose (cap_util_transpose), exactly as upstream's call site prepares it (cholinv.hpp:114-120).
int cap_summa_dtrmm(cap_summa_plan* p, int side, int uplo, int trans, int diag, double alpha, const double* T, int64_t ldt, int t_packed,
                    double* B, int64_t ldb, void* stream) {
  if (!p || !T || !B || ldb < p->ml) return CAP_ERR_ARG;
  if (uplo != CAP_UPPER || diag != CAP_NONUNIT) return CAP_ERR_UNSUPPORTED;       // every upstream call site
  hipStream_t s0 = cap_stream(stream);
  const int64_t ml = p->ml, nl = p->nl, kl = p->kl;
  const bool left = side == CAP_LEFT;
  if ((left && kl != ml) || (!left && kl != nl) || (!t_packed && ldt < kl) || nl > 65535) return CAP_ERR_ARG;
  const int tmode = t_packed ? SLOT_PACKED : SLOT_TRI;
  CAP_TRY(ensure_tri(p, left && t_packed ? kl : 0, !left && t_packed ? kl : 0));
  if (left) {
    // acc[ly, lx] = sum over kp, lk of  op(Tpiece)[ly, lk] B(x, kp)[lk, lx];  NoTrans: Tpiece = T(kp, y); Trans: the exchanged piece
    // on (kp, y) is T(y, kp) = T[lk d + kp, ly d + y], used transposed
    CAP_TRY(summa_core(p, trans, CAP_NOTRANS, Slot{T, ldt, tmode, kl, kl}, Slot{B, ldb, SLOT_RECT, kl, nl}, ml, nl, kl, alpha,
                       trans == CAP_TRANS ? 16 : 32, 1, s0));
  } else {
    // acc[ly, lx] = sum B(kp, y)[ly, lk] op(Tpiece)[lk, lx];  NoTrans: Tpiece = T(x, kp); Trans: the exchanged piece on (x, kp) is T(kp, x)
    CAP_TRY(summa_core(p, CAP_NOTRANS, trans, Slot{B, ldb, SLOT_RECT, ml, kl}, Slot{T, ldt, tmode, kl, kl}, ml, nl, kl, alpha,
                       trans == CAP_TRANS ? 0 : 8, 0, s0));
  }
  cap_acc_w(B, ldb, ml, nl); cap_acc_r(p->acc, ml, ml, nl);
  hipLaunchKernelGGL(combine_kernel, dim3((unsigned)std::min<int64_t>(cap_ceil_div(ml, 256), 1024), (unsigned)nl), dim3(256), 0, s0, B, ldb,
                     p->acc, ml, ml, nl, 0.0);
  CAP_HIP(hipGetLastError());
  return summa_join(p, s0);
}

namespace {
// C (rect, full local square - what upstream leaves for a rect C, summa.hpp:146-151 - or packed upper: local upper triangle)
__global__ void combine_packed_kernel(double* Cp, const double* acc, int64_t n, double beta) {
  const int64_t col = blockIdx.y;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row <= col; row += (int64_t)gridDim.x * blockDim.x) {
    double* q = Cp + col * (col + 1) / 2 + row;
    const double a = acc[row + col * n];
    *q = beta == 0.0 ? a : beta * (*q) + a;
  }
}
}  // namespace

// matmult::summa::invoke, SYRK overload (summa.hpp:85-161): C <- alpha A^T A + beta C (trans == CAP_TRANS, A k x n: the form
// cholinv.hpp:128-133 uses) or alpha A A^T + beta C (NoTrans, A n x k) on element-cyclic pieces.  Like upstream it is a GEMM of
// A's piece with the piece of the TRANSPOSE PARTNER (a copy of A exchanged through util::transpose, summa.hpp:91-92) - a rank's
// block of the symmetric result is not symmetric itself.  Plan: (m = n, n = n, k).  C_local: nl x nl rect (c_packed = 0: the
// whole local square is written) or packed upper (c_packed = 1: the local upper triangle).
int cap_summa_dsyrk(cap_summa_plan* p, int uplo, int trans, double alpha, const double* A, int64_t lda, double beta, double* C,
                    int64_t ldc, int c_packed, void* stream) {
  if (!p || !A || !C || p->ml != p->nl || (!c_packed && ldc < p->nl)) return CAP_ERR_ARG;
  if (uplo != CAP_UPPER) return CAP_ERR_UNSUPPORTED;
  hipStream_t s0 = cap_stream(stream);
  const int64_t nl = p->nl, kl = p->kl;
  if (nl > 65535) return CAP_ERR_UNSUPPORTED;
  const bool tr = trans == CAP_TRANS;
  const int64_t ra = tr ? kl : nl, ca = tr ? nl : kl;      // stored dims of A's piece
  if (lda < ra) return CAP_ERR_ARG;
  // the exchanged copy (upstream: MatrixSrcType B = A; util::transpose(B), summa.hpp:91)
  if (!p->xch[0]) for (int i = 0; i < 2; i++) CAP_HIP(hipMalloc((void**)&p->xch[i], sizeof(double) * nl * kl));
  CAP_TRY(cap_copy_rect(A, lda, p->xch[0], ra, ra, ca, s0));
  CAP_TRY(cap_util_transpose(p->topo, p->xch[0], p->xch[1], ra * ca, stream));
  if (tr) {
    // acc[ly, lx] = sum_k A[k, gy] A[k, gx]:  A slot = exchanged piece on (kp, y) = A(y, kp) = A[lk d + kp, ly d + y] used transposed;
    //                                       B slot = A(x, kp) = A[lk d + kp, lx d + x]
    CAP_TRY(summa_core(p, CAP_TRANS, CAP_NOTRANS, Slot{p->xch[0], kl, SLOT_RECT, kl, nl}, Slot{A, lda, SLOT_RECT, kl, nl}, nl, nl, kl, alpha, 0, 1, s0));
  } else {
    // acc[ly, lx] = sum_k A[gy, k] A[gx, k]:  A slot = A(kp, y) = A[ly d + y, lk d + kp];  B slot = exchanged piece on (x, kp) = A(kp, x) =
    //                                       A[lx d + x, lk d + kp] used transposed
    CAP_TRY(summa_core(p, CAP_NOTRANS, CAP_TRANS, Slot{A, lda, SLOT_RECT, nl, kl}, Slot{p->xch[0], nl, SLOT_RECT, nl, kl}, nl, nl, kl, alpha, 0, 0, s0));
  }
  if (c_packed) { cap_acc_rw(C, 0, nl * (nl + 1) / 2, 1); cap_acc_r(p->acc, nl, nl, nl, 1); }
  else { cap_acc(beta != 0.0 ? CAP_ACC_RW : CAP_ACC_W, C, ldc, nl, nl); cap_acc_r(p->acc, nl, nl, nl); }
  if (c_packed) hipLaunchKernelGGL(combine_packed_kernel, dim3((unsigned)std::min<int64_t>(cap_ceil_div(nl, 256), 1024), (unsigned)nl), dim3(256), 0, s0, C, p->acc, nl, beta);
  else hipLaunchKernelGGL(combine_kernel, dim3((unsigned)std::min<int64_t>(cap_ceil_div(nl, 256), 1024), (unsigned)nl), dim3(256), 0, s0, C, ldc, p->acc, nl, nl, nl, beta);
  CAP_HIP(hipGetLastError());
  return summa_join(p, s0);
}

}  // extern "C"
