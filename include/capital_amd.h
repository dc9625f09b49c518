/* capital_amd.h - C ABI of the MI355X-native CAPITAL hot path (libcapital_amd.so).
 *
 * This is the drop-in boundary.  The reference (tbennun/capital) has no FFI: its seam is
 * C++ static-template call signatures.  Each entry point below names the reference
 * interface it replaces (file:line relative to the reference tree).  INTEGRATION.md shows
 * the reference-side binding a maintainer would add.
 *
 * Conventions
 *  - all matrices are COLUMN-MAJOR fp64 in DEVICE memory (HBM), caller owned;
 *  - dimensions / leading dimensions are int64_t (blas/interface.h:58-66 uses int64_t);
 *  - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call is
 *    asynchronous on that stream unless stated otherwise;
 *  - every function returns a cap_status (0 = ok).  The reference has no error channel
 *    (LAPACKE info is dropped, lapack/interface.hpp:39,54); here POTRF's `info` is
 *    propagated through a device-resident int the caller can read back.
 *  - enums use the reference's own numeric values (blas/engine.h:23-52,
 *    lapack/engine.h:23-52) so ArgPacks map 1:1.
 */
#ifndef CAPITAL_AMD_H_
#define CAPITAL_AMD_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
  CAP_OK = 0,
  CAP_ERR_ARG = 1,        /* bad argument */
  CAP_ERR_HIP = 2,        /* a HIP runtime call failed */
  CAP_ERR_NOT_SPD = 3,    /* potrf met a non-positive pivot (info > 0) */
  CAP_ERR_UNSUPPORTED = 4,
  CAP_ERR_COMM = 5,       /* RCCL failure */
  CAP_ERR_ALLOC = 6
} cap_status;

/* blas/engine.h:23-52 */
enum { CAP_NOTRANS = 0, CAP_TRANS = 1 };
enum { CAP_LEFT = 0, CAP_RIGHT = 1 };
enum { CAP_LOWER = 0, CAP_UPPER = 1 };
enum { CAP_NONUNIT = 0, CAP_UNIT = 1 };

const char* cap_status_string(int status);
/* library / device info: fills name (<= len bytes), CU count, HBM bytes. */
int cap_device_info(char* name, int len, int* cus, int64_t* hbm_bytes);

/* ------------------------------------------------------------------------------------
 * Operator seam  (replaces blas::engine / lapack::engine, device pointers + status)
 * ---------------------------------------------------------------------------------- */

/* blas::engine::_gemm  - blas/interface.h:58-60, interface.hpp:43-59 (cblas_dgemm).
 * C[m x n] = alpha * op(A)[m x k] * op(B)[k x n] + beta * C.
 * n <= 8 with op(B) = B and m >= 1024 (a few right-hand sides against a big operand: refinement residuals, TRSM block steps)
 * runs on streaming kernels that read op(A) once; their transposed form sums in 64-k blocks with Kahan's correction.      */
int cap_dgemm(int transa, int transb, int64_t m, int64_t n, int64_t k, double alpha,
              const double* A, int64_t lda, const double* B, int64_t ldb, double beta,
              double* C, int64_t ldc, void* stream);

/* blas::engine::_syrk  - blas/interface.h:65-66, interface.hpp:81-97 (cblas_dsyrk).
 * C[n x n](uplo triangle only) = alpha * op(A) op(A)^T + beta * C;
 * trans == CAP_TRANS: A is k x n and C = alpha*A^T*A + beta*C (the form cacqr.hpp:15 and
 * the trailing update cholinv.hpp:128-137 use).                                        */
int cap_dsyrk(int uplo, int trans, int64_t n, int64_t k, double alpha, const double* A,
              int64_t lda, double beta, double* C, int64_t ldc, void* stream);

/* blas::engine::_trmm  - blas/interface.h:62-63, interface.hpp:61-79 (cblas_dtrmm).
 * B[m x n] = alpha * op(T) * B (side = LEFT, T m x m) or alpha * B * op(T) (RIGHT, T n x n).
 * In place like the reference; `work` is device scratch of >= cap_dtrmm_work_size doubles
 * (the reference hides the same copy inside MKL).  uplo = UPPER, diag = NONUNIT only (all
 * upstream call sites); other values return CAP_ERR_UNSUPPORTED.               */
int cap_dtrmm(int side, int uplo, int trans, int diag, int64_t m, int64_t n, double alpha,
              const double* T, int64_t ldt, double* B, int64_t ldb, double* work, void* stream);
int64_t cap_dtrmm_work_size(int side, int64_t m, int64_t n);

/* Real triangular solve (not in the reference, which inverts then multiplies - SURVEY 2b;
 * trsm/diaginvert/diaginvert.hpp:7-10 is a static_assert stub).  Solves
 * op(T) X = alpha B (LEFT) or X op(T) = alpha B (RIGHT) in place; T upper, non-unit.  Done as
 * "invert the triangle (recursive TRTRI), then one MFMA GEMM" - upstream's invert-then-multiply.
 * `work`: device scratch >= cap_dtrsm_work_size doubles.                                   */
int cap_dtrsm(int side, int uplo, int trans, int64_t m, int64_t n, double alpha, const double* T,
              int64_t ldt, double* B, int64_t ldb, double* work, void* stream);
int64_t cap_dtrsm_work_size(int side, int64_t m, int64_t n);

/* lapack::engine::_potrf - lapack/interface.h:49-50, interface.hpp:30-43 (LAPACKE_dpotrf).
 * In-place A = R^T R (uplo = UPPER; LOWER returns CAP_ERR_UNSUPPORTED - upstream removed
 * 'L' too, cholinv.hpp:9) of the n x n block; the other triangle is not referenced.  `info` (device int, may be NULL): 0 or 1-based index of the first
 * non-positive pivot.  `work`: device scratch >= cap_dpotrf_work_size(n) doubles.         */
int cap_dpotrf(int uplo, int64_t n, double* A, int64_t lda, int* info, double* work, void* stream);
int64_t cap_dpotrf_work_size(int64_t n);

/* lapack::engine::_trtri - lapack/interface.h:52-53, interface.hpp:45-58 (LAPACKE_dtrtri).
 * In-place inverse of the triangular n x n block (non-unit).  work >= cap_dtrtri_work_size. */
int cap_dtrtri(int uplo, int64_t n, double* A, int64_t lda, double* work, void* stream);
int64_t cap_dtrtri_work_size(int64_t n);

/* ------------------------------------------------------------------------------------
 * Matrix descriptor helpers (replaces src/matrix/: generators, serialize, structure)
 * ---------------------------------------------------------------------------------- */

/* matrix<T,U,Structure> descriptor - matrix.h:9-97: global dims (X = columns, Y = rows, upstream's naming), process
 * grid, local element-cyclic dims ceil(global / grid) (matrix.hpp:8-11), one column-major HBM buffer and its ownership:
 * cap_desc_create allocates (zero-filled; matrix.hpp:5-50,141-155), cap_desc_create_view is the injection constructor
 * (matrix.hpp:52-74: the caller keeps ownership of `device_data`), destroy frees only what the descriptor owns
 * (matrix.hpp:157-169).  import / export move the local piece between HOST memory (column-major, ld_host) and HBM
 * through two pinned 64 MiB chunk buffers, overlapping the host-side copy of one chunk with the PCIe transfer of the
 * other; a host pointer that is already pinned is copied directly.  import is ordered on `stream`; export blocks
 * until the host buffer is complete.  cap_desc_get: 0 global cols, 1 global rows, 2 local cols, 3 local rows, 4 ld,
 * 5 owns, 6 grid x, 7 grid y, 8 local element count.                                                                  */
typedef struct cap_desc cap_desc;
int cap_desc_create(cap_desc** desc, int64_t global_cols, int64_t global_rows, int64_t grid_x, int64_t grid_y);
int cap_desc_create_view(cap_desc** desc, int64_t global_cols, int64_t global_rows, int64_t grid_x, int64_t grid_y,
                         double* device_data, int64_t ld);
/* BLOCK-CYCLIC kind (north_star's "2D block-cyclic matrix descriptor"; upstream has only the element-cyclic map, matrix.hpp:8-11): nb x nb
 * blocks, block (I, J) on process (I mod Pr, J mod Pc) as local block (I div Pr, J div Pc).  The descriptor is the piece of process
 * (pr, pc): the VALID local rows x columns, compact, column-major - what cap_dist2d_factor / cap_dist2d_get_R take, and with Pr = 1 the
 * block columns of a multi-rank cap_cholinv_plan / cap_dist_plan.  device_data = NULL: allocated, zero-filled and owned; otherwise the
 * injection constructor (matrix.hpp:52-74) with leading dimension ld.  An empty piece (more processes than blocks) is valid.       */
int cap_desc_create_bc(cap_desc** desc, int64_t global_cols, int64_t global_rows, int64_t nb, int Pr, int Pc, int pr, int pc,
                       double* device_data, int64_t ld);
/* element-cyclic descriptors do not store their grid position (upstream passes (x, y) to the generators again, matrix.h:65-68);
 * the GLOBAL import / export below need it.                                                                                      */
int cap_desc_set_position(cap_desc* desc, int64_t x, int64_t y);
int cap_desc_destroy(cap_desc* desc);
double* cap_desc_data(cap_desc* desc);
int64_t cap_desc_get(const cap_desc* desc, int field);
int cap_desc_import_host(cap_desc* desc, const double* host, int64_t ld_host, void* stream);
int cap_desc_export_host(cap_desc* desc, double* host, int64_t ld_host, void* stream);
/* The caller holds the GLOBAL matrix in host memory (column-major, ld_host >= global rows: upstream's matrix<> on one rank, or what
 * a host application hands to a distributed solver): import picks this process's blocks (block-cyclic kind) or elements
 * (element-cyclic kind with cap_desc_set_position) out of it, export writes them back to their global places and touches nothing
 * else - every rank exporting into its own zero-filled copy and summing the copies assembles the global result.  Through the same
 * two pinned 64 MiB buffers: host threads pack / unpack one range of local columns while the previous one is on the PCIe link.
 * cap_desc_get adds: 9 kind (0 element-cyclic, 1 block-cyclic), 10 nb, 11 my grid column (pc / x; -1 unknown), 12 my grid row.  */
int cap_desc_import_host_global(cap_desc* desc, const double* host_global, int64_t ld_host, void* stream);
int cap_desc_export_host_global(cap_desc* desc, double* host_global, int64_t ld_host, void* stream);

/* rect::_distribute_symmetric - structure.hpp:68-103.  Fills the local element-cyclic piece
 * (grid position x,y of a d x d grid) of the N x N SPD test matrix directly on the GPU:
 * A[gy,gx] = u(max + N*min) (+N on the diagonal), u = srand48/drand48 closed form.
 * local: ceil(N/d) x ceil(N/d) column-major with leading dimension ld; padding zero-filled. */
int cap_fill_symmetric(double* local, int64_t ld, int64_t n_global, int64_t x, int64_t y, int64_t d,
                       int diagonally_dominant, void* stream);
/* rect::_distribute_random - structure.hpp:105-129 (sequential drand48 stream per rank,
 * key = rank / c; jump-ahead LCG so every element is computed independently, bit-exact).   */
int cap_fill_random(double* local, int64_t ld, int64_t m_global, int64_t n_global, int64_t x, int64_t y,
                    int64_t dx, int64_t dy, int64_t key, void* stream);

/* serialize<S1,S2>::invoke - serialize.hpp:12-150: copy a window between rect / packed-upper
 * buffers.  src/dst are either column-major rect (packed = 0, leading dim ld) or packed upper
 * (packed = 1: column x at x(x+1)/2, structure.h:39; ld ignored).  Copies the n x n window's
 * upper triangle (tri_only = 1) or the full rows x cols window (tri_only = 0).
 * zero_lower = 1 additionally zero-fills the strictly lower part of a rect destination.     */
int cap_copy_window(const double* src, int src_packed, int64_t src_ld, int64_t src_row0, int64_t src_col0,
                    double* dst, int dst_packed, int64_t dst_ld, int64_t dst_row0, int64_t dst_col0,
                    int64_t rows, int64_t cols, int tri_only, int zero_lower, void* stream);

/* Element-cyclic layout of the reference (matrix.hpp:8-11; util::block_to_cyclic_* / cyclic_to_local,
 * util.hpp:56-230): piece (x, y) of a dx x dy grid holds global rows y, y+dy, ... and columns x, x+dx, ...
 * (ceil sizes, zero padded).  import scatters one piece into a dense m x n matrix, export extracts it - so a
 * caller holding upstream-style cyclic pieces can assemble / split the dense operand of the GPU plans.      */
int cap_cyclic_import(const double* piece, int64_t ldp, double* dense, int64_t ldd, int64_t m, int64_t n, int64_t x, int64_t y,
                      int64_t dx, int64_t dy, void* stream);
int cap_cyclic_export(const double* dense, int64_t ldd, double* piece, int64_t ldp, int64_t m, int64_t n, int64_t x, int64_t y,
                      int64_t dx, int64_t dy, void* stream);

/* util::remove_triangle - util.hpp:266-318: zero the entries of a local element-cyclic piece
 * that are globally strictly below (dir 'U') / above ('L') the diagonal.                   */
int cap_remove_triangle(double* local, int64_t ld, int64_t rows_local, int64_t cols_local, int64_t x, int64_t y,
                        int64_t d, int dir_upper, void* stream);

/* util::residual_local pieces (util.hpp:25-53) + the validators' GEMMs
 * (test/cholesky/validate.hpp:33-46): writes out[0] = sum_{upper}(R^T R - A)^2,
 * out[1] = sum_{upper} A^2 (device doubles) for single-rank (d = 1) matrices.
 * work >= n*n doubles.                                                                    */
int cap_cholesky_residual_terms(const double* A, int64_t lda, const double* R, int64_t ldr, int64_t n,
                                double* work, double* out2, void* stream);
/* sum of squares of (alpha*X + beta*Y) over an m x n window, optional identity subtraction:
 * out[0] = sum (X - (sub_identity ? I : 0))^2 ; used by qr residual / orthogonality
 * (test/qr/validate.hpp:24-31,46-51).                                                      */
int cap_sumsq(const double* X, int64_t ldx, int64_t m, int64_t n, int sub_identity, int upper_only,
              double* out1, void* stream);

/* ------------------------------------------------------------------------------------
 * Communicators (replaces the MPI_Comm handles of topo::square / topo::rect, util/topology.h:16-143)
 * RCCL over xGMI, one process per GPU.  The 128-byte unique id is produced on rank 0 by
 * cap_comm_unique_id and shipped to the other ranks by the host (torch.distributed / MPI).
 * A communicator made by cap_comm_create issues RCCL calls for every collective, also when
 * size == 1; cap_comm_create_self is the RCCL-free single-rank form.
 * ---------------------------------------------------------------------------------- */
typedef struct cap_comm cap_comm;
int cap_comm_unique_id(void* id128);
int cap_comm_create(cap_comm** comm, const void* id128, int rank, int size, void* stream);
int cap_comm_create_self(cap_comm** comm);           /* P = 1, no RCCL */
/* Host-staged communicator: the three collectives are provided by the caller (device pointers in,
 * 0 = success).  The callback must order itself behind `stream` (and only that stream) before it
 * touches the buffers.  Used by the tests to run the multi-rank schedules with several processes
 * sharing one GPU (gloo over host memory); the product uses cap_comm_create (RCCL).              */
typedef int (*cap_allgather_fn)(void* ctx, const double* send, double* recv, int64_t count_per_rank, void* stream);
typedef int (*cap_bcast_fn)(void* ctx, double* buf, int64_t count, int root, void* stream);
typedef int (*cap_allreduce_fn)(void* ctx, double* buf, int64_t count, void* stream);
int cap_comm_create_callbacks(cap_comm** comm, int rank, int size, cap_allgather_fn allgather, cap_bcast_fn bcast,
                              cap_allreduce_fn allreduce, void* ctx);
/* MPI_Comm_split (topology.h:28-39,84-126) -> ncclCommSplit: collective over `comm`; color < 0 joins no
 * group (*out = NULL).  MPI_Comm_dup (topology.h:54-59): an independent communicator over the same ranks. */
int cap_comm_split(cap_comm* comm, int color, int key, cap_comm** out);
int cap_comm_dup(cap_comm* comm, cap_comm** out);
int cap_comm_destroy(cap_comm* comm);
int cap_comm_rank(const cap_comm* comm);
int cap_comm_size(const cap_comm* comm);
int cap_comm_backend(const cap_comm* comm);          /* 0 self, 1 RCCL, 2 host-staged */
/* what RCCL itself reports for this communicator: rank count, this rank, bound device (bench.py prints it as n_ranks_seen) */
int cap_comm_query(const cap_comm* comm, int* nranks, int* rank, int* device);
/* MPI_Allreduce(IN_PLACE, SUM) summa.hpp:236 | MPI_Reduce(SUM) to root cacqr.hpp:98 | MPI_Bcast summa.hpp:185
 * | MPI_Allgather policy.h:176 | MPI_Barrier bench/cholesky/cholinv.cpp:47 (drains the stream).          */
int cap_comm_allreduce_sum(cap_comm* comm, double* buf, int64_t count, void* stream);
int cap_comm_reduce_sum(cap_comm* comm, double* buf, int64_t count, int root, void* stream);
int cap_comm_bcast(cap_comm* comm, double* buf, int64_t count, int root, void* stream);
int cap_comm_allgather(cap_comm* comm, const double* send, double* recv, int64_t count_per_rank, void* stream);
int cap_comm_barrier(cap_comm* comm, void* stream);
/* Personalised all-to-all (what upstream assembles from MPI_Allgather + util::block_to_cyclic_* / cyclic_to_local,
 * util.hpp:56-230): rank r receives the `sendcounts[r]` doubles every rank addressed to it.  counts / displacements are HOST
 * arrays of cap_comm_size int64 (in doubles); grouped ncclSend / ncclRecv, the self piece is a device copy.
 * cap_comm_exchange = MPI_Sendrecv_replace with one partner (util::transpose, util.hpp:232-247): swaps `count` doubles of
 * `buf` with rank `partner` (tmp: device scratch of `count` doubles; partner == own rank: no-op).                       */
int cap_comm_alltoallv(cap_comm* comm, const double* send, const int64_t* sendcounts, const int64_t* sdispls, double* recv,
                       const int64_t* recvcounts, const int64_t* rdispls, void* stream);
int cap_comm_exchange(cap_comm* comm, double* buf, double* tmp, int64_t count, int partner, void* stream);
/* host-staged communicators (tests): the caller's all-to-all; 0 = success.  The self piece is copied by the library. */
typedef int (*cap_alltoallv_fn)(void* ctx, const double* send, const int64_t* sendcounts, const int64_t* sdispls, double* recv,
                                const int64_t* recvcounts, const int64_t* rdispls, void* stream);
int cap_comm_set_alltoallv_callback(cap_comm* comm, cap_alltoallv_fn fn);

/* Grid bundles: topo::square (kind 0, d x d x c, topology.h:67-143) and topo::rect (kind 1, c x d x c,
 * topology.h:16-65) - the row / column / depth / slice (/ column_contig / column_alt / cube)
 * sub-communicators split off `world` exactly as upstream's constructors do, plus the grid coordinates.
 * Collective over `world` (which stays owned by the caller).  Square layouts 1-2 are rejected
 * (numerically wrong upstream).  cap_topo_coords is the pure rank -> (d, x, y, z) map.               */
typedef struct cap_topo cap_topo;
int cap_topo_coords(int kind, int rank, int size, int c, int* d, int* x, int* y, int* z);
int cap_topo_create(cap_topo** topo, int kind, cap_comm* world, int c, int layout, int num_chunks);
/* bundle over caller-built sub-communicators: comms[7] = row, column, depth, slice, column_contig,
 * column_alt, cube (NULL where the kind has none); not owned.                                        */
int cap_topo_create_from(cap_topo** topo, int kind, cap_comm* world, int c, int layout, int num_chunks,
                         cap_comm** comms, int ncomms);
int cap_topo_destroy(cap_topo* topo);
/* which: 0 world, 1 row, 2 column, 3 depth, 4 slice, 5 column_contig, 6 column_alt, 7 cube */
cap_comm* cap_topo_comm(cap_topo* topo, int which);
/* field: 0 rank, 1 size, 2 c, 3 d, 4 x, 5 y, 6 z, 7 layout, 8 num_chunks, 9 kind */
int cap_topo_get(const cap_topo* topo, int field);

/* Distributed redistribution between the reference's element-cyclic pieces and the block-cyclic layouts of the multi-GPU
 * plans (csrc/redist.hip) - how a caller holding upstream-style pieces on P ranks reaches cap_dist_* / cap_dist2d_* / cap_dmp_*
 * and gets R / R^-1 back the way construct_R / construct_Rinv return them (cholinv.hpp:30-46).
 *   cyclic side: rank = z + c x + c d y of topo::square's d x d x c grid (topology.h:75-83) holds piece (x, y): global rows
 *                y, y + d, ..., columns x, x + d, ... (ceil(n / d) each, zero padded; matrix.hpp:8-11), replicated over z;
 *   block-cyclic side: block (I, J) of nb x nb on process (I mod Pr, J mod Pc), rank = pr Pc + pc, local block (I div Pr,
 *                J div Pc), compact local array (valid rows x valid columns); Pr = 1: the block columns of cap_dist_*.
 * One all-to-all over `world` (cap_comm_alltoallv) + one gather / scatter launch per peer; the index sets are derived on both
 * sides from (n, nb, d, Pr, Pc).  cyclic_to_bc: destination t reads from layer z = t mod c (the replicas share the work);
 * bc_to_cyclic: every rank of every layer receives its piece.  Collective, asynchronous on `stream`.
 * cap_redist_get: 0 cyclic piece edge, 1 / 2 valid local rows / columns of my block-cyclic piece, 3 d, 4 c, 5 x, 6 y, 7 z,
 * 8 Pr, 9 Pc, 10 pr, 11 pc, 12 / 13 doubles sent in direction cyclic->bc / bc->cyclic, 14 / 15 doubles received.        */
typedef struct cap_redist_plan cap_redist_plan;
int cap_redist_plan_create(cap_redist_plan** plan, int64_t n, int64_t nb, cap_comm* world, int c, int Pr);
int cap_redist_plan_destroy(cap_redist_plan* plan);
int64_t cap_redist_get(const cap_redist_plan* plan, int which);
/* pure index helper (no GPU): doubles rank `from` sends to rank `to`; dir 0 = cyclic -> block-cyclic, 1 = the way back */
int64_t cap_redist_message_elems(int64_t n, int64_t nb, int P, int c, int Pr, int from, int to, int dir);
int cap_redistribute_cyclic_to_bc(cap_redist_plan* plan, const double* piece, int64_t ldp, double* bc_local, int64_t ldb, void* stream);
int cap_redistribute_bc_to_cyclic(cap_redist_plan* plan, const double* bc_local, int64_t ldb, double* piece, int64_t ldp, void* stream);

/* ------------------------------------------------------------------------------------
 * Algorithm seam (replaces src/alg/cholesky/cholinv, src/alg/qr/cacqr)
 * ---------------------------------------------------------------------------------- */

/* cholesky::cholinv<...>::info + factor - cholinv.h:16-53, cholinv.hpp:6-28.
 * A plan handle (like `info`: create once, factor many times; owns R, Rinv and workspace).
 * Knobs keep the reference's meaning:
 *   complete_inv: 1 = full R^-1; 0 = skip the root-level Rinv12 (cholinv.hpp:147);
 *                 -1 (extension) = do not build R^-1 at all: blocked right-looking Cholesky
 *                 with real DTRSM/DSYRK - the headline "fp64 Cholesky" path;
 *   split:        root partition is n >> split (cholinv.hpp:107);
 *   bc_mult_dim:  base-case size knob (cholinv.hpp:15-18) -> panel width of the GPU schedule;
 *   dir:          'U' only, as upstream (cholinv.hpp:9).
 * comm = NULL or a size-1 communicator: single-GPU plan, A is the whole n x n matrix.
 * comm of size P > 1: the multi-GPU schedule of cap_dist_* behind the same handle - A, get_R / get_Rinv and
 * the *_ptr accessors then refer to THIS RANK's block-cyclic columns (global block column J, width nb, on
 * rank J % P; cap_bc_num_local_cols columns, all n rows; option "nb" sets the width); complete_inv = 0 / 1
 * also build R^-1 there (cap_dist_get_Rinv).
 * Option "cyclic_c" = c (multi-GPU plans; c = depth of topo::square's d x d x c grid, comm size = c d d): the plan speaks the
 * REFERENCE's layout end to end - factor's A is this rank's element-cyclic piece (ceil(n/d) x ceil(n/d), matrix.hpp:8-11),
 * get_R / get_Rinv write this rank's piece of R / R^-1 (what construct_R / construct_Rinv return on every rank of every layer,
 * cholinv.hpp:30-46; zero below the GLOBAL diagonal, i.e. already util::remove_triangle'd) - through the distributed
 * redistribution of cap_redistribute_* (one all-to-all each way).  get "piece" = the piece edge.  With it the knobs mean what they
 * mean upstream ON THAT GRID: the base-case dimension starts at c d (cholinv.hpp:15-18: with complete_inv = 0 and bc_mult_dim = 0 the
 * 2 x 2 x 2 grid partitions the root and leaves its block of R^-1 empty where one process would invert the whole base case), and
 * the root partition is taken on the local dimension, (ceil(n / d) >> split) d rows (cholinv.hpp:107).                          */
typedef struct cap_cholinv_plan cap_cholinv_plan;
int cap_cholinv_plan_create(cap_cholinv_plan** plan, int64_t n, int complete_inv, int64_t split,
                            int64_t bc_mult_dim, char dir, cap_comm* comm);
int cap_cholinv_plan_destroy(cap_cholinv_plan* plan);
/* factor: A (n x n col-major, lda) is read-only (only its upper triangle is consumed,
 * cholinv.hpp:13).  Results stay resident in the plan.                                     */
int cap_cholinv_factor(cap_cholinv_plan* plan, const double* A, int64_t lda, void* stream);
/* construct_R / construct_Rinv - cholinv.hpp:30-46: copy the upper-triangular result into a
 * caller rect buffer (strictly lower part zero-filled).                                    */
int cap_cholinv_get_R(cap_cholinv_plan* plan, double* out, int64_t ld, void* stream);
int cap_cholinv_get_Rinv(cap_cholinv_plan* plan, double* out, int64_t ld, void* stream);
/* The same three calls on descriptors - cholinv::factor(const Matrix& A, info&, Topo&&) and construct_R / construct_Rinv returning a
 * matrix (cholinv.h:46-53).  The descriptor must describe exactly the piece the plan works on, else CAP_ERR_ARG: a single-GPU plan
 * takes a 1 x 1-grid descriptor of the whole matrix; a multi-rank plan takes the block-cyclic kind with Pr = 1, Pc = P, pc = my rank
 * and the plan's nb (cap_desc_create_bc), or - option "cyclic_c" - upstream's element-cyclic kind on the d x d grid.               */
int cap_cholinv_factor_desc(cap_cholinv_plan* plan, const cap_desc* A, void* stream);
int cap_cholinv_get_R_desc(cap_cholinv_plan* plan, cap_desc* R, void* stream);
int cap_cholinv_get_Rinv_desc(cap_cholinv_plan* plan, cap_desc* Rinv, void* stream);
/* device pointers to the resident factors (leading dimension returned through *ld).        */
double* cap_cholinv_R_ptr(cap_cholinv_plan* plan, int64_t* ld);
double* cap_cholinv_Rinv_ptr(cap_cholinv_plan* plan, int64_t* ld);
/* host-readable status of the last factor: 0, or 1-based index of the failing pivot.  A launch of the one-launch
 * diagonal-block chain (option "chain_coop") whose workgroups were never all resident gives up after ~3 s of polling, and the
 * recovery launch behind it restores that diagonal block and re-runs it on two workgroups (counted in option
 * "chain_fallbacks"); -64 is only left if that re-run could not complete either (a stream restricted to fewer than two CUs).
 * Synchronises the stream.                                                                 */
int cap_cholinv_info(cap_cholinv_plan* plan, void* stream, int64_t* info);
/* tuning knobs of the GPU schedule: "nb" (panel width), "leaf", "lookahead", "outer" (strip height = K of
 * the bulk updates), "tail" (columns left below which strips are nb wide), "depth2" (look-ahead depth 2),
 * "occ1_m" (columns left below which bulk updates run one workgroup per CU so the diagonal-block chain
 * always finds a slot; 0 = never), "inner_la" (column-split look-ahead, off), "reserve" (CU-masked chain
 * stream, off), "serial_m", "fastdiag", "profile", "use_sb" (strip buffers: the solved block rows of a strip are written
 * K-contiguously into a ring of three buffers every update reads from, R receives a copy off the panel stream; on),
 * complete_inv >= 0: "inv_fast" (blocked sweep + inverse tree instead of the plain recursion of cholinv.hpp:85-165; on),
 * "inv_overlap" (tree nodes are enqueued as their inputs become final; on), "inv_start_m" (columns left below which the
 * tree starts), "fuse_copy" (only the first strip's rows of A are copied into R, the updates of step 0 read their C input
 * from A; on, bit-identical), "reserve_m" (with "reserve": the CU masks only apply once at most reserve_m columns are left;
 * measured slower, off), "chain_coop" (PER PLAN since round 5: resident workgroups of the one-launch diagonal-block chain - the
 * whole 64-blocked factor phase of a diagonal block + the inverse levels up to 256 as one launch whose workgroups meet at a
 * counter in device memory, csrc/leaf.hip; process default 32 (CAP_CHAIN_COOP), clamped to what the device holds at once; 0 = one
 * launch per 64-column step, bit-identical; -1 = back to the process default), get only: "chain_fallbacks" (diagonal blocks of
 * this process that the recovery launch had to re-run on this device; synchronises).
 * Multi-GPU plans forward to cap_dist_set_option.                                                                         */
int cap_cholinv_set_option(cap_cholinv_plan* plan, const char* key, int64_t value);
int64_t cap_cholinv_get_option(cap_cholinv_plan* plan, const char* key);
/* Diagnostics of the one-launch diagonal-block chain (no counterpart upstream: its base case is one LAPACKE_dpotrf + dtrtri on the
 * host, policy.h:307-414).  cap_chain_fallbacks: diagonal blocks of this process that the recovery launch restored and re-ran on the
 * current device because a workgroup gave up waiting for its peers (synchronises the device).  cap_chain_inject_timeouts(count):
 * TEST HOOK - the next `count` chain launches on the current device give up at their first meeting, so the recovery path runs.   */
int64_t cap_chain_fallbacks(void);
int cap_chain_inject_timeouts(int count);
/* Live measurement of the dominant kernel (trailing-update DSYRK) of the LAST factor call, enabled
 * with cap_cholinv_set_option(plan, "profile", 1): number of launches, their summed duration in ms
 * (HIP events recorded on the stream each launch went to) and summed algorithmic flops
 * (m(m+1)k per launch).  Synchronises on the recorded events.                               */
int cap_cholinv_profile(cap_cholinv_plan* plan, int64_t* launches, double* ms_total, double* flops_total);
/* the same launches of the LAST factor call one by one: up to cap entries of (ms, algorithmic flops); *count = launches */
int cap_cholinv_profile_launches(cap_cholinv_plan* plan, double* ms_out, double* flops_out, int64_t cap, int64_t* count);

/* Multi-GPU blocked Cholesky on a 1 x P block-column-cyclic matrix (the 2D block-cyclic descriptor with
 * Pr = 1): global block column J (width nb) lives on rank J % P as local block J / P; rows are not
 * distributed.  Per step: the owner factors + inverts the diagonal block, broadcasts
 * [R(k,k+1) | Dinv(k+1)], every rank solves its own part of block row k with one GEMM, the solved rows
 * are all-gathered and each rank applies the rank-nb update to its own columns (upper staircase only).
 * Replaces the MPI_Bcast / MPI_Allgather schedule of summa.hpp:163-253 + policy.h:160-305 with RCCL
 * over xGMI; one process per GPU.  Inputs/outputs are the LOCAL column blocks (n rows, ld >= n).     */
typedef struct cap_dist_plan cap_dist_plan;
/* nb: block width (multiple of 128; 0 = 512).  Any n: the plan pads to a multiple of nb with an identity tail. */
int cap_dist_plan_create(cap_dist_plan** plan, int64_t n, int64_t nb, cap_comm* comm);
int cap_dist_plan_destroy(cap_dist_plan* plan);
int64_t cap_dist_local_cols(const cap_dist_plan* plan);                 /* columns stored on this rank   */
int cap_dist_factor(cap_dist_plan* plan, const double* Alocal, int64_t lda, void* stream);
double* cap_dist_R_ptr(cap_dist_plan* plan, int64_t* ld);               /* local columns of R, ld = padded n; entries
                                                                           below the global diagonal are scratch */
int cap_dist_get_R(cap_dist_plan* plan, double* out, int64_t ld, void* stream);   /* construct_R: n x local_cols, zero below
                                                                                      the global diagonal              */
/* Options "complete_inv" = 0 / 1 (+ "split"): the factor call also leaves this rank's block columns of R^-1 - upstream's
 * R + R^-1 semantics on P > 1 (cholinv.hpp:85-165) - STREAMED with the sweep: as soon as block row k is solved, the owner of
 * block column k finishes that column of R^-1 (X[0:k+1, k] Dinv(k)), broadcasts it ((k+1) nb x nb doubles) and every rank
 * updates its own block columns J > k with its OWN piece of the solved row (X[0:k+1, J] -= X[0:k+1, k] R[k, J]) - one MFMA GEMM
 * per step on the plan's inverse stream, (n^3/3)/P flops per rank, half the bytes of an all-gather of R, no replicated R:
 * per-rank memory is the local columns of R and R^-1 plus two n x nb buffers (allocated when the option is set, so a failure is
 * reported before any collective).  "safe" = 1 runs the steps after the sweep on the one communicator.
 * complete_inv = 0 leaves Ri[0 : n >> split, n >> split : n] empty (cholinv.hpp:107,147).                              */
int cap_dist_get_Rinv(cap_dist_plan* plan, double* out, int64_t ld, void* stream);   /* construct_Rinv, same layout as get_R */
double* cap_dist_Rinv_ptr(cap_dist_plan* plan, int64_t* ld);
/* 0, or the smallest failing pivot (1-based) reported by any rank.  Collective; call it on the stream
 * cap_dist_factor ran on.                                                                              */
int cap_dist_info(cap_dist_plan* plan, void* stream, int64_t* info);
/* knobs: "strip" (block rows per bulk update, 1|2), "depth2" (split bulk updates), "occ1_m" (bulk updates of at most
 * occ1_m^2 rows x local columns run one workgroup per CU), "profile", "safe" (one communicator + one communication
 * stream), "ipc" (strip exchange as IPC peer copies; get "ipc_active" tells whether the peers could be mapped),
 * "complete_inv" / "split" (R^-1, see cap_dist_get_Rinv), "root_n1" (the root partition given explicitly instead of n >> split:
 * upstream cuts its LOCAL dimension, (ceil(n / d) >> split) d global rows on a d x d x c grid, cholinv.hpp:107 - a cholinv plan
 * with "cyclic_c" sets it, together with the grid's base-case rule cholinv.hpp:15-18; 0 = n >> split), "jitter_us" / "jitter_seed" (stress testing: random spin kernels in
 * front of every launch group); get only: "count_gemm" / "count_chain" / "count_copy" / "count_coll" = launches and
 * collectives of the last factor call on this rank.                                                                     */
int cap_dist_set_option(cap_dist_plan* plan, const char* key, int64_t value);
int64_t cap_dist_get_option(const cap_dist_plan* plan, const char* key);
int cap_dist_profile(cap_dist_plan* plan, int64_t* launches, double* ms_total, double* flops_total);
/* profile mode, per stream role: out6 = busy ms of the diagonal-block chains, block-row solves, HEAD updates (panel stream),
 * message broadcasts, strip exchanges (communication streams) and bulk updates (caller's stream) of the LAST factor call. */
int cap_dist_profile_streams(cap_dist_plan* plan, double* out6);
/* the bulk updates of the LAST factor call in profile mode, one by one: up to cap entries of (ms, algorithmic flops); *count = launches */
int cap_dist_profile_launches(cap_dist_plan* plan, double* ms_out, double* flops_out, int64_t cap, int64_t* count);
/* profile mode, complete_inv >= 0: out3 = busy ms of the streamed inverse's launch groups, ms of it left after the sweep's join
 * (what the overlap did not hide), ms of the whole factor call.                                                          */
int cap_dist_profile_inverse(cap_dist_plan* plan, double* out3);
/* Non-blocking progress of the factor call in flight (host watchdog of bench.py): out9 = leading complete events among
 * fact, msg, rowdone (per block row), solved, gather, head2, rest (per strip), then the block-row and strip counts.
 * Option "safe" = 1 runs the same schedule with ONE communicator and ONE communication stream (collectives in program order). */
int cap_dist_progress(cap_dist_plan* plan, int64_t* out9);
/* distribute_symmetric (structure.hpp:68-103) for this layout: fills the local block columns.            */
int cap_fill_symmetric_bc(double* local, int64_t ld, int64_t n, int64_t nb, int P, int p, int diagonally_dominant,
                          void* stream);
/* pure index helpers (no GPU): owner / local block / local column offset of global block column J    */
int cap_bc_owner(int64_t J, int P);
int64_t cap_bc_local_block(int64_t J, int P);
int64_t cap_bc_num_local_cols(int64_t n, int64_t nb, int P, int p);

/* The same factorization on a 2D Pr x Pc BLOCK-CYCLIC process grid (csrc/dist2d.hip): block (I, J) of nb x nb elements on
 * process (I % Pr, J % Pc) as local block (I / Pr, J / Pc), rank = pr * Pc + pc, local arrays column-major; Pr must divide
 * Pc (1 x P, 2 x 2, 2 x 4, 4 x 4).  Strips of two block rows like cap_dist_*: per block row the diagonal block on its owner,
 * [R(a,b) | Dinv] along the owner's process row, block-row solve there, the solved row down the process columns into a
 * K-contiguous strip buffer (B operand); per strip the pieces I = pr mod Pr re-broadcast along the process rows by the Pc / Pr
 * columns that hold them (A operand) and ONE K = 2 nb staircase MFMA update of the local blocks b < I <= J, look-ahead depth 2.
 * Replaces topo::square's row / column communicators + summa::distribute (topology.h:67-143, summa.hpp:163-221).
 * row / col: communicators of my process row (ordered by pc) / column (ordered by pr), or NULL to split them off `world`
 * (ncclCommSplit).  Alocal / get_R: the valid local piece (cap_bc2d_local_extent rows x columns, column-major).
 * cap_dist2d_get: 0 valid local rows, 1 valid local columns, 2 Pr, 3 Pc, 4 pr, 5 pc, 6 nb, 7 padded n, 8..11 = MFMA kernels,
 * diagonal-block chains, copy kernels, collectives issued by the last factor call on this rank; 12 = IPC moves active.    */
typedef struct cap_dist2d_plan cap_dist2d_plan;
int cap_dist2d_plan_create(cap_dist2d_plan** plan, int64_t n, int64_t nb, cap_comm* world, int Pr, cap_comm* row, cap_comm* col);
int cap_dist2d_plan_destroy(cap_dist2d_plan* plan);
int64_t cap_dist2d_get(const cap_dist2d_plan* plan, int which);
int cap_dist2d_factor(cap_dist2d_plan* plan, const double* Alocal, int64_t lda, void* stream);
double* cap_dist2d_R_ptr(cap_dist2d_plan* plan, int64_t* ld);
int cap_dist2d_get_R(cap_dist2d_plan* plan, double* out, int64_t ld, void* stream);
int cap_dist2d_info(cap_dist2d_plan* plan, void* stream, int64_t* info);
/* options: "strip" (block rows per K = strip nb update, 1 | 2; default 2 from 8 block rows on), "depth2" (bulk update split so that
 * the rows of strip t + 2 release the panel stream early; on), "occ1_m", "safe" (one communicator family, one communication stream),
 * "complete_inv" = 0 / 1 + "split": the factor call also leaves my piece of R^-1 (cap_dist2d_get_Rinv), streamed with the sweep as in
 * cap_dist_*: per block row Dinv(k) down the owner's process column, the finished block column of R^-1 along the process rows, one
 * local MFMA GEMM with my piece of the solved row - no replicated R.  "ipc" = 1: both operand moves (the solved row down the process
 * columns, the strip pieces along the process rows) as IPC peer copies on SDMA engines with two 8-byte all-reduces each as barriers,
 * like the 1 x P plan's strip exchange (cap_dist2d_get(plan, 12) tells whether the peers could be mapped).                  */
int cap_dist2d_set_option(cap_dist2d_plan* plan, const char* key, int64_t value);
/* descriptor forms: A / R / Rinv are block-cyclic descriptors (cap_desc_create_bc) of THIS plan's n, nb, Pr x Pc and position -
 * checked, CAP_ERR_ARG otherwise.  Host matrix -> pieces -> factor -> host R: cap_desc_import_host_global, cap_dist2d_factor_desc,
 * cap_dist2d_get_R_desc, cap_desc_export_host_global.                                                                            */
int cap_dist2d_factor_desc(cap_dist2d_plan* plan, const cap_desc* A, void* stream);
int cap_dist2d_get_R_desc(cap_dist2d_plan* plan, cap_desc* R, void* stream);
int cap_dist2d_get_Rinv_desc(cap_dist2d_plan* plan, cap_desc* Rinv, void* stream);
int cap_dist2d_get_Rinv(cap_dist2d_plan* plan, double* out, int64_t ld, void* stream);
double* cap_dist2d_Rinv_ptr(cap_dist2d_plan* plan, int64_t* ld);
int64_t cap_bc2d_local_extent(int64_t n, int64_t nb, int Pr, int Pc, int pr, int pc, int which);   /* which: 0 rows, 1 columns */
/* pure index helper of the update kernel's staircase enumeration: local row tiles (128 rows) of process row pr of Pr, from
 * local row block rlb0 on, whose global tile index relative to block J0 is <= X (nbT = nb / 128)                        */
int cap_bc2d_rows_le(int X, int nbT, int J0, int Pr, int pr, int rlb0);
int cap_fill_symmetric_bc2d(double* local, int64_t ld, int64_t n, int64_t nb, int Pr, int Pc, int pr, int pc,
                            int diagonally_dominant, void* stream);

/* matmult::summa::invoke, GEMM overload (summa.hpp:6-44, distribute :163-221, collect :223-253; driver
 * bench/matmult/summa_gemm.cpp:7-55): C = alpha A B + beta C on the d x d x c grid of a topo::square bundle, operands
 * are the element-cyclic local pieces (ceil(M/d) x ceil(K/d) etc., zero padded).  Layer z walks the inner process
 * indices z, z + c, ... (c == d: upstream's single step; c == 1: 2D SUMMA); row / column broadcasts run on their own
 * streams and communicators, B moves in `num_chunks` column chunks overlapped with the local MFMA GEMMs
 * (upstream's Ibcast pipelining, summa.hpp:195-215); partial products are summed over `depth` when c > 1.          */
typedef struct cap_summa_plan cap_summa_plan;
int cap_summa_plan_create(cap_summa_plan** plan, cap_topo* topo, int64_t m, int64_t n, int64_t k, int num_chunks);
int cap_summa_plan_destroy(cap_summa_plan* plan);
void cap_summa_local_dims(const cap_summa_plan* plan, int64_t* ml, int64_t* nl, int64_t* kl);
int cap_summa_dgemm(cap_summa_plan* plan, double alpha, const double* A_local, int64_t lda, const double* B_local,
                    int64_t ldb, double beta, double* C_local, int64_t ldc, void* stream);
/* util::transpose (util.hpp:232-247): MPI_Sendrecv_replace with the transpose partner (x, y, z) <-> (y, x, z) of topo::square.
 * `count` doubles of `buf` are swapped (the received piece is the partner's piece as stored, NOT transposed); tmp: device
 * scratch of `count` doubles.                                                                                            */
int cap_util_transpose(cap_topo* topo, double* buf, double* tmp, int64_t count, void* stream);
/* matmult::summa::invoke, TRMM overload (summa.hpp:46-83; call sites cholinv.hpp:114-120,148-154): in place
 * B <- alpha op(T) B (side LEFT, T m x m; plan created with (m, n, k = m)) or alpha B op(T) (RIGHT, T n x n; plan (m, n, k = n))
 * on element-cyclic pieces.  T_local = my piece of the globally upper-triangular T: rect storage (t_packed = 0, only its upper
 * triangle is referenced) or upstream's packed-upper storage (t_packed = 1: column x at x (x + 1) / 2, structure.h:39; the
 * packed image is what travels).  trans = CAP_TRANS: T_local must be the piece AFTER cap_util_transpose, exactly as upstream's
 * call site prepares it.  uplo = UPPER, diag = NONUNIT only (every upstream call site); local products run on the MFMA GEMM
 * with K ranges cut at the triangle.                                                                                     */
int cap_summa_dtrmm(cap_summa_plan* plan, int side, int uplo, int trans, int diag, double alpha, const double* T_local, int64_t ldt,
                    int t_packed, double* B_local, int64_t ldb, void* stream);
/* matmult::summa::invoke, SYRK overload (summa.hpp:85-161; call site cholinv.hpp:128-133): C <- alpha A^T A + beta C (trans =
 * CAP_TRANS, A k x n) or alpha A A^T + beta C (NoTrans, A n x k); plan created with (m = n, n = n, k).  Like upstream a GEMM of
 * A's piece with the transpose partner's piece (internal copy + cap_util_transpose, summa.hpp:91-92).  C_local: nl x nl rect
 * (c_packed = 0: the whole local square is written, as upstream does for a rect C) or packed upper (c_packed = 1).        */
int cap_summa_dsyrk(cap_summa_plan* plan, int uplo, int trans, double alpha, const double* A_local, int64_t lda, double beta,
                    double* C_local, int64_t ldc, int c_packed, void* stream);

/* qr::cacqr<...>::info + factor, 1D path - cacqr.h:18-49, cacqr.hpp:5-29,172-193,217-248.
 * A is the local row-cyclic piece (m_local x n, column-major); R (n x n) is replicated;
 * num_iter = 1 (CholeskyQR) or 2 (CholeskyQR2).  comm == NULL -> single rank.              */
typedef struct cap_cacqr_plan cap_cacqr_plan;
int cap_cacqr_plan_create(cap_cacqr_plan** plan, int64_t m_local, int64_t n, int num_iter, cap_comm* comm);
/* The 3D / tunable-grid path (cacqr.hpp:44-170: sweep_3d, sweep_tune, solve; invoke_3d :195-215) on a topo::rect
 * bundle (c x d x c; c == d is the 3D cube): A_local = ceil(M/d) x (N/c) element-cyclic piece (rows y mod d, columns
 * x mod c, replicated over the layers).  Row broadcast + Gram block + all-reduces over the process column, dense
 * Gram assembled on every rank, Cholesky factor and inverse computed redundantly per GPU, Q R^-1 as one term per
 * layer summed over `depth`.  Q_ptr is the local piece of Q, R_ptr the dense replicated R, cap_cacqr_R_piece the
 * c x c cyclic piece upstream keeps.                                                                              */
int cap_cacqr_plan_create_grid(cap_cacqr_plan** plan, int64_t m_global, int64_t n_global, int num_iter, cap_topo* topo);
int64_t cap_cacqr_local_cols(const cap_cacqr_plan* plan);
int cap_cacqr_R_piece(cap_cacqr_plan* plan, double* out, int64_t ld, void* stream);
int cap_cacqr_plan_destroy(cap_cacqr_plan* plan);
int cap_cacqr_factor(cap_cacqr_plan* plan, const double* A, int64_t lda, void* stream);
double* cap_cacqr_Q_ptr(cap_cacqr_plan* plan, int64_t* ld);
double* cap_cacqr_R_ptr(cap_cacqr_plan* plan, int64_t* ld);
int cap_cacqr_info(cap_cacqr_plan* plan, void* stream, int64_t* info);

/* ------------------------------------------------------------------------------------
 * Mixed-precision Cholesky solve (BASELINE config 5; not in the reference, whose solve path is the
 * stub trsm/diaginvert/diaginvert.hpp:7-10): factor in low precision - every O(n^3) flop on
 * v_mfma_f32_32x32x16_bf16 with fp32 accumulation and fp32 storage, the O(nb n^2) panel work in fp64 -
 * then solve A X = B to fp64 accuracy by iterative refinement (fp64 residual GEMM + blocked fp64
 * TRSMs with the promoted factor).  n must be a multiple of 128.  `A` is the same fp64 matrix in both
 * calls; solve needs it FULL symmetric.  The fp64 path (cap_cholinv_* + cap_dtrsm) is its oracle.
 * ---------------------------------------------------------------------------------- */
typedef struct cap_mpchol_plan cap_mpchol_plan;
int cap_mpchol_plan_create(cap_mpchol_plan** plan, int64_t n, int64_t nrhs_max);
int cap_mpchol_plan_destroy(cap_mpchol_plan* plan);
int cap_mpchol_factor(cap_mpchol_plan* plan, const double* A, int64_t lda, void* stream);
int cap_mpchol_info(cap_mpchol_plan* plan, void* stream, int64_t* info);
/* up to max_iter refinement sweeps until ||B - A X||_F / ||B||_F <= tol; *iters sweeps used, *relres the final value.
 * Synchronises the stream once per sweep.                                                                          */
int cap_mpchol_solve(cap_mpchol_plan* plan, const double* A, int64_t lda, const double* B, int64_t ldb, double* X,
                     int64_t ldx, int64_t nrhs, int max_iter, double tol, int* iters, double* relres, void* stream);
float* cap_mpchol_R32_ptr(cap_mpchol_plan* plan, int64_t* ld);          /* the fp32 factor (upper, n x n) */
/* Live measurement of the bf16 trailing updates (the dominant kernel, bf16_tn_kernel) of the LAST factor call, enabled with
 * cap_mpchol_set_option(plan, "profile", 1): launches, summed duration (ms, HIP events on the launch stream), summed
 * algorithmic flops (2 K per updated element) and bytes (fp32 C read + write, bf16 panel once).  Same protocol as
 * cap_cholinv_profile.  Other options: "strip" = panels (1024 rows each) contracted per bf16 update, 1 | 2 (default 2: K = 2048);
 * "split" = column-split schedule (near columns of every block-row solve / head update on the panel stream, the far ones on a
 * third stream; default 1, bit-identical to 0).                                                                           */
int cap_mpchol_set_option(cap_mpchol_plan* plan, const char* key, int64_t value);
/* The bf16 trailing update by itself (tests, tools/bf16_bench.py): C32[m x n] += alpha A^T B, A: k x m, B: k x n bf16, both
 * K-contiguous (lda / ldb in elements), fp32 atomics into C; tri = 1: square problem, elements with row <= col only.
 * variant 0 = the 128 x 128-tile kernel, 1 = the wave-specialised 256 x 128-tile kernel with the three-deep LDS ring and the tile
 * loop (m % 256 == n % 128 == k % 64 == 0, else CAP_ERR_UNSUPPORTED), -1 = the dispatcher the factorization uses (options
 * "update_kernel" 0 | 1, "update_tpw" = supertile steps per workgroup, "update_min_tiles" of cap_mpchol_set_option; process-wide).
 * tpw > 0 overrides the chunk length.                                                                                     */
int cap_bf16_update(int variant, int64_t m, int64_t n, int64_t k, float alpha, const void* A16, int64_t lda, const void* B16,
                    int64_t ldb, float* C, int64_t ldc, int tri, int tpw, void* stream);
int cap_mpchol_profile(cap_mpchol_plan* plan, int64_t* launches, double* ms_total, double* flops_total, double* bytes_total);

/* The same solve on P GPUs (BASELINE config 5: N = 131072 on 8 MI355X; csrc/dist_mixed.hip): the bf16-MFMA factorization on
 * the 1 x P block-column-cyclic layout of cap_dist_* - fp64 diagonal blocks and block-row solves, Dinv broadcast, the bf16
 * panel pieces all-gathered (a quarter of the fp64 schedule's bytes), staircase bf16 update of the local fp32 columns - and
 * fp64 iterative refinement with a distributed solve: forward / backward block substitution over the block columns (one
 * nb x nrhs broadcast / all-reduce per block) and the residual from each rank's own columns of the symmetric A.
 * n % 128 == 0; nb = block width = K of the bf16 update (power of two >= 128, 0 = 1024); Alocal = this rank's block columns
 * (n rows, cap_dmp_local_cols columns); B and X are n x nrhs and the same on every rank.                                */
typedef struct cap_dmp_plan cap_dmp_plan;
int cap_dmp_plan_create(cap_dmp_plan** plan, int64_t n, int64_t nb, int64_t nrhs_max, cap_comm* comm);
int cap_dmp_plan_destroy(cap_dmp_plan* plan);
int64_t cap_dmp_local_cols(const cap_dmp_plan* plan);
int cap_dmp_factor(cap_dmp_plan* plan, const double* Alocal, int64_t lda, void* stream);
int cap_dmp_info(cap_dmp_plan* plan, void* stream, int64_t* info);
int cap_dmp_solve(cap_dmp_plan* plan, const double* Alocal, int64_t lda, const double* B, int64_t ldb, double* X, int64_t ldx,
                  int64_t nrhs, int max_iter, double tol, int* iters, double* relres, void* stream);
float* cap_dmp_R32_ptr(cap_dmp_plan* plan, int64_t* ld);                /* my block columns of the fp32 factor, ld = padded n */

#ifdef __cplusplus
}
#endif
#endif /* CAPITAL_AMD_H_ */
