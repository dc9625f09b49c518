// Argument structs of the kernels that take one (gfx950 only).  Shared by the kernels' translation units and by the CPU kernel models of
// tests/hipshim (kernels_cpu.cpp), which decode the very bytes a launch hands to hipLaunchKernel - a mirror of its own would drift.
// Anonymous namespace: the kernels' mangled names (rocprofv3 reports, profiles/) stay what they were when the structs lived in the .hip files.
#pragma once
#include <stdint.h>
#if !defined(__HIPCC__) && !defined(__HIP__)
typedef unsigned short __bf16;      // (host-only builds: the pointer types below only need a 2-byte element)
#endif

namespace {

// ---- gemm.hip
struct GemmArgs {
  const double* A; const double* B; double* C;
  int64_t lda, ldb, ldc;
  int64_t M, N, K;
  double alpha, beta;
  int tm, tn;       // tiles in M, N
  int tri;          // element mask: 0 full, 1 upper (row<=col), 2 lower
  int etri;         // tile enumeration: 0 full grid, 1/2 triangular (square tile spaces only)
  int chunk;        // logical tiles per XCD
  int nsm, nsn;     // supertiles in M, N
  int st;           // supertile edge in tiles (tiles of one supertile are co-scheduled on one XCD)
  // full-grid products whose K range depends on the tile row (aupt / aupn) or column (bupper): supertiles of stm x stn tiles
  // (64 in all), enumerated so that every XCD's contiguous range covers ALL values of the K-determining index - a square
  // 8 x 8 walk hands XCD 0 the short-K and XCD 7 the long-K tiles of a triangular operand (measured: the tree's R^-1
  // products ran at the speed of the dense product).  sorder 1: supertile columns fastest.
  int stm, stn, sorder;
  // split-K (tall-skinny Gram / few-tile problems): blockIdx.y = K slice; partial tiles go to
  // slab kz of P (column-major, ld = M) and a second kernel reduces them deterministically
  int ksplit; int64_t kchunk; double* P; int64_t slab;
  // distributed trailing update (dist.hip): C = my block-cyclic block columns (1 x P grid, block width
  // nbT tiles), rows are global.  stair: upper mask follows the staircase  row_tile <= global tile of
  // my local column tile;  gather: operand A is the all-gathered block row, stored as P pieces in
  // (rank, local block) order - row tile ti lives in piece (J % P) at local block J / P - gstart[r].
  int aupt, aupn;   // op(A) comes from an upper-triangular A (Trans / NoTrans form): see tn_dma_tile
  int bupper;       // op(B) is upper triangular (k x n, zero for k > column): K range of a column tile stops at its diagonal
  int* ctr;         // persistent launches: 8 per-XCD slot counters (zeroed on the stream before the launch)
  int hiprio;       // panel-stream launches: raise the waves' issue priority (they share CUs with the bulk update)
  int stair, gather, sP, sp, snbT, sJ0, slb0;
  // rows of C block-cyclic over rP process rows as well (Pr x Pc layout): local row block b holds global block
  // rp + rP (rlb0 + b).  rP = 1, rp = 0, rlb0 = sJ0 is the 1 x P layout (rows global, origin at block sJ0).
  int rP, rp, rlb0;
  int64_t gpiece; int gstart[8];
  // epilogue of the beta == 1 update form as fire-and-forget fp64 atomic adds executed in L2 (global_atomic_add_f64):
  // no C read-back into registers, no load latency on the tile's critical path.  Every C element has exactly one
  // writer (no split-K on this path), so the result is the same single rounding fl(C + alpha*acc) as the load/add/store
  // form and stays run-to-run deterministic.
  int atomic_c;
  int usebuf;       // operand rows of a tile fit a 32-bit byte offset: LDS-DMA through buffer descriptors (scalar offsets)
  int skip;         // launch the block-skipping instantiation (triangular operands, few-tile SYRK)
  // beta != 0 with the C input read from another matrix (C = alpha op(A) op(B) + beta Cin): the first trailing updates of a
  // factorization read A and write R, which replaces the n x n copy in front of it.  Cin == C, ldcin == ldc otherwise.
  const double* Cin; int64_t ldcin;
};

// ---- gemm.hip
struct SmallArgs {
  const double* A; const double* B; double* C;
  int64_t lda, ldb, ldc;
  int M, N, K;
  double alpha, beta;
  int tri, hiprio;
  int64_t sa, sb, sc;   // batch strides (blockIdx.z)
};

// ---- gemm.hip
// A32 != nullptr: op(A) is read from an fp32 array of the same shape (lda in ELEMENTS) and widened in registers - the same fp64 arithmetic on half the bytes
struct SkinnyArgs { const double* A; const double* B; double* C; int64_t lda, ldb, ldc; int64_t M, K; int N; double alpha, beta; const float* A32; };

// ---- leaf.hip
struct Panel64Fold {
  double* Dnext; int64_t ldn;            // Dinv_{i+1} (strictly lower part zero-filled)
  int* info; int info_base;              // first non-positive pivot -> info_base + 1-based index
  const double* cj_src; double* cj_dst; int64_t cj_ld; int cj_cols;
  int direct;
};

// ---- leaf.hip
struct Chain64 {
  double* R; int64_t ldr; double* Ri; int64_t ldi; int nblk; int* info; int info_base; int* ctr; int fence;
  int hmax;             // the inverse is assembled in the same launch up to pairs of hmax x hmax blocks (0: not at all; <= 256)
  long long* trace;     // nullptr, or [64 workgroups][32 steps][8]: 100 MHz stamps (step start, S done, released, U done, leaf done, released)
  // recovery (round 5).  ctr[3] is the state word of the slot: 0 normal; 1 = a workgroup of the primary launch gave up waiting for its
  // peers (every workgroup that sees it stops meeting and leaves); 2 = the recovery launch has taken over; 3 = it gave up too.  The
  // recovery launch (recover = 1, two workgroups, enqueued behind every primary launch) returns at once if it finds 0; else it restores the block from `backup` (upper 64 x 64 blocks, packed
  // column by column of blocks, written by the primary launch before it touched the block), clears info's -64, counts the event in
  // fallbacks[0] and runs the same sweep.  fallbacks[1] > 0: test hook, the primary launch gives up at its first meeting.
  double* backup; int* fallbacks; int recover;
};

// ---- mixed.hip
struct BfArgs {
  const __bf16* A; const __bf16* B; float* C;
  int64_t lda, ldb, ldc;           // elements
  int64_t M, N, K;
  float alpha;
  int tri;                         // 1: only tiles / elements with row <= col (square problems)
  int tm, tn, chunk;
  // st > 0: tiles are enumerated supertile by supertile (st x st tiles, column-major inside and across; the upper triangle of
  // supertiles for square tri problems), so the 64 / 32 tiles an XCD runs at a time share 2 st panel slices through its L2
  // instead of st^2 + 1 (a plain column-major walk: every tile of a column has its own A slice); nsm = supertile rows
  int st, nsm;
  // distributed trailing update (dist_mixed.hip; the bf16 twin of GemmArgs::stair / gather in gemm.hip): C = my block-cyclic
  // block columns (1 x P grid, block width snbT tiles), rows global from block sJ0 on; upper mask along the staircase
  // row tile <= global tile of my column tile; operand A = the all-gathered bf16 block row, P pieces in (rank, local block) order
  int stair, sP, sp, snbT, sJ0, slb0;
  int64_t gpiece; int gstart[8];
};

// ---- mixed.hip
struct Bf2Args {
  const __bf16* A; const __bf16* B; float* C;
  int64_t lda, ldb, ldc;                 // elements
  int tm, tn, nk;                        // 256-row tiles, 128-column tiles, K / 64
  int tri;                               // square problem, same origin for rows and columns: tiles / elements with row <= col only
  int nsi, nsuper, tpw;                  // supertile rows (rectangular walk), supertiles in all, supertile steps per workgroup
  float alpha;
};

// ---- cqr_kernels.hip
struct GramArgs { const double* Q; int64_t ld, m, chunk; double* P; };

// ---- cqr_kernels.hip
struct ApplyArgs { const double* Qin; int64_t ldin; const double* Ri; double* Qout; int64_t ldout; int ntiles; int contig; };

}  // namespace
