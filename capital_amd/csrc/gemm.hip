// fp64 GEMM / SYRK on v_mfma_f64_16x16x4_f64 for gfx950 (MI355X).
//
// Replaces blas::engine::_gemm / _syrk (reference blas/interface.hpp:43-59, 81-97, which
// call cblas_dgemm / cblas_dsyrk).  Column-major, device pointers.
//
// Kernel shape (v1):
//   workgroup 256 threads = 4 waves (2 x 2), C tile 128 x 128, wave tile 64 x 64
//   = 4 x 4 MFMA blocks of 16 x 16 (128 accumulator registers), K tile BK = 16.
//   Operands are staged global -> VGPR (16-byte loads) -> LDS (double buffered, one
//   barrier per K tile).  The LDS image of an operand follows its GLOBAL contiguity so
//   that both the staging writes (ds_write_b128) and the fragment reads (ds_read_b64)
//   are bank-conflict free:
//     K-contiguous operand (op(A)=A^T, op(B)=B):  [outer][BK+2]   (LDK = 18)
//     M-contiguous operand (op(A)=A,   op(B)=B^T): [k][128+16]    (LDO = 144)
//   The MFMA is issued with the operands swapped (D' = B-frag x A-frag) so that a lane
//   holds C[i = lane&15][j = (lane>>4)+4r]: 16 consecutive rows of a column -> 128-byte
//   store segments in column-major C.
//   Blocks are remapped XCD-aware: block b runs on XCD b%8; each XCD walks a contiguous
//   range of supertiles (2x2 tiles; 8x8 for the NN products with a triangular operand) so its private L2 sees compact A/B panels.
#include <algorithm>
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "common.h"
#include "kargs.h"
#include "gemm_index.h"
#include "tile_dma.h"

namespace {

constexpr int BM = 128, BN = 128, BK = 16, NTHREADS = 256;
constexpr int LDK = BK + 2;       // K-contiguous LDS row stride (doubles)
constexpr int LDO = 128 + 16;     // outer-contiguous LDS row stride (doubles)
constexpr int TILE_ELEMS = 128 * LDK;  // == BK * LDO == 2304 doubles
static_assert(128 * LDK == BK * LDO, "both LDS layouts use the same footprint");
constexpr int ST = 8;             // supertile edge (tiles) of the NN products with a triangular operand
constexpr int ST_SMALL = 2;       // ... of everything else, see cap_gemm_launch

// struct GemmArgs: csrc/kargs.h (shared with the CPU kernel models of tests/hipshim)

// tile enumeration and staircase index maps (stair_*, a_tile_base, slot_to_tile): csrc/gemm_index.h - shared with the CPU kernel models
// of tests/hipshim, which walk a launch's blocks through the very same functions

// Stage one operand tile (128 outer x BK) from global into registers.
// KC: element (o, k) at P[k + o*ld];  !KC: element (o, k) at P[o + k*ld].
// EDGE: 0 = aligned interior (no checks), 1 = ragged extents but 16-byte aligned even geometry (predicated
// vector loads), 2 = anything (scalar loads).
template <bool KC, int EDGE>
__device__ __forceinline__ void load_tile(const double* __restrict__ P, int64_t ld, int64_t o0, int64_t k0,
                                          int64_t omax, int64_t kmax, d2 (&r)[4]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    int c = t + NTHREADS * s;
    if (KC) {
      int o = c >> 3, k2 = (c & 7) * 2;
      const double* p = P + (o0 + o) * ld + k0 + k2;
      if (EDGE == 0) {
        r[s] = *reinterpret_cast<const d2*>(p);
      } else if (EDGE == 1) {          // kmax even: a k pair is in or out as a whole
        r[s] = (o0 + o < omax && k0 + k2 < kmax) ? *reinterpret_cast<const d2*>(p) : (d2){0.0, 0.0};
      } else {
        double v0 = 0, v1 = 0;
        if (o0 + o < omax) {
          if (k0 + k2 < kmax) v0 = p[0];
          if (k0 + k2 + 1 < kmax) v1 = p[1];
        }
        r[s] = (d2){v0, v1};
      }
    } else {
      int k = c >> 6, o2 = (c & 63) * 2;
      const double* p = P + (k0 + k) * ld + o0 + o2;
      if (EDGE == 0) {
        r[s] = *reinterpret_cast<const d2*>(p);
      } else if (EDGE == 1) {          // omax even: an outer pair is in or out as a whole
        r[s] = (k0 + k < kmax && o0 + o2 < omax) ? *reinterpret_cast<const d2*>(p) : (d2){0.0, 0.0};
      } else {
        double v0 = 0, v1 = 0;
        if (k0 + k < kmax) {
          if (o0 + o2 < omax) v0 = p[0];
          if (o0 + o2 + 1 < omax) v1 = p[1];
        }
        r[s] = (d2){v0, v1};
      }
    }
  }
}

template <bool KC>
__device__ __forceinline__ void store_tile(double* __restrict__ lds, const d2 (&r)[4]) {
  const int t = threadIdx.x;
#pragma unroll
  for (int s = 0; s < 4; s++) {
    int c = t + NTHREADS * s;
    int off = KC ? ((c >> 3) * LDK + (c & 7) * 2) : ((c >> 6) * LDO + (c & 63) * 2);
    *reinterpret_cast<d2*>(lds + off) = r[s];
  }
}

// TAG only changes the kernel's NAME (same code): TAG 1 = the trailing update of the blocked
// Cholesky, so rocprofv3 --stats reports the dominant kernel separately from panel-sized launches.
template <bool A_KC, bool B_KC, int EDGE, int TAG>
__global__ void __launch_bounds__(NTHREADS, 2) dgemm_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // block -> XCD-aware logical slot.  Under split-K every K slice would put the same few tiles on the same
  // XCDs (block b runs on XCD b % 8 for every blockIdx.y when gridDim.x % 8 == 0): rotate by the slice index.
  const int b = (int)((blockIdx.x + blockIdx.y) % gridDim.x);
  const int L = (b & 7) * g.chunk + (b >> 3);
  int ti, tj;
  if ((b >> 3) >= g.chunk || !slot_to_tile(g, L, ti, tj)) return;
  if (g.hiprio) __builtin_amdgcn_s_setprio(3);

  const int64_t i0 = (int64_t)ti * BM, j0 = (int64_t)tj * BN;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wi = (wid & 1) * 64, wj = (wid >> 1) * 64;
  const int lr = lane & 15, kg = lane >> 4;

  // LDS carve: [A buf0][B buf0][A buf1][B buf1]
  auto sA = [&](int buf) -> double* { return smem + buf * 2 * TILE_ELEMS; };
  auto sB = [&](int buf) -> double* { return smem + buf * 2 * TILE_ELEMS + TILE_ELEMS; };

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  const int kz = blockIdx.y;
  const int64_t kbeg = (int64_t)kz * g.kchunk;
  const int64_t kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
  const int nk = (int)((kend - kbeg + BK - 1) / BK);
  d2 ra[4], rb[4];
  if (nk > 0) {
    load_tile<A_KC, EDGE>(g.A, g.lda, i0, kbeg, g.M, kend, ra);
    load_tile<B_KC, EDGE>(g.B, g.ldb, j0, kbeg, g.N, kend, rb);
    store_tile<A_KC>(sA(0), ra);
    store_tile<B_KC>(sB(0), rb);
  }
  __syncthreads();

  // per-lane fragment base offsets inside an LDS tile
  const int a_base = A_KC ? ((wi + lr) * LDK + kg) : (kg * LDO + wi + lr);
  const int b_base = B_KC ? ((wj + lr) * LDK + kg) : (kg * LDO + wj + lr);
  constexpr int A_SI = A_KC ? 16 * LDK : 16;      // +16 outer
  constexpr int A_SK = A_KC ? 4 : 4 * LDO;        // +4 in k
  constexpr int B_SI = B_KC ? 16 * LDK : 16;
  constexpr int B_SK = B_KC ? 4 : 4 * LDO;

  for (int kt = 0; kt < nk; kt++) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      load_tile<A_KC, EDGE>(g.A, g.lda, i0, kbeg + (int64_t)(kt + 1) * BK, g.M, kend, ra);
      load_tile<B_KC, EDGE>(g.B, g.ldb, j0, kbeg + (int64_t)(kt + 1) * BK, g.N, kend, rb);
    }
    const double* pa = sA(cur) + a_base;
    const double* pb = sB(cur) + b_base;
#pragma unroll
    for (int ks = 0; ks < BK / 4; ks++) {
      double fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; i++) fa[i] = pa[i * A_SI + ks * A_SK];
#pragma unroll
      for (int j = 0; j < 4; j++) fb[j] = pb[j * B_SI + ks * B_SK];
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) {
      store_tile<A_KC>(sA(cur ^ 1), ra);
      store_tile<B_KC>(sB(cur ^ 1), rb);
    }
    __syncthreads();
  }

  // epilogue: lane holds C[i0+wi+16i+lr][j0+wj+16j+kg+4r]
  const double alpha = g.alpha, beta = g.beta;
#pragma unroll
  for (int i = 0; i < 4; i++) {
    const int64_t row = i0 + wi + 16 * i + lr;
#pragma unroll
    for (int j = 0; j < 4; j++) {
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int64_t col = j0 + wj + 16 * j + kg + 4 * r;
        bool ok = true;
        if (EDGE != 0) ok = (row < g.M) && (col < g.N);
        if (g.tri == 1) ok = ok && (row <= col);
        if (g.tri == 2) ok = ok && (row >= col);
        if (ok) {
          double v = alpha * acc[i][j][r];
          if (g.ksplit > 1) {
            g.P[(int64_t)kz * g.slab + row + col * g.M] = v;
          } else {
            double* pc = g.C + row + col * g.ldc;
            if (beta != 0.0) v += beta * (*pc);
            *pc = v;
          }
        }
      }
    }
  }
}


// ---------------------------------------------------------------------------------------------
// v2 hot kernel: TN (both operands K-contiguous), aligned shapes.  Differences from the generic
// kernel above:
//  * operand tiles go HBM/L2 -> LDS directly (global_load_lds_dwordx4, 1 KiB per wave-instruction):
//    no staging VGPRs, no ds_write pass, no vmcnt->ds_write dependency in the loop;
//  * the LDS image is the lane-linear [row][8 x 16 B] tile, bank-conflict free through an XOR
//    swizzle applied to the per-lane SOURCE address and to the fragment read (chunk ^= (row>>1)&7);
//  * fragments are read with ds_read_b128: a lane gets k = 2kg, 2kg+1 of an 8-deep half tile; the
//    MFMA pair {x, y} then contracts k = {0,2,4,6} and {1,3,5,7} - the same permutation on both
//    operands, so the sum over k is unchanged;
//  * fragment registers are double buffered per half tile and the single barrier per K tile sits
//    in the middle of the tile, so LDS latency, DMA landing and barrier skew all hide behind 32 MFMAs.
// ---------------------------------------------------------------------------------------------
constexpr int DMA_TILE = 128 * BK;   // doubles per operand tile (16 KiB), unpadded

__device__ __forceinline__ void dma_tile(const double* __restrict__ P, int64_t ld, int64_t o0, int64_t k0, double* lds_tile) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int rsub = lane >> 3, p = lane & 7;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int r0 = (wid * 4 + q) * 8;
    const int row = r0 + rsub;
    const int c = p ^ ((row >> 1) & 7);
    const double* src = P + (o0 + row) * ld + k0 + c * 2;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + r0 * BK), 16, 0, 0);
  }
}

// M-contiguous A operand (op(A) = A, column-major m x k): a K tile is 16 rows (k) of 128 consecutive doubles.
// One wave-instruction moves one k row (1 KiB); the LDS image is [k][128], with the two 128-byte halves of a
// row swapped when (k >> 1) is odd (source-side XOR) so that the four k rows a fragment read touches
// alternate between the two halves of the 64-bank row: ds_read_b64 stays conflict free.
__device__ __forceinline__ void dma_tile_mc(const double* __restrict__ P, int64_t ld, int64_t i0, int64_t k0, double* lds_tile) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int kr = wid * 4 + q;
    const int c = lane ^ (((kr >> 1) & 1) << 3);
    const double* src = P + (k0 + kr) * ld + i0 + c * 2;
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(lds_tile + kr * 128), 16, 0, 0);
  }
}

// DIAG (timing experiments only, results are wrong): 1 = no DMA inside the loop, 2 = no DMA and no barrier,
// 3 = no DMA, no barrier, no fragment reads (pure MFMA stream with the kernel's register pattern)
// SKIP: operands / results with triangular structure - the 16 x 16 blocks of a wave that only see zeros (or lie below
// the diagonal of a SYRK tile) are left out of the MFMA stream with wave-uniform branches.  Separate instantiation:
// the branches would break the single-basic-block software pipeline of the bulk update.
template <bool A_MC, int DIAG, bool BUF, bool SKIP>
__device__ __forceinline__ void tn_dma_tile(const GemmArgs& g, const int ti, const int tj, const int kz, double* smem) {
  const int64_t i0 = (int64_t)ti * BM, j0 = (int64_t)tj * BN;
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int wi = (wid & 1) * 64, wj = (wid >> 1) * 64;
  const int lr = lane & 15, kg = lane >> 4;

  // LDS carve: [A0][B0][A1][B1], 16 KiB each
  auto sA = [&](int buf) -> double* { return smem + buf * 2 * DMA_TILE; };
  auto sB = [&](int buf) -> double* { return smem + buf * 2 * DMA_TILE + DMA_TILE; };

  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  int64_t kbeg = (int64_t)kz * g.kchunk;
  int64_t kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
  // triangular operands stored as full squares with explicit zeros: skip the K range that only multiplies zeros
  if (g.bupper && kend > j0 + BN) kend = j0 + BN;      // op(B)[k][j] = 0 for k > j   (B upper, NoTrans)
  if (g.aupt && kend > i0 + BM) kend = i0 + BM;        // op(A)[i][k] = A[k][i] = 0 for k > i   (A upper, Trans)
  if (g.aupn && kbeg < i0) kbeg = i0;                  // op(A)[i][k] = A[i][k] = 0 for k < i   (A upper, NoTrans)
  const int nk = kend > kbeg ? (int)((kend - kbeg) / BK) : 0;

  // per-lane fragment offsets (doubles) for the two half tiles; +16 rows = +16*BK doubles per block
  const int sw = lr >> 1;
  const int a_off0 = (wi + lr) * BK + ((0 + kg) ^ sw) * 2, a_off1 = (wi + lr) * BK + ((4 + kg) ^ sw) * 2;
  const int b_off0 = (wj + lr) * BK + ((0 + kg) ^ sw) * 2, b_off1 = (wj + lr) * BK + ((4 + kg) ^ sw) * 2;

  d2 fa0[4], fb0[4], fa1[4], fb1[4];
  // M-contiguous A: x/y steps take k = 8h + 2kg and 8h + 2kg + 1 (the k permutation of the B operand's b128 read)
  const int amc_flip = ((kg & 1) << 4);                 // rows with (k >> 1) odd are stored half-swapped
  auto read_frags = [&](const double* tA, const double* tB, int aoff, int boff, d2 (&fa)[4], d2 (&fb)[4]) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      if (A_MC) {
        const int h8 = (aoff == a_off0) ? 0 : 8;       // a_off0 / a_off1 select the half tile
        const int m = (wi + 16 * i + lr) ^ amc_flip;
        fa[i] = (d2){tA[(h8 + 2 * kg) * 128 + m], tA[(h8 + 2 * kg + 1) * 128 + m]};
      } else {
        fa[i] = *reinterpret_cast<const d2*>(tA + aoff + i * 16 * BK);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; j++) fb[j] = *reinterpret_cast<const d2*>(tB + boff + j * 16 * BK);
  };
  // SKIP bookkeeping (wave-uniform): origin of the wave's blocks relative to the tile and the structure flags
  const int wid_u = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi_u = (wid_u & 1) * 64, wj_u = (wid_u >> 1) * 64;
  const bool dtri = SKIP && g.tri == 1 && !g.stair && ti == tj;        // diagonal tile of an upper SYRK: blocks below the diagonal are dead
  auto mma32 = [&](const d2 (&fa)[4], const d2 (&fb)[4], int64_t kb) {    // kb: first k of this 8-deep half tile
    if (!SKIP) {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j].x, fa[i].x, acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j].y, fa[i].y, acc[i][j], 0, 0, 0);
    } else {
      // block column j holds columns c0 = j0 + wj + 16 j ...; block row i holds rows r0 = i0 + wi + 16 i ...
      int jlo = 0, ilo = 0, ihi = 3;
      if (g.bupper) { const int64_t d = kb - j0 - wj_u; jlo = d > 0 ? (int)(d >> 4) : 0; }          // op(B)[k][c] = 0 for k > c
      if (g.aupt) { const int64_t d = kb - i0 - wi_u; ilo = d > 0 ? (int)(d >> 4) : 0; }            // op(A)[r][k] = 0 for k > r
      if (g.aupn) { const int64_t d = kb + 7 - i0 - wi_u; ihi = d < 0 ? -1 : (d >> 4) > 3 ? 3 : (int)(d >> 4); }   // op(A)[r][k] = 0 for k < r
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 4; j++) {
          const bool on = j >= jlo && i >= ilo && i <= ihi && !(dtri && wi_u + 16 * i > wj_u + 16 * j);
          if (on) {
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j].x, fa[i].x, acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j].y, fa[i].y, acc[i][j], 0, 0, 0);
          }
        }
    }
  };

  // Software pipeline, rotated so that every fragment set is consumed in the basic block that issued
  // its reads or right behind a barrier (hipcc's waitcnt pass falls back to lgkmcnt(0) on loop-carried
  // LDS reads otherwise):
  //   prologue : DMA t0 | sync | read F0(t0,h0) | DMA t1 | read F1(t0,h1) | MMA(F0)
  //   loop kt  : sync | read F0(t(kt+1),h0) | MMA(F1) | DMA t(kt+2) | read F1(t(kt+1),h1) | MMA(F0)
  //   tail     : MMA(F1)
  // The barrier at the top of iteration kt proves: all reads of tile kt are done (buffer kt&1 may be
  // refilled) and tile kt+1 has landed (each wave drained its own DMA share with vmcnt(0) before it).
  const double* Abase = a_tile_base(g, ti);   // rows of this tile start at Abase (row offset 0 below)
  // buffer-addressed DMA (scalar offsets) whenever the tile's rows fit a 31-bit byte offset; else 64-bit global addresses
  constexpr bool ubuf = BUF;
  const int wid_s = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  DmaBuf dA, dB;
  if (BUF && !A_MC) dA = dma_buf_make(Abase + kbeg, g.lda);
  if (BUF) dB = dma_buf_make(g.B + j0 * g.ldb + kbeg, g.ldb);
  // half = 0: first part of the K tile's pieces (issued in phase A), 1: second part (phase B), 2: everything
  auto dma_a = [&](int kt_, double* dst, int half) {
    if (A_MC) { if (half != 1) dma_tile_mc(g.A, g.lda, i0, kbeg + (int64_t)kt_ * BK, dst); }
    else if (ubuf) { if (half != 1) dma_tile_buf<0, 4>(dA, wid_s, (uint32_t)kt_ * BK * 8, dst); }
    else { if (half != 1) dma_tile(Abase, g.lda, 0, kbeg + (int64_t)kt_ * BK, dst); }
  };
  auto dma_b = [&](int kt_, double* dst, int half) {
    if (ubuf) { if (half != 0) dma_tile_buf<0, 4>(dB, wid_s, (uint32_t)kt_ * BK * 8, dst); }
    else { if (half != 0) dma_tile(g.B, g.ldb, j0, kbeg + (int64_t)kt_ * BK, dst); }
  };
  if (nk > 0) {
    dma_a(0, sA(0), 2);
    dma_b(0, sB(0), 2);
    __syncthreads();
    read_frags(sA(0), sB(0), a_off0, b_off0, fa0, fb0);
    {
      const int k1 = nk > 1 ? 1 : 0;
      dma_a(k1, sA(1), 2);
      dma_b(k1, sB(1), 2);
    }
    read_frags(sA(0), sB(0), a_off1, b_off1, fa1, fb1);
    __builtin_amdgcn_sched_barrier(0);
    mma32(fa0, fb0, kbeg);
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt + 1 < nk; kt++) {
      const int nxt = (kt + 1) & 1;
      __builtin_amdgcn_s_waitcnt(0xc07f);                // F1 (issued during the previous MFMA block) is complete
      if (DIAG < 2) __syncthreads();
      // ---- phase A: refill the buffer tile kt just vacated, fetch F0 of tile kt+1, contract 2nd half of tile kt.
      // The DMA pieces and LDS reads are interleaved INTO the MFMA stream (one of each per 4 MFMAs) so that
      // a single wave keeps the matrix pipe busy while it issues them (sched_group_barrier pipeline).
      const int kn = (kt + 2 < nk) ? kt + 2 : nk - 1;    // clamp: the last refill is redundant but branch-free
      if (DIAG == 0) dma_a(kn, sA(nxt ^ 1), 0);          // A operand's pieces ride in phase A, B's in phase B
      if (DIAG < 3) read_frags(sA(nxt), sB(nxt), a_off0, b_off0, fa0, fb0);
      mma32(fa1, fb1, kbeg + (int64_t)kt * BK + 8);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);   // 4 MFMA
        if (q & 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);   // 1 VMEM (LDS-DMA piece) per 8 MFMAs
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // 1 DS read
      }
      __builtin_amdgcn_sched_barrier(0);
      // ---- phase B: fetch F1 of tile kt+1, contract its first half
      __builtin_amdgcn_s_waitcnt(0xc07f);                // F0 has landed (last read issued >= 4 MFMAs ago)
      if (DIAG == 0) { dma_a(kn, sA(nxt ^ 1), 1); dma_b(kn, sB(nxt ^ 1), 2); }
      if (DIAG < 3) read_frags(sA(nxt), sB(nxt), a_off1, b_off1, fa1, fb1);
      mma32(fa0, fb0, kbeg + (int64_t)(kt + 1) * BK);
#pragma unroll
      for (int q = 0; q < 8; q++) {
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        if (q & 1) __builtin_amdgcn_sched_group_barrier(0x010, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    mma32(fa1, fb1, kbeg + (int64_t)(nk - 1) * BK + 8);
  }

  // epilogue: lane holds C[i0+wi+16i+lr][j0+wj+16j+kg+4r]; all branches are block-uniform
  const double alpha = g.alpha, beta = g.beta;
  const bool diag_tile = g.stair ? (stair_gti(g, ti) == stair_gtj(g, tj)) : ((g.tri != 0) && (ti == tj));
  auto epilogue = [&](auto masked, auto with_beta, auto split) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int64_t row = i0 + wi + 16 * i + lr;
      double cin[4][4];
      if (with_beta.value && !split.value) {
#pragma unroll
        for (int j = 0; j < 4; j++)
#pragma unroll
          for (int r = 0; r < 4; r++) {
            const int64_t col = j0 + wj + 16 * j + kg + 4 * r;
            // diagonal tiles: compare WITHIN-tile offsets (under the staircase view rows and columns have different origins)
            bool ok = !masked.value || (g.tri == 1 ? (row - i0) <= (col - j0) : (row - i0) >= (col - j0));
            // (the NN instantiations sit at 252 VGPRs: they never get a separate C input and must not carry it)
            cin[j][r] = ok ? (A_MC ? g.C[row + col * g.ldc] : g.Cin[row + col * g.ldcin]) : 0.0;
          }
      }
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int64_t col = j0 + wj + 16 * j + kg + 4 * r;
          bool ok = !masked.value || (g.tri == 1 ? (row - i0) <= (col - j0) : (row - i0) >= (col - j0));
          double v = alpha * acc[i][j][r];
          if (split.value) {
            if (ok) g.P[(int64_t)kz * g.slab + row + col * g.M] = v;
          } else {
            if (with_beta.value) v += beta * cin[j][r];
            if (ok) g.C[row + col * g.ldc] = v;
          }
        }
    }
  };
  auto epilogue_atomic = [&](auto masked) {
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int64_t row = i0 + wi + 16 * i + lr;
#pragma unroll
      for (int j = 0; j < 4; j++)
#pragma unroll
        for (int r = 0; r < 4; r++) {
          const int64_t col = j0 + wj + 16 * j + kg + 4 * r;
          bool ok = !masked.value || (g.tri == 1 ? (row - i0) <= (col - j0) : (row - i0) >= (col - j0));
          if (ok) __builtin_amdgcn_global_atomic_fadd_f64((__attribute__((address_space(1))) double*)(g.C + row + col * g.ldc), alpha * acc[i][j][r]);
        }
    }
  };
  using T = std::true_type; using F = std::false_type;
  if (g.ksplit > 1) { if (diag_tile) epilogue(T{}, F{}, T{}); else epilogue(F{}, F{}, T{}); }
  else if (g.atomic_c) { if (diag_tile) epilogue_atomic(T{}); else epilogue_atomic(F{}); }
  else if (beta != 0.0) { if (diag_tile) epilogue(T{}, T{}, F{}); else epilogue(F{}, T{}, F{}); }
  else { if (diag_tile) epilogue(T{}, F{}, F{}); else epilogue(F{}, F{}, F{}); }
}

// One C tile per workgroup.  (A persistent variant that pulled tiles from per-XCD atomic queues was measured in round 1:
// +2.4 % on the bulk kernel alone, -10 % end to end because the latency-bound panel chain starves behind workgroups
// that never retire; it also needed > 256 VGPRs.  It was removed.)
template <int TAG, bool A_MC = false, int DIAG = 0, bool BUF = false, bool SKIP = false>
__global__ void __launch_bounds__(NTHREADS, 2) dgemm_tn_dma_kernel(const GemmArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int b = (int)((blockIdx.x + blockIdx.y) % gridDim.x);   // split-K: rotate tiles over the XCDs (see dgemm_kernel)
  const int L = (b & 7) * g.chunk + (b >> 3);
  int ti, tj;
  if ((b >> 3) >= g.chunk || !slot_to_tile(g, L, ti, tj)) return;
  if (g.hiprio) __builtin_amdgcn_s_setprio(3);
  tn_dma_tile<A_MC, DIAG, BUF, SKIP>(g, ti, tj, (int)blockIdx.y, smem);
}

int launch_nn_dma(const GemmArgs& g, int grid, hipStream_t stream) {
  size_t lds = 4 * DMA_TILE * sizeof(double);
  if (g.skip) hipLaunchKernelGGL((dgemm_tn_dma_kernel<0, true, 0, false, true>), dim3(grid, g.ksplit), dim3(NTHREADS), lds, stream, g);
  else hipLaunchKernelGGL((dgemm_tn_dma_kernel<0, true>), dim3(grid, g.ksplit), dim3(NTHREADS), lds, stream, g);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

template <int TAG>
int launch_tn_dma(GemmArgs g, int grid, hipStream_t stream, int persist_wgs = 0) {
  // persist_wgs == -1 (the only value still honoured): ask for 96 KiB of LDS so that only ONE workgroup of this launch fits a CU AND a
  // workgroup of the one-launch diagonal-block chain (68.5 KiB) does NOT fit next to it.  Used for the bulk updates of the chain-bound
  // tail (cholinv.hip, option occ1_m).  Round 5 measured why this works (profiles/r05_chain_fp64_occ1_lds{96,88}.log): the chain only
  // reaches its isolated speed (0.31 ms per 512-block) on CUs it has to itself - next to one bulk workgroup every fp64 VALU operation of
  // its leaf waits for the fp64 matrix pipe (1.9 ms per block at 88 KiB, where the two co-reside; N = 32768: 61.7 TF against 63.4).
  static const size_t occ1_lds = (size_t)(CAP_ENV("CAP_OCC1_LDS_KB") ? atoi(CAP_ENV("CAP_OCC1_LDS_KB")) : 96) * 1024;
  size_t lds = persist_wgs == -1 ? std::max<size_t>(occ1_lds, 4 * DMA_TILE * sizeof(double)) : 4 * DMA_TILE * sizeof(double);
  if constexpr (CAP_EXPERIMENTS && TAG == 0) {                       // timing experiments only (results are wrong)
    static const int diag_env = CAP_ENV("CAP_DIAG") ? atoi(CAP_ENV("CAP_DIAG")) : 0;
    if (diag_env == 1) {
      hipLaunchKernelGGL((dgemm_tn_dma_kernel<0, false, 1, true>), dim3(grid, g.ksplit), dim3(NTHREADS), lds, stream, g);
      CAP_HIP(hipGetLastError());
      return CAP_OK;
    }
  }
  if (g.skip && g.usebuf) hipLaunchKernelGGL((dgemm_tn_dma_kernel<0, false, 0, true, true>), dim3(grid, g.ksplit), dim3(NTHREADS), lds, stream, g);
  else if (g.skip) hipLaunchKernelGGL((dgemm_tn_dma_kernel<0, false, 0, false, true>), dim3(grid, g.ksplit), dim3(NTHREADS), lds, stream, g);
  else if (g.usebuf) hipLaunchKernelGGL((dgemm_tn_dma_kernel<TAG, false, 0, true>), dim3(grid, g.ksplit), dim3(NTHREADS), lds, stream, g);
  else hipLaunchKernelGGL((dgemm_tn_dma_kernel<TAG, false, 0, false>), dim3(grid, g.ksplit), dim3(NTHREADS), lds, stream, g);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

template <bool A_KC, bool B_KC, int TAG = 0>
int launch_variant(const GemmArgs& g, int edge, int grid, hipStream_t stream) {
  size_t lds = 4 * TILE_ELEMS * sizeof(double);
  if (edge == 2)
    hipLaunchKernelGGL((dgemm_kernel<A_KC, B_KC, 2, TAG>), dim3(grid, g.ksplit), dim3(NTHREADS), lds, stream, g);
  else if (edge == 1)
    hipLaunchKernelGGL((dgemm_kernel<A_KC, B_KC, 1, TAG>), dim3(grid, g.ksplit), dim3(NTHREADS), lds, stream, g);
  else
    hipLaunchKernelGGL((dgemm_kernel<A_KC, B_KC, 0, TAG>), dim3(grid, g.ksplit), dim3(NTHREADS), lds, stream, g);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}


// ---------------------------------------------------------------------------------------------
// Small-problem kernel (the diagonal-block recursion issues dozens of 64...256-sized products that are
// pure latency): 64 x 64 C tile per 256-thread workgroup, 4 waves x (2 x 2 MFMA blocks), K chunks of 32
// staged through LDS as [k][64 + 16] (fragment reads: 16 consecutive doubles per k row, the +16 pad puts
// the second k row of a half-wave on the other 32 banks).  Scalar predicated loads: any shape / alignment.
// ---------------------------------------------------------------------------------------------
constexpr int SM_T = 64, SM_K = 64, SM_LD = SM_T + 16;
constexpr int SM_PER_THREAD = (SM_T * SM_K) / 256;   // elements of each operand chunk staged per thread

// struct SmallArgs: csrc/kargs.h (shared with the CPU kernel models of tests/hipshim)

template <bool TA, bool TB>
__global__ void __launch_bounds__(256) dgemm_small_kernel(SmallArgs g) {
  extern __shared__ __attribute__((aligned(16))) double sm_lds[];
  g.A += (int64_t)blockIdx.z * g.sa; g.B += (int64_t)blockIdx.z * g.sb; g.C += (int64_t)blockIdx.z * g.sc;
  double* As = sm_lds;
  double* Bs = sm_lds + SM_K * SM_LD;
  const int ti = blockIdx.x, tj = blockIdx.y;
  if (g.tri == 1 && ti > tj) return;
  if (g.tri == 2 && ti < tj) return;
  if (g.hiprio) __builtin_amdgcn_s_setprio(3);
  const int t = threadIdx.x, lane = t & 63, wid = t >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  const int wi = (wid & 1) * 32, wj = (wid >> 1) * 32;
  const int i0 = ti * SM_T, j0 = tj * SM_T;
  d4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++) acc[i][j] = (d4){0.0, 0.0, 0.0, 0.0};

  // global -> registers for one K chunk (threads walk the contiguous global axis; all loads of a chunk are
  // issued back to back, and the NEXT chunk's loads fly while the current one is contracted)
  double ra[SM_PER_THREAD], rb[SM_PER_THREAD];
  auto fetch = [&](int k0) {
#pragma unroll
    for (int s = 0; s < SM_PER_THREAD; s++) {
      const int e = t + 256 * s;
      int m, k;
      if (TA) { k = e % SM_K; m = e / SM_K; } else { m = e % SM_T; k = e / SM_T; }
      ra[s] = (i0 + m < g.M && k0 + k < g.K) ? (TA ? g.A[(int64_t)(i0 + m) * g.lda + k0 + k] : g.A[(int64_t)(k0 + k) * g.lda + i0 + m]) : 0.0;
      int n, kb;
      if (TB) { n = e % SM_T; kb = e / SM_T; } else { kb = e % SM_K; n = e / SM_K; }
      rb[s] = (j0 + n < g.N && k0 + kb < g.K) ? (TB ? g.B[(int64_t)(k0 + kb) * g.ldb + j0 + n] : g.B[(int64_t)(j0 + n) * g.ldb + k0 + kb]) : 0.0;
    }
  };
  auto stash = [&]() {
#pragma unroll
    for (int s = 0; s < SM_PER_THREAD; s++) {
      const int e = t + 256 * s;
      int m, k;
      if (TA) { k = e % SM_K; m = e / SM_K; } else { m = e % SM_T; k = e / SM_T; }
      As[k * SM_LD + m] = ra[s];
      int n, kb;
      if (TB) { n = e % SM_T; kb = e / SM_T; } else { kb = e % SM_K; n = e / SM_K; }
      Bs[kb * SM_LD + n] = rb[s];
    }
  };

  fetch(0);
  for (int k0 = 0; k0 < g.K; k0 += SM_K) {
    stash();
    __syncthreads();
    if (k0 + SM_K < g.K) fetch(k0 + SM_K);
#pragma unroll
    for (int ks = 0; ks < SM_K / 4; ks++) {
      double fa[2], fb[2];
#pragma unroll
      for (int i = 0; i < 2; i++) fa[i] = As[(ks * 4 + kg) * SM_LD + wi + 16 * i + lr];
#pragma unroll
      for (int j = 0; j < 2; j++) fb[j] = Bs[(ks * 4 + kg) * SM_LD + wj + 16 * j + lr];
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j], fa[i], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }
  // lane holds C[i0+wi+16i+lr][j0+wj+16j+kg+4r]
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int row = i0 + wi + 16 * i + lr;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int r = 0; r < 4; r++) {
        const int col = j0 + wj + 16 * j + kg + 4 * r;
        bool ok = row < g.M && col < g.N;
        if (g.tri == 1) ok = ok && row <= col;
        if (g.tri == 2) ok = ok && row >= col;
        if (ok) {
          double* pc = g.C + row + (int64_t)col * g.ldc;
          double v = g.alpha * acc[i][j][r];
          if (g.beta != 0.0) v += g.beta * (*pc);
          *pc = v;
        }
      }
  }
}

// access notes (common.h) of C = alpha op(A) op(B) + beta C [element mask tri; C input from Cin when given]
void note_product(int transa, int transb, int64_t m, int64_t n, int64_t k, double alpha, const double* A, int64_t lda, const double* B,
                  int64_t ldb, double beta, double* C, int64_t ldc, int tri, const double* Cin = nullptr, int64_t ldcin = 0) {
  if (!cap_acc_on()) return;
  if (alpha != 0.0 && k > 0) {
    if (transa == CAP_TRANS) cap_acc_r(A, lda, k, m); else cap_acc_r(A, lda, m, k);
    if (transb == CAP_TRANS) cap_acc_r(B, ldb, n, k); else cap_acc_r(B, ldb, k, n);
  }
  if (Cin && Cin != C) { cap_acc_r(Cin, ldcin, m, n, tri); cap_acc_w(C, ldc, m, n, tri); }
  else cap_acc(beta != 0.0 ? CAP_ACC_RW : CAP_ACC_W, C, ldc, m, n, tri);
}

int launch_small(int transa, int transb, int64_t m, int64_t n, int64_t k, double alpha, const double* A, int64_t lda,
                 const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int tri, int hiprio, hipStream_t stream,
                 int nbatch = 1, int64_t sa = 0, int64_t sb = 0, int64_t sc = 0) {
  SmallArgs g{A, B, C, lda, ldb, ldc, (int)m, (int)n, (int)k, alpha, beta, tri, hiprio, sa, sb, sc};
  if (cap_acc_on())
    for (int z = 0; z < nbatch; z++) note_product(transa, transb, m, n, k, alpha, A + z * sa, lda, B + z * sb, ldb, beta, C + z * sc, ldc, tri);
  dim3 grid((unsigned)cap_ceil_div(m, SM_T), (unsigned)cap_ceil_div(n, SM_T), (unsigned)nbatch);
  const bool ta = transa == CAP_TRANS, tb = transb == CAP_TRANS;
  const size_t lds = 2 * SM_K * SM_LD * sizeof(double);
  if (ta && !tb) hipLaunchKernelGGL((dgemm_small_kernel<true, false>), grid, dim3(256), lds, stream, g);
  else if (!ta && !tb) hipLaunchKernelGGL((dgemm_small_kernel<false, false>), grid, dim3(256), lds, stream, g);
  else if (ta && tb) hipLaunchKernelGGL((dgemm_small_kernel<true, true>), grid, dim3(256), lds, stream, g);
  else hipLaunchKernelGGL((dgemm_small_kernel<false, true>), grid, dim3(256), lds, stream, g);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

// C = beta*C (alpha == 0 or k == 0), honouring the triangle mask
// ---------------------------------------------------------------------------------------------
// Skinny products: C[M x N] = alpha op(A) B + beta C with N <= 8 right-hand sides (the refinement sweeps of the mixed-precision
// solve, TRSM / TRMM with a few columns).  The 128 x 128 tile kernels would pad N to 128 and run 16 x the flops: a block step
// of the triangular solve became MFMA- and latency-bound although the operation streams op(A) exactly once.  Both kernels read
// A once, coalesced, with eight independent loads in flight per lane, and are bound by HBM.
//   TN (A: K x M, K-contiguous): one wave per 16 output rows on v_mfma_f64_16x16x4_f64 - the MFMA is the 64-lane reduction: lane
//      (lr, kg) feeds column lr of A and right-hand side lr at k = k0 + 2 kg, + 1 (the k permutation of the tile kernels);
//   NN (A: M x K, M-contiguous): one lane per output row, 8 accumulators, B[k, :] is wave-uniform; the four waves of a workgroup
//      split K and meet in LDS.
// ---------------------------------------------------------------------------------------------
// struct SkinnyArgs: csrc/kargs.h (shared with the CPU kernel models of tests/hipshim)

typedef float f2 __attribute__((ext_vector_type(2)));
template <bool F32>
__global__ void __launch_bounds__(256) dgemm_tn_skinny_kernel(const SkinnyArgs g) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int lr = lane & 15, kg = lane >> 4;
  const int64_t j0 = ((int64_t)blockIdx.x * 4 + wid) * 16;
  if (j0 >= g.M) return;
  const double* __restrict__ pa = g.A + (j0 + lr) * g.lda + 2 * kg;
  const float* __restrict__ pa32 = g.A32 + (j0 + lr) * g.lda + 2 * kg;          // (F32 only)
  auto lda2 = [&](int64_t k) -> d2 {                                          // op(A)[k, k + 1] of my column, widened when it is stored in fp32
    if (F32) { const f2 v = *reinterpret_cast<const f2*>(pa32 + k); return (d2){(double)v.x, (double)v.y}; }
    return *reinterpret_cast<const d2*>(pa + k);
  };
  const double* __restrict__ pb = g.B + (int64_t)(lr < g.N ? lr : 0) * g.ldb + 2 * kg;
  const bool bon = lr < g.N;
  // Blocked, compensated summation: every 64 k start from zero accumulators (chains of 8 MFMAs) and are added to the running
  // sum with Kahan's correction.  A plain chain of K / 4 MFMA accumulations carries a rounding error of ~sqrt(K / 4) ulp of the
  // partial sums - 5e-15 of ||b|| in the residual b - A x at K = 65536 (measured; the tile kernels and rocBLAS share that
  // chain) - and the refinement converges to the fixed point of its residual kernel: this kernel IS the attainable accuracy.
  d4 sum = {0.0, 0.0, 0.0, 0.0}, comp = {0.0, 0.0, 0.0, 0.0};
  auto kahan = [&](const d4& t) {
#pragma unroll
    for (int r = 0; r < 4; r++) {
      const double y = t[r] - comp[r];
      const double s2 = sum[r] + y;
      comp[r] = (s2 - sum[r]) - y;
      sum[r] = s2;
    }
  };
  int64_t k0 = 0;
  for (; k0 + 64 <= g.K; k0 += 64) {          // 8 x (8 k): sixteen 16-byte loads in flight per lane
    d2 a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; u++) a[u] = lda2(k0 + 8 * u);
#pragma unroll
    for (int u = 0; u < 8; u++) b[u] = bon ? *reinterpret_cast<const d2*>(pb + k0 + 8 * u) : (d2){0.0, 0.0};
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int u = 0; u < 8; u++) {
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].x, b[u].x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].y, b[u].y, acc1, 0, 0, 0);
    }
    kahan(acc0 + acc1);
  }
  if (k0 < g.K) {
    d4 acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = {0.0, 0.0, 0.0, 0.0};
    for (; k0 < g.K; k0 += 8) {
      const d2 a = lda2(k0);
      const d2 b = bon ? *reinterpret_cast<const d2*>(pb + k0) : (d2){0.0, 0.0};
      acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, b.x, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, b.y, acc1, 0, 0, 0);
    }
    kahan(acc0 + acc1);
  }
  if (!bon) return;                           // D layout: row = kg + 4 r (output row), column = lr (right-hand side)
#pragma unroll
  for (int r = 0; r < 4; r++) {
    double* pc = g.C + j0 + kg + 4 * r + (int64_t)lr * g.ldc;
    double v = g.alpha * sum[r];
    if (g.beta != 0.0) v += g.beta * (*pc);
    *pc = v;
  }
}

// (round 6: sixteen waves split K instead of four and every wave requests its next 8 k while it multiplies the current 8 - the first version
//  waited for each group of 8 loads before it asked for the next: 40 of the 50 us of a 1024 x 1024 block step were load latency in series)
#ifndef NNS_CFG
#define NNS_CFG 1608
#endif
constexpr int NNS_W = NNS_CFG / 100, NNS_U = NNS_CFG % 100;
template <bool F32>
__global__ void __launch_bounds__(64 * NNS_W) dgemm_nn_skinny_kernel(const SkinnyArgs g) {
  __shared__ double part[NNS_W][8][64];
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int64_t row = (int64_t)blockIdx.x * 64 + lane;
  const int64_t kq = ((g.K + 8 * NNS_W - 1) / (8 * NNS_W)) * 8;   // K share of a wave, multiple of 8
  const int64_t kb = std::min<int64_t>(wid * kq, g.K), ke = std::min<int64_t>(kb + kq, g.K);
  const double* __restrict__ pa = g.A + row;
  const float* __restrict__ pa32 = g.A32 + row;                                // (F32 only)
  const double* __restrict__ pb = g.B;
  double acc[8];
#pragma unroll
  for (int c = 0; c < 8; c++) acc[c] = 0.0;
  int64_t k = kb;
  double a0[NNS_U], a1[NNS_U];
  auto load = [&](double (&a)[NNS_U], int64_t kk) {
#pragma unroll
    for (int u = 0; u < NNS_U; u++) a[u] = F32 ? (double)pa32[(kk + u) * g.lda] : pa[(kk + u) * g.lda];
  };
  auto fma = [&](const double (&a)[NNS_U], int64_t kk) {
#pragma unroll
    for (int c = 0; c < 8; c++) {
      if (c < g.N) {
#pragma unroll
        for (int u = 0; u < NNS_U; u++) acc[c] += a[u] * pb[kk + u + (int64_t)c * g.ldb];  // wave-uniform operand; k ascending per accumulator
      }
    }
  };
  if (k + NNS_U <= ke) {
    load(a0, k);
    for (; k + 3 * NNS_U <= ke; k += 2 * NNS_U) {         // two groups per trip: the buffers swap roles without register copies
      load(a1, k + NNS_U); fma(a0, k);
      load(a0, k + 2 * NNS_U); fma(a1, k + NNS_U);
    }
    if (k + 2 * NNS_U <= ke) { load(a1, k + NNS_U); fma(a0, k); fma(a1, k + NNS_U); k += 2 * NNS_U; }
    else { fma(a0, k); k += NNS_U; }
  }
  for (; k < ke; k++) {
    const double a = F32 ? (double)pa32[k * g.lda] : pa[k * g.lda];
#pragma unroll
    for (int c = 0; c < 8; c++) if (c < g.N) acc[c] += a * pb[k + (int64_t)c * g.ldb];
  }
#pragma unroll
  for (int c = 0; c < 8; c++) part[wid][c][lane] = acc[c];
  __syncthreads();
  // thread t: row t & 63, right-hand side t >> 6; the waves' partial sums are added in wave order (deterministic)
  const int rr = threadIdx.x & 63, c = threadIdx.x >> 6;
  if (c < g.N) {
    double* pc = g.C + (int64_t)blockIdx.x * 64 + rr + (int64_t)c * g.ldc;
    double sum = part[0][c][rr];
#pragma unroll
    for (int w = 1; w < NNS_W; w++) sum += part[w][c][rr];
    double v = g.alpha * sum;
    if (g.beta != 0.0) v += g.beta * (*pc);
    *pc = v;
  }
}

// returns CAP_ERR_UNSUPPORTED when the shape does not qualify (the caller then takes the tile kernels)
int launch_skinny(int transa, int64_t m, int64_t n, int64_t k, double alpha, const double* A, int64_t lda, const double* B, int64_t ldb,
                  double beta, double* C, int64_t ldc, hipStream_t stream, const float* A32 = nullptr) {
  static const int on = CAP_ENV("CAP_SKINNY") ? atoi(CAP_ENV("CAP_SKINNY")) : 1;
  if (!on || n < 1 || n > 8 || m < 64 || k < 8 || (k % 8)) return CAP_ERR_UNSUPPORTED;
  SkinnyArgs g{A, B, C, lda, ldb, ldc, m, k, (int)n, alpha, beta, A32};
  auto note = [&]() {                                     // access notes of the stand-in's race checker (op(A) in fp32: 4-byte elements)
    if (!A32) { note_product(transa, CAP_NOTRANS, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 0); return; }
    if (!cap_acc_on()) return;
    if (transa == CAP_TRANS) cap_acc_r(A32, lda, k, m, 0, 4); else cap_acc_r(A32, lda, m, k, 0, 4);
    cap_acc_r(B, ldb, k, n); cap_acc(beta != 0.0 ? CAP_ACC_RW : CAP_ACC_W, C, ldc, m, n, 0);
  };
  if (transa == CAP_TRANS) {
    if ((m % 16) || (lda & 1) || (ldb & 1) || (((uintptr_t)(A32 ? (const void*)A32 : (const void*)A)) & (A32 ? 7 : 15)) || (((uintptr_t)B) & 15)) return CAP_ERR_UNSUPPORTED;
    note();
    if (A32) hipLaunchKernelGGL(dgemm_tn_skinny_kernel<true>, dim3((unsigned)cap_ceil_div(m, 64)), dim3(256), 0, stream, g);
    else hipLaunchKernelGGL(dgemm_tn_skinny_kernel<false>, dim3((unsigned)cap_ceil_div(m, 64)), dim3(256), 0, stream, g);
  } else {
    if (m % 64) return CAP_ERR_UNSUPPORTED;
    note();
    if (A32) hipLaunchKernelGGL(dgemm_nn_skinny_kernel<true>, dim3((unsigned)(m / 64)), dim3(64 * NNS_W), 0, stream, g);
    else hipLaunchKernelGGL(dgemm_nn_skinny_kernel<false>, dim3((unsigned)(m / 64)), dim3(64 * NNS_W), 0, stream, g);
  }
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

__global__ void scale_kernel(double* C, int64_t ldc, int64_t m, int64_t n, double beta, int tri) {
  int64_t col = blockIdx.y;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < m; row += (int64_t)gridDim.x * blockDim.x) {
    if (tri == 1 && row > col) continue;
    if (tri == 2 && row < col) continue;
    double* p = C + row + col * ldc;
    *p = (beta == 0.0) ? 0.0 : beta * (*p);
  }
}

// C = beta*C + sum_z P[z]  (fixed summation order -> run-to-run deterministic)
__global__ void splitk_reduce_kernel(double* C, int64_t ldc, const double* P, int64_t slab, int ksplit, int64_t m, int64_t n,
                                     double beta, int tri) {
  int64_t col = blockIdx.y;
  for (int64_t row = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; row < m; row += (int64_t)gridDim.x * blockDim.x) {
    if (tri == 1 && row > col) continue;
    if (tri == 2 && row < col) continue;
    double s = 0.0;
    for (int z = 0; z < ksplit; z++) s += P[(int64_t)z * slab + row + col * m];
    double* pc = C + row + col * ldc;
    *pc = (beta == 0.0) ? s : s + beta * (*pc);
  }
}

// library-owned scratch for split-K partials: ONE buffer per (device, stream), because two split-K products in
// flight on different streams must not share slabs.  Growth uses the stream-ordered allocator (no device-wide sync);
// the map itself is guarded by a mutex (the operator seam may be called from several host threads).
struct ScratchBuf { double* p; int64_t elems; };
std::mutex g_scratch_mu;
std::map<std::pair<int, hipStream_t>, ScratchBuf> g_scratch;

}  // namespace

double* cap_scratch(int64_t elems, hipStream_t stream) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  ScratchBuf& b = g_scratch[std::make_pair(dev, stream)];
  if (elems <= b.elems) return b.p;
  if (b.p) { (void)hipFreeAsync(b.p, stream); b.p = nullptr; b.elems = 0; }   // ordered behind the stream's earlier readers
  double* np_ = nullptr;
  if (hipMallocAsync((void**)&np_, sizeof(double) * elems, stream) != hipSuccess) return nullptr;
  b.p = np_; b.elems = elems;
  return b.p;
}

// the stream is about to be destroyed (cap_stream_destroy, common.h): its split-K scratch goes back to the pool, ordered behind its work
void cap_scratch_release(hipStream_t stream) {
  // found by the stream HANDLE alone: a plan may be destroyed while another device is current (the (device, stream) lookup leaked it then)
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  for (auto it = g_scratch.begin(); it != g_scratch.end(); ++it)
    if (it->first.second == stream) {
      if (it->second.p) (void)hipFreeAsync(it->second.p, stream);
      g_scratch.erase(it);
      return;
    }
}

// C[m x n] = alpha op(A32) B + beta C with op(A) stored in fp32 (lda in elements) and n <= 8: the streaming kernels with the operand widened in
// registers - the same fp64 sums as on the promoted copy, half the bytes.  CAP_ERR_UNSUPPORTED when the shape does not qualify.
int cap_skinny_f32a_launch(int transa, int64_t m, int64_t n, int64_t k, double alpha, const float* A32, int64_t lda, const double* B, int64_t ldb,
                           double beta, double* C, int64_t ldc, hipStream_t stream) {
  if (!A32 || m < 1024) return CAP_ERR_UNSUPPORTED;
  return launch_skinny(transa, m, n, k, alpha, nullptr, lda, B, ldb, beta, C, ldc, stream, A32);
}

int cap_gemm_launch(int transa, int transb, int64_t m, int64_t n, int64_t k, double alpha, const double* A,
                    int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, int tri,
                    hipStream_t stream, int tag, int persist_wgs, const double* Cin, int64_t ldcin) {
  if (m < 0 || n < 0 || k < 0) return CAP_ERR_ARG;
  if (Cin == C) Cin = nullptr;
  if (Cin && (ldcin < m || beta == 0.0 || alpha == 0.0 || k == 0)) return CAP_ERR_ARG;
  if (m == 0 || n == 0) return CAP_OK;
  if (ldc < m) return CAP_ERR_ARG;
  if (alpha == 0.0 || k == 0) {
    if (beta == 1.0) return CAP_OK;
    dim3 grid((unsigned)cap_ceil_div(m, 256), (unsigned)n);
    if (n > 65535) return CAP_ERR_UNSUPPORTED;
    note_product(transa, transb, m, n, 0, 0.0, A, lda, B, ldb, beta, C, ldc, tri);
    hipLaunchKernelGGL(scale_kernel, grid, dim3(256), 0, stream, C, ldc, m, n, beta, tri);
    CAP_HIP(hipGetLastError());
    return CAP_OK;
  }
  if (!A || !B || !C) return CAP_ERR_ARG;
  // op(A) is m x k: stored k x m when transposed
  if (transa == CAP_TRANS ? lda < k : lda < m) return CAP_ERR_ARG;
  if (transb == CAP_TRANS ? ldb < n : ldb < k) return CAP_ERR_ARG;

  // a few right-hand sides against a big operand: streaming kernels (no padding of N to a 128-wide tile)
  if (n <= 8 && tri == 0 && transb != CAP_TRANS && !Cin && m >= 1024) {
    const int st = launch_skinny(transa, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, stream);
    if (st != CAP_ERR_UNSUPPORTED) return st;
  }
  // latency-bound little products (diagonal-block recursion): 64 x 64 tiles, no setup cost
  if (m <= 512 && n <= 512 && k <= 1024 && m * n <= 256 * 256 && !Cin)
    return launch_small(transa, transb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri, (tag & 2) ? 1 : 0, stream);

  GemmArgs g;
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.M = m; g.N = n; g.K = k; g.alpha = alpha; g.beta = beta; g.tri = tri;
  g.hiprio = (tag & 2) ? 1 : 0; g.bupper = (tag & 8) ? 1 : 0;
  g.aupt = ((tag & 16) && transa == CAP_TRANS) ? 1 : 0; g.aupn = ((tag & 32) && transa != CAP_TRANS) ? 1 : 0;
  const bool no_atomic = (tag & CAP_TAG_NO_ATOMIC) != 0 || Cin != nullptr;
  g.Cin = Cin ? Cin : C; g.ldcin = Cin ? ldcin : ldc;
  tag &= 1; g.ctr = nullptr;
  g.atomic_c = 0; g.usebuf = 0; g.skip = 0;
  g.stair = 0; g.gather = 0; g.sP = 1; g.sp = 0; g.snbT = 1; g.sJ0 = 0; g.slb0 = 0; g.gpiece = 0; g.rP = 1; g.rp = 0; g.rlb0 = 0;
  for (int i = 0; i < 8; i++) g.gstart[i] = 0;
  g.tm = (int)cap_ceil_div(m, BM); g.tn = (int)cap_ceil_div(n, BN);
  // Supertile edge.  Round 6 (profiles/r06_experiments.md section 10): edge 2 instead of 8 makes the factorizations 0.7 - 2 % faster (N = 65536 / 32768:
  // the XCDs' slot ranges end more evenly and the diagonal supertiles carry 1 empty slot of 4 instead of 28 of 64), edge 4 would cut the
  // operand re-fetch by 30 % for half of that gain; the NN products with a triangular operand (the inverse tree) keep 8 - their band mapping
  // below is built on it and they lose 5 % without it.
  static const int st_env = CAP_ENV("CAP_ST") ? atoi(CAP_ENV("CAP_ST")) : 0;
  g.st = st_env > 0 ? st_env : ((g.bupper || g.aupn) ? ST : ST_SMALL);
  g.stm = g.st; g.stn = g.st; g.sorder = 0;
  g.nsm = (int)cap_ceil_div(g.tm, g.st); g.nsn = (int)cap_ceil_div(g.tn, g.st);
  int64_t nsuper;
  g.etri = (tri != 0 && g.nsm == g.nsn) ? tri : 0;
  static const int xbal_env = CAP_ENV("CAP_XCD_BALANCE") ? atoi(CAP_ENV("CAP_XCD_BALANCE")) : 1;
  if (g.etri == 0 && xbal_env && g.st == 8 && (g.bupper || g.aupt || g.aupn) && g.tm * g.tn >= 64) {
    // triangular operand: make the K-determining tile index the one every XCD's range walks completely (see GemmArgs::stm)
    if (g.bupper && !(g.aupt || g.aupn)) {            // K range grows with the column tile: XCD x = a band of tile rows, all columns
      g.stm = std::max(1, std::min(8, g.tm / 8)); g.stn = std::max(1, std::min(g.tn, 64 / g.stm)); g.sorder = 1;
    } else if (!g.bupper) {                           // K range depends on the row tile: XCD x = a band of tile columns, all rows
      g.stn = std::max(1, std::min(8, g.tn / 8)); g.stm = std::max(1, std::min(g.tm, 64 / g.stn)); g.sorder = 0;
    }
    g.nsm = (int)cap_ceil_div(g.tm, g.stm); g.nsn = (int)cap_ceil_div(g.tn, g.stn);
  }
  if (g.etri == 0) nsuper = (int64_t)g.nsm * g.nsn;   // strips keep the element mask but walk the full grid
  else nsuper = (int64_t)g.nsm * (g.nsm + 1) / 2;      // square tile space: enumerate the supertile triangle
  int64_t slots = nsuper * g.stm * g.stn;
  g.chunk = (int)cap_ceil_div(slots, 8);
  int64_t grid = (int64_t)g.chunk * 8;
  if (grid > 0x7fffffff) return CAP_ERR_UNSUPPORTED;

  // split-K when the tile grid cannot fill the chip and K is long (Gram matrices, cacqr.hpp:15)
  g.ksplit = 1; g.kchunk = cap_round_up(k, BK); g.P = nullptr; g.slab = 0;
  {
    int64_t tiles = (tri == 0) ? (int64_t)g.tm * g.tn : ((int64_t)g.tm * (g.tm + 1)) / 2;
    if (tiles < 128 && k >= 4096 && n <= 65535) {
      // whole rounds of the chip's 512 workgroup slots, at least three of them: the tiles of a small SYRK are unequal
      // (diagonal tiles skip their dead blocks) and a half-empty last round costs as much as a full one
      int64_t want = cap_ceil_div(1536, tiles);
      int64_t maxs = k / 1024;
      int64_t ks = want < maxs ? want : maxs;
      if (ks > 1) {
        int64_t kc = cap_round_up(cap_ceil_div(k, ks), BK);
        ks = cap_ceil_div(k, kc);
        double* P = cap_scratch(ks * m * n, stream);
        if (P && ks > 1) { g.ksplit = (int)ks; g.kchunk = kc; g.P = P; g.slab = m * n; }
      }
    }
  }

  {
    static const int atomic_env = CAP_ENV("CAP_ATOMIC_C") ? atoi(CAP_ENV("CAP_ATOMIC_C")) : 1;
    g.atomic_c = (atomic_env && !no_atomic && beta == 1.0 && g.ksplit == 1) ? 1 : 0;
    static const int skip_env = CAP_ENV("CAP_SKIP") ? atoi(CAP_ENV("CAP_SKIP")) : 1;
    g.skip = (skip_env && tag != 1 && (g.bupper || g.aupt || g.aupn || (tri == 1 && g.tm <= 8))) ? 1 : 0;
    static const int buf_env = CAP_ENV("CAP_DMA_BUF") ? atoi(CAP_ENV("CAP_DMA_BUF")) : 1;
    g.usebuf = (buf_env && 128 * lda * 8 + k * 8 < 0xfffffff0LL && 128 * ldb * 8 + k * 8 < 0xfffffff0LL) ? 1 : 0;
  }
  const bool a_kc = (transa == CAP_TRANS);   // op(A)=A^T: k contiguous
  const bool b_kc = (transb != CAP_TRANS);   // op(B)=B:   k contiguous
  auto aligned16 = [](const void* p) { return (((uintptr_t)p) & 15) == 0; };
  // 0: aligned interior; 1: ragged extents with 16-byte-aligned even geometry (vector loads); 2: scalar
  int edge = 0;
  if ((m % BM) || (n % BN) || (k % BK)) edge = 1;
  {
    bool vec_ok = !(lda & 1) && !(ldb & 1) && aligned16(A) && aligned16(B) && !(k & 1);
    if (!a_kc && (m & 1)) vec_ok = false;     // outer-contiguous operands need an even outer extent
    if (!b_kc && (n & 1)) vec_ok = false;
    if (!vec_ok) edge = 2;
  }

  int st;
  constexpr bool use_v1 = false;       // (round 1's register-staged kernels for the aligned TN / NN forms: CAP_GEMM_V1 is gone)
  // a separate C input is only implemented in the LDS-DMA kernels' load / add / store epilogue
  if (Cin && !(a_kc && b_kc && !edge && !use_v1 && g.ksplit == 1)) return CAP_ERR_UNSUPPORTED;     // (the NN form has a_kc == false)
  // (split-K: the product's notes ride on the first of its two launches, the partial slabs are this stream's own scratch)
  note_product(transa, transb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, tri, Cin, ldcin);
  if (a_kc && b_kc && !edge && !use_v1) st = (tag == 1) ? launch_tn_dma<1>(g, (int)grid, stream, persist_wgs) : launch_tn_dma<0>(g, (int)grid, stream, 0);
  else if (!a_kc && b_kc && !edge && !use_v1 && g.ksplit == 1) st = launch_nn_dma(g, (int)grid, stream);
  else if (a_kc && b_kc && tag == 1) st = launch_variant<true, true, 1>(g, edge, (int)grid, stream);
  else if (a_kc && b_kc) st = launch_variant<true, true>(g, edge, (int)grid, stream);
  else if (a_kc && !b_kc) st = launch_variant<true, false>(g, edge, (int)grid, stream);
  else if (!a_kc && b_kc) st = launch_variant<false, true>(g, edge, (int)grid, stream);
  else st = launch_variant<false, false>(g, edge, (int)grid, stream);
  if (st != CAP_OK || g.ksplit == 1) return st;
  cap_acc_none();
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)cap_ceil_div(m, 256), (unsigned)n), dim3(256), 0, stream, C, ldc,
                     g.P, g.slab, g.ksplit, m, n, beta, tri);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

// Distributed trailing update on a 1 x P block-column-cyclic matrix (see GemmArgs::stair):
//   C[m x nloc] -= G^T * B  restricted to the global upper triangle, K = nb.
// G: gathered block row (P pieces of `piece` doubles, each nb x cols_r column-major with ld = k),
// B: this rank's own solved columns (nb x nloc, ld = k), C: local columns, rows are global (ldc).
// Requires m, nloc multiples of 128, nb multiple of 128, k multiple of 16.
int cap_dist_update_launch(int64_t m, int64_t nloc, int64_t k, const double* G, int64_t piece, const int* gstart,
                           const double* B, double* C, int64_t ldc, int P, int p, int nb, int J0, int lb0,
                           hipStream_t stream, int persist_wgs, int Pr, int pr, int rlb0) {
  if (m <= 0 || nloc <= 0) return CAP_OK;
  if ((m % BM) || (nloc % BN) || (k % BK) || (nb % 128) || P < 1 || P > 8 || Pr < 1 || pr < 0 || pr >= Pr) return CAP_ERR_UNSUPPORTED;
  if (Pr == 1) rlb0 = J0;                                  // rows are global: origin at block J0
  GemmArgs g;
  g.A = G; g.B = B; g.C = C; g.lda = k; g.ldb = k; g.ldc = ldc;
  g.M = m; g.N = nloc; g.K = k; g.alpha = -1.0; g.beta = 1.0; g.tri = 1; g.etri = 0;
  g.tm = (int)(m / BM); g.tn = (int)(nloc / BN);
  // (round 6: 16 tiles per supertile instead of 64 - the distributed update alone gains 3 - 4 % in rank shapes of a 1 x 8 plan, the replayed rank 1.7 %;
  //  the one-rank plans do not care; profiles/r06_experiments.md section 10)
  constexpr int DIST_ST = 4;
  g.st = DIST_ST; g.stm = DIST_ST; g.stn = DIST_ST; g.sorder = 0;
  // block-column-shaped supertiles where the staircase is steep (1 x P with P >= 4: P row tiles per local column tile), see stair_cnt
  if (Pr == 1 && P >= 4 && nb / 128 >= 2 && nb / 128 <= 8 && (DIST_ST * DIST_ST) % (nb / 128) == 0 && DIST_ST * DIST_ST >= nb / 128) { g.stn = nb / 128; g.stm = DIST_ST * DIST_ST / g.stn; }
  g.nsm = (int)cap_ceil_div(g.tm, g.stm); g.nsn = (int)cap_ceil_div(g.tn, g.stn);
  g.ksplit = 1; g.kchunk = k; g.P = nullptr; g.slab = 0;
  g.hiprio = 0; g.ctr = nullptr; g.bupper = 0; g.aupt = 0; g.aupn = 0;
  {
    static const int atomic_env = CAP_ENV("CAP_ATOMIC_C") ? atoi(CAP_ENV("CAP_ATOMIC_C")) : 1;
    g.atomic_c = atomic_env ? 1 : 0;
    g.usebuf = (128 * k * 8 + k * 8 < 0x7fffffffLL) ? 1 : 0; g.skip = 0; g.Cin = C; g.ldcin = ldc;
  }
  g.stair = 1; g.gather = 1; g.sP = P; g.sp = p; g.snbT = nb / 128; g.sJ0 = J0; g.slb0 = lb0; g.gpiece = piece;
  g.rP = Pr; g.rp = pr; g.rlb0 = rlb0;
  for (int i = 0; i < 8; i++) g.gstart[i] = i < P ? gstart[i] : 0;
  // only supertiles under the staircase are enumerated, so the 8 XCD ranges carry equal work
  g.etri = 3;
  int64_t nsuper = 0;
  for (int sj = 0; sj < g.nsn; sj++) nsuper += stair_cnt(sj, g.stm, g.stn, g.tn, g.nsm, p, P, lb0, nb / 128, J0, Pr, pr, rlb0);
  if (nsuper == 0) return CAP_OK;      // Pr > 1: every local row block may lie below every local column block (nothing to update)
  int64_t slots = nsuper * g.stm * g.stn;
  g.chunk = (int)cap_ceil_div(slots, 8);
  if (cap_acc_on()) {
    // access notes: per local column block J the row blocks I < J whole and I == J above the diagonal; the A operand's column chunk
    // of every local row block that has a partner, B whole
    const int64_t ncb = nloc / nb, nrb = m / nb;
    const int64_t Jlast = p + (int64_t)P * (lb0 + ncb - 1);
    for (int64_t cb = 0; cb < ncb; cb++) {
      const int64_t J = p + (int64_t)P * (lb0 + cb);
      int64_t full = 0; bool diag = false;
      for (int64_t rb = 0; rb < nrb; rb++) {
        const int64_t I = pr + (int64_t)Pr * (rlb0 + rb);
        if (I < J) full = rb + 1; else { diag = I == J; break; }
      }
      cap_acc_rw(C + cb * nb * ldc, ldc, full * nb, nb);
      if (diag) cap_acc_rw(C + full * nb + cb * nb * ldc, ldc, nb, nb, 1);
    }
    for (int64_t rb = 0; rb < nrb; rb++) {
      const int64_t I = pr + (int64_t)Pr * (rlb0 + rb);
      if (I > Jlast) break;
      const int64_t r = (I % P) / Pr, lb = I / P - gstart[r];
      cap_acc_r(G + r * piece + lb * nb * k, 0, k * nb, 1);
    }
    cap_acc_r(B, 0, k * nloc, 1);
  }
  return launch_tn_dma<1>(g, g.chunk * 8, stream, persist_wgs);
}

// batched small products (blockIdx.z = batch, affine pointer strides) - the level-by-level inverse assembly
int cap_gemm_small_batched(int transa, int transb, int64_t m, int64_t n, int64_t k, double alpha, const double* A, int64_t lda,
                           int64_t sa, const double* B, int64_t ldb, int64_t sb, double beta, double* C, int64_t ldc, int64_t sc,
                           int nbatch, hipStream_t stream) {
  if (m <= 0 || n <= 0 || nbatch <= 0) return CAP_OK;
  if (nbatch > 65535) return CAP_ERR_UNSUPPORTED;
  return launch_small(transa, transb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 0, 1, stream, nbatch, sa, sb, sc);
}

// pure index helper behind the staircase enumeration of the distributed update (tests check it against a brute-force count):
// number of LOCAL row tiles (128 rows) of process row pr of Pr, starting at local row block rlb0, whose global tile index
// relative to block J0 is <= X; nbT = nb / 128 tiles per block
extern "C" int cap_bc2d_rows_le(int X, int nbT, int J0, int Pr, int pr, int rlb0) { return stair_rows_le(X, nbT, J0, Pr, pr, rlb0); }

extern "C" int cap_dgemm(int transa, int transb, int64_t m, int64_t n, int64_t k, double alpha, const double* A,
                         int64_t lda, const double* B, int64_t ldb, double beta, double* C, int64_t ldc, void* stream) {
  // caller memory: the atomic epilogue only on plain device allocations (fp64 hardware atomics are silently
  // dropped on fine-grained / managed / host-mapped buffers)
  const int tag = (beta == 1.0 && !cap_plain_device_ptr(C)) ? CAP_TAG_NO_ATOMIC : 0;
  return cap_gemm_launch(transa, transb, m, n, k, alpha, A, lda, B, ldb, beta, C, ldc, 0, cap_stream(stream), tag);
}

extern "C" int cap_dsyrk(int uplo, int trans, int64_t n, int64_t k, double alpha, const double* A, int64_t lda,
                         double beta, double* C, int64_t ldc, void* stream) {
  // trans: C = alpha*A^T*A + beta*C with A k x n;  notrans: C = alpha*A*A^T + beta*C with A n x k
  int tri = (uplo == CAP_UPPER) ? 1 : 2;
  const int tag = (beta == 1.0 && !cap_plain_device_ptr(C)) ? CAP_TAG_NO_ATOMIC : 0;
  if (trans == CAP_TRANS)
    return cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, n, n, k, alpha, A, lda, A, lda, beta, C, ldc, tri, cap_stream(stream), tag);
  return cap_gemm_launch(CAP_NOTRANS, CAP_TRANS, n, n, k, alpha, A, lda, A, lda, beta, C, ldc, tri, cap_stream(stream), tag);
}
