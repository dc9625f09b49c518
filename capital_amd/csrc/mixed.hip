// Mixed-precision Cholesky solve: bf16 MFMA factorization + fp64 iterative refinement (BASELINE config 5).
//
// Not in the reference: its only solve path is the stub trsm/diaginvert (diaginvert.hpp:7-10, static_assert) and its
// factorization is fp64 only.  The fp64 path of this library (cholinv.hip + cap_dtrsm) is the oracle of this one.
//
//   factor   R32 (fp32 storage, upper) from A (fp64): blocked right-looking, panel width nb = 1024
//              diagonal block   -> fp64, the existing in-LDS / MFMA fp64 cholinv chain (R_kk and its inverse), back to fp32
//              block row        -> fp64, S = Dinv^T * row  (fp64 MFMA GEMM), stored as fp32 (the factor) and bf16 (the panel)
//              trailing update  -> C32 -= P16^T P16 on v_mfma_f32_32x32x16_bf16, fp32 accumulate, upper tiles only
//            i.e. ALL O(N^3) flops run at bf16 MFMA rate; the O(nb N^2) panel work stays fp64.
//   solve    x = R^-1 R^-T b with the fp64-promoted factor and the blocked TRSM (diagonal-block inverses cached in the
//            plan), then classical iterative refinement in fp64:  r = b - A x (fp64 MFMA GEMM), d = R^-1 R^-T r, x += d,
//            until ||r||_F / ||b||_F <= tol.  Converges when kappa(A) * eps_bf16-factor < 1.
//
// bf16 tile kernel (same shape as the fp64 one, gemm.hip): 128 x 128 C tile per 256-thread workgroup, 4 waves x (2 x 2
// blocks of 32 x 32), K tile = 64 bf16 = 128 bytes per row, so the LDS image, its XOR swizzle and the buffer-addressed
// LDS-DMA are byte-for-byte those of the fp64 kernel (tile_dma.h); fragments by ds_read_b128 (8 bf16 = one MFMA operand);
// operands swapped so a lane owns 32 consecutive rows of a column -> the fp32 atomic-add epilogue touches whole lines.
// Roofline of the update: C traffic 8 B per 2K flop -> K/4 flop/B; K = 1024: 1.5 PFLOP/s at 6 TB/s - HBM-bound below the
// 2.5 PFLOP/s MFMA peak, and this 128-wide tile additionally pulls 32 KiB of operands per 16 MFMAs through L2.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <new>
#include <type_traits>
#include <vector>

#include "common.h"
#include "kargs.h"
#include "mixed_kernels.h"
#include "tile_dma.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int TB = 128;            // C tile edge
constexpr int KB = 64;             // K tile (bf16 elements) = 128 bytes per row
constexpr int TILE_D = 128 * 16;   // doubles per operand tile image (16 KiB)

// struct BfArgs: csrc/kargs.h (shared with the CPU kernel models of tests/hipshim)

// C[M x N] += alpha * A^T B,  A: K x M, B: K x N (both K-contiguous bf16), C fp32 column-major.  M, N % 128 == 0, K % 64 == 0.
// Epilogue: fire-and-forget fp32 atomics.  Every C tile has ONE writer per launch, so a plain read - add - write works too; round 5
// measured it (buffer-addressed loads behind the last MFMAs, then stores): 786 TF against 800 alone, 171.0 ms against 170.0 in the
// N = 65536 factorization (profiles/r05_bf16_bench.log, r05_mixed_rmw.log) - this kernel is bound by its LDS traffic (16 KiB + 16 KiB
// staged and 64 KiB of fragment reads per 64 MFMAs), not by how C is written - and removed it again.
__global__ void __launch_bounds__(256, 2) bf16_tn_kernel(const BfArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  // block -> tile: XCD b % 8 walks a contiguous range of the tile list (column-major; upper triangle for tri)
  const int b = (int)blockIdx.x;
  const int L = (b & 7) * g.chunk + (b >> 3);
  int ti, tj;
  if ((b >> 3) >= g.chunk) return;
  if (!g.stair && g.st > 0) {
    const int per = g.st * g.st, sl = L / per, q = L - sl * per;
    int si, sj;
    if (g.tri && g.tm == g.tn) {
      sj = (int)((__builtin_sqrtf(8.0f * (float)sl + 1.0f) - 1.0f) * 0.5f);
      while ((sj + 1) * (sj + 2) / 2 <= sl) sj++;
      while (sj * (sj + 1) / 2 > sl) sj--;
      si = sl - sj * (sj + 1) / 2;
    } else {
      si = sl % g.nsm; sj = sl / g.nsm;
    }
    ti = si * g.st + q % g.st; tj = sj * g.st + q / g.st;
    if (ti >= g.tm || tj >= g.tn || (g.tri && ti > tj)) return;
  } else if (!g.stair && g.tri && g.tm == g.tn) {
    tj = (int)((__builtin_sqrtf(8.0f * (float)L + 1.0f) - 1.0f) * 0.5f);
    while ((tj + 1) * (tj + 2) / 2 <= L) tj++;
    while (tj * (tj + 1) / 2 > L) tj--;
    ti = L - tj * (tj + 1) / 2;
    if (tj >= g.tn) return;
  } else {
    ti = L % g.tm; tj = L / g.tm;
    if (tj >= g.tn) return;
    if (!g.stair && g.tri && ti > tj) return;            // strip of a triangular update: only tiles on / above the diagonal
  }
  int gtj = tj;                                          // global tile column (relative to the row origin) under the staircase view
  const __bf16* Abase = g.A + (int64_t)ti * TB * g.lda;
  if (g.stair) {
    const int J = g.sp + g.sP * (g.slb0 + tj / g.snbT);
    gtj = (J - g.sJ0) * g.snbT + tj % g.snbT;
    if (ti > gtj) return;
    const int I = g.sJ0 + ti / g.snbT, r = I % g.sP, lb = I / g.sP - g.gstart[r];
    Abase = g.A + (int64_t)r * g.gpiece + ((int64_t)(lb * g.snbT + ti % g.snbT) * TB) * g.lda;
  }
  const int64_t i0 = (int64_t)ti * TB, j0 = (int64_t)tj * TB;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wi = (wid & 1) * 64, wj = (wid >> 1) * 64;
  const int r32 = lane & 31, kg = lane >> 5;
  auto sA = [&](int buf) -> double* { return smem + buf * 2 * TILE_D; };
  auto sB = [&](int buf) -> double* { return smem + buf * 2 * TILE_D + TILE_D; };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;

  const int nk = (int)(g.K / KB);
  // the bf16 panels are addressed as "double" matrices of K / 4 doubles per row: same 128-byte rows, same swizzle
  DmaBuf dA = dma_buf_make(reinterpret_cast<const double*>(Abase), g.lda / 4);
  DmaBuf dB = dma_buf_make(reinterpret_cast<const double*>(g.B + j0 * g.ldb), g.ldb / 4);
  // fragment byte offsets: row (block origin + r32), chunk (2 s + kg) ^ ((row >> 1) & 7) = (2 s) ^ t
  const int t = kg ^ ((r32 >> 1) & 7);
  const int a_row = (wi + r32) * 128, b_row = (wj + r32) * 128;
  bf16x8 fa[2][2][2], fb[2][2][2];          // [half][step in half][block]
  auto read_half = [&](const double* tA, const double* tB, int h, bf16x8 (&xa)[2][2], bf16x8 (&xb)[2][2]) {
    const char* pa = reinterpret_cast<const char*>(tA); const char* pb = reinterpret_cast<const char*>(tB);
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const int off = (((2 * (2 * h + s)) ^ t) << 4);
#pragma unroll
      for (int blk = 0; blk < 2; blk++) {
        xa[s][blk] = *reinterpret_cast<const bf16x8*>(pa + a_row + blk * 32 * 128 + off);
        xb[s][blk] = *reinterpret_cast<const bf16x8*>(pb + b_row + blk * 32 * 128 + off);
      }
    }
  };
  auto mma_half = [&](const bf16x8 (&xa)[2][2], const bf16x8 (&xb)[2][2]) {
#pragma unroll
    for (int s = 0; s < 2; s++)
#pragma unroll
      for (int i = 0; i < 2; i++)
#pragma unroll
        for (int j = 0; j < 2; j++)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[s][j], xa[s][i], acc[i][j], 0, 0, 0);   // swapped: lane = C row
  };

  if (nk > 0) {
    dma_tile_buf<0, 4>(dA, wid, 0u, sA(0)); dma_tile_buf<0, 4>(dB, wid, 0u, sB(0));
    __syncthreads();
    read_half(sA(0), sB(0), 0, fa[0], fb[0]);
    { const uint32_t k1 = (nk > 1 ? 1u : 0u) * 128u; dma_tile_buf<0, 4>(dA, wid, k1, sA(1)); dma_tile_buf<0, 4>(dB, wid, k1, sB(1)); }
    read_half(sA(0), sB(0), 1, fa[1], fb[1]);
    mma_half(fa[0], fb[0]);
    for (int kt = 0; kt + 1 < nk; kt++) {
      const int nxt = (kt + 1) & 1;
      __builtin_amdgcn_s_waitcnt(0xc07f);
      __syncthreads();                                   // tile kt+1 has landed, tile kt's buffer is free
      const uint32_t kn = (uint32_t)((kt + 2 < nk) ? kt + 2 : nk - 1) * 128u;
      dma_tile_buf<0, 4>(dA, wid, kn, sA(nxt ^ 1));
      read_half(sA(nxt), sB(nxt), 0, fa[0], fb[0]);
      mma_half(fa[1], fb[1]);
      __builtin_amdgcn_s_waitcnt(0xc07f);
      dma_tile_buf<0, 4>(dB, wid, kn, sB(nxt ^ 1));
      read_half(sA(nxt), sB(nxt), 1, fa[1], fb[1]);
      mma_half(fa[0], fb[0]);
    }
    mma_half(fa[1], fb[1]);
  }

  // epilogue: lane holds C[i0 + wi + 32 i + r32][j0 + wj + 32 j + (e & 3) + 8 (e >> 2) + 4 kg]: 32 consecutive rows per half wave
  // the diagonal mask compares WITHIN-tile offsets: under the staircase view a diagonal tile (ti == gtj) sits at a local column
  // tile tj != ti, so absolute offsets would drop the whole tile (the fp64 twin tn_dma_tile in gemm.hip does the same)
  const bool diag = g.stair ? (ti == gtj) : (g.tri && ti == tj);
#pragma unroll
  for (int i = 0; i < 2; i++) {
    const int lrow = wi + 32 * i + r32;
    const int64_t row = i0 + lrow;
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int lcol = wj + 32 * j + (e & 3) + 8 * (e >> 2) + 4 * kg;
        const int64_t col = j0 + lcol;
        if (!diag || lrow <= lcol)
          __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(g.C + row + col * g.ldc), g.alpha * acc[i][j][e]);
      }
  }
}

// (One workgroup per CU for the chain-bound strips - the fp64 plan's occ1 form, 96 KiB of LDS - was measured here too in round 5: the
// factorization does not move, 164.1 against 164.2 ms at N = 65536, and loses from 24576 rows up: profiles/r05_mixed_occ1.log.)
inline void launch_bf16_tn_kernel(const BfArgs& g, unsigned grid, hipStream_t s) {
  hipLaunchKernelGGL(bf16_tn_kernel, dim3(grid), dim3(256), 4 * TILE_D * sizeof(double), s, g);
}

int launch_bf16_tn(int64_t m, int64_t n, int64_t k, float alpha, const __bf16* A, int64_t lda, const __bf16* B, int64_t ldb, float* C,
                   int64_t ldc, int tri, hipStream_t s) {
  if (m <= 0 || n <= 0 || k <= 0) return CAP_OK;
  if ((m % TB) || (n % TB) || (k % KB) || (lda % 8) || (ldb % 8) || (tri && m > n)) return CAP_ERR_UNSUPPORTED;
  if (128 * lda * 2 + k * 2 >= 0xfffffff0LL || 128 * ldb * 2 + k * 2 >= 0xfffffff0LL) return CAP_ERR_UNSUPPORTED;
  BfArgs g;
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = m; g.N = n; g.K = k; g.alpha = alpha; g.tri = tri;
  g.stair = 0; g.sP = 1; g.sp = 0; g.snbT = 1; g.sJ0 = 0; g.slb0 = 0; g.gpiece = 0;
  for (int i = 0; i < 8; i++) g.gstart[i] = 0;
  static const int st_env = CAP_ENV("CAP_BF16_ST") ? atoi(CAP_ENV("CAP_BF16_ST")) : 8;     // supertile edge (BfArgs::st), 0 = column-major walk
  g.st = 0; g.nsm = 1;
  g.tm = (int)(m / TB); g.tn = (int)(n / TB);
  int64_t tiles = (tri && m == n) ? (int64_t)g.tn * (g.tn + 1) / 2 : (int64_t)g.tm * g.tn;
  if (st_env > 0) {
    g.st = st_env; g.nsm = (int)cap_ceil_div(g.tm, g.st);
    const int64_t nsn = cap_ceil_div(g.tn, g.st);
    tiles = ((tri && m == n) ? nsn * (nsn + 1) / 2 : (int64_t)g.nsm * nsn) * g.st * g.st;
  }
  g.chunk = (int)cap_ceil_div(tiles, 8);
  // access notes: C is accumulated with device-scope fp32 atomics (launches on two streams may add into the same elements)
  cap_acc_r(A, lda, k, m, 0, 2); cap_acc_r(B, ldb, k, n, 0, 2); cap_acc(CAP_ACC_ATOMIC, C, ldc, m, n, tri ? 1 : 0, 4);
  launch_bf16_tn_kernel(g, (unsigned)(g.chunk * 8), s);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

// =====================================================================================================================================
// bf16 update, second generation (round 4): what the round-3 measurements asked for (profiles/r03_bf16_update_pmc.txt: matrix pipe busy
// 40 %, waves waiting 52 % of their cycles; K = 1024 -> 2048 showed a fixed cost of 0.45 K-1024-loops per 128 x 128 tile).
//   * 256 x 128 C tile per 512-thread workgroup: 48 KiB of operands per 64-deep K tile for 4.2 MFLOP (87 flop/B through L2, 64 before)
//   * wave specialisation: waves 0-3 compute (128 x 64 each: 8 accumulators of 32 x 32), waves 4-7 only move data (LDS-DMA).
//     The two roles never share a memory counter: the loaders' vmcnt counts DMA pieces only (12 per wave and stage, waited for with a
//     COUNTED s_waitcnt - loads return in order), the compute waves' vmcnt counts their fire-and-forget fp32 atomics only and is never
//     waited for.  (One wave doing both cannot use a counted wait: loads and atomics complete out of order with respect to each other.)
//   * three-deep LDS ring of 48 KiB stages (144 of 160 KiB, one workgroup per CU): stage g + 2 is requested when stage g is handed
//     over - a prefetch distance of two K tiles (~2000 cycles of MFMA work) against the ~1-2 us the L2 misses take under load
//   * ONE raw s_barrier per K tile; fragments one k-step ahead in registers; the last k-step of a stage runs behind the barrier
//     and covers the first fragment reads of the next stage
//   * tile loop: a workgroup walks `tpw` supertile steps (a chunk of up to tpw tiles); the ring runs on across tile boundaries, the
//     accumulators' 128 atomics per lane are issued behind the next tile's first fragment reads and drain while it computes.
//     Round 3's kernels paid prologue DMA + epilogue drain per tile with the whole chip in lockstep (every tile takes the same time, so
//     all 512 resident tiles reached their read-modify-write of C - 32 MB, an HBM-bound burst - together while the matrix pipes idled).
//   * XCD-aware order: the 32 workgroups an XCD runs side by side work on the 4 x 8 tiles of one 1024 x 1024 supertile (4 A slices +
//     8 B slices per K tile for 32 tiles), supertiles round-robin over the XCDs, upper-triangular enumeration for the symmetric update.
// struct Bf2Args: csrc/kargs.h (shared with the CPU kernel models of tests/hipshim)
constexpr int V2_STAGE = 49152, V2_AIMG = 32768;      // bytes: A image 256 rows x 128 B, B image 128 rows x 128 B

// step i of workgroup (xcd, slot, round): which tile, if any
__device__ __forceinline__ bool v2_tile_at(const Bf2Args& g, int xcd, int slot, int round, int i, int& ti, int& tj) {
  const int q = xcd + 8 * (round * g.tpw + i);
  if (i >= g.tpw || q >= g.nsuper) return false;
  int si, sj;
  if (g.tri) {
    sj = (int)((__builtin_sqrtf(8.0f * (float)q + 1.0f) - 1.0f) * 0.5f);
    while ((sj + 1) * (sj + 2) / 2 <= q) sj++;
    while (sj * (sj + 1) / 2 > q) sj--;
    si = q - sj * (sj + 1) / 2;
  } else {
    si = q % g.nsi; sj = q / g.nsi;
  }
  ti = 4 * si + (slot & 3); tj = 8 * sj + (slot >> 2);
  return ti < g.tm && tj < g.tn && !(g.tri && tj < 2 * ti);      // (tj < 2 ti: every row of the tile lies below its columns)
}
__device__ __forceinline__ int v2_next(const Bf2Args& g, int xcd, int slot, int round, int i, int& ti, int& tj) {
  while (i < g.tpw && !v2_tile_at(g, xcd, slot, round, i, ti, tj)) i++;
  return i;
}

// SCHED 1: the fragment reads of k-step s + 1 are all issued before the MFMAs of k-step s (sched_barrier: hipcc otherwise sinks them
//          behind the MFMAs that free their registers, 2 - 3 MFMAs ahead of their use - one compute wave per SIMD has nobody to cover that)
// DBG (timing surgery, wrong results, CAP_EXPERIMENTS builds only): 1 no atomics, 2 no DMA after the ring is primed, 4 no MFMAs
// PFI > 0: the loaders also pull the C tile through L2 (LDS-DMA into a 1 KiB dummy slot) during the last PFI K tiles of a tile, so that the
//          atomics that follow find their lines in L2 (measured: the atomics of a launch alone take as long as K = 2048 of MFMA work,
//          at a third of the HBM rate - every one of them is an L2 miss)
template <int SCHED, int DBG, int PFI>
__global__ void __launch_bounds__(512, 2) bf16_tn_v2_kernel(const Bf2Args g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  char* lds = reinterpret_cast<char*>(smem);
  const int b = (int)blockIdx.x, xcd = b & 7, ell = b >> 3, slot = ell & 31, round = ell >> 5;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // my tiles: count them (every wave runs the same scalar walk: no LDS list, the loaders must not read LDS inside their loop)
  int nt = 0;
  { int ti, tj; for (int i = 0; i < g.tpw; i++) nt += v2_tile_at(g, xcd, slot, round, i, ti, tj) ? 1 : 0; }
  if (nt == 0) return;
  const int G = nt * g.nk;                         // stages (K tiles) of this workgroup

  if (wid >= 4) {
    // ------------------------------------------------------------------ loaders: 12 LDS-DMA pieces (1 KiB each) per wave and stage
    const int lw = wid - 4;
    DmaBuf dA = dma_buf_make(reinterpret_cast<const double*>(g.A), g.lda / 4);
    DmaBuf dB = dma_buf_make(reinterpret_cast<const double*>(g.B), g.ldb / 4);
    int li, lti = 0, ltj = 0, lkt = 0, issued = 0, sl = 0;      // issue cursor: step, tile, K tile; stages issued; ring slot of the next stage
    li = v2_next(g, xcd, slot, round, 0, lti, ltj);
    constexpr int NPF = PFI > 0 ? 32 / PFI : 0;                // C columns (1 KiB each: 256 rows) per loader wave and K tile
    __amdgpu_buffer_rsrc_t rC = dA.rsrc;
    const bool pf_on = PFI > 0 && g.nk >= PFI;
    int pf_last = 0;
    auto retarget = [&]() {
      dA.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + (int64_t)lti * 256 * g.lda), 0, (int)0xffffffffu, 0x00020000);
      dB.rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.B + (int64_t)ltj * 128 * g.ldb), 0, (int)0xffffffffu, 0x00020000);
      if (PFI > 0) rC = __builtin_amdgcn_make_buffer_rsrc((void*)(g.C + (int64_t)lti * 256 + (int64_t)ltj * 128 * g.ldc), 0, (int)0xffffffffu, 0x00020000);
    };
    retarget();
    auto issue = [&]() {
      char* st = lds + sl * V2_STAGE;
      const uint32_t kb = (uint32_t)lkt * 128u;
      pf_last = 0;
      if (PFI > 0 && pf_on && lkt >= g.nk - PFI) {                // C columns [32 lw + NPF (lkt - (nk - PFI)), + NPF) of this tile, BEFORE the stage's pieces
        const uint32_t c0 = (uint32_t)(lw * 32 + NPF * (lkt - (g.nk - PFI)));
#pragma unroll
        for (int q = 0; q < NPF; q++)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rC, (__attribute__((address_space(3))) void*)(lds + 3 * V2_STAGE), 16, lane * 16,
                                                   (int)((c0 + q) * (uint32_t)(g.ldc * 4)), 0, 0);
        pf_last = 1;
      }
#pragma unroll
      for (int q = 0; q < 8; q++) {
        const uint32_t g8 = (uint32_t)(lw * 8 + q);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dA.rsrc, (__attribute__((address_space(3))) void*)(st + g8 * 1024), 16,
                                                 (int)((q & 1) ? dA.voff_odd : dA.voff_even), (int)(g8 * dA.rowgrp + kb), 0, 0);
      }
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const uint32_t g8 = (uint32_t)(lw * 4 + q);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dB.rsrc, (__attribute__((address_space(3))) void*)(st + V2_AIMG + g8 * 1024), 16,
                                                 (int)((q & 1) ? dB.voff_odd : dB.voff_even), (int)(g8 * dB.rowgrp + kb), 0, 0);
      }
      issued++; sl = sl == 2 ? 0 : sl + 1;
      if (++lkt == g.nk) { lkt = 0; li = v2_next(g, xcd, slot, round, li + 1, lti, ltj); if (li < g.tpw) retarget(); }
    };
    issue();
    if (G > 1) issue();
    for (int gs = 0; gs < G; gs++) {
      // stage gs has landed once at most the pieces of stage gs + 1 (12 of mine) are outstanding
      // (everything issued behind stage gs's last piece: the C prefetches + 12 pieces of the most recent issue)
      if (issued > gs + 1) {
        if (PFI > 0 && pf_last) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(12 + NPF) : "memory");
        else asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
      } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_barrier" ::: "memory");                   // B(gs): stage gs is visible; the readers have left slot (gs - 1) % 3
      if (issued < G) {                                          // stage gs + 2 into that slot
        if ((DBG & 2) && issued >= 3) issued++;
        else issue();
      }
    }
    asm volatile("s_barrier" ::: "memory");                       // B(G): pairs with the compute waves' barrier behind their last stage
    return;
  }

  // -------------------------------------------------------------------- compute waves: 128 x 64 of the tile each
  const int wi = (wid & 1) * 128, wj = (wid >> 1) * 64;
  const int r32 = lane & 31, kg = lane >> 5;
  const int t = kg ^ ((r32 >> 1) & 7);
  const int a_off = (wi + r32) * 128, b_off = V2_AIMG + (wj + r32) * 128;
  f32x16 acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 2; j++)
#pragma unroll
      for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
  bf16x8 fa0[4], fa1[4], fb0[2], fb1[2];
  auto rd = [&](const char* st, int ks, bf16x8 (&xa)[4], bf16x8 (&xb)[2]) {
    const int off = ((2 * ks) ^ t) << 4;
#pragma unroll
    for (int i = 0; i < 4; i++) xa[i] = *reinterpret_cast<const bf16x8*>(st + a_off + i * 32 * 128 + off);
#pragma unroll
    for (int j = 0; j < 2; j++) xb[j] = *reinterpret_cast<const bf16x8*>(st + b_off + j * 32 * 128 + off);
  };
  auto mma = [&](const bf16x8 (&xa)[4], const bf16x8 (&xb)[2]) {
    if (SCHED) __builtin_amdgcn_sched_barrier(0);
    if (DBG & 4) {
#pragma unroll
      for (int i = 0; i < 4; i++) asm volatile("" :: "v"(xa[i]));
#pragma unroll
      for (int j = 0; j < 2; j++) asm volatile("" :: "v"(xb[j]));
    } else {
#pragma unroll
      for (int i = 0; i < 4; i++)
#pragma unroll
        for (int j = 0; j < 2; j++) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(xb[j], xa[i], acc[i][j], 0, 0, 0);   // swapped: lane = C row
    }
    if (SCHED) __builtin_amdgcn_sched_barrier(0);
  };
  int ci, cti = 0, ctj = 0, ckt = 0, sl = 0;
  ci = v2_next(g, xcd, slot, round, 0, cti, ctj);
  asm volatile("s_barrier" ::: "memory");                       // B(0)
  rd(lds, 0, fa0, fb0);
  for (int gs = 0; gs < G; gs++) {
    const char* st = lds + sl * V2_STAGE;
    rd(st, 1, fa1, fb1); mma(fa0, fb0);
    rd(st, 2, fa0, fb0); mma(fa1, fb1);
    rd(st, 3, fa1, fb1); mma(fa0, fb0);
    sl = sl == 2 ? 0 : sl + 1;
    {
      // every fragment read of this stage has returned before the loaders may refill its slot.  The barrier is UNCONDITIONAL (the
      // loaders run one extra one after their last stage; behind the last stage these reads fetch stale bytes nobody uses): with a
      // barrier-free path joining here hipcc would re-wait for the fragments of k-step 3 - i.e. for the reads just issued
      // (the wait is the BUILTIN so that hipcc's own counter model knows the fragments of k-step 3 have landed - behind an asm wait
      //  it re-waits for them after the barrier, i.e. for the next stage's first reads, in front of the MFMAs meant to cover those)
      __builtin_amdgcn_s_waitcnt(0xc07f);                                  // lgkmcnt(0)
      asm volatile("s_barrier" ::: "memory");                              // B(gs + 1)
      rd(lds + sl * V2_STAGE, 0, fa0, fb0);
    }
    mma(fa1, fb1);
    if (++ckt == g.nk) {
      // tile done: fire-and-forget atomics (lane = 32 consecutive rows of a column per half wave -> whole 128-B lines), then on to
      // the next tile; the first fragments of its stage 0 are already on their way
      const int64_t i0 = (int64_t)cti * 256, j0 = (int64_t)ctj * 128;
      const int dd = g.tri ? (int)(j0 - i0) : 1 << 30;           // columns lead the rows by dd: mask row <= col  <=>  lrow <= lcol + dd
      const bool diag = g.tri && dd < 256;                       // wave-uniform: only the two tiles per tile row that touch the diagonal
      float* Cl = g.C + (i0 + wi + r32) + (j0 + wj + 4 * kg) * g.ldc;      // this lane's first element
      if (DBG & 1) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) { asm volatile("" :: "v"(acc[i][j][e])); acc[i][j][e] = 0.0f; }
      } else if (!diag) {
#pragma unroll
        for (int i = 0; i < 4; i++)
#pragma unroll
          for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
              __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(Cl + 32 * i + (int64_t)(32 * j + (e & 3) + 8 * (e >> 2)) * g.ldc),
                                                      g.alpha * acc[i][j][e]);
              acc[i][j][e] = 0.0f;
            }
      } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
          const int lrow = wi + 32 * i + r32;
#pragma unroll
          for (int j = 0; j < 2; j++)
#pragma unroll
            for (int e = 0; e < 16; e++) {
              const int lcol = wj + 32 * j + (e & 3) + 8 * (e >> 2) + 4 * kg;
              if (lrow <= lcol + dd)
                __builtin_amdgcn_global_atomic_fadd_f32((__attribute__((address_space(1))) float*)(Cl + 32 * i + (int64_t)(32 * j + (e & 3) + 8 * (e >> 2)) * g.ldc),
                                                        g.alpha * acc[i][j][e]);
              acc[i][j][e] = 0.0f;
            }
        }
      }
      ckt = 0;
      ci = v2_next(g, xcd, slot, round, ci + 1, cti, ctj);
    }
  }
}

// process-wide choice of the update kernel: 0 = round-3 kernel only, 1 = second-generation kernel wherever it applies (default),
// tpw = supertile steps per workgroup
// Default 0: measured inside the N = 65536 factorization (profiles/r04_experiments.log) the new kernel's updates run at 700-720 TF
// against 630 for the old one, but its workgroups hold 145 KiB of LDS for a whole chunk, the diagonal-block chain finds no CU to
// start on, and the factorization - bound by that chain (panel stream) at every strip - gets 5 % SLOWER (211 vs 199 ms).  Stand-alone
// (update-bound callers, K >= 2048) it is the faster kernel: 855 vs 770 TF at K = 2048, 1017 vs 894 at K = 4096.
static int g_bf16_variant = CAP_ENV("CAP_BF16_V2") ? atoi(CAP_ENV("CAP_BF16_V2")) : 6;
static int g_bf16_tpw = CAP_ENV("CAP_BF16_TPW") ? atoi(CAP_ENV("CAP_BF16_TPW")) : 8;
static int64_t g_bf16_min_tiles = CAP_ENV("CAP_BF16_V2_MIN") ? atoll(CAP_ENV("CAP_BF16_V2_MIN")) : 1024;

static int64_t g_bf16_v3_min = 64;      // smallest launch (in 256 x 256 tiles) the third-generation kernel takes
static int64_t g_bf16_head_min = 32;     // smallest panel-stream update it takes (-1: none, the 128-tile kernel)
static int g_bf16_v3_st = 4;             // its supertile edge in tiles
static int g_bf16_sched = CAP_ENV("CAP_BF16_SCHED") ? atoi(CAP_ENV("CAP_BF16_SCHED")) : 1;
static int g_bf16_dbg = 0;              // timing surgery (CAP_EXPERIMENTS builds): set through cap_bf16_update(variant = 100 + DBG)

template <int SCHED, int DBG, int PFI = 0>
int launch_bf16_v2_t(const Bf2Args& g, unsigned grid, hipStream_t s) {
  static bool attr_set[16] = {};                            // per device: the attribute belongs to the device's copy of the function
  constexpr int LDS_BYTES = 3 * V2_STAGE + 1024;            // ring + the dummy slot of the C prefetch
  int dev = 0;
  CAP_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= 16 || !attr_set[dev]) {
    CAP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(bf16_tn_v2_kernel<SCHED, DBG, PFI>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    if (dev >= 0 && dev < 16) attr_set[dev] = true;
  }
  hipLaunchKernelGGL((bf16_tn_v2_kernel<SCHED, DBG, PFI>), dim3(grid), dim3(512), LDS_BYTES, s, g);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

int launch_bf16_v2(int64_t m, int64_t n, int64_t k, float alpha, const __bf16* A, int64_t lda, const __bf16* B, int64_t ldb, float* C, int64_t ldc,
                   int tri, hipStream_t s) {
  Bf2Args g;
  g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.alpha = alpha; g.tri = tri ? 1 : 0;
  g.tm = (int)(m / 256); g.tn = (int)(n / 128); g.nk = (int)(k / 64);
  const int64_t nsi = cap_ceil_div(g.tm, 4), nsj = cap_ceil_div(g.tn, 8);
  g.nsi = (int)nsi;
  g.nsuper = (int)(tri ? nsj * (nsj + 1) / 2 : nsi * nsj);
  const int64_t per_xcd = cap_ceil_div(g.nsuper, 8);
  g.tpw = (int)std::max<int64_t>(1, std::min<int64_t>(g_bf16_tpw, per_xcd));
  const int64_t rounds = cap_ceil_div(per_xcd, g.tpw);
  const unsigned grid = (unsigned)(256 * rounds);
  cap_acc_r(A, lda, k, m, 0, 2); cap_acc_r(B, ldb, k, n, 0, 2); cap_acc(CAP_ACC_ATOMIC, C, ldc, m, n, tri ? 1 : 0, 4);
  if constexpr (CAP_EXPERIMENTS) {
    switch (g_bf16_dbg) {
      case 1: return launch_bf16_v2_t<1, 1>(g, grid, s);
      case 2: return launch_bf16_v2_t<1, 2>(g, grid, s);
      case 3: return launch_bf16_v2_t<1, 3>(g, grid, s);
      case 4: return launch_bf16_v2_t<1, 4>(g, grid, s);
      case 6: return launch_bf16_v2_t<1, 6>(g, grid, s);
      case 16: return launch_bf16_v2_t<1, 0, 2>(g, grid, s);       // C prefetch over the last 2 / 4 / 8 K tiles
      case 17: return launch_bf16_v2_t<1, 0, 4>(g, grid, s);
      case 18: return launch_bf16_v2_t<1, 0, 8>(g, grid, s);
      case 20: return launch_bf16_v2_t<1, 4, 4>(g, grid, s);       // ... without the MFMAs
      default: break;
    }
  }
  return g_bf16_sched ? launch_bf16_v2_t<1, 0>(g, grid, s) : launch_bf16_v2_t<0, 0>(g, grid, s);
}

// dispatcher of the single-GPU updates: the second-generation kernel needs whole 256 x 128 tiles and enough of them to fill the chip
int launch_bf16_update(int64_t m, int64_t n, int64_t k, float alpha, const __bf16* A, int64_t lda, const __bf16* B, int64_t ldb, float* C,
                       int64_t ldc, int tri, hipStream_t s) {
  if (m <= 0 || n <= 0 || k <= 0) return CAP_OK;
  const bool ok2 = g_bf16_variant == 1 && m % 256 == 0 && n % 128 == 0 && k % 64 == 0 && lda % 8 == 0 && ldb % 8 == 0 && (!tri || m == n) &&
                   (m / 256) * (n / 128) / (tri ? 2 : 1) >= g_bf16_min_tiles &&
                   256 * lda * 2 + k * 2 < 0xfffffff0LL && 128 * ldb * 2 + k * 2 < 0xfffffff0LL;
  if (ok2) return launch_bf16_v2(m, n, k, alpha, A, lda, B, ldb, C, ldc, tri, s);
  // third generation (bf16_tn3.hip): whole 256 x 256 tiles and at least g_bf16_v3_min of them (below that the 128-wide tiles fill the chip better)
  if (g_bf16_variant >= 3 && cap_bf16_tn3_applies(m, n, k, lda, ldb, tri, g_bf16_variant) && (m / 256) * (n / 256) / ((tri && m == n) ? 2 : 1) >= g_bf16_v3_min)
    return cap_bf16_tn3_launch(m, n, k, alpha, A, lda, B, ldb, C, ldc, tri, g_bf16_variant, g_bf16_v3_st, s);
  return launch_bf16_tn(m, n, k, alpha, A, lda, B, ldb, C, ldc, tri, s);
}
// the panel-stream updates (heads, in-strip updates): the same dispatcher with its own threshold (g_bf16_head_min tiles of 256 x 256)
int launch_bf16_head(int64_t m, int64_t n, int64_t k, float alpha, const __bf16* A, int64_t lda, const __bf16* B, int64_t ldb, float* C,
                     int64_t ldc, int tri, hipStream_t s) {
  if (g_bf16_variant >= 3 && g_bf16_head_min >= 0 && cap_bf16_tn3_applies(m, n, k, lda, ldb, tri, g_bf16_variant) &&
      (m / 256) * (n / 256) / ((tri && m == n) ? 2 : 1) >= g_bf16_head_min)
    return cap_bf16_tn3_launch(m, n, k, alpha, A, lda, B, ldb, C, ldc, tri, g_bf16_variant, g_bf16_v3_st, s);
  return launch_bf16_tn(m, n, k, alpha, A, lda, B, ldb, C, ldc, tri, s);
}
}  // namespace

// Testing / benchmarking entry of the bf16 update (tests/test_gpu_mixed.py, tools/bf16_bench.py): C32[m x n] += alpha A^T B with A: k x m,
// B: k x n bf16 (K-contiguous, lda / ldb elements), upper triangle only when tri.  variant 0 = round-3 kernel, 1 = second generation
// (CAP_ERR_UNSUPPORTED where it does not apply), -1 = the dispatcher the factorization uses.  tpw > 0 overrides the chunk length.
extern "C" int cap_bf16_update(int variant, int64_t m, int64_t n, int64_t k, float alpha, const void* A16, int64_t lda, const void* B16, int64_t ldb,
                               float* C, int64_t ldc, int tri, int tpw, void* stream) {
  if (!A16 || !B16 || !C || m < 0 || n < 0 || k < 0) return CAP_ERR_ARG;
  if (tpw > 0 && (variant < 3 || (variant > 6 && variant < 300))) g_bf16_tpw = tpw;
  const __bf16* A = (const __bf16*)A16; const __bf16* B = (const __bf16*)B16;
  hipStream_t s = cap_stream(stream);
  if (variant < 0) return launch_bf16_update(m, n, k, alpha, A, lda, B, ldb, C, ldc, tri, s);
  if (variant == 0) return launch_bf16_tn(m, n, k, alpha, A, lda, B, ldb, C, ldc, tri, s);
  if (variant >= 3 && variant <= 6)      // third generation (6: wide staging with 32-MFMA phases): LDS ring of 3 / 4 stages of 32 k, 5 = two stages of 64 k; tpw carries the supertile edge here (0: default)
    return cap_bf16_tn3_launch(m, n, k, alpha, A, lda, B, ldb, C, ldc, tri, variant, tpw > 0 ? tpw : g_bf16_v3_st, s);
  if (variant >= 300 && variant < 316)   // timing surgery on the third generation (experiment builds only): 300 + DBG; 500 + DBG: wide staging
    return cap_bf16_tn3_launch(m, n, k, alpha, A, lda, B, ldb, C, ldc, tri, 3, tpw > 0 ? tpw : g_bf16_v3_st, s, variant - 300);
  if (variant >= 500 && variant < 516)
    return cap_bf16_tn3_launch(m, n, k, alpha, A, lda, B, ldb, C, ldc, tri, 5, tpw > 0 ? tpw : g_bf16_v3_st, s, variant - 500);
  if (variant >= 600 && variant < 616)
    return cap_bf16_tn3_launch(m, n, k, alpha, A, lda, B, ldb, C, ldc, tri, 6, tpw > 0 ? tpw : g_bf16_v3_st, s, variant - 600);
  if (m % 256 || n % 128 || k % 64 || lda % 8 || ldb % 8 || (tri && m != n) || m == 0 || n == 0 || k == 0) return CAP_ERR_UNSUPPORTED;
  // variant 1: the production schedule; 2: without the forced read-ahead; 100 + DBG: timing surgery (experiment builds only)
  g_bf16_sched = variant == 2 ? 0 : 1;
  g_bf16_dbg = variant >= 100 ? variant - 100 : 0;
  if (g_bf16_dbg && !CAP_EXPERIMENTS) return CAP_ERR_UNSUPPORTED;
  const int st = launch_bf16_v2(m, n, k, alpha, A, lda, B, ldb, C, ldc, tri, s);
  g_bf16_sched = 1; g_bf16_dbg = 0;
  return st;
}

// Distributed bf16 trailing update on a 1 x P block-column-cyclic fp32 matrix (dist_mixed.hip): C32[m x nloc] -= G^T B restricted
// to the global upper triangle.  G: gathered bf16 block row (P pieces of `piece` elements, each k x cols_r, ld = k), B: my own
// columns of it (k x nloc, ld = k), C: my local columns from local block lb0 on, rows global from block J0 on (ldc).
int cap_bf16_dist_update_launch(int64_t m, int64_t nloc, int64_t k, const void* G16, int64_t piece, const int* gstart, const void* B16,
                                float* C, int64_t ldc, int P, int p, int nb, int J0, int lb0, hipStream_t s) {
  if (m <= 0 || nloc <= 0) return CAP_OK;
  if ((m % TB) || (nloc % TB) || (k % KB) || (nb % TB) || P < 1 || P > 8) return CAP_ERR_UNSUPPORTED;
  if (128 * k * 2 + k * 2 >= 0xfffffff0LL) return CAP_ERR_UNSUPPORTED;
  BfArgs g;
  g.A = (const __bf16*)G16; g.B = (const __bf16*)B16; g.C = C; g.lda = k; g.ldb = k; g.ldc = ldc; g.M = m; g.N = nloc; g.K = k;
  g.alpha = -1.0f; g.tri = 1; g.tm = (int)(m / TB); g.tn = (int)(nloc / TB);
  g.stair = 1; g.sP = P; g.sp = p; g.snbT = nb / TB; g.sJ0 = J0; g.slb0 = lb0; g.gpiece = piece; g.st = 0; g.nsm = 1;
  for (int i = 0; i < 8; i++) g.gstart[i] = i < P ? gstart[i] : 0;
  g.chunk = (int)cap_ceil_div((int64_t)g.tm * g.tn, 8);       // full grid; workgroups below the staircase return at once
  if (cap_acc_on()) {
    // access notes (cf. cap_dist_update_launch, gemm.hip): per local column block J the row blocks I < J whole and I == J above the
    // diagonal (fp32 atomics), the gathered operand's chunk of every row block that has a partner, my own columns whole
    const int64_t ncb = nloc / nb, nrb = m / nb;
    const int64_t Jlast = p + (int64_t)P * (lb0 + ncb - 1);
    for (int64_t cb = 0; cb < ncb; cb++) {
      const int64_t J = p + (int64_t)P * (lb0 + cb);
      const int64_t full = std::max<int64_t>(0, std::min<int64_t>(nrb, J - J0));
      cap_acc(CAP_ACC_ATOMIC, C + cb * nb * ldc, ldc, full * nb, nb, 0, 4);
      if (J - J0 >= 0 && J - J0 < nrb) cap_acc(CAP_ACC_ATOMIC, C + full * nb + cb * nb * ldc, ldc, nb, nb, 1, 4);
    }
    for (int64_t rb = 0; rb < nrb; rb++) {
      const int64_t I = J0 + rb;
      if (I > Jlast) break;
      const int64_t r = I % P, lb = I / P - gstart[r];
      cap_acc_r((const __bf16*)G16 + r * piece + lb * nb * k, 0, k * nb, 1, 0, 2);
    }
    cap_acc_r(B16, 0, k * nloc, 1, 0, 2);
  }
  launch_bf16_tn_kernel(g, (unsigned)(g.chunk * 8), s);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

struct cap_mpchol_plan {
  int64_t n, nb, nrhs_cap;          // nrhs_cap: internal right-hand-side width (multiple of 128)
  float* R32; double* R64; __bf16* P16[2]; int64_t strip;
  // column-split schedule (cap_mpchol_factor): third stream, its events, near-column solve scratch (2 x nb x nb)
  int split; bool split_ready; hipStream_t s_far; hipEvent_t ev_c, ev_ns, ev_hf, ev_fs[2], ev_far[2], ev_join2; double* Tn;
  // option "reserve" = r > 0: r CUs (bit i of the mask -> XCD i % 8) are kept free of everything but the diagonal-block chains:
  // the updates run on a stream masked OFF those CUs, the panel / far streams too, the chain's latency-bound launches (leaf, fused
  // 64-steps, inverse merges: <= 15 workgroups of 84 KiB LDS each) on a stream masked to exactly those CUs - they start at once and
  // never share a SIMD with bf16-MFMA waves (measured under contention: 91 us per fused step against 30 us alone, 88 of a 198 ms factor)
  int reserve; hipStream_t s_bulk, s_chain; hipEvent_t ev_user, ev_bulk_done, ev_ch[2]; bool res_ready;
  int chain_coop;                   // resident workgroups of the one-launch diagonal-block chain for THIS plan (-1: process default)
  int pair_rest;                    // split schedule: the far part of the bulk update takes two strips at a time (K = 4 nb), see mp_factor_impl
  int cnt_paired;                   // ... how many such launches the last factor call made
  // block-row solve on the bf16 pipe (option "solve3", default on): split operands, see mixed_kernels.h
  int solve3; __bf16* A3[2]; __bf16* B3far; __bf16* B3near;
  hipStream_t s_panel; hipEvent_t ev_rest[2], ev_panel[2], ev_fork, ev_join; bool streams_ready;
  double* D64; double* Dinv; double* T64; double* S64; double* W; int64_t wcap;
  double* Inv; int64_t tb; double* Xt; double* Wt;          // blocked TRSM state
  double* Xw; double* Rw; double* Bw; double* norms;
  int* info_dev; bool have_r64;
  // live profile of the bf16 trailing updates on the caller's stream (HIP events; cap_mpchol_profile)
  int profile; std::vector<hipEvent_t>* prof_ev; std::vector<double>* prof_flops; std::vector<double>* prof_bytes; int prof_used;
};

namespace {
// a helper stream of the factorization: high priority on the whole chip, or - reserve mode - masked off the chain's CUs
int mp_make_stream(cap_mpchol_plan* p, hipStream_t* out) {
  if (p->reserve > 0) {
    uint32_t mask[8];
    for (int i = 0; i < 8; i++) mask[i] = 0xffffffffu;
    for (int b = 0; b < p->reserve && b < 128; b++) mask[b / 32] &= ~(1u << (b % 32));
    CAP_HIP(hipExtStreamCreateWithCUMask(out, 8, mask));
    return CAP_OK;
  }
  int lo = 0, hi = 0;
  CAP_HIP(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CAP_HIP(hipStreamCreateWithPriority(out, hipStreamNonBlocking, hi));
  return CAP_OK;
}
int mp_ensure_reserved(cap_mpchol_plan* p) {
  if (p->res_ready || p->reserve <= 0) return CAP_OK;
  CAP_TRY(mp_make_stream(p, &p->s_bulk));
  uint32_t mask[8];
  for (int i = 0; i < 8; i++) mask[i] = 0u;
  for (int b = 0; b < p->reserve && b < 128; b++) mask[b / 32] |= 1u << (b % 32);
  CAP_HIP(hipExtStreamCreateWithCUMask(&p->s_chain, 8, mask));
  for (hipEvent_t* e : {&p->ev_user, &p->ev_bulk_done, &p->ev_ch[0], &p->ev_ch[1]}) CAP_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
  p->res_ready = true;
  return CAP_OK;
}
void mp_release_streams(cap_mpchol_plan* p) {
  if (p->split_ready) {
    (void)hipStreamSynchronize(p->s_far); cap_stream_destroy(p->s_far);
    for (hipEvent_t e : {p->ev_c, p->ev_ns, p->ev_hf, p->ev_fs[0], p->ev_fs[1], p->ev_far[0], p->ev_far[1], p->ev_join2}) (void)hipEventDestroy(e);
    if (p->Tn) (void)hipFree(p->Tn);
    p->Tn = nullptr; p->split_ready = false;
  }
  if (p->streams_ready) {
    (void)hipStreamSynchronize(p->s_panel); cap_stream_destroy(p->s_panel);
    for (int i = 0; i < 2; i++) { (void)hipEventDestroy(p->ev_rest[i]); (void)hipEventDestroy(p->ev_panel[i]); }
    (void)hipEventDestroy(p->ev_fork); (void)hipEventDestroy(p->ev_join);
    p->streams_ready = false;
  }
  if (p->res_ready) {
    (void)hipStreamSynchronize(p->s_bulk); cap_stream_destroy(p->s_bulk);
    (void)hipStreamSynchronize(p->s_chain); cap_stream_destroy(p->s_chain);
    for (hipEvent_t e : {p->ev_user, p->ev_bulk_done, p->ev_ch[0], p->ev_ch[1]}) (void)hipEventDestroy(e);
    p->res_ready = false;
  }
}
}  // namespace

static int mp_factor_impl(cap_mpchol_plan* p, const double* A, int64_t lda, hipStream_t s0);

extern "C" {

int cap_mpchol_plan_create(cap_mpchol_plan** plan, int64_t n, int64_t nrhs_max) {
  if (!plan || n <= 0 || nrhs_max <= 0) return CAP_ERR_ARG;
  if (n % 128) return CAP_ERR_UNSUPPORTED;                     // the bf16 tile kernel works on whole 128 x 128 tiles
  cap_mpchol_plan* p = new (std::nothrow) cap_mpchol_plan();
  if (!p) return CAP_ERR_ALLOC;
  memset(p, 0, sizeof(*p));
  p->n = n; p->nb = 1024; p->nrhs_cap = cap_round_up(nrhs_max, 128);
  p->strip = CAP_ENV("CAP_MP_STRIP") ? atoll(CAP_ENV("CAP_MP_STRIP")) : 2;
  p->split = CAP_ENV("CAP_MP_SPLIT") ? atoi(CAP_ENV("CAP_MP_SPLIT")) : 1;
  p->solve3 = CAP_ENV("CAP_MP_SOLVE3") ? atoi(CAP_ENV("CAP_MP_SOLVE3")) : 1;
  p->reserve = CAP_ENV("CAP_MP_RESERVE") ? atoi(CAP_ENV("CAP_MP_RESERVE")) : 0;
  p->chain_coop = -1;
  p->pair_rest = CAP_ENV("CAP_MP_PAIR_REST") ? atoi(CAP_ENV("CAP_MP_PAIR_REST")) : 1;
  p->cnt_paired = 0;
  // largest power of two <= min(n, 1024) (>= 128 because n % 128 == 0): the fused diagonal-block chain and the bf16 tile
  // kernel (k % 64, m % 128) need it; the last panel of a non-power-of-two n is a shorter multiple of 128
  while (p->nb > n) p->nb /= 2;
  p->wcap = cap_rec_work_size(p->nb);
  p->tb = p->nb;                      // the TRSM blocks ARE the panels: their inverses fall out of the factorization
  const int64_t nb = p->nb, w = p->nrhs_cap, nblk = cap_ceil_div(n, p->tb);
  hipError_t e = hipMalloc((void**)&p->R32, sizeof(float) * n * n);
  if (e == hipSuccess) e = hipMemset(p->R32, 0, sizeof(float) * n * n);        // the strictly-lower part stays zero for the plan's life
  if (e == hipSuccess) e = hipMalloc((void**)&p->R64, sizeof(double) * n * n);
  if (e == hipSuccess) e = hipMemset(p->R64, 0, sizeof(double) * n * n);      // nothing ever writes below its diagonal
  for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMalloc((void**)&p->P16[i], sizeof(__bf16) * 4 * nb * n);     // pair buffers: two strips each, ld = 4 nb (2 nb in the single-stream schedule)
  if (e == hipSuccess) e = hipMalloc((void**)&p->D64, sizeof(double) * (2 * nb * nb + 2 * nb * n + p->wcap));
  if (e == hipSuccess) e = hipMalloc((void**)&p->Inv, sizeof(double) * (nblk * p->tb * p->tb + p->tb * w));
  if (e == hipSuccess) e = hipMalloc((void**)&p->Xw, sizeof(double) * (3 * n * w + 8));
  if (e == hipSuccess) e = hipMalloc((void**)&p->info_dev, sizeof(int));
  for (int i = 0; i < 2 && e == hipSuccess; i++) e = hipMalloc((void**)&p->A3[i], sizeof(__bf16) * 3 * nb * nb);
  if (e == hipSuccess) e = hipMalloc((void**)&p->B3far, sizeof(__bf16) * 3 * nb * n);
  if (e == hipSuccess) e = hipMalloc((void**)&p->B3near, sizeof(__bf16) * 3 * nb * nb);
  if (e != hipSuccess) { cap_mpchol_plan_destroy(p); return CAP_ERR_ALLOC; }
  p->Dinv = p->D64 + nb * nb; p->T64 = p->Dinv + nb * nb; p->S64 = p->T64 + nb * n; p->W = p->S64 + nb * n;
  p->Xt = p->Inv + nblk * p->tb * p->tb; p->Wt = nullptr;
  p->Rw = p->Xw + n * w; p->Bw = p->Rw + n * w; p->norms = p->Bw + n * w;
  *plan = p;
  return CAP_OK;
}

int cap_mpchol_plan_destroy(cap_mpchol_plan* p) {
  if (!p) return CAP_OK;
  if (p->prof_ev) { for (hipEvent_t e : *p->prof_ev) (void)hipEventDestroy(e); delete p->prof_ev; delete p->prof_flops; delete p->prof_bytes; }
  for (void* q : {(void*)p->R32, (void*)p->R64, (void*)p->P16[0], (void*)p->P16[1], (void*)p->D64, (void*)p->Inv, (void*)p->Xw, (void*)p->info_dev,
                  (void*)p->A3[0], (void*)p->A3[1], (void*)p->B3far, (void*)p->B3near})
    if (q) (void)hipFree(q);
  mp_release_streams(p);
  delete p;
  return CAP_OK;
}

// A: n x n fp64, upper triangle consumed.  Leaves the fp32 factor, its fp64 promotion and the TRSM block inverses in the plan.
// Two streams: the caller's carries the bf16 trailing updates, a high-priority one the fp64 panel work of the NEXT panel
// (it only needs the head of the previous update: the rows of that panel), so the latency-bound diagonal-block chain
// and the fp64 row solve hide behind the bf16 MFMA update once that is the longer of the two.
int cap_mpchol_factor(cap_mpchol_plan* p, const double* A, int64_t lda, void* stream) {
  if (!p || !A || lda < p->n) return CAP_ERR_ARG;
  CapChainScope chain_scope(p->chain_coop);     // this plan's own "chain_coop" for every launch made from this call
  hipStream_t su = cap_stream(stream);
  if (p->reserve <= 0) return mp_factor_impl(p, A, lda, su);
  // reserve mode: the whole factorization runs on the plan's masked streams, forked off / joined into the caller's stream
  CAP_TRY(mp_ensure_reserved(p));
  CAP_HIP(hipEventRecord(p->ev_user, su));
  CAP_HIP(hipStreamWaitEvent(p->s_bulk, p->ev_user, 0));
  const int st = mp_factor_impl(p, A, lda, p->s_bulk);
  CAP_HIP(hipEventRecord(p->ev_bulk_done, p->s_bulk));
  CAP_HIP(hipStreamWaitEvent(su, p->ev_bulk_done, 0));
  return st;
}

}  // extern "C"

static int mp_factor_impl(cap_mpchol_plan* p, const double* A, int64_t lda, hipStream_t s0) {
  const int64_t n = p->n, nb = p->nb;
  if (!p->streams_ready) {
    CAP_TRY(mp_make_stream(p, &p->s_panel));
    for (int i = 0; i < 2; i++) {
      CAP_HIP(hipEventCreateWithFlags(&p->ev_rest[i], hipEventDisableTiming));
      CAP_HIP(hipEventCreateWithFlags(&p->ev_panel[i], hipEventDisableTiming));
    }
    CAP_HIP(hipEventCreateWithFlags(&p->ev_fork, hipEventDisableTiming));
    CAP_HIP(hipEventCreateWithFlags(&p->ev_join, hipEventDisableTiming));
    p->streams_ready = true;
  }
  hipStream_t s1 = p->s_panel;
  p->prof_used = 0;
  if (p->prof_flops) { p->prof_flops->clear(); p->prof_bytes->clear(); }
  CAP_HIP(hipMemsetAsync(p->info_dev, 0, sizeof(int), s0));
  launch_f64_to_f32_upper(s0, A, lda, p->R32, n, n);
  CAP_HIP(hipGetLastError());
  CAP_HIP(hipEventRecord(p->ev_fork, s0));
  CAP_HIP(hipStreamWaitEvent(s1, p->ev_fork, 0));

  // Strips of `sp` panels (option "strip", default 2): the bf16 update below a strip contracts K = sp * nb rows at once.  At
  // K = 1024 a 128 x 128 tile holds 3.4 us of bf16 MFMA work against a cold first DMA and 16 K fp32 atomics: the update ran at
  // 0.19 of the bf16 peak and 0.23 of HBM, i.e. on per-tile fixed cost; K = 2048 halves the C traffic and the tile count.
  // Strip buffer (bf16, K-contiguous): element (r, c) = solved row (strip start + r), global column c, at SP[c * ldp + r].
  const int64_t npan = cap_ceil_div(n, nb), sp = std::max<int64_t>(1, std::min<int64_t>(p->strip, 2)), ldp = 2 * nb;
  const int64_t nstrip = cap_ceil_div(npan, sp);

  // ---- column-split schedule (option "split", default): the second half of this factorization is bound by the panel stream
  // (kernel trace, N = 65536: 208 of 247 ms busy - diagonal-block chains 95 ms, fp64 row solves 60 ms, conversions 23 ms, bf16
  // heads 19 ms - while the big updates need 144 ms).  Only the NEXT diagonal block has to be up to date before its chain can
  // start, so every block-row solve and every head update is split into the columns of the next panel ("near", panel stream)
  // and everything to the right ("far", a third stream): the chain of panel k+1 runs next to the far solve / far update of
  // panel k.  Sums per C element keep their order (head of panel a, head of the strip, rest; one launch each), so the factor
  // does not depend on the option bit for bit.
  if (p->split && sp == 2) {
    if (!p->split_ready) {
      CAP_TRY(mp_make_stream(p, &p->s_far));
      for (hipEvent_t* e : {&p->ev_c, &p->ev_ns, &p->ev_hf, &p->ev_fs[0], &p->ev_fs[1], &p->ev_far[0], &p->ev_far[1], &p->ev_join2})
        CAP_HIP(hipEventCreateWithFlags(e, hipEventDisableTiming));
      CAP_HIP(hipMalloc((void**)&p->Tn, sizeof(double) * 2 * nb * nb));
      p->split_ready = true;
    }
    hipStream_t s2 = p->s_far;
    CAP_HIP(hipStreamWaitEvent(s2, p->ev_fork, 0));
    const int64_t ldp = 4 * nb;     // pair buffers (this schedule only; the single-stream schedule below keeps ld = 2 nb)
    double* Tn = p->Tn; double* Sn = p->Tn + nb * nb;
    bool have_hf = false;
    // block-row solve of panel k on the columns [c0, c1): X = Dinv_k^T R32[rows k, c0:c1] in fp64 -> fp32 (factor) + bf16 (strip buffer)
    auto solve_cols = [&](int64_t k, int64_t c0, int64_t c1, double* T, double* S, __bf16* SP, int64_t roff, hipStream_t s) -> int {
      const int64_t j0 = k * nb, jb = std::min(nb, n - j0), w = c1 - c0;
      if (w <= 0) return CAP_OK;
      float* Row32 = p->R32 + j0 + c0 * n;
      if (p->solve3 && jb == nb) {
        // S = Dinv^T Row on the bf16 pipe with split operands (K = 3 jb): the row is split into B3 and cleared, the product is
        // accumulated straight back into it (fp32 atomics of the update kernel), then rounded into the strip buffer
        __bf16* B3 = (T == Tn) ? p->B3near : p->B3far;             // the near / far solves of a panel run side by side
        launch_split3_row(s, Row32, n, B3, jb, w, 1);
        CAP_TRY(launch_bf16_update(jb, w, 3 * jb, 1.0f, p->A3[k & 1], 3 * jb, B3, 3 * jb, Row32, n, 0, s));
        launch_f32_to_bf16(s, Row32, n, SP + roff + c0 * ldp, ldp, jb, w);
        CAP_HIP(hipGetLastError());
        return CAP_OK;
      }
      launch_f32_to_f64(s, Row32, n, T, jb, jb, w, 0);
      CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, jb, w, jb, 1.0, p->Inv + k * nb * nb, nb, T, jb, 0.0, S, jb, 0, s, 2 | 16));
      launch_f64_to_f32_bf16(s, S, jb, Row32, n, SP + roff + c0 * ldp, ldp, jb, w, 0);
      CAP_HIP(hipGetLastError());
      return CAP_OK;
    };
    auto chain = [&](int64_t k) -> int {
      const int64_t j0 = k * nb, jb = std::min(nb, n - j0);
      float* D32 = p->R32 + j0 + j0 * n;
      double* Dinv = p->Inv + k * nb * nb;
      launch_f32_to_f64(s1, D32, n, p->D64, jb, jb, jb, 1);
      CAP_HIP(hipMemsetAsync(Dinv, 0, sizeof(double) * nb * nb, s1));
      if (p->res_ready) {           // the latency-bound launches on the reserved CUs
        CAP_HIP(hipEventRecord(p->ev_ch[0], s1));
        CAP_HIP(hipStreamWaitEvent(p->s_chain, p->ev_ch[0], 0));
        cap_chain_coop_cap(p->reserve);          // no more resident workgroups than the masked stream has CUs
        const int st_chain = cap_rec_cholinv_full(p->D64, jb, Dinv, nb, jb, p->W, p->wcap, p->info_dev, p->s_chain, j0);
        cap_chain_coop_cap(0);
        CAP_TRY(st_chain);
        CAP_HIP(hipEventRecord(p->ev_ch[1], p->s_chain));
        CAP_HIP(hipStreamWaitEvent(s1, p->ev_ch[1], 0));
      } else
      CAP_TRY(cap_rec_cholinv_full(p->D64, jb, Dinv, nb, jb, p->W, p->wcap, p->info_dev, s1, j0));
      launch_f64_to_f32_bf16(s1, p->D64, jb, D32, n, (__bf16*)nullptr, (int64_t)0, jb, jb, 1);
      if (p->solve3 && jb == nb) launch_split3_tri(s1, Dinv, nb, p->A3[k & 1], jb);
      CAP_HIP(hipGetLastError());
      return CAP_OK;
    };
    bool deferred = false;          // paired far update: an even strip left the region below strip t + 2 to the next strip
    p->cnt_paired = 0;
    for (int64_t t = 0; t < nstrip; t++) {
      // strip t lives in pair buffer (t / 2) % 2 at row offset (t % 2) 2 nb (ld = 4 nb): strips 2 j and 2 j + 1 are K-contiguous
      __bf16* SP = p->P16[(t >> 1) & 1] + (t & 1) * 2 * nb;
      const int64_t ka = 2 * t, kb = std::min(npan, ka + 2) - 1;           // first / last panel of the strip (kb == ka: one panel)
      for (int64_t k = ka; k <= kb; k++) {
        const int64_t j1 = std::min(n, (k + 1) * nb), j1n = std::min(n, j1 + nb), roff = (k - ka) * nb;
        CAP_TRY(chain(k));
        CAP_HIP(hipEventRecord(p->ev_c, s1));
        CAP_HIP(hipStreamWaitEvent(s2, p->ev_c, 0));
        // near solve: the rows of panel k have received every earlier update on the near columns once the last far update is in
        if (have_hf) CAP_HIP(hipStreamWaitEvent(s1, p->ev_hf, 0));
        CAP_TRY(solve_cols(k, j1, j1n, Tn, Sn, SP, roff, s1));
        CAP_HIP(hipEventRecord(p->ev_ns, s1));
        CAP_TRY(solve_cols(k, j1n, n, p->T64, p->S64, SP, roff, s2));
        CAP_HIP(hipEventRecord(p->ev_fs[k & 1], s2));
        {
          // fp64 promotion of the finished block row for the refinement sweeps, off the critical path (far stream) - it used to be one
          // 15 ms pass over the whole factor behind the last panel
          const int64_t j0 = k * nb, jb = std::min(nb, n - j0);
          if (n > j1n) launch_f32_to_f64(s2, p->R32 + j0 + j1n * n, n, p->R64 + j0 + j1n * n, n, jb, n - j1n, 0);
          CAP_HIP(hipStreamWaitEvent(s2, p->ev_ns, 0));
          launch_f32_to_f64(s2, p->R32 + j0 + j0 * n, n, p->R64 + j0 + j0 * n, n, jb, jb, 1);
          if (j1n > j1) launch_f32_to_f64(s2, p->R32 + j0 + j1 * n, n, p->R64 + j0 + j1 * n, n, jb, j1n - j1, 0);
          CAP_HIP(hipGetLastError());
        }
        if (k < kb) {
          // head of panel a inside the strip: rows of panel b.  near = its diagonal block, far = the rectangle to the right
          const int64_t rb = j1n - j1;
          const __bf16* Pa = SP + roff;                                    // this panel's rows of the strip buffer, column c at c * ldp
          CAP_TRY(launch_bf16_head(rb, rb, nb, -1.0f, Pa + j1 * ldp, ldp, Pa + j1 * ldp, ldp, p->R32 + j1 + j1 * n, n, 1, s1));
          CAP_HIP(hipStreamWaitEvent(s2, p->ev_ns, 0));
          if (n > j1n) CAP_TRY(launch_bf16_head(rb, n - j1n, nb, -1.0f, Pa + j1 * ldp, ldp, Pa + j1n * ldp, ldp, p->R32 + j1 + j1n * n, n, 0, s2));
          CAP_HIP(hipEventRecord(p->ev_hf, s2)); have_hf = true;
        }
      }
      const int64_t Js = std::min(n, (kb + 1) * nb), m = n - Js;
      CAP_HIP(hipEventRecord(p->ev_panel[t & 1], s1));                      // near solves of the strip are in
      CAP_HIP(hipEventRecord(p->ev_far[t & 1], s2));                        // ... and the far ones
      if (m <= 0) break;
      const int64_t K = Js - ka * nb;                                      // rows of the strip (2 nb: a ragged strip is the last one)
      const __bf16* S = SP + Js * ldp;
      const int64_t hb = std::min(2 * nb, m), ra = std::min(nb, m), rbn = hb - ra;    // rows of the next strip / its two panels
      // head of the strip on the next strip's rows.  near: the diagonal block of its first panel (panel stream: the next chain
      // waits for nothing else); far: the rest of those rows.  Both wait for rest(t-1), which touched the same rows.
      if (t > 0) { CAP_HIP(hipStreamWaitEvent(s1, p->ev_rest[(t - 1) & 1], 0)); CAP_HIP(hipStreamWaitEvent(s2, p->ev_rest[(t - 1) & 1], 0)); }
      if (kb > ka) CAP_HIP(hipStreamWaitEvent(s1, p->ev_fs[ka & 1], 0));    // the first panel's rows at these columns come from its FAR solve
      CAP_TRY(launch_bf16_head(ra, ra, K, -1.0f, S, ldp, S, ldp, p->R32 + Js + Js * n, n, 1, s1));
      CAP_HIP(hipStreamWaitEvent(s2, p->ev_ns, 0));                         // the last panel's rows at the near columns (A operand of rows a')
      if (m > ra) {
        CAP_TRY(launch_bf16_head(ra, m - ra, K, -1.0f, S, ldp, S + ra * ldp, ldp, p->R32 + Js + (Js + ra) * n, n, 0, s2));
        if (rbn > 0) CAP_TRY(launch_bf16_head(rbn, m - ra, K, -1.0f, S + ra * ldp, ldp, S + ra * ldp, ldp, p->R32 + (Js + ra) * (n + 1), n, 1, s2));
      }
      CAP_HIP(hipEventRecord(p->ev_hf, s2)); have_hf = true;
      // rest: rows below the next strip, caller's stream
      CAP_HIP(hipStreamWaitEvent(s0, p->ev_panel[t & 1], 0));
      CAP_HIP(hipStreamWaitEvent(s0, p->ev_far[t & 1], 0));
      bool rest_recorded = false, head_only = false;
      if (m > hb) {
        hipEvent_t e0 = nullptr, e1 = nullptr;
        if (p->profile && p->prof_ev) {
          if ((size_t)p->prof_used + 2 > p->prof_ev->size())
            for (int i = 0; i < 64; i++) { hipEvent_t e; CAP_HIP(hipEventCreate(&e)); p->prof_ev->push_back(e); }
          e0 = (*p->prof_ev)[p->prof_used]; e1 = (*p->prof_ev)[p->prof_used + 1];
          CAP_HIP(hipEventRecord(e0, s0));
        }
        // Paired far update (round 5, option "pair_rest"; the fp64 plan's right_looking does the same): the rows below strip t + 2 are
        // not read before strip t + 2 is factored, so an EVEN strip t only brings strip t + 2's rows up to date (one more head, K = 2 nb)
        // and leaves the region below to the ODD strip t + 1, whose own rest is exactly that region: ONE product with K = 4 nb over
        // the two strips (the halves of a pair buffer are K-contiguous) - half the C traffic and half the tiles' prologue / epilogue
        // for the bulk of the flops.
        const int64_t m3 = m - hb, J3 = Js + hb;
        int64_t Kr = K; const __bf16* Sr = S + hb * ldp;                     // operand of the rest launch
        bool launch_rest = true;
        if (deferred) {
          // rows of strip t - 1 and strip t.  The rows of strip t + 2 come first and release the other streams: the head of strip t + 1
          // only has to wait for THEM (it touches the same rows), not for the whole K = 4 nb launch - without this second level of
          // look-ahead the paired launch (twice a plain rest) and the next strip's panel work took turns instead of overlapping and
          // the factorization gained nothing from a bulk that ran 15 % faster (profiles/r05_mixed_pair_first.log)
          Kr = K + 2 * nb; Sr = p->P16[(t >> 1) & 1] + J3 * ldp;
          const int64_t hh = std::min<int64_t>(2 * nb, m3);
          CAP_TRY(launch_bf16_head(hh, m3, Kr, -1.0f, Sr, ldp, Sr, ldp, p->R32 + J3 * (n + 1), n, 1, s0));
          CAP_HIP(hipEventRecord(p->ev_rest[t & 1], s0)); rest_recorded = true;
          launch_rest = false;
          if (m3 > hh) CAP_TRY(launch_bf16_update(m3 - hh, m3 - hh, Kr, -1.0f, Sr + hh * ldp, ldp, Sr + hh * ldp, ldp, p->R32 + (J3 + hh) * (n + 1), n, 1, s0));
          deferred = false; p->cnt_paired++;
        } else if (p->pair_rest && (t & 1) == 0 && K == 2 * nb && hb == 2 * nb && m3 > 2 * nb) {
          // head of strip t on strip t + 2's rows; the region below waits for strip t + 1
          CAP_TRY(launch_bf16_head(2 * nb, m3, K, -1.0f, Sr, ldp, Sr, ldp, p->R32 + J3 * (n + 1), n, 1, s0));
          deferred = true; launch_rest = false; head_only = true;
        }
        if (launch_rest) CAP_TRY(launch_bf16_update(m3, m3, Kr, -1.0f, Sr, ldp, Sr, ldp, p->R32 + J3 * (n + 1), n, 1, s0));
        if (e0) {
          CAP_HIP(hipEventRecord(e1, s0));
          p->prof_used += 2;
          // (an even strip of a pair: its launch is the head on strip t + 2's rows, 2 nb x m3 of the upper staircase)
          const double mm = (double)m3, hh = (double)(2 * nb);
          const double elems = head_only ? hh * mm - 0.5 * hh * (hh - 1.0) : 0.5 * mm * (mm + 1.0);
          p->prof_flops->push_back(2.0 * (double)Kr * elems);
          p->prof_bytes->push_back(8.0 * elems + 2.0 * (double)Kr * mm);
        }
      }
      if (!rest_recorded) CAP_HIP(hipEventRecord(p->ev_rest[t & 1], s0));
    }
    CAP_HIP(hipEventRecord(p->ev_join2, s2));
    CAP_HIP(hipStreamWaitEvent(s0, p->ev_join2, 0));
    CAP_HIP(hipEventRecord(p->ev_join, s1));
    CAP_HIP(hipStreamWaitEvent(s0, p->ev_join, 0));
    p->have_r64 = true;                        // promoted block row by block row above
    return CAP_OK;
  }

  // fp64 panel work of panel k on stream s: diagonal block (R_kk, inverse), block row solve; writes rows [roff, roff + jb) of SP
  auto panel = [&](int64_t k, hipStream_t s, __bf16* SP, int64_t roff) -> int {
    const int64_t j0 = k * nb, jb = std::min(nb, n - j0), j1 = j0 + jb, m = n - j1;
    float* D32 = p->R32 + j0 + j0 * n;
    double* Dinv = p->Inv + k * nb * nb;                  // kept: the diagonal-block inverse of the blocked TRSM
    launch_f32_to_f64(s, D32, n, p->D64, jb, jb, jb, 1);
    CAP_HIP(hipMemsetAsync(Dinv, 0, sizeof(double) * nb * nb, s));
    CAP_TRY(cap_rec_cholinv_full(p->D64, jb, Dinv, nb, jb, p->W, p->wcap, p->info_dev, s, j0));
    launch_f64_to_f32_bf16(s, p->D64, jb, D32, n, (__bf16*)nullptr, (int64_t)0, jb, jb, 1);
    CAP_HIP(hipGetLastError());
    if (m > 0 && p->solve3 && jb == nb) {
      // the block-row solve on the bf16 pipe with split operands - the same three kernels, per element the same sums, as the
      // column-split schedule's solve_cols (the two schedules stay bit-identical)
      float* Row32 = p->R32 + j0 + j1 * n;
      launch_split3_tri(s, Dinv, nb, p->A3[k & 1], jb);
      launch_split3_row(s, Row32, n, p->B3far, jb, m, 1);
      CAP_TRY(launch_bf16_update(jb, m, 3 * jb, 1.0f, p->A3[k & 1], 3 * jb, p->B3far, 3 * jb, Row32, n, 0, s));
      launch_f32_to_bf16(s, Row32, n, SP + roff + j1 * ldp, ldp, jb, m);
      CAP_HIP(hipGetLastError());
    } else if (m > 0) {
      float* Row32 = p->R32 + j0 + j1 * n;
      launch_f32_to_f64(s, Row32, n, p->T64, jb, jb, m, 0);
      CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, jb, m, jb, 1.0, Dinv, nb, p->T64, jb, 0.0, p->S64, jb, 0, s, 2 | 16));
      launch_f64_to_f32_bf16(s, p->S64, jb, Row32, n, SP + roff + j1 * ldp, ldp, jb, m, 0);
      CAP_HIP(hipGetLastError());
    }
    return CAP_OK;
  };
  // factor strip t on stream s: its panels one after the other, each followed by the K = nb update of the strip's remaining rows
  auto strip = [&](int64_t t, hipStream_t s) -> int {
    __bf16* SP = p->P16[t & 1];
    const int64_t k0 = t * sp, k1 = std::min(npan, k0 + sp), Jend = std::min(n, k1 * nb);
    for (int64_t k = k0; k < k1; k++) {
      CAP_TRY(panel(k, s, SP, (k - k0) * nb));
      const int64_t j1 = std::min(n, (k + 1) * nb), rows_left = Jend - j1;
      if (rows_left > 0) {         // rows of the strip below panel k, all columns to their right: K = nb, this panel's rows of SP
        const __bf16* Pk = SP + (k - k0) * nb + j1 * ldp;
        CAP_TRY(launch_bf16_head(rows_left, n - j1, nb, -1.0f, Pk, ldp, Pk, ldp, p->R32 + j1 + j1 * n, n, 1, s));
      }
    }
    return CAP_OK;
  };

  // update(t): R32[Js:, Js:] -= S^T S (upper) with S = the strip's solved rows (K = sp nb), split into HEAD (the rows of strip
  // t+1, panel stream) and REST (rows below, caller's stream).  strip(t+1) needs head(t) (same stream) and rest(t-1) (it touched
  // those rows too; waiting for it before head(t) also keeps the order of the atomic adds - hence the bits - fixed) and frees
  // P16[(t+1) & 1].
  CAP_TRY(strip(0, s1));
  CAP_HIP(hipEventRecord(p->ev_panel[0], s1));
  for (int64_t t = 0; t < nstrip; t++) {
    const int64_t Js = std::min(n, (t + 1) * sp * nb), m = n - Js;
    if (m <= 0) break;
    const int64_t K = Js - t * sp * nb;                   // rows of strip t (a multiple of nb: only the last strip can be ragged)
    const __bf16* S = p->P16[t & 1] + Js * ldp;
    const int64_t hb = std::min(sp * nb, m);              // rows of the next strip
    if (t > 0) CAP_HIP(hipStreamWaitEvent(s1, p->ev_rest[(t - 1) & 1], 0));
    CAP_TRY(launch_bf16_head(hb, m, K, -1.0f, S, ldp, S, ldp, p->R32 + Js + Js * n, n, 1, s1));
    CAP_HIP(hipStreamWaitEvent(s0, p->ev_panel[t & 1], 0));
    if (m > hb) {
      hipEvent_t e0 = nullptr, e1 = nullptr;
      if (p->profile && p->prof_ev) {
        if ((size_t)p->prof_used + 2 > p->prof_ev->size())
          for (int i = 0; i < 64; i++) { hipEvent_t e; CAP_HIP(hipEventCreate(&e)); p->prof_ev->push_back(e); }
        e0 = (*p->prof_ev)[p->prof_used]; e1 = (*p->prof_ev)[p->prof_used + 1];
        CAP_HIP(hipEventRecord(e0, s0));
      }
      CAP_TRY(launch_bf16_update(m - hb, m - hb, K, -1.0f, S + hb * ldp, ldp, S + hb * ldp, ldp, p->R32 + (Js + hb) * (n + 1), n, 1, s0));
      if (e0) {
        CAP_HIP(hipEventRecord(e1, s0));
        p->prof_used += 2;
        // algorithmic work of one launch: 2 K flop and 8 B (fp32 read + write) per element of the upper triangle, plus the panel once
        const double mm = (double)(m - hb), elems = 0.5 * mm * (mm + 1.0);
        p->prof_flops->push_back(2.0 * (double)K * elems);
        p->prof_bytes->push_back(8.0 * elems + 2.0 * (double)K * mm);
      }
    }
    CAP_HIP(hipEventRecord(p->ev_rest[t & 1], s0));
    if (t + 1 < nstrip) {
      CAP_TRY(strip(t + 1, s1));
      CAP_HIP(hipEventRecord(p->ev_panel[(t + 1) & 1], s1));
    }
  }
  CAP_HIP(hipEventRecord(p->ev_join, s1));
  CAP_HIP(hipStreamWaitEvent(s0, p->ev_join, 0));
  // fp64 promotion of the factor for the refinement sweeps (the TRSM block inverses were stored panel by panel)
  launch_f32_to_f64(s0, p->R32, n, p->R64, n, n, n, 1);
  CAP_HIP(hipGetLastError());
  p->have_r64 = true;
  return CAP_OK;
}

extern "C" {

// "profile" = 1: bracket every bf16 trailing update on the caller's stream with HIP events (read back by cap_mpchol_profile)
int cap_mpchol_set_option(cap_mpchol_plan* p, const char* key, int64_t value) {
  if (!p || !key) return CAP_ERR_ARG;
  if (!strcmp(key, "profile")) {
    p->profile = value != 0;
    if (p->profile && !p->prof_ev) {
      p->prof_ev = new std::vector<hipEvent_t>(); p->prof_flops = new std::vector<double>(); p->prof_bytes = new std::vector<double>();
    }
    return CAP_OK;
  }
  if (!strcmp(key, "split")) { p->split = value != 0; return CAP_OK; }     // column-split schedule (near / far columns), see cap_mpchol_factor
  if (!strcmp(key, "reserve")) {     // CUs kept for the diagonal-block chains (0 = off; a multiple of 8 keeps the XCDs balanced)
    if (value < 0 || value > 64) return CAP_ERR_ARG;
    if (value != p->reserve) { mp_release_streams(p); p->reserve = (int)value; }
    return CAP_OK;
  }
  if (!strcmp(key, "chain_coop")) { if (value < -1 || value > 256) return CAP_ERR_ARG; p->chain_coop = (int)value; return CAP_OK; }   // per plan, see cap_cholinv_set_option
  if (!strcmp(key, "solve3")) { p->solve3 = value != 0; return CAP_OK; }   // block-row solves on the bf16 pipe with split operands (split schedule)
  // process-wide A/B switches of the bf16 update (see launch_bf16_update): which kernel, chunk length, smallest launch for the new one
  if (!strcmp(key, "update_kernel")) { if (value < 0 || value > 6 || value == 2) return CAP_ERR_ARG; g_bf16_variant = (int)value; return CAP_OK; }
  if (!strcmp(key, "update_v3_min_tiles")) { if (value < 0) return CAP_ERR_ARG; g_bf16_v3_min = value; return CAP_OK; }
  if (!strcmp(key, "update_v3_head_min_tiles")) { if (value < -1) return CAP_ERR_ARG; g_bf16_head_min = value; return CAP_OK; }
  if (!strcmp(key, "update_v3_st")) { if (value < 1 || value > 64) return CAP_ERR_ARG; g_bf16_v3_st = (int)value; return CAP_OK; }
  if (!strcmp(key, "update_tpw")) { if (value < 1 || value > 64) return CAP_ERR_ARG; g_bf16_tpw = (int)value; return CAP_OK; }
  if (!strcmp(key, "update_min_tiles")) { if (value < 0) return CAP_ERR_ARG; g_bf16_min_tiles = value; return CAP_OK; }
  if (!strcmp(key, "strip")) { if (value < 1 || value > 2) return CAP_ERR_ARG; p->strip = value; return CAP_OK; }   // panels per bf16 update
  if (!strcmp(key, "pair_rest")) { p->pair_rest = value != 0; return CAP_OK; }     // K = 4 nb far updates (split schedule), see mp_factor_impl
  return CAP_ERR_ARG;
}

// the bf16 trailing updates (bf16_tn_kernel, rows below the next panel) of the LAST factor call: launches, their summed duration
// (ms), summed algorithmic flops (2 K per element of the updated upper triangle) and bytes (fp32 C read + write + the bf16 panel)
int cap_mpchol_profile(cap_mpchol_plan* p, int64_t* launches, double* ms_total, double* flops_total, double* bytes_total) {
  if (!p || !launches || !ms_total || !flops_total || !bytes_total) return CAP_ERR_ARG;
  *launches = 0; *ms_total = 0; *flops_total = 0; *bytes_total = 0;
  if (!p->prof_ev) return CAP_OK;
  for (int i = 0; i + 1 < p->prof_used; i += 2) {
    CAP_HIP(hipEventSynchronize((*p->prof_ev)[i + 1]));
    float ms = 0;
    CAP_HIP(hipEventElapsedTime(&ms, (*p->prof_ev)[i], (*p->prof_ev)[i + 1]));
    *ms_total += ms; *flops_total += (*p->prof_flops)[i / 2]; *bytes_total += (*p->prof_bytes)[i / 2]; (*launches)++;
  }
  return CAP_OK;
}

float* cap_mpchol_R32_ptr(cap_mpchol_plan* p, int64_t* ld) { if (!p) return nullptr; if (ld) *ld = p->n; return p->R32; }

int cap_mpchol_info(cap_mpchol_plan* p, void* stream, int64_t* info) {
  if (!p || !info) return CAP_ERR_ARG;
  CAP_TRY(cap_drain_streams({p->s_panel, p->s_far, p->s_bulk, p->s_chain}));
  int h = 0;
  CAP_HIP(hipMemcpyAsync(&h, p->info_dev, sizeof(int), hipMemcpyDeviceToHost, cap_stream(stream)));
  CAP_HIP(hipStreamSynchronize(cap_stream(stream)));
  *info = h;
  return h == 0 ? CAP_OK : CAP_ERR_NOT_SPD;
}

// Solve A X = B (nrhs columns) to fp64 accuracy.  A must be the FULL symmetric matrix (the residual is a plain GEMM).
// Returns the sweeps used and the final ||B - A X||_F / ||B||_F; CAP_OK also when max_iter was reached (check relres).
// Stops early when the residual stagnates below 1e-10 (the attainable level is ~ n eps ||A|| ||X|| / ||B||).
// Synchronises the stream once per sweep (the convergence test is a host decision).
int cap_mpchol_solve(cap_mpchol_plan* p, const double* A, int64_t lda, const double* B, int64_t ldb, double* X, int64_t ldx, int64_t nrhs,
                     int max_iter, double tol, int* iters, double* relres, void* stream) {
  if (!p || !A || !B || !X || lda < p->n || ldb < p->n || ldx < p->n || nrhs <= 0 || nrhs > p->nrhs_cap || !p->have_r64) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  // working width: up to 8 right-hand sides run on the skinny streaming kernels (gemm.hip: op(A) is read once, no padding to a
  // 128-wide tile - the residual and every block step of the two substitutions are then bound by HBM, not by 16 x padded flops)
  const int64_t n = p->n, w = nrhs <= 8 ? 8 : cap_round_up(nrhs, 128);
  auto apply_Ainv = [&](double* V) -> int {      // V <- R^-1 R^-T V  (n x w)
    // up to 8 right-hand sides: the block updates stream the fp32 factor itself (R64 is its exact promotion: the same sums, half the bytes)
    const float* T32 = w <= 8 ? p->R32 : nullptr;
    CAP_TRY(cap_trsm_apply(CAP_LEFT, CAP_TRANS, n, w, p->R64, n, p->Inv, p->tb, V, n, p->Xt, s, 0, T32, n));
    return cap_trsm_apply(CAP_LEFT, CAP_NOTRANS, n, w, p->R64, n, p->Inv, p->tb, V, n, p->Xt, s, 0, T32, n);
  };
  // zero-padded working copies: B, X (= first solve), residual
  CAP_HIP(hipMemsetAsync(p->Bw, 0, sizeof(double) * n * w, s));
  CAP_TRY(cap_copy_rect(B, ldb, p->Bw, n, n, nrhs, s));
  CAP_HIP(hipMemcpyAsync(p->Xw, p->Bw, sizeof(double) * n * w, hipMemcpyDeviceToDevice, s));
  CAP_TRY(apply_Ainv(p->Xw));
  CAP_TRY(cap_sumsq(p->Bw, n, n, w, 0, 0, p->norms, stream));
  double h[2] = {0, 0};
  int it = 0; double rr = 0, prev = 1e300;
  for (;;) {
    // r = b - A x  (A symmetric: A^T form = the fast K-contiguous kernel)
    CAP_HIP(hipMemcpyAsync(p->Rw, p->Bw, sizeof(double) * n * w, hipMemcpyDeviceToDevice, s));
    CAP_TRY(cap_gemm_launch(CAP_TRANS, CAP_NOTRANS, n, w, n, -1.0, A, lda, p->Xw, n, 1.0, p->Rw, n, 0, s));
    CAP_TRY(cap_sumsq(p->Rw, n, n, w, 0, 0, p->norms + 1, stream));
    CAP_HIP(hipMemcpyAsync(h, p->norms, sizeof(double) * 2, hipMemcpyDeviceToHost, s));
    CAP_HIP(hipStreamSynchronize(s));
    rr = h[0] > 0 ? std::sqrt(h[1] / h[0]) : 0.0;
    // converged, out of sweeps, broken (NaN), or stagnating at the attainable accuracy (less than 2x progress per sweep)
    if (rr <= tol || it >= max_iter || !(rr == rr) || (it >= 2 && rr > 0.5 * prev && rr < 1e-10)) break;
    prev = rr;
    CAP_TRY(apply_Ainv(p->Rw));                  // correction
    cap_acc_rw(p->Xw, n, n, w); cap_acc_r(p->Rw, n, n, w);
    hipLaunchKernelGGL(axpy_cols_kernel, dim3((unsigned)std::min<int64_t>(cap_ceil_div(n, 256), 4096), (unsigned)w), dim3(256), 0, s, p->Xw, n,
                       p->Rw, n, n, w);
    CAP_HIP(hipGetLastError());
    it++;
  }
  CAP_TRY(cap_copy_rect(p->Xw, n, X, ldx, n, nrhs, s));
  if (iters) *iters = it;
  if (relres) *relres = rr;
  return CAP_OK;
}

}  // extern "C"
