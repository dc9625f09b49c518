// Dedicated kernels of the CholeskyQR sweep for n = 256 columns (BASELINE config 4: 2^24 x 256 over 8 GPUs = 2^21 x 256 per
// GPU).  Both passes stream a panel nobody re-reads; with the generic 128 x 128 tile kernel they were bound by LDS-DMA
// latency x bytes in flight and by per-tile fixed cost (16 K tiles per tile), not by MFMA:
//
//  gram256    G = Q^T Q (upper).  One 8-wave workgroup per CU owns a contiguous range of rows and the WHOLE 256 x 256
//             upper triangle: a K tile (16 rows of Q = 32 KiB) is staged ONCE through a 4-stage LDS ring (3 tiles = 96 KiB
//             in flight per CU, counted vmcnt), wave w accumulates the block columns w and 15 - w (17 blocks of 16 x 16
//             each: perfectly balanced, 136 of 136 useful blocks).  Partial Grams go to one slab per workgroup and are
//             summed in a fixed order (deterministic).  Replaces the split-K DSYRK form of cacqr.hpp:15.
//  qrapply256 Qout = Qin * Rinv (Rinv upper; cacqr.hpp:24-25: dtrmm Right/Upper/NoTrans).  One persistent 8-wave workgroup
//             per CU walks row tiles of 128 rows; the K loop runs on across row-tile boundaries through a 3-stage ring, so
//             the pipeline is filled once per workgroup, not once per 16 K tiles; 16 x 16 blocks of Rinv below the
//             diagonal are skipped (column waves 0 & 3 and 1 & 2 share a SIMD: balanced).
#include <algorithm>

#include "common.h"
#include "kargs.h"
#include "tile_dma.h"

namespace {

constexpr int GN = 256, BK = 16;

// ================================================================================================ gram256
constexpr int G_STAGES = 4;
constexpr int G_TILE = GN * BK;        // doubles per stage (32 KiB): [256 Q-columns][16 k], 16-byte chunks XOR-swizzled

// struct GramArgs: csrc/kargs.h (shared with the CPU kernel models of tests/hipshim)

template <int W>
__device__ __forceinline__ void gram_wave(const GramArgs& g, double* smem, int nk, const DmaBuf& d, int half, int piece0) {
  constexpr int CA = W, CB = 15 - W;                          // this wave's two block columns: CA + 1 + CB + 1 = 17 blocks
  const int lane = threadIdx.x & 63;
  const int lr = lane & 15, kg = lane >> 4, sw = lr >> 1;
  d4 accA[CA + 1], accB[CB + 1];
#pragma unroll
  for (int r = 0; r <= CA; r++) accA[r] = (d4){0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r <= CB; r++) accB[r] = (d4){0, 0, 0, 0};
  auto issue = [&](int t) {
    const int tc = t < nk ? t : nk - 1;                       // clamped: redundant refills keep the vmcnt bookkeeping uniform
    dma_tile_buf<0, 4>(d, piece0, (uint32_t)tc * BK * 8, smem + (t & (G_STAGES - 1)) * G_TILE + half * 128 * BK);
  };
  if (nk > 0) { issue(0); issue(1); issue(2); }
  for (int t = 0; t < nk; t++) {
    // tile t has landed once only the 8 younger pieces of this wave are outstanding; after the barrier everybody's share is in LDS
    __builtin_amdgcn_s_waitcnt(8 | (7 << 4) | (0 << 8));
    __builtin_amdgcn_s_barrier();
    issue(t + 3);                                             // into the stage of tile t-1: its reads all happened before this barrier
    const double* tile = smem + (t & (G_STAGES - 1)) * G_TILE;
#pragma unroll
    for (int h = 0; h < 2; h++) {
      d2 f[CB + 1];                                           // CB >= CA for W < 8: block rows 0..CB cover both block columns
#pragma unroll
      for (int r = 0; r <= CB; r++) f[r] = *reinterpret_cast<const d2*>(tile + (16 * r + lr) * BK + (((h * 4 + kg) ^ sw) << 1));
#pragma unroll
      for (int r = 0; r <= CA; r++) accA[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[CA].x, f[r].x, accA[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r <= CB; r++) accB[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[CB].x, f[r].x, accB[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r <= CA; r++) accA[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[CA].y, f[r].y, accA[r], 0, 0, 0);
#pragma unroll
      for (int r = 0; r <= CB; r++) accB[r] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[CB].y, f[r].y, accB[r], 0, 0, 0);
    }
  }
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));       // the clamped tail refills must not outlive the workgroup's LDS
  // slab: block (r, c): lane holds rows 16 r + lr, columns 16 c + kg + 4 j  (operands swapped like the tile kernel)
  double* P = g.P + (int64_t)blockIdx.x * GN * GN;
#pragma unroll
  for (int r = 0; r <= CA; r++)
#pragma unroll
    for (int j = 0; j < 4; j++) P[16 * r + lr + (int64_t)(16 * CA + kg + 4 * j) * GN] = accA[r][j];
#pragma unroll
  for (int r = 0; r <= CB; r++)
#pragma unroll
    for (int j = 0; j < 4; j++) P[16 * r + lr + (int64_t)(16 * CB + kg + 4 * j) * GN] = accB[r][j];
}

__global__ void __launch_bounds__(512, 2) gram256_kernel(const GramArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int64_t r0 = (int64_t)blockIdx.x * g.chunk;
  const int64_t rows = std::max<int64_t>(0, std::min<int64_t>(g.chunk, g.m - r0));
  const int nk = (int)(rows / BK);
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = wid >> 2, piece0 = wid & 3;
  // one descriptor per half of the columns: 256 columns x ld x 8 B would overflow a 32-bit offset at ld = 2^21
  const DmaBuf d = dma_buf_make(g.Q + r0 + (int64_t)half * 128 * g.ld, g.ld);
  switch (wid) {
    case 0: gram_wave<0>(g, smem, nk, d, half, piece0); break;
    case 1: gram_wave<1>(g, smem, nk, d, half, piece0); break;
    case 2: gram_wave<2>(g, smem, nk, d, half, piece0); break;
    case 3: gram_wave<3>(g, smem, nk, d, half, piece0); break;
    case 4: gram_wave<4>(g, smem, nk, d, half, piece0); break;
    case 5: gram_wave<5>(g, smem, nk, d, half, piece0); break;
    case 6: gram_wave<6>(g, smem, nk, d, half, piece0); break;
    default: gram_wave<7>(g, smem, nk, d, half, piece0); break;
  }
}

// G(upper) = sum over the slabs in a fixed association order (deterministic): 4 slab groups per element, 4 independent partial
// sums per thread so that 16 loads per element are in flight, groups combined through LDS
__global__ void __launch_bounds__(1024) gram256_reduce_kernel(const double* P, int nslab, double* G, int64_t ldg) {
  __shared__ double part[4][GN];
  const int col = blockIdx.x, row = threadIdx.x & (GN - 1), grp = threadIdx.x >> 8;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  if (row <= col) {
    const double* p = P + row + (int64_t)col * GN;
    int z = grp;
    for (; z + 12 < nslab; z += 16) {
      s0 += p[(int64_t)z * GN * GN]; s1 += p[(int64_t)(z + 4) * GN * GN];
      s2 += p[(int64_t)(z + 8) * GN * GN]; s3 += p[(int64_t)(z + 12) * GN * GN];
    }
    for (; z < nslab; z += 4) s0 += p[(int64_t)z * GN * GN];
  }
  part[grp][row] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  // the strictly-lower part is written too (zeros): the callers move / multiply G as a dense square
  if (grp == 0) G[row + (int64_t)col * ldg] = row <= col ? (part[0][row] + part[1][row]) + (part[2][row] + part[3][row]) : 0.0;
}

// ================================================================================================ qrapply256
constexpr int TA = 128 * BK, TB = GN * BK;       // A tile [16 k][128 m] (16 KiB), B tile [256 cols][16 k] (32 KiB)
constexpr int A_NST = 4, B_NST = 3;              // 4 x 16 KiB + 3 x 32 KiB = 160 KiB: the whole LDS of the CU
// A (the streamed panel, straight from HBM) is requested 3 steps ahead, B (Rinv, L2-resident) 2 steps ahead.  Per step a wave
// issues [4 pieces of B, 2 pieces of A]; vmcnt retires in order, so "tile t has landed" = at most the 8 younger pieces
// A(t+1), B(t+1), A(t+2) outstanding.

// struct ApplyArgs: csrc/kargs.h (shared with the CPU kernel models of tests/hipshim)

template <int DIAG, int PIPE>
__global__ void __launch_bounds__(512, 2) qrapply256_kernel(const ApplyArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  // contig: workgroup b owns `per` consecutive row tiles, so per column it streams one contiguous range and its address
  // translations (256 columns in, 256 out, 16 MiB apart at m = 2^21) stay the same for the whole launch; otherwise tiles are
  // dealt round-robin (b, b + G, ...)
  const int per = (g.ntiles + G - 1) / G;
  const int t0 = g.contig ? b * per : b, tstride = g.contig ? 1 : G;
  const int nmine = g.contig ? max(0, min(per, g.ntiles - t0)) : (b < g.ntiles ? (g.ntiles - 1 - b) / G + 1 : 0);
  const int U = nmine * 16;                                    // pipeline steps: (my row tile, K tile)
  if (U == 0) return;
  // column wave wn owns the block columns wn, wn + 4, wn + 8, wn + 12 (cyclic: at K tile kt only block columns >= kt are
  // non-zero in an upper-triangular Rinv, so every wave loses work at the same pace); waves w and w + 4 share a SIMD:
  // pairing column waves (0, 3) and (1, 2) there evens out the remaining +-1 block
  const int wq = wid >> 1;
  const int wn = wq == 2 ? 3 : (wq == 3 ? 2 : wq);
  const int wi = (wid & 1) * 64;
  const int lr = lane & 15, kg = lane >> 4, sw = lr >> 1;
  const DmaBuf dB = dma_buf_make(g.Ri, GN);
  double* sAbase = smem; double* sBbase = smem + A_NST * TA;
  auto issue_a = [&](int u) {
    const int uc = u < U ? u : U - 1;
    const int kt = uc & 15;
    const int64_t i0 = (int64_t)(t0 + (uc >> 4) * tstride) * 128;
    double* st = sAbase + (u & (A_NST - 1)) * TA;
    // rows of the stage = k, 128 consecutive doubles of Q each; one wave-instruction per k row, halves swapped when (k >> 1) is odd
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int kr = wid * 2 + q;
      const int c = lane ^ (((kr >> 1) & 1) << 3);
      const double* src = g.Qin + (int64_t)(kt * BK + kr) * g.ldin + i0 + c * 2;
      __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                       (__attribute__((address_space(3))) void*)(st + kr * 128), 16, 0, 0);
    }
  };
  // K tile kt of the upper-triangular Rinv is zero in the columns < 16 kt: only the pieces (8 columns each) p >= 2 kt are moved, dealt
  // round-robin over the 8 waves - wave w takes p = 2 kt + w + 8 q, q < 4 - (kt >> 2), clamped to 31 (the few duplicates rewrite the same
  // bytes) - so every wave issues the SAME number of pieces per step and the counted vmcnt below stays wave-uniform.  37 % fewer LDS-DMA
  // pieces of B (40 instead of 64 per wave and row tile); the stale columns of a stage are never used (CQR_MMA skips their blocks).
  auto nbp_of = [&](int u) { return 4 - ((((u < U ? u : U - 1)) & 15) >> 2); };
  auto issue_b = [&](int u) {
    const int uc = u < U ? u : U - 1, kt = uc & 15;
    const int nbp = 4 - (kt >> 2);
    double* st = sBbase + (u % B_NST) * TB;
    const uint32_t voff = (wid & 1) ? dB.voff_odd : dB.voff_even;      // piece parity = wave parity (2 kt and 8 q are even)
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (q < nbp) {
        const int pp = 2 * kt + wid + 8 * q;
        const uint32_t g8 = (uint32_t)(pp < 31 ? pp : (31 - ((31 - wid) & 1)));   // clamp to the last piece of MY parity (30 or 31)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(dB.rsrc, (__attribute__((address_space(3))) void*)(st + g8 * 8 * 16), 16, (int)voff,
                                                 (int)(g8 * dB.rowgrp + (uint32_t)kt * BK * 8), 0, 0);
      }
  };
  d4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; i++)
#pragma unroll
    for (int j = 0; j < 4; j++) acc[i][j] = (d4){0, 0, 0, 0};
  const int amc_flip = (kg & 1) << 4;
  auto frags = [&](int u, int h, d2 (&fa)[4], d2 (&fb)[4]) {
    const double* tA = sAbase + (u & (A_NST - 1)) * TA;
    const double* tB = sBbase + (u % B_NST) * TB;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      const int mm = (wi + 16 * i + lr) ^ amc_flip;
      fa[i] = (d2){tA[(8 * h + 2 * kg) * 128 + mm], tA[(8 * h + 2 * kg + 1) * 128 + mm]};
    }
#pragma unroll
    for (int j = 0; j < 4; j++) fb[j] = *reinterpret_cast<const d2*>(tB + (16 * (wn + 4 * j) + lr) * BK + (((h * 4 + kg) ^ sw) << 1));
  };
  // (MMA / STORE_COL are macros, not lambdas: with acc captured by reference the compiler kept three quarters of it in scratch)
#define CQR_MMA(fa, fb, jlo)                                                                                                   \
  if (!(DIAG & 2)) {                                                                                                           \
    _Pragma("unroll") for (int j = 0; j < 4; j++)                                                                              \
      if (j >= (jlo)) {                                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                          \
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j].x, fa[i].x, acc[i][j], 0, 0, 0);                              \
        _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                          \
          acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j].y, fa[i].y, acc[i][j], 0, 0, 0);                              \
      }                                                                                                                        \
  }
  // vmcnt retires loads and stores in issue order (the compiler itself counts past younger stores on gfx950), so "tile t has
  // landed" = at most the younger pieces outstanding: 8 DMA pieces (A(t+1), B(t+1), A(t+2)) plus the 16 stores of the two steps
  // before the wait, if this wave finished a block column there (it does every 4th step: kt % 4 == wn)
  // nby = B pieces of the ONE younger B tile in flight at this wait (1 .. 4): the younger loads are 4 pieces of A + nby of B
  auto wait_tile = [&](int ulast, int nby) {        // ulast = the last step whose stores were issued before this wait
    const bool st_young = !(DIAG & 1) && ((ulast >= 0 && (ulast & 3) == wn) || (ulast >= 1 && ((ulast - 1) & 3) == wn));
#define CQR_WAITV(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (0 << 8) | (((n) >> 4) << 14))
    if (st_young) { if (nby >= 4) CQR_WAITV(24); else if (nby == 3) CQR_WAITV(23); else if (nby == 2) CQR_WAITV(22); else CQR_WAITV(21); }
    else { if (nby >= 4) CQR_WAITV(8); else if (nby == 3) CQR_WAITV(7); else if (nby == 2) CQR_WAITV(6); else CQR_WAITV(5); }
#undef CQR_WAITV
    __builtin_amdgcn_s_barrier();
  };
  // Block column kt = wn + 4 j is complete after K tile kt (K tiles beyond it only meet zeros of Rinv): store it right away, so
  // the writes trickle out over the row tile instead of draining as one 256 KiB burst.  Lane holds
  // Qout[i0 + wi + 16 i + lr][16 kt + kg + 4 r].  In place (Qout == Qin) is fine: these columns of this row tile were consumed
  // as K tile kt, later K tiles lie to the right.
#define CQR_STORE_COL(u)                                                                                                       \
  if ((((u) & 15) & 3) == wn && (!(DIAG & 1) || g.ntiles < 0)) {                                                               \
    const int kt_ = (u) & 15;                                                                                                  \
    const int64_t i0_ = (int64_t)(t0 + ((u) >> 4) * tstride) * 128;                                                                  \
    _Pragma("unroll") for (int j = 0; j < 4; j++)                                                                              \
      if (j == (kt_ >> 2)) {                                                                                                   \
        _Pragma("unroll") for (int i = 0; i < 4; i++) {                                                                        \
          const int64_t row = i0_ + wi + 16 * i + lr;                                                                          \
          _Pragma("unroll") for (int r = 0; r < 4; r++) {                                                                      \
            g.Qout[row + (int64_t)(16 * kt_ + kg + 4 * r) * g.ldout] = acc[i][j][r];                                           \
            acc[i][j][r] = 0.0;                                                                                                \
          }                                                                                                                    \
        }                                                                                                                      \
      }                                                                                                                        \
  }
  issue_a(0); issue_b(0); issue_a(1); issue_b(1); issue_a(2);
  {
    // The tile hand-over (wait + barrier + refill + first fragment reads of the NEXT tile) sits in the middle of a step, between
    // the two k-halves of the current tile, so every wave has half a step of MFMAs queued behind its LDS reads.
    d2 fa0[4], fb0[4], fa1[4], fb1[4];
    wait_tile(-1, nbp_of(1));                 // tile 0 has landed: younger A(1), A(2), B(1)
    issue_b(2); issue_a(3);
    frags(0, 0, fa0, fb0);
    for (int u = 0; u < U; u++) {
      const int kt = u & 15;
      const int jlo = kt > wn ? (kt - wn + 3) >> 2 : 0;          // block column cb = wn + 4 j is active at K tile kt iff cb >= kt
      frags(u, 1, fa1, fb1);
      CQR_MMA(fa0, fb0, jlo)
      __builtin_amdgcn_s_waitcnt(63 | (7 << 4) | (0 << 8) | (3 << 14));   // lgkmcnt(0): my reads of tile u are done -> its stages may be refilled
      wait_tile(u - 1, nbp_of(u + 2));        // tile u + 1 has landed: younger A(u + 2), A(u + 3), B(u + 2)
      issue_b(u + 3);
      issue_a(u + 4);
      frags(u + 1, 0, fa0, fb0);
      CQR_MMA(fa1, fb1, jlo)
      CQR_STORE_COL(u)
    }
  }
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
#undef CQR_MMA
#undef CQR_STORE_COL
}

// (Round 6: starting the workgroups (b % 16) sixteenths of a row tile apart - so that the chip sees the AVERAGE of the MFMA-heavy early and the
// HBM-heavy late K steps of a row tile instead of all workgroups moving through them together - changes nothing: 2.96 ms with any stagger between
// 0 and 6 us per sixteenth, 3.0 - 3.05 ms beyond; profiles/r06_experiments.md.)
// (A variant that handled the K tiles (s, 15 - s) of a row tile in one step - 17 block columns of work in every step, half the barriers -
// was built and measured in round 4: 4 % SLOWER on the whole CholeskyQR2 call, its hand-over sits at the top of a step where this kernel
// hides it between the two k-halves of a tile; profiles/r04_experiments.log section 5.  Removed in round 5.)

}  // namespace

// G (ldg >= 256, upper triangle written) = Q^T Q for Q m x 256 (ld), m % 16 == 0.  work: >= cap_gram256_work(m) doubles.
int64_t cap_gram256_slabs(int64_t m) {
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  return std::max<int64_t>(1, std::min<int64_t>(cus, m / 512));     // at least 32 K tiles per workgroup
}
int64_t cap_gram256_work(int64_t m) { return cap_gram256_slabs(m) * GN * GN; }

int cap_gram256_launch(const double* Q, int64_t ld, int64_t m, double* G, int64_t ldg, double* work, hipStream_t s) {
  if (!Q || !G || !work || m <= 0 || (m % BK) || ld < m || ldg < GN || (ld & 1) || ((uintptr_t)Q & 15)) return CAP_ERR_UNSUPPORTED;
  if (128 * ld * 8 >= 0xfffffff0LL) return CAP_ERR_UNSUPPORTED;
  const int64_t nslab = cap_gram256_slabs(m);
  GramArgs g{Q, ld, m, cap_round_up(cap_ceil_div(m, nslab), BK), work};
  cap_acc_r(Q, ld, m, GN); cap_acc_w(work, 0, nslab * GN * GN, 1);
  hipLaunchKernelGGL(gram256_kernel, dim3((unsigned)nslab), dim3(512), G_STAGES * G_TILE * sizeof(double), s, g);
  cap_acc_r(work, 0, nslab * GN * GN, 1); cap_acc_w(G, ldg, GN, GN);
  hipLaunchKernelGGL(gram256_reduce_kernel, dim3(GN), dim3(4 * GN), 0, s, work, (int)nslab, G, ldg);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}

// Qout (m x 256, ldout) = Qin (m x 256, ldin) * Ri (256 x 256 upper, ld 256, strictly-lower part zero); m % 128 == 0
int cap_qrapply256_launch(const double* Qin, int64_t ldin, const double* Ri, double* Qout, int64_t ldout, int64_t m, hipStream_t s) {
  if (!Qin || !Ri || !Qout || m <= 0 || (m % 128) || ldin < m || ldout < m || (ldin & 1) || ((uintptr_t)Qin & 15) || ((uintptr_t)Ri & 15))
    return CAP_ERR_UNSUPPORTED;
  int dev = 0, cus = 256;
  CAP_HIP(hipGetDevice(&dev));
  (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  static const int contig = CAP_ENV("CAP_CQR_CONTIG") ? atoi(CAP_ENV("CAP_CQR_CONTIG")) : 1;
  ApplyArgs g{Qin, ldin, Ri, Qout, ldout, (int)(m / 128), contig};
  const int grid = (int)std::min<int64_t>(cus, g.ntiles);
  // CAP_CQR_DIAG is timing surgery only (1 = no stores, 2 = no MFMA: results are wrong)
  const size_t lds = (A_NST * TA + B_NST * TB) * sizeof(double);
  const dim3 gr((unsigned)grid), bl(512);
  cap_acc_r(Qin, ldin, m, 256); cap_acc_r(Ri, 256, 256, 256, 1); cap_acc_w(Qout, ldout, m, 256);
  if constexpr (CAP_EXPERIMENTS) {
    static const int diag = CAP_ENV("CAP_CQR_DIAG") ? atoi(CAP_ENV("CAP_CQR_DIAG")) : 0;
    if (diag == 1) hipLaunchKernelGGL((qrapply256_kernel<1, 1>), gr, bl, lds, s, g);
    else if (diag == 2) hipLaunchKernelGGL((qrapply256_kernel<2, 1>), gr, bl, lds, s, g);
    else if (diag == 3) hipLaunchKernelGGL((qrapply256_kernel<3, 1>), gr, bl, lds, s, g);
    if (diag >= 1 && diag <= 3) { CAP_HIP(hipGetLastError()); return CAP_OK; }
  }
  hipLaunchKernelGGL((qrapply256_kernel<0, 1>), gr, bl, lds, s, g);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}
