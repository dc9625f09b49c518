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
//             per CU walks row tiles of 128 rows; the K loop runs on across row-tile boundaries through the LDS rings, so
//             the pipeline is filled once per workgroup, not once per 16 K tiles; 16 x 16 blocks of Rinv below the
//             diagonal are skipped.  Wave = 32 rows x the 8 block columns of one parity: every SIMD carries the same work
//             at every K tile; LDS reads, LDS-DMA requests and stores ride in the issue slots between the wave's own MFMAs.
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
// A (the streamed panel, straight from HBM) is requested 3 steps ahead, B (Rinv, L2-resident) 2 steps ahead.  Per step a wave issues
// [nq <= 4 pieces of B, 2 pieces of A]; vmcnt retires in order, so "tile t has landed" = at most the younger requests A(t+1), B(t+1), A(t+2)
// (and the stores issued since) outstanding.

// struct ApplyArgs: csrc/kargs.h (shared with the CPU kernel models of tests/hipshim)

// Wave w = (parity p = w >> 2, row slice rs = w & 3): rows 32 rs .. 32 rs + 31 of the row tile x the 8 block columns cb = 2 j + p.
// Waves w and w + 4 share a SIMD, so every SIMD carries even + odd block columns of one row slice: 16 - kt block columns of work at K tile
// kt on EVERY SIMD (the cyclic column split of the first kernel left 72 units of 512 cycles on the busier SIMD pair against 68 here).
// Measured on this part (tools/micro/mfma_valu.hip): while one wave issues fp64 MFMAs back to back, the other wave of its SIMD issues NO vector
// instruction at all (VALU, LDS, VMEM - at any s_setprio), so nothing of a wave's hand-over can hide behind the OTHER wave's MFMAs.  What is free
// are the issue slots between a wave's OWN MFMAs (one per 64 cycles): every LDS read, LDS-DMA request and store of a step is placed there
// (sched_group_barrier), the MFMAs of a half step come first after every barrier, and all addressing is scalar (buffer resources rebuilt per step
// on the SALU, one 32-bit lane offset each) so that the vector ALU sees ~ 10 instructions per step.
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
template <int DIAG>
__global__ void __launch_bounds__(512, 2) qrapply256_kernel(const ApplyArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int G = (int)gridDim.x, b = (int)blockIdx.x;
  const int per = (g.ntiles + G - 1) / G;
  // contig: workgroup b owns `per` consecutive row tiles, so per column it streams one contiguous range and its address translations
  // (256 columns in, 256 out, 16 MiB apart at m = 2^21) stay the same for the whole launch; otherwise tiles are dealt round-robin (b, b + G, ...)
  const int t0 = g.contig ? b * per : b, tstride = g.contig ? 1 : G;
  const int nmine = g.contig ? max(0, min(per, g.ntiles - t0)) : (b < g.ntiles ? (g.ntiles - 1 - b) / G + 1 : 0);
  const int U = nmine * 16;
  if (U == 0) return;
  const int par = wid >> 2, rs = wid & 3;
  const int lr = lane & 15, kg = lane >> 4, sw = lr >> 1;
  const DmaBuf dB = dma_buf_make(g.Ri, GN);
  const uint32_t voffB = (wid & 1) ? dB.voff_odd : dB.voff_even;            // piece parity = wave parity (2 kt and 8 q are even)
  const uint32_t voffA = (uint32_t)((lane ^ ((wid & 1) << 3)) << 4);         // k rows 2 wid, 2 wid + 1: halves swapped when (k >> 1) = wid is odd
  const uint32_t ldin8 = (uint32_t)(g.ldin * 8), ldout8 = (uint32_t)(g.ldout * 8);
  char* const ldsA = reinterpret_cast<char*>(smem);
  char* const ldsB = reinterpret_cast<char*>(smem + A_NST * TA);
  // per-lane parts of the fragment addresses (bytes); everything else is an immediate or the stage base
  uint32_t pA0 = (uint32_t)(((2 * kg) * 128 + 32 * rs + 16 * (0 ^ (kg & 1)) + lr) * 8);
  uint32_t pA1 = (uint32_t)(((2 * kg) * 128 + 32 * rs + 16 * (1 ^ (kg & 1)) + lr) * 8);
  const uint32_t pB0 = (uint32_t)(lr * 128 + (((0 + kg) ^ sw) << 4) + 2048 * par);
  const uint32_t pB1 = (uint32_t)(lr * 128 + (((4 + kg) ^ sw) << 4) + 2048 * par);
  const uint32_t voffO = (uint32_t)((32 * rs + lr) * 8) + (uint32_t)kg * ldout8;   // lane holds Qout[32 rs + 16 i + lr][16 kt + kg + 4 r]
  auto tile_row0 = [&](int uc) { return (int64_t)(t0 + (uc >> 4) * tstride) * 128; };
  auto issue_a = [&](int u) {                                  // k rows 2 wid, 2 wid + 1 of K tile u: 1 KiB each, one wave-instruction
    const int uc = u < U ? u : U - 1;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(g.Qin + (int64_t)((uc & 15) * BK) * g.ldin + tile_row0(uc)), 0, (int)0xffffffffu, 0x00020000);
    char* st = ldsA + (u & (A_NST - 1)) * (TA * 8);
#pragma unroll
    for (int q = 0; q < 2; q++) {
      const int kr = wid * 2 + q;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(st + kr * 1024), 16, (int)voffA, (int)((uint32_t)kr * ldin8), 0, 0);
    }
  };
  // K tile kt of the upper-triangular Rinv is zero in the columns < 16 kt: only the pieces (8 columns each) p >= 2 kt are moved, dealt round-robin
  // over the 8 waves - wave w takes p = 2 kt + w + 8 q, q < nq = 4 - (kt >> 2), clamped to the last piece of its parity (the few duplicates
  // rewrite the same bytes) - so every wave issues the SAME number of requests per step and the counted vmcnt stays wave-uniform.
#define CQY_ISSUE_B(u, NQ)                                                                                                     \
  {                                                                                                                            \
    const int uc_ = (u) < U ? (u) : U - 1, ktb_ = uc_ & 15;                                                                    \
    char* stb_ = ldsB + ((u) % B_NST) * (TB * 8);                                                                              \
    _Pragma("unroll") for (int q = 0; q < (NQ); q++) {                                                                         \
      const int pp = 2 * ktb_ + wid + 8 * q;                                                                                   \
      const uint32_t g8 = (uint32_t)(pp < 31 ? pp : (31 - ((31 - wid) & 1)));                                                  \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(dB.rsrc, (__attribute__((address_space(3))) void*)(stb_ + g8 * 1024), 16, (int)voffB, \
                                               (int)(g8 * dB.rowgrp + (uint32_t)ktb_ * BK * 8), 0, 0);                         \
    }                                                                                                                          \
  }
  d4 acc[2][8];                                                // (never zeroed: the first MFMA of a row tile takes C = 0)
  d2 fa0[2], fb0[8], fa1[2], fb1[8];
  // fragments of K tile u, k half h: A rows (8 h + 2 kg, + 1) x my 2 row blocks; Rinv columns 16 (2 j + par) + lr, 16-byte chunk (4 h + kg) ^ sw
#define CQY_READS(u, h, fa, fb, JR)                                                                                               \
  {                                                                                                                            \
    /* (the lane addresses are made opaque: otherwise every lane address + constant is hoisted out of the loop into a register of its own) */ \
    const uint32_t oA = (uint32_t)(((u) & (A_NST - 1)) * (TA * 8) + (h) * 8192), oB = (uint32_t)(A_NST * TA * 8 + ((u) % B_NST) * (TB * 8)); \
    uint32_t vB = ((h) ? pB1 : pB0) + oB;                                                                                      \
    asm volatile("" : "+v"(pA0), "+v"(pA1), "+v"(vB));                                                                         \
    const char* l0 = reinterpret_cast<const char*>(smem);                                                                      \
    fa[0] = (d2){*reinterpret_cast<const double*>(l0 + pA0 + oA), *reinterpret_cast<const double*>(l0 + pA0 + oA + 1024)};     \
    fa[1] = (d2){*reinterpret_cast<const double*>(l0 + pA1 + oA), *reinterpret_cast<const double*>(l0 + pA1 + oA + 1024)};     \
    _Pragma("unroll") for (int j = (JR); j < 8; j++) fb[j] = *reinterpret_cast<const d2*>(l0 + vB + 4096 * j);                 \
  }
#define CQY_MMA(fa, fb, J, FIRST)                                                                                              \
  if (!(DIAG & 2)) {                                                                                                           \
    _Pragma("unroll") for (int j = (J); j < 8; j++) {                                                                          \
      _Pragma("unroll") for (int i = 0; i < 2; i++)                                                                            \
        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j].x, fa[i].x, (FIRST) ? (d4){0.0, 0.0, 0.0, 0.0} : acc[i][j], 0, 0, 0); \
      _Pragma("unroll") for (int i = 0; i < 2; i++) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fb[j].y, fa[i].y, acc[i][j], 0, 0, 0); \
    }                                                                                                                          \
  }
  // one LDS read (or LDS-DMA request) rides behind every MFMA; the MFMA comes first
#define CQY_SCHED_DS()                                                                                                         \
  _Pragma("unroll") for (int q_ = 0; q_ < 12; q_++) {                                                                          \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);                                                                         \
  }
#define CQY_SCHED_VM()                                                                                                         \
  _Pragma("unroll") for (int q_ = 0; q_ < 6; q_++) {                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);                                                                         \
  }
  // the finished block column is the FIRST of the half step's MFMA order (4 MFMAs); its 8 stores follow the LDS-DMA requests (the counted
  // vmcnt of the hand-over relies on this order: requests, then stores - capital_amd/build.py check_qrapply_requests() reads the kernel's
  // machine code after every build and fails if a half step holds anything else)
#define CQY_SCHED_ST()                                                                                                         \
  _Pragma("unroll") for (int q_ = 0; q_ < 8; q_++) {                                                                           \
    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);                                                                         \
  }
#define CQY_WAIT(vm, lgkm) __builtin_amdgcn_s_waitcnt(((vm) & 15) | (7 << 4) | (((lgkm) & 15) << 8) | (((vm) >> 4) << 14))
  // Block column kt = 2 J + par is complete after K tile kt (J = this step's first active column index): the wave stores its 32 x 16 piece.
  // In place is fine: these columns of this row tile were K tile kt, already consumed from LDS.
#define CQY_STORE(u, J)                                                                                                        \
  if ((J) < 8 && (!(DIAG & 1) || g.ntiles < 0)) {                                                        \
    const int kt_ = (u) & 15;                                                                                                  \
    const __amdgpu_buffer_rsrc_t rO = __builtin_amdgcn_make_buffer_rsrc((void*)(g.Qout + (int64_t)(16 * kt_) * g.ldout + tile_row0(u)), 0, (int)0xffffffffu, 0x00020000); \
    _Pragma("unroll") for (int i = 0; i < 2; i++) {                                                                            \
      /* (scalar copies first: __builtin_bit_cast applied to a vector ELEMENT expression reads element 0 whatever the index) */     \
      const double e0_ = acc[i][(J) < 8 ? (J) : 0][0], e1_ = acc[i][(J) < 8 ? (J) : 0][1];                                     \
      const double e2_ = acc[i][(J) < 8 ? (J) : 0][2], e3_ = acc[i][(J) < 8 ? (J) : 0][3];                                     \
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, e0_), rO, (int)(voffO + 128 * i), 0, 0);                 \
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, e1_), rO, (int)(voffO + 128 * i), (int)(4 * ldout8), 0); \
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, e2_), rO, (int)(voffO + 128 * i), (int)(8 * ldout8), 0); \
      __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, e3_), rO, (int)(voffO + 128 * i), (int)(12 * ldout8), 0); \
    }                                                                                                                          \
  }
  // One K step, K tile index KT and column parity PAR at compile time (so the first active column J, the stored column and the A stage are
  // constants and a row tile is 16 straight-line steps; a switch over J inside one loop body made the register allocator spill).
  // The tile hand-over (wait + barrier) sits between the two k halves.  Younger than tile u + 1's pieces at the wait: A(u+2), B(u+2), A(u+3)
  // = 8 requests and the 8 stores of the one step among u - 2, u - 1 whose block column has my parity.
#define CQY_STEP(KT, PAR)                                                                                                      \
  {                                                                                                                            \
    constexpr int J_ = ((KT) - (PAR) + 1) >> 1;                                   /* first active column of K tile KT */         \
    constexpr int JN_ = (KT) == 15 ? 0 : ((KT) + 1 - (PAR) + 1) >> 1;             /* ... of the next K tile */                   \
    constexpr int NQ3_ = 4 - ((((KT) + 3) & 15) >> 2), NQ2_ = 4 - ((((KT) + 2) & 15) >> 2);   /* Rinv pieces per wave of tiles u + 3, u + 2 */ \
    const int u = u0 + (KT);                                                                                                   \
    CQY_MMA(fa0, fb0, J_, (KT) == 0)                                                                                           \
    CQY_READS(u, 1, fa1, fb1, J_)                                                                                              \
    CQY_SCHED_DS()                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
    /* younger than tile u + 1's requests: A(u+2), B(u+2), A(u+3) and the 8 stores of the one step among u - 2, u - 1 with my parity */ \
    if ((DIAG & 1) || (u0 == 0 && ((KT) == 0 || ((KT) == 1 && (PAR) == 1)))) CQY_WAIT(4 + NQ2_, 0); else CQY_WAIT(12 + NQ2_, 0); \
    __builtin_amdgcn_s_barrier();                                                                                              \
    CQY_MMA(fa1, fb1, J_, false)                                                                                               \
    CQY_ISSUE_B(u + 3, NQ3_)                                                                                                   \
    issue_a(u + 4);                                                                                                            \
    if (((KT) & 1) == (PAR)) CQY_STORE(u, J_)                                                                                  \
    CQY_READS(u + 1, 0, fa0, fb0, JN_)                                                                                         \
    __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);                                                                         \
    CQY_SCHED_VM()                                                                                                             \
    if (((KT) & 1) == (PAR)) CQY_SCHED_ST()                                                                                    \
    CQY_SCHED_DS()                                                                                                             \
    __builtin_amdgcn_sched_barrier(0);                                                                                         \
  }
#define CQY_ROWTILES(PAR)                                                                                                      \
  for (int u0 = 0; u0 < U; u0 += 16) {                                                                                         \
    CQY_STEP(0, PAR) CQY_STEP(1, PAR) CQY_STEP(2, PAR) CQY_STEP(3, PAR) CQY_STEP(4, PAR) CQY_STEP(5, PAR) CQY_STEP(6, PAR) CQY_STEP(7, PAR)     \
    CQY_STEP(8, PAR) CQY_STEP(9, PAR) CQY_STEP(10, PAR) CQY_STEP(11, PAR) CQY_STEP(12, PAR) CQY_STEP(13, PAR) CQY_STEP(14, PAR) CQY_STEP(15, PAR) \
  }
  issue_a(0); CQY_ISSUE_B(0, 4) issue_a(1); CQY_ISSUE_B(1, 4) issue_a(2);
  CQY_WAIT(8, 15);                          // tile 0 has landed: younger A(1), B(1), A(2)
  __builtin_amdgcn_s_barrier();
  CQY_ISSUE_B(2, 4) issue_a(3);
  CQY_READS(0, 0, fa0, fb0, 0)
  if (par == 0) { CQY_ROWTILES(0) } else { CQY_ROWTILES(1) }
  __builtin_amdgcn_s_waitcnt(0 | (7 << 4) | (15 << 8));
#undef CQY_READS
#undef CQY_ISSUE_B
#undef CQY_MMA
#undef CQY_SCHED_DS
#undef CQY_SCHED_VM
#undef CQY_WAIT
#undef CQY_STORE
#undef CQY_STEP
#undef CQY_ROWTILES
}

// History of this kernel (profiles/r06_experiments.md section 3, profiles/HISTORY.md):
//  - rounds 3-5: column waves (wave = 64 rows x block columns wn, wn + 4, wn + 8, wn + 12), hand-over work issued in a burst before the MFMAs of
//    a half step: 2.75 - 2.96 ms.  Per-step stamps (round 6) showed ~ 1.7 K of ~ 8 K cycles per step with the matrix pipe idle: after every barrier
//    the first wave of a SIMD issued its LDS-DMA requests and fragment reads before its first MFMA, and when it had finished its MFMAs the
//    second wave did the same - nothing of it overlapped, because of the starvation rule in the kernel's header comment.
//  - a two-group variant (one group loads while the other multiplies, as in the bf16 update) was 3.4 ms for the same reason.
//  - K-tile pairs (s, 15 - s) per step (round 4): 4 % slower.  Start stagger of the workgroups, non-temporal stores: no effect.
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
    if (diag == 1) hipLaunchKernelGGL((qrapply256_kernel<1>), gr, bl, lds, s, g);
    else if (diag == 2) hipLaunchKernelGGL((qrapply256_kernel<2>), gr, bl, lds, s, g);
    else if (diag == 3) hipLaunchKernelGGL((qrapply256_kernel<3>), gr, bl, lds, s, g);
    if (diag >= 1 && diag <= 3) { CAP_HIP(hipGetLastError()); return CAP_OK; }
  }
  hipLaunchKernelGGL((qrapply256_kernel<0>), gr, bl, lds, s, g);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}
