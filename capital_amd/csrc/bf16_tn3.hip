// bf16 trailing update, third generation (round 6): C32 += alpha A^T B on v_mfma_f32_32x32x16_bf16 with a C-STATIONARY 256 x 256 tile and a
// plain read - add - store epilogue (every C tile has one writer per launch: no atomics).
//
// Why: the 128 x 128 kernel (mixed.hip, bf16_tn_kernel) pulls 1 KiB of fragments out of LDS per MFMA (64 x 64 per wave) and 64 flop per
// byte through L2 - at the 2.5 PFLOP/s bf16 peak that is 39 TB/s of L2 -> LDS traffic, more than the chip's L2s deliver; it is LDS- and
// L2-bound at 0.25 - 0.32 of peak (profiles/r05_pmc_cqr_bf16.txt).  The second-generation kernel (256 x 128, loader waves) halves neither
// per compute wave and pays for its atomics.  Here:
//   * 256 x 256 tile per 512-thread workgroup, 8 waves as 2 (rows) x 4 (columns), 128 x 64 per wave = 8 accumulators of 32 x 32 (128
//     registers): 768 B of fragments per MFMA, 128 flop per byte through L2;
//   * ping-pong: the two row groups of waves (one wave of each per SIMD) run one PHASE apart.  A phase is either "load" (LDS-DMA pieces
//     of a stage ahead + the 12 ds_read_b128 of the wave's next 16 MFMAs) or "compute" (16 MFMAs = 512 matrix-pipe cycles); while group 0
//     computes, group 1 loads, so every SIMD always has one wave feeding the matrix pipe.  One raw s_barrier per phase; LDS-DMA stays in
//     flight across barriers (counted vmcnt);
//   * operands swapped in the MFMA (lane = C row): the epilogue's loads and stores touch 32 consecutive rows = whole 128-byte lines.
// Two stagings of K:
//   bf16_tn3_kernel<NST>  stages of 32 k (64-byte rows: 16 KiB per operand image) through a ring of NST stages (96 / 128 KiB); row r stores
//                         logical 16-byte chunk c at position c ^ ((r >> 2) & 3); one DMA piece = 16 rows
//   bf16_tn3w_kernel      stages of 64 k (128-byte rows = whole cache lines per row: every DMA piece is 8 full lines instead of 16 half
//                         lines), two stages = 128 KiB; chunk c of row r at c ^ ((r >> 1) & 7) (the image of tile_dma.h); a stage is
//                         consumed in two halves of 32 k, i.e. four phases per stage
// In both every 16-lane group of a ds_read_b128 covers all 64 banks once and the XOR sits on the SOURCE address of the lane-linear DMA.
// Same accumulation order per C element as the 128 x 128 kernel (k ascending in steps of 16, one final add into C): bit-identical results.
#include <algorithm>

#include "common.h"
#include "kargs.h"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int T3 = 256;              // C tile edge
constexpr int K3 = 32;               // narrow staging: K per stage (bf16 elements)
constexpr int RB3 = 64;              //   bytes per image row
constexpr int IMG3 = T3 * RB3;       //   one operand image: 16 KiB
constexpr int STAGE3 = 2 * IMG3;     //   A image + B image
constexpr int K3W = 64;              // wide staging: K per stage
constexpr int RB3W = 128;
constexpr int IMG3W = T3 * RB3W;     //   32 KiB
constexpr int STAGE3W = 2 * IMG3W;   //   64 KiB

// vmcnt(n) with the other counters left alone / with lgkmcnt(0)
#define CAP_VMCNT(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (15 << 8) | (((n) >> 4) << 14))
#define CAP_VMCNT_LGKM0(n) __builtin_amdgcn_s_waitcnt(((n) & 15) | (7 << 4) | (0 << 8) | (((n) >> 4) << 14))

// block -> tile: XCD b % 8 walks a contiguous range of the tile list, supertile by supertile (st x st tiles; the upper triangle of
// supertiles for square tri problems) - the walk of bf16_tn_kernel with 256-wide tiles.  false: this workgroup has no tile
__device__ __forceinline__ bool tn3_tile(const BfArgs& g, int& ti, int& tj) {
  const int b = (int)blockIdx.x;
  if ((b >> 3) >= g.chunk) return false;
  const int L = (b & 7) * g.chunk + (b >> 3);
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
  return !(ti >= g.tm || tj >= g.tn || (g.tri && ti > tj));
}

// epilogue: C += alpha * acc, plain read - add - store.  Lane holds C[i0 + grp 128 + 32 i + r32][j0 + w4 64 + 32 j + (e & 3) + 8 (e >> 2) + 4 kg];
// DEPTH blocks of C in flight (16 loads per lane each): block b + DEPTH is requested as soon as block b has been added and stored.
// Buffer addressing: ONE 32-bit per-lane offset, everything else scalar (64-bit per-element addresses cost two registers each).
template <int DEPTH = 4>
__device__ __forceinline__ void tn3_epilogue(const BfArgs& g, f32x16 (&acc)[4][2], int ti, int tj, int grp, int w4, int r32, int kg) {
  const int64_t i0 = (int64_t)ti * T3, j0 = (int64_t)tj * T3;
  const bool diag = g.tri && ti == tj;                 // wave-uniform: only these tiles mask (row <= col, tile-relative)
  const __amdgpu_buffer_rsrc_t rc = __builtin_amdgcn_make_buffer_rsrc((void*)(g.C + i0 + j0 * g.ldc), 0, (int)0xffffffffu, 0x00020000);
  const int ldc4 = (int)(g.ldc * 4);                   // (256 columns x ldc x 4 B < 2^32: checked by the launcher)
  const int voff = (grp * 128 + r32) * 4 + (w4 * 64 + 4 * kg) * ldc4;
  const float alpha = g.alpha;
  float cb[DEPTH][16];
#define CAP_C3_OFF(blk, e) (128 * ((blk) >> 1) + (32 * ((blk) & 1) + ((e) & 3) + 8 * ((e) >> 2)) * ldc4)
#define CAP_C3_LOAD(blk, buf)                                                                                        \
  _Pragma("unroll") for (int e = 0; e < 16; e++)                                                                     \
    cb[buf][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rc, voff, CAP_C3_OFF(blk, e), 0));
#pragma unroll
  for (int blk = 0; blk < DEPTH; blk++) { CAP_C3_LOAD(blk, blk) }
#pragma unroll
  for (int blk = 0; blk < 8; blk++) {
    const int i = blk >> 1, j = blk & 1;
    if (!diag) {
#pragma unroll
      for (int e = 0; e < 16; e++)
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cb[blk % DEPTH][e] + alpha * acc[i][j][e]), rc, voff, CAP_C3_OFF(blk, e), 0);
    } else {
      const int lrow = grp * 128 + 32 * i + r32;
#pragma unroll
      for (int e = 0; e < 16; e++) {
        const int lcol = w4 * 64 + 32 * j + (e & 3) + 8 * (e >> 2) + 4 * kg;
        if (lrow <= lcol)
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, cb[blk % DEPTH][e] + alpha * acc[i][j][e]), rc, voff, CAP_C3_OFF(blk, e), 0);
      }
    }
    if (blk + DEPTH < 8) { CAP_C3_LOAD(blk + DEPTH, blk % DEPTH) }
  }
#undef CAP_C3_OFF
#undef CAP_C3_LOAD
}

// the 16 MFMAs of a phase: two k-steps x (4 x 2) blocks, operands swapped (lane = C row)
// DBG & 4 (timing surgery): the fragments are only "used"
#define CAP_MMA3()                                                                                                   \
  if (DBG & 4) {                                                                                                     \
    _Pragma("unroll") for (int s = 0; s < 2; s++) {                                                                  \
      _Pragma("unroll") for (int i = 0; i < 4; i++) asm volatile("" ::"v"(fa[s][i]));                                \
      _Pragma("unroll") for (int j = 0; j < 2; j++) asm volatile("" ::"v"(fb[s][j]));                                \
    }                                                                                                                \
  } else {                                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                                   \
    _Pragma("unroll") for (int s = 0; s < 2; s++)                                                                    \
      _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                  \
        _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s][j], fa[s][i], acc[i][j], 0, 0, 0);               \
    __builtin_amdgcn_s_setprio(0);                                                                                   \
  }
#define CAP_ACC3_ZERO()                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                      \
    _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                    \
      _Pragma("unroll") for (int e = 0; e < 16; e++) acc[i][j][e] = 0.0f;
#define CAP_ACC3_KEEP()                                                                                              \
  _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                      \
    _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                    \
      _Pragma("unroll") for (int e = 0; e < 16; e++) asm volatile("" ::"v"(acc[i][j][e]));

// DBG (timing surgery, WRONG results, CAP_EXPERIMENTS builds only): 1 no LDS-DMA after the prologue, 2 fragment reads only for stage 0,
// 4 no MFMAs, 8 no epilogue
template <int NST, int DBG = 0>
__global__ void __launch_bounds__(512, 2) bf16_tn3_kernel(const BfArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  char* lds = reinterpret_cast<char*>(smem);
  int ti, tj;
  if (!tn3_tile(g, ti, tj)) return;
  const int64_t i0 = (int64_t)ti * T3, j0 = (int64_t)tj * T3;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wid >> 2, w4 = wid & 3;           // row group (0: rows 0-127, 1: rows 128-255), column wave (64 columns each)
  const int r32 = lane & 31, kg = lane >> 5;
  const int nk = (int)(g.K / K3);

  // ---- LDS-DMA: group 0 moves the A image of a stage, group 1 the B image; wave w4 the pieces 4 w4 .. 4 w4 + 3 (16 rows each)
  const __bf16* src = grp ? g.B + j0 * g.ldb : g.A + i0 * g.lda;
  const uint32_t rowbytes = (uint32_t)((grp ? g.ldb : g.lda) * 2);
  const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (int)0xffffffffu, 0x00020000);
  const uint32_t voff = (uint32_t)(lane >> 2) * rowbytes + (uint32_t)((((lane & 3) ^ ((lane >> 4) & 3))) << 4);
  const uint32_t piece0 = (uint32_t)(w4 * 4);
  auto issue = [&](int st) {                         // stage st (< nk) into slot st % NST
    char* dst = lds + (st % NST) * STAGE3 + grp * IMG3;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t p = piece0 + q;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, (int)voff,
                                               (int)(p * 16u * rowbytes + (uint32_t)st * RB3), 0, 0);
    }
  };
  // "stage v has landed" = at most the pieces of the younger stages this wave has issued are outstanding (loads retire in order)
  auto wait_landed = [&](int younger, bool lgkm0) {
    if (lgkm0) {
      if (younger >= 2) CAP_VMCNT_LGKM0(8); else if (younger == 1) CAP_VMCNT_LGKM0(4); else CAP_VMCNT_LGKM0(0);
    } else {
      if (younger >= 2) CAP_VMCNT(8); else if (younger == 1) CAP_VMCNT(4); else CAP_VMCNT(0);
    }
  };

  // ---- fragments: A rows grp * 128 + 32 i + r32, B rows w4 * 64 + 32 j + r32; k-step s reads logical chunk 2 s + kg
  const int t = kg ^ ((r32 >> 2) & 3);
  const int a_row = (grp * 128 + r32) * RB3, b_row = IMG3 + (w4 * 64 + r32) * RB3;
  const int o0 = t << 4, o1 = (t ^ 2) << 4;
  bf16x8 fa[2][4], fb[2][2];
  auto read_frags = [&](int st) {
    const char* base = lds + (st % NST) * STAGE3;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      fa[0][i] = *reinterpret_cast<const bf16x8*>(base + a_row + o0 + i * 32 * RB3);
      fa[1][i] = *reinterpret_cast<const bf16x8*>(base + a_row + o1 + i * 32 * RB3);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      fb[0][j] = *reinterpret_cast<const bf16x8*>(base + b_row + o0 + j * 32 * RB3);
      fb[1][j] = *reinterpret_cast<const bf16x8*>(base + b_row + o1 + j * 32 * RB3);
    }
  };
  f32x16 acc[4][2];
  CAP_ACC3_ZERO()

  // ---- prologue: stages 0 .. NST - 2 on their way, stage 0 landed and visible
  const int npro = nk < NST - 1 ? nk : NST - 1;
  for (int st = 0; st < npro; st++) issue(st);
  wait_landed(npro - 1, false);
  __builtin_amdgcn_s_barrier();
  // younger stages in flight when stage t + 1 must have landed: stages t + 2 .. min(nk - 1, t + NST - 1)
  auto younger_at = [&](int tt) { const int hi = tt + NST - 1 < nk - 1 ? tt + NST - 1 : nk - 1; return hi - (tt + 1) > 0 ? hi - (tt + 1) : 0; };
  if (grp == 0) {
    for (int tt = 0; tt < nk; tt++) {
      // load phase 2 tt: the slot of stage tt - 1 was read for the last time (by group 1) before the barrier just passed
      if (!(DBG & 1) && tt + NST - 1 < nk) issue(tt + NST - 1);
      if (!(DBG & 2) || tt == 0) read_frags(tt);
      CAP_VMCNT_LGKM0(63);                            // lgkmcnt(0): my reads of stage tt are done
      __builtin_amdgcn_s_barrier();
      // compute phase 2 tt + 1
      CAP_MMA3()
      wait_landed(younger_at(tt), false);             // my share of stage tt + 1 has landed
      __builtin_amdgcn_s_barrier();
    }
  } else {
    __builtin_amdgcn_s_barrier();                     // one phase behind group 0
    for (int tt = 0; tt < nk; tt++) {
      // load phase 2 tt + 1
      if (!(DBG & 1) && tt + NST - 1 < nk) issue(tt + NST - 1);
      if (!(DBG & 2) || tt == 0) read_frags(tt);
      wait_landed(younger_at(tt), true);              // my reads of stage tt are done, my share of stage tt + 1 has landed
      __builtin_amdgcn_s_barrier();
      // compute phase 2 tt + 2
      CAP_MMA3()
      if (tt + 1 < nk) __builtin_amdgcn_s_barrier();
    }
  }
  if (DBG & 8) { CAP_ACC3_KEEP() return; }
  tn3_epilogue(g, acc, ti, tj, grp, w4, r32, kg);
}

// Wide staging: a stage = 64 k of the whole tile (A image 256 rows x 128 B + B image 256 rows x 128 B = 64 KiB), two slots.  Stage t is
// consumed in two halves (k-steps 0-1, 2-3): four phases.  Group 0: 4t load half 0 (+ its DMA of stage t + 1), 4t + 1 compute, 4t + 2 load
// half 1, 4t + 3 compute; group 1 one phase later.  Slot of stage t + 1 = slot of stage t - 1, read for the last time by group 1 in phase
// 4t - 1, so the DMA may start in phase 4t (group 0) / 4t + 1 (group 1).  Every wave moves 4 pieces of B and 4 pieces of A (the rows
// grp * 128 + 32 w4 .. + 31 of either image), B FIRST: group 1's B rows are needed by group 0 in phase 4t + 4 (flight 2.5 phases), its A
// rows (128-255) only by group 1 itself in phase 4t + 5, so only a quarter of the stage has the short flight.
template <int DBG = 0>
__global__ void __launch_bounds__(512, 2) bf16_tn3w_kernel(const BfArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  char* lds = reinterpret_cast<char*>(smem);
  int ti, tj;
  if (!tn3_tile(g, ti, tj)) return;
  const int64_t i0 = (int64_t)ti * T3, j0 = (int64_t)tj * T3;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wid >> 2, w4 = wid & 3;
  const int r32 = lane & 31, kg = lane >> 5;
  const int nk = (int)(g.K / K3W);

  // ---- LDS-DMA: a piece = 8 rows x 128 B
  const int rsub = lane >> 3, p8 = lane & 7;
  const uint32_t rbA = (uint32_t)(g.lda * 2), rbB = (uint32_t)(g.ldb * 2);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + i0 * g.lda), 0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)(g.B + j0 * g.ldb), 0, (int)0xffffffffu, 0x00020000);
  // row = 8 piece + rsub: (row >> 1) & 7 = (4 (piece & 1) + (rsub >> 1)) & 7 -> even / odd pieces differ in bit 2 of the XOR
  const uint32_t sw_e = (uint32_t)((p8 ^ (rsub >> 1)) << 4), sw_o = (uint32_t)((p8 ^ ((rsub >> 1) ^ 4)) << 4);
  const uint32_t vA_e = rsub * rbA + sw_e, vA_o = rsub * rbA + sw_o, vB_e = rsub * rbB + sw_e, vB_o = rsub * rbB + sw_o;
  const uint32_t piece0 = (uint32_t)(grp * 16 + w4 * 4);           // first of my 4 pieces in either image (32 pieces each)
  auto issue = [&](int st) {                                        // stage st (< nk) into slot st & 1: B pieces first, then A pieces
    char* dst = lds + (st & 1) * STAGE3W;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t p = piece0 + q;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(dst + IMG3W + p * 1024), 16, (int)((q & 1) ? vB_o : vB_e),
                                               (int)(p * 8u * rbB + (uint32_t)st * RB3W), 0, 0);
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t p = piece0 + q;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, (int)((q & 1) ? vA_o : vA_e),
                                               (int)(p * 8u * rbA + (uint32_t)st * RB3W), 0, 0);
    }
  };

  // ---- fragments: k-step s (0 .. 3) reads logical chunk 2 s + kg of rows (block origin + r32): position (2 s) ^ t, t = kg ^ ((r32 >> 1) & 7)
  const int t = kg ^ ((r32 >> 1) & 7);
  const int a_row = (grp * 128 + r32) * RB3W, b_row = IMG3W + (w4 * 64 + r32) * RB3W;
  bf16x8 fa[2][4], fb[2][2];
  auto read_frags = [&](int st, int h) {
    const char* base = lds + (st & 1) * STAGE3W;
    const int o0 = ((4 * h) ^ t) << 4, o1 = ((4 * h + 2) ^ t) << 4;
#pragma unroll
    for (int i = 0; i < 4; i++) {
      fa[0][i] = *reinterpret_cast<const bf16x8*>(base + a_row + o0 + i * 32 * RB3W);
      fa[1][i] = *reinterpret_cast<const bf16x8*>(base + a_row + o1 + i * 32 * RB3W);
    }
#pragma unroll
    for (int j = 0; j < 2; j++) {
      fb[0][j] = *reinterpret_cast<const bf16x8*>(base + b_row + o0 + j * 32 * RB3W);
      fb[1][j] = *reinterpret_cast<const bf16x8*>(base + b_row + o1 + j * 32 * RB3W);
    }
  };
  f32x16 acc[4][2];
  CAP_ACC3_ZERO()

  issue(0);
  CAP_VMCNT(0);
  __builtin_amdgcn_s_barrier();
  if (grp == 0) {
    for (int tt = 0; tt < nk; tt++) {
      // phase 4 tt: load half 0; stage tt + 1 goes into the slot group 1 left before the barrier just passed
      if (!(DBG & 1) && tt + 1 < nk) issue(tt + 1);
      if (!(DBG & 2) || tt == 0) read_frags(tt, 0);
      CAP_VMCNT_LGKM0(63);
      __builtin_amdgcn_s_barrier();
      CAP_MMA3()                                      // phase 4 tt + 1
      __builtin_amdgcn_s_barrier();
      if (!(DBG & 2)) read_frags(tt, 1);              // phase 4 tt + 2
      CAP_VMCNT_LGKM0(63);
      __builtin_amdgcn_s_barrier();
      CAP_MMA3()                                      // phase 4 tt + 3
      CAP_VMCNT(0);                                   // my 8 pieces of stage tt + 1 have landed
      __builtin_amdgcn_s_barrier();
    }
  } else {
    __builtin_amdgcn_s_barrier();                     // one phase behind group 0
    for (int tt = 0; tt < nk; tt++) {
      // phase 4 tt + 1: load half 0 (+ my DMA of stage tt + 1; my A pieces of stage tt landed before the barrier just passed)
      if (!(DBG & 1) && tt + 1 < nk) issue(tt + 1);
      if (!(DBG & 2) || tt == 0) read_frags(tt, 0);
      CAP_VMCNT_LGKM0(63);
      __builtin_amdgcn_s_barrier();
      CAP_MMA3()                                      // phase 4 tt + 2
      __builtin_amdgcn_s_barrier();
      if (!(DBG & 2)) read_frags(tt, 1);              // phase 4 tt + 3
      CAP_VMCNT_LGKM0(4);                             // my B pieces of stage tt + 1 have landed (group 0 reads them in phase 4 tt + 4); A may fly on
      __builtin_amdgcn_s_barrier();
      CAP_MMA3()                                      // phase 4 tt + 4
      if (tt + 1 < nk) {
        CAP_VMCNT(0);                                 // my A pieces (rows 128-255: read by group 1 from phase 4 tt + 5 on)
        __builtin_amdgcn_s_barrier();
      }
    }
  }
  if (DBG & 8) { CAP_ACC3_KEEP() return; }
  tn3_epilogue(g, acc, ti, tj, grp, w4, r32, kg);
}

// Long phases (variant 6): the wide staging with ONE load phase and ONE compute phase per 64-k stage - 32 MFMAs (1024 matrix-pipe cycles) between
// barriers instead of 16.  The MFMA + barrier skeleton of the 16-MFMA phases reaches 0.72 of peak: ~ 200 cycles per phase are hand-over (the two
// wave groups alternate strictly), and they halve here; the price is 96 fragment registers (4 k-steps) instead of 48.  Group 0 moves ALL of B and its
// own A rows (12 pieces per wave) at the start of its load phase 2t (flight: 2 phases = 2048 cycles until group 0 reads them in phase 2t + 2);
// group 1 moves only its own A rows (4 pieces per wave) in phase 2t + 1 and reads them in phase 2t + 3.
template <int DBG = 0>
__global__ void __launch_bounds__(512, 2) bf16_tn3x_kernel(const BfArgs g) {
  extern __shared__ __attribute__((aligned(16))) double smem[];
  char* lds = reinterpret_cast<char*>(smem);
  int ti, tj;
  if (!tn3_tile(g, ti, tj)) return;
  const int64_t i0 = (int64_t)ti * T3, j0 = (int64_t)tj * T3;
  const int lane = threadIdx.x & 63;
  const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int grp = wid >> 2, w4 = wid & 3;
  const int r32 = lane & 31, kg = lane >> 5;
  const int nk = (int)(g.K / K3W);

  const int rsub = lane >> 3, p8 = lane & 7;
  const uint32_t rbA = (uint32_t)(g.lda * 2), rbB = (uint32_t)(g.ldb * 2);
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)(g.A + i0 * g.lda), 0, (int)0xffffffffu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)(g.B + j0 * g.ldb), 0, (int)0xffffffffu, 0x00020000);
  const uint32_t sw_e = (uint32_t)((p8 ^ (rsub >> 1)) << 4), sw_o = (uint32_t)((p8 ^ ((rsub >> 1) ^ 4)) << 4);
  const uint32_t vA_e = rsub * rbA + sw_e, vA_o = rsub * rbA + sw_o, vB_e = rsub * rbB + sw_e, vB_o = rsub * rbB + sw_o;
  auto issue = [&](int st) {
    char* dst = lds + (st & 1) * STAGE3W;
    if (grp == 0) {
#pragma unroll
      for (int q = 0; q < 8; q++) {                     // B: pieces 8 w4 .. 8 w4 + 7 of 32
        const uint32_t p = (uint32_t)(w4 * 8 + q);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rB, (__attribute__((address_space(3))) void*)(dst + IMG3W + p * 1024), 16, (int)((q & 1) ? vB_o : vB_e),
                                                 (int)(p * 8u * rbB + (uint32_t)st * RB3W), 0, 0);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {                       // A: my group's rows, pieces grp * 16 + 4 w4 + q
      const uint32_t p = (uint32_t)(grp * 16 + w4 * 4 + q);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, (int)((q & 1) ? vA_o : vA_e),
                                               (int)(p * 8u * rbA + (uint32_t)st * RB3W), 0, 0);
    }
  };
  const int t = kg ^ ((r32 >> 1) & 7);
  const int a_row = (grp * 128 + r32) * RB3W, b_row = IMG3W + (w4 * 64 + r32) * RB3W;
  bf16x8 fa[4][4], fb[4][2];
  auto read_frags = [&](int st) {
    const char* base = lds + (st & 1) * STAGE3W;
#pragma unroll
    for (int s = 0; s < 4; s++) {
      const int o = ((2 * s) ^ t) << 4;
#pragma unroll
      for (int i = 0; i < 4; i++) fa[s][i] = *reinterpret_cast<const bf16x8*>(base + a_row + o + i * 32 * RB3W);
#pragma unroll
      for (int j = 0; j < 2; j++) fb[s][j] = *reinterpret_cast<const bf16x8*>(base + b_row + o + j * 32 * RB3W);
    }
  };
  f32x16 acc[4][2];
  CAP_ACC3_ZERO()
#define CAP_MMA3X()                                                                                                  \
  if (DBG & 4) {                                                                                                     \
    _Pragma("unroll") for (int s = 0; s < 4; s++) {                                                                  \
      _Pragma("unroll") for (int i = 0; i < 4; i++) asm volatile("" ::"v"(fa[s][i]));                                \
      _Pragma("unroll") for (int j = 0; j < 2; j++) asm volatile("" ::"v"(fb[s][j]));                                \
    }                                                                                                                \
  } else {                                                                                                           \
    __builtin_amdgcn_s_setprio(1);                                                                                   \
    _Pragma("unroll") for (int s = 0; s < 4; s++)                                                                    \
      _Pragma("unroll") for (int i = 0; i < 4; i++)                                                                  \
        _Pragma("unroll") for (int j = 0; j < 2; j++)                                                                \
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[s][j], fa[s][i], acc[i][j], 0, 0, 0);               \
    __builtin_amdgcn_s_setprio(0);                                                                                   \
  }
  issue(0);
  CAP_VMCNT(0);
  __builtin_amdgcn_s_barrier();
  if (grp == 0) {
    for (int tt = 0; tt < nk; tt++) {
      if (!(DBG & 1) && tt + 1 < nk) issue(tt + 1);   // phase 2 tt: the slot was read for the last time (group 1, phase 2 tt - 1) before the barrier just passed
      if (!(DBG & 2) || tt == 0) read_frags(tt);
      CAP_VMCNT_LGKM0(63);
      __builtin_amdgcn_s_barrier();
      CAP_MMA3X()                                     // phase 2 tt + 1
      CAP_VMCNT(0);                                   // my 12 pieces of stage tt + 1 have landed
      __builtin_amdgcn_s_barrier();
    }
  } else {
    __builtin_amdgcn_s_barrier();
    for (int tt = 0; tt < nk; tt++) {
      if (!(DBG & 1) && tt + 1 < nk) issue(tt + 1);   // phase 2 tt + 1: my A rows of stage tt + 1 (read by me in phase 2 tt + 3)
      if (!(DBG & 2) || tt == 0) read_frags(tt);
      CAP_VMCNT_LGKM0(63);
      __builtin_amdgcn_s_barrier();
      CAP_MMA3X()                                     // phase 2 tt + 2
      if (tt + 1 < nk) {
        CAP_VMCNT(0);                                 // my A pieces of stage tt + 1
        __builtin_amdgcn_s_barrier();
      }
    }
  }
#undef CAP_MMA3X
  if (DBG & 8) { CAP_ACC3_KEEP() return; }
  tn3_epilogue(g, acc, ti, tj, grp, w4, r32, kg);
}
#undef CAP_MMA3
#undef CAP_ACC3_ZERO
#undef CAP_ACC3_KEEP

template <typename KernelT>
int launch_tn3_any(KernelT kernel, int lds_bytes, bool& attr_done, const BfArgs& g, unsigned grid, hipStream_t s) {
  if (!attr_done) {
    CAP_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes));
    attr_done = true;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(512), lds_bytes, s, g);
  CAP_HIP(hipGetLastError());
  return CAP_OK;
}
// (the attribute belongs to the device's copy of the function: one flag per device and instantiation)
template <int NST, int DBG = 0>
int launch_tn3(const BfArgs& g, unsigned grid, hipStream_t s) {
  static bool attr_set[16] = {};
  int dev = 0;
  CAP_HIP(hipGetDevice(&dev));
  bool scratch = false;
  return launch_tn3_any(bf16_tn3_kernel<NST, DBG>, NST * STAGE3, (dev >= 0 && dev < 16) ? attr_set[dev] : scratch, g, grid, s);
}
template <int DBG = 0>
int launch_tn3x(const BfArgs& g, unsigned grid, hipStream_t s) {
  static bool attr_set[16] = {};
  int dev = 0;
  CAP_HIP(hipGetDevice(&dev));
  bool scratch = false;
  return launch_tn3_any(bf16_tn3x_kernel<DBG>, 2 * STAGE3W, (dev >= 0 && dev < 16) ? attr_set[dev] : scratch, g, grid, s);
}
template <int DBG = 0>
int launch_tn3w(const BfArgs& g, unsigned grid, hipStream_t s) {
  static bool attr_set[16] = {};
  int dev = 0;
  CAP_HIP(hipGetDevice(&dev));
  bool scratch = false;
  return launch_tn3_any(bf16_tn3w_kernel<DBG>, 2 * STAGE3W, (dev >= 0 && dev < 16) ? attr_set[dev] : scratch, g, grid, s);
}

}  // namespace

// nst 3 / 4: narrow staging (K % 32 == 0); nst 5: wide staging (K % 64 == 0)
bool cap_bf16_tn3_applies(int64_t m, int64_t n, int64_t k, int64_t lda, int64_t ldb, int tri, int nst) {
  return m > 0 && n > 0 && k > 0 && m % T3 == 0 && n % T3 == 0 && k % (nst >= 5 ? K3W : K3) == 0 && lda % 8 == 0 && ldb % 8 == 0 && (!tri || m <= n) &&
         256 * lda * 2 + k * 2 < 0xfffffff0LL && 256 * ldb * 2 + k * 2 < 0xfffffff0LL;
}

// C32[m x n] += alpha A^T B (A: k x m, B: k x n bf16, K-contiguous), upper tiles / elements only when tri.  nst = 3 / 4: ring of 3 / 4 stages
// of 32 k (96 / 128 KiB), 5: two stages of 64 k (128 KiB); st = supertile edge in tiles.
int cap_bf16_tn3_launch(int64_t m, int64_t n, int64_t k, float alpha, const void* A16, int64_t lda, const void* B16, int64_t ldb, float* C,
                        int64_t ldc, int tri, int nst, int st, hipStream_t s, int dbg) {
  if (m <= 0 || n <= 0 || k <= 0) return CAP_OK;
  if (!cap_bf16_tn3_applies(m, n, k, lda, ldb, tri, nst) || st < 1 || 257 * ldc * 4 >= 0x7ffffff0LL) return CAP_ERR_UNSUPPORTED;
  BfArgs g;
  g.A = (const __bf16*)A16; g.B = (const __bf16*)B16; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = m; g.N = n; g.K = k; g.alpha = alpha; g.tri = tri;
  g.stair = 0; g.sP = 1; g.sp = 0; g.snbT = 1; g.sJ0 = 0; g.slb0 = 0; g.gpiece = 0;
  for (int i = 0; i < 8; i++) g.gstart[i] = 0;
  g.tm = (int)(m / T3); g.tn = (int)(n / T3);
  g.st = st; g.nsm = (int)cap_ceil_div(g.tm, st);
  const int64_t nsn = cap_ceil_div(g.tn, st);
  const int64_t tiles = ((tri && m == n) ? nsn * (nsn + 1) / 2 : (int64_t)g.nsm * nsn) * st * st;
  g.chunk = (int)cap_ceil_div(tiles, 8);
  // access notes: every C tile has one writer and is read and written in place
  cap_acc_r(A16, lda, k, m, 0, 2); cap_acc_r(B16, ldb, k, n, 0, 2); cap_acc(CAP_ACC_RW, C, ldc, m, n, tri ? 1 : 0, 4);
  const unsigned grid = (unsigned)(g.chunk * 8);
  if constexpr (CAP_EXPERIMENTS) {
    const bool wide = nst == 5;
    if (nst == 6) switch (dbg) {
      case 1: return launch_tn3x<1>(g, grid, s);
      case 8: return launch_tn3x<8>(g, grid, s);
      case 9: return launch_tn3x<9>(g, grid, s);
      case 11: return launch_tn3x<11>(g, grid, s);
      case 12: return launch_tn3x<12>(g, grid, s);
      default: break;
    }
    switch (dbg) {
      case 1: return wide ? launch_tn3w<1>(g, grid, s) : launch_tn3<3, 1>(g, grid, s);
      case 2: return wide ? launch_tn3w<2>(g, grid, s) : launch_tn3<3, 2>(g, grid, s);
      case 3: return wide ? launch_tn3w<3>(g, grid, s) : launch_tn3<3, 3>(g, grid, s);
      case 4: return wide ? launch_tn3w<4>(g, grid, s) : launch_tn3<3, 4>(g, grid, s);
      case 8: return wide ? launch_tn3w<8>(g, grid, s) : launch_tn3<3, 8>(g, grid, s);
      case 9: return wide ? launch_tn3w<9>(g, grid, s) : launch_tn3<3, 9>(g, grid, s);
      case 11: return wide ? launch_tn3w<11>(g, grid, s) : launch_tn3<3, 11>(g, grid, s);
      case 12: return wide ? launch_tn3w<12>(g, grid, s) : launch_tn3<3, 12>(g, grid, s);
      default: break;
    }
  }
  if (dbg) return CAP_ERR_UNSUPPORTED;
  if (nst == 6) return launch_tn3x<0>(g, grid, s);
  if (nst == 5) return launch_tn3w<0>(g, grid, s);
  return nst >= 4 ? launch_tn3<4>(g, grid, s) : launch_tn3<3>(g, grid, s);
}
