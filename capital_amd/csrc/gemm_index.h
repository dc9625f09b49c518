// Tile enumeration and staircase index maps of the fp64 GEMM kernels (gemm.hip), host- and device-callable: the kernels walk their
// blocks through these functions, and so do the CPU kernel models of tests/hipshim (kernels_cpu.cpp) - what a launch's grid covers is
// decided by the same code on both sides.
#pragma once
#include "kargs.h"
#if defined(__HIPCC__) || defined(__HIP__)
#define CAP_HD __host__ __device__ __forceinline__
#else
#define CAP_HD static inline
#endif

namespace {

// global tile index (relative to the row origin) of local column tile tj under the staircase view
CAP_HD int stair_gtj(const GemmArgs& g, int tj) {
  const int J = g.sp + g.sP * (g.slb0 + tj / g.snbT);
  return (J - g.sJ0) * g.snbT + tj % g.snbT;
}
// global tile index (relative to the row origin sJ0) of local row tile ti under the staircase view
CAP_HD int stair_gti(const GemmArgs& g, int ti) {
  const int I = g.rp + g.rP * (g.rlb0 + ti / g.snbT);
  return (I - g.sJ0) * g.snbT + ti % g.snbT;
}
// base of A's rows for row tile ti (K-contiguous operand, lda doubles per row).  gather: global block I of the row tile
// sits in piece (I % sP) / rP (the contributors of a process row are the columns pc' = rp mod rP, + rP, ...: all of them
// when rP = 1) at that contributor's local block I / sP - gstart[piece]
CAP_HD const double* a_tile_base(const GemmArgs& g, int ti) {
  if (!g.gather) return g.A + (int64_t)ti * 128 * g.lda;
  const int I = g.rp + g.rP * (g.rlb0 + ti / g.snbT), r = (I % g.sP) / g.rP, lb = I / g.sP - g.gstart[r];
  return g.A + (int64_t)r * g.gpiece + ((int64_t)(lb * g.snbT + ti % g.snbT) * 128) * g.lda;
}

// staircase enumeration (etri == 3): supertile column sj holds this many supertiles with at least one valid tile
CAP_HD int stair_gtj_hd(int sp, int sP, int slb0, int snbT, int sJ0, int tj) {
  const int J = sp + sP * (slb0 + tj / snbT);
  return (J - sJ0) * snbT + tj % snbT;
}
// number of LOCAL row tiles whose global tile index is <= X (rows block-cyclic over rP process rows, see GemmArgs::rP)
CAP_HD int stair_rows_le(int X, int snbT, int sJ0, int rP, int rp, int rlb0) {
  if (X < 0) return 0;
  const int Xb = X / snbT + sJ0, Xo = X % snbT;          // global block of tile X, offset inside it
  // local blocks b >= 0 with  rp + rP (rlb0 + b) < Xb  are complete
  int full = Xb - rp > 0 ? (Xb - rp + rP - 1) / rP - rlb0 : -rlb0;
  if (full < 0) full = 0;
  int cnt = full * snbT;
  if (Xb >= rp && (Xb - rp) % rP == 0 && (Xb - rp) / rP >= rlb0) cnt += Xo + 1;     // the block of X itself is local
  return cnt;
}
// (supertiles of stm x stn tiles: on a 1 x P layout the staircase climbs P row tiles per local column tile, so a supertile ONE block
//  column wide and 64 / snbT tiles tall leaves one partial supertile per block column where a square one leaves ~ P of them - the XCDs'
//  contiguous slot ranges then carry equal work)
CAP_HD int stair_cnt(int sj, int stm, int stn, int tn, int nsm, int sp, int sP, int slb0, int snbT, int sJ0,
                                                  int rP, int rp, int rlb0) {
  int tjm = sj * stn + stn - 1;
  if (tjm > tn - 1) tjm = tn - 1;
  const int rows = stair_rows_le(stair_gtj_hd(sp, sP, slb0, snbT, sJ0, tjm), snbT, sJ0, rP, rp, rlb0);
  const int c = (rows + stm - 1) / stm;
  return c < nsm ? c : nsm;
}

// logical slot -> tile coordinates (returns false when the slot is empty)
CAP_HD bool slot_to_tile(const GemmArgs& g, int L, int& ti, int& tj) {
  const int STM = g.stm, STN = g.stn;
  int S = L / (STM * STN), w = L % (STM * STN);
  int si, sj;
  if (g.etri == 3) {  // staircase: walk the supertile columns, only supertiles that hold valid tiles are numbered
    sj = 0;
    for (; sj < g.nsn; sj++) {
      const int c = stair_cnt(sj, STM, STN, g.tn, g.nsm, g.sp, g.sP, g.slb0, g.snbT, g.sJ0, g.rP, g.rp, g.rlb0);
      if (S < c) break;
      S -= c;
    }
    si = S;
  } else if (g.etri == 1) {  // upper triangle of supertiles, column-major: S = sj(sj+1)/2 + si
    sj = (int)((__builtin_sqrtf(8.0f * (float)S + 1.0f) - 1.0f) * 0.5f);
    while ((sj + 1) * (sj + 2) / 2 <= S) sj++;
    while (sj * (sj + 1) / 2 > S) sj--;
    si = S - sj * (sj + 1) / 2;
  } else if (g.etri == 2) {
    si = (int)((__builtin_sqrtf(8.0f * (float)S + 1.0f) - 1.0f) * 0.5f);
    while ((si + 1) * (si + 2) / 2 <= S) si++;
    while (si * (si + 1) / 2 > S) si--;
    sj = S - si * (si + 1) / 2;
  } else if (g.sorder) {
    sj = S % g.nsn; si = S / g.nsn;
  } else {
    si = S % g.nsm; sj = S / g.nsm;
  }
  if (si >= g.nsm || sj >= g.nsn) return false;
  ti = si * STM + (w % STM); tj = sj * STN + (w / STM);
  if (ti >= g.tm || tj >= g.tn) return false;
  if (g.stair) return stair_gti(g, ti) <= stair_gtj(g, tj);
  if (g.tri == 1 && ti > tj) return false;
  if (g.tri == 2 && ti < tj) return false;
  return true;
}

}  // namespace
