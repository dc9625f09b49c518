// TEST INFRASTRUCTURE - CPU models of the library's kernels for the recording stand-in (hipshim.cpp, compute mode).
//
// In compute mode a launch is not only a line in the trace: the stand-in hands the kernel's name, its grid and the very argument bytes
// the library passed to hipLaunchKernel to shim_cpu_kernel below, which does on the host what the kernel does on the GPU - in enqueue
// order (a legal order: every happens-before edge of a trace points forward in it, and the replay shows there are no races).  The host
// side of the library - plans, index arithmetic, every pointer, leading dimension, flag and grid size it computes - then produces a real
// factor on a machine without a GPU, which tests/hipshim/run_compute.py compares with NumPy.
//
// What is modelled is the kernels' CONTRACT as the sources state it (capital_amd/csrc/*.hip), launch geometry included: the GEMM models
// walk the launch's blocks through the library's own tile enumeration (csrc/gemm_index.h) and apply the K-range trimming of the
// triangular-operand hints at the granularity the kernels use (tiles, 16 x 16 blocks of the skipping variants) - a grid that misses a
// tile or a hint on an operand that is not triangular in memory gives a wrong factor here as it would on the GPU.  Arithmetic order is
// NOT modelled (results agree to rounding, not bit for bit), nor anything inside a launch (LDS, waves, the chain's meetings).
// Nothing here is product code; nothing here is used to produce a result the library returns.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "kargs.h"
#include "gemm_index.h"

namespace {

template <class T> T arg(void** a, int i) { T v; memcpy(&v, a[i], sizeof(T)); return v; }

// the i-th template argument of a mangled kernel name "...kernelILb1ELi0E...E": bools and ints only
long tmpl(const std::string& name, int idx) {
  size_t p = name.find("kernelI");
  if (p == std::string::npos) return 0;
  p += 7;
  for (int i = 0; p < name.size() && name[p] == 'L'; i++) {
    size_t e = name.find('E', p);
    if (e == std::string::npos) break;
    std::string tok = name.substr(p + 2, e - p - 2);          // after "Lb" / "Li"
    long v = 0; bool neg = false;
    for (char c : tok) { if (c == 'n') neg = true; else if (c >= '0' && c <= '9') v = v * 10 + (c - '0'); }
    if (i == idx) return neg ? -v : v;
    p = e + 1;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------ fp64 products
struct Acc { std::vector<double> v; int m, n; Acc(int m_, int n_) : v((size_t)m_ * n_, 0.0), m(m_), n(n_) {} double& at(int i, int j) { return v[(size_t)i + (size_t)j * m]; } };

// acc[i][j] += sum_{k in [k0, k1)} a(i, k) b(k, j): a / b given by base pointer + two strides (row/outer stride, k stride)
void mma(Acc& c, int i0, int i1, int j0, int j1, const double* A, int64_t as_o, int64_t as_k, const double* B, int64_t bs_o, int64_t bs_k, int64_t k0, int64_t k1) {
  if (k1 <= k0) return;
  if (as_o == 1) {                                     // rows contiguous (op(A) = A): one axpy per (k, j)
    for (int j = j0; j < j1; j++) {
      double* cj = &c.at(0, j);
      for (int64_t k = k0; k < k1; k++) {
        const double b = B[j * bs_o + k * bs_k]; const double* ak = A + k * as_k;
#pragma omp simd
        for (int i = i0; i < i1; i++) cj[i] += ak[i] * b;
      }
    }
    return;
  }
  for (int j = j0; j < j1; j++) {
    const double* bj = B + j * bs_o;
    for (int i = i0; i < i1; i++) {
      const double* ai = A + i * as_o;
      double s = 0.0;
      if (as_k == 1 && bs_k == 1) {
#pragma omp simd reduction(+ : s)
        for (int64_t k = k0; k < k1; k++) s += ai[k] * bj[k];
      }
      else { for (int64_t k = k0; k < k1; k++) s += ai[k * as_k] * bj[k * bs_k]; }
      c.at(i, j) += s;
    }
  }
}

constexpr int TB = 128, TK = 16;

// dgemm_kernel<A_KC, B_KC, EDGE, TAG>: the register-staged kernels (any shape)
void k_dgemm(const std::string& name, void** a, unsigned gx, unsigned gy) {
  const GemmArgs g = arg<GemmArgs>(a, 0);
  const bool akc = tmpl(name, 0), bkc = tmpl(name, 1);
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (unsigned kz = 0; kz < gy; kz++)
    for (unsigned bx = 0; bx < gx; bx++) {
      const int b = (int)((bx + kz) % gx), L = (b & 7) * g.chunk + (b >> 3);
      int ti, tj;
      if ((b >> 3) >= g.chunk || !slot_to_tile(g, L, ti, tj)) continue;
      const int64_t i0 = (int64_t)ti * TB, j0 = (int64_t)tj * TB;
      const int mi = (int)std::min<int64_t>(TB, g.M - i0), nj = (int)std::min<int64_t>(TB, g.N - j0);
      const int64_t kbeg = (int64_t)kz * g.kchunk, kend = std::min<int64_t>(kbeg + g.kchunk, g.K);
      Acc c(TB, TB);
      const double* A = akc ? g.A + i0 * g.lda : g.A + i0;
      const double* B = bkc ? g.B + j0 * g.ldb : g.B + j0;
      mma(c, 0, mi, 0, nj, A, akc ? g.lda : 1, akc ? 1 : g.lda, B, bkc ? g.ldb : 1, bkc ? 1 : g.ldb, kbeg, kend);
      for (int j = 0; j < nj; j++)
        for (int i = 0; i < mi; i++) {
          const int64_t row = i0 + i, col = j0 + j;
          if (g.tri == 1 && row > col) continue;
          if (g.tri == 2 && row < col) continue;
          const double v = g.alpha * c.at(i, j);
          if (g.ksplit > 1) g.P[(int64_t)kz * g.slab + row + col * g.M] = v;
          else { double* pc = g.C + row + col * g.ldc; *pc = g.beta != 0.0 ? v + g.beta * (*pc) : v; }
        }
    }
}

// dgemm_tn_dma_kernel<TAG, A_MC, DIAG, BUF, SKIP>: the LDS-DMA kernels (aligned shapes), tile_dma semantics of gemm.hip
// SHIM_FAULT="kind:n" (tests/test_cpu_compute.py): the n-th launch of the LDS-DMA GEMM family runs with ONE thing wrong, the way a host
// bug would hand it over - 1: the tile grid is one tile column short, 2: B's leading dimension is two
// elements too long, 3: a dense operand carries the "upper triangular" hint (its K range is cut at the diagonal), 4: K is one K tile short - so that the harness can be shown to notice
static long fault_seen = 0;
void k_dgemm_dma(const std::string& name, void** a, unsigned gx, unsigned gy) {
  GemmArgs g = arg<GemmArgs>(a, 0);
  const bool amc = tmpl(name, 1), skip = tmpl(name, 4);
  static const char* fault = getenv("SHIM_FAULT");
  if (fault) {
    const int kind = atoi(fault); const char* c = strchr(fault, ':'); const long nth = c ? atol(c + 1) : 0;
    const bool applies = (kind == 1 && g.tn > 1) || kind == 2 || (kind == 3 && !g.aupt && !g.aupn && !g.bupper && !amc && g.K > 128) || (kind == 4 && g.K > 16);
    if (applies && fault_seen++ == nth) {          // the n-th launch the fault makes a difference to
      if (kind == 1) g.tn -= 1;
      if (kind == 2) g.ldb += 2;
      if (kind == 3) g.aupt = 1;
      if (kind == 4) g.K -= 16;
    }
  }
#pragma omp parallel for collapse(2) schedule(dynamic)
  for (unsigned kz = 0; kz < gy; kz++)
    for (unsigned bx = 0; bx < gx; bx++) {
      const int b = (int)((bx + kz) % gx), L = (b & 7) * g.chunk + (b >> 3);
      int ti, tj;
      if ((b >> 3) >= g.chunk || !slot_to_tile(g, L, ti, tj)) continue;
      const int64_t i0 = (int64_t)ti * TB, j0 = (int64_t)tj * TB;
      int64_t kbeg = (int64_t)kz * g.kchunk, kend = std::min<int64_t>(kbeg + g.kchunk, g.K);
      if (g.bupper && kend > j0 + TB) kend = j0 + TB;
      if (g.aupt && kend > i0 + TB) kend = i0 + TB;
      if (g.aupn && kbeg < i0) kbeg = i0;
      const int nk = kend > kbeg ? (int)((kend - kbeg) / TK) : 0;
      kend = kbeg + (int64_t)nk * TK;
      Acc c(TB, TB);
      const double* A = amc ? g.A + i0 : a_tile_base(g, ti);
      const int64_t as_o = amc ? 1 : g.lda, as_k = amc ? g.lda : 1;
      const double* B = g.B + j0 * g.ldb;
      if (!skip) mma(c, 0, TB, 0, TB, A, as_o, as_k, B, g.ldb, 1, kbeg, kend);
      else {
        const bool dtri = g.tri == 1 && !g.stair && ti == tj;
        for (int64_t kb = kbeg; kb < kend; kb += 8)
          for (int bi = 0; bi < 8; bi++)
            for (int bj = 0; bj < 8; bj++) {
              const int wi = (bi / 4) * 64, i = bi % 4, wj = (bj / 4) * 64, j = bj % 4;
              int jlo = 0, ilo = 0, ihi = 3;
              if (g.bupper) { const int64_t d = kb - j0 - wj; jlo = d > 0 ? (int)(d >> 4) : 0; }
              if (g.aupt) { const int64_t d = kb - i0 - wi; ilo = d > 0 ? (int)(d >> 4) : 0; }
              if (g.aupn) { const int64_t d = kb + 7 - i0 - wi; ihi = d < 0 ? -1 : (d >> 4) > 3 ? 3 : (int)(d >> 4); }
              const bool on = j >= jlo && i >= ilo && i <= ihi && !(dtri && wi + 16 * i > wj + 16 * j);
              if (on) mma(c, 16 * bi, 16 * bi + 16, 16 * bj, 16 * bj + 16, A, as_o, as_k, B, g.ldb, 1, kb, kb + 8);
            }
      }
      const bool diag_tile = g.stair ? (stair_gti(g, ti) == stair_gtj(g, tj)) : ((g.tri != 0) && (ti == tj));
      for (int j = 0; j < TB; j++)
        for (int i = 0; i < TB; i++) {
          if (diag_tile && (g.tri == 1 ? i > j : i < j)) continue;
          const int64_t row = i0 + i, col = j0 + j;
          const double v = g.alpha * c.at(i, j);
          if (g.ksplit > 1) g.P[(int64_t)kz * g.slab + row + col * g.M] = v;
          else if (g.atomic_c) g.C[row + col * g.ldc] += v;
          else if (g.beta != 0.0) g.C[row + col * g.ldc] = v + g.beta * (amc ? g.C[row + col * g.ldc] : g.Cin[row + col * g.ldcin]);
          else g.C[row + col * g.ldc] = v;
        }
    }
}

void k_dgemm_small(const std::string& name, void** a, unsigned gx, unsigned gy, unsigned gz) {
  const SmallArgs g0 = arg<SmallArgs>(a, 0);
  const bool ta = tmpl(name, 0), tb = tmpl(name, 1);
  for (unsigned z = 0; z < gz; z++) {
    const double* A = g0.A + (int64_t)z * g0.sa; const double* B = g0.B + (int64_t)z * g0.sb; double* C = g0.C + (int64_t)z * g0.sc;
    for (unsigned tj = 0; tj < gy; tj++)
      for (unsigned ti = 0; ti < gx; ti++) {
        if (g0.tri == 1 && ti > tj) continue;
        if (g0.tri == 2 && ti < tj) continue;
        const int i0 = ti * 64, j0 = tj * 64, mi = std::min(64, g0.M - i0), nj = std::min(64, g0.N - j0);
        if (mi <= 0 || nj <= 0) continue;
        Acc c(64, 64);
        mma(c, 0, mi, 0, nj, ta ? A + (int64_t)i0 * g0.lda : A + i0, ta ? g0.lda : 1, ta ? 1 : g0.lda,
            tb ? B + j0 : B + (int64_t)j0 * g0.ldb, tb ? 1 : g0.ldb, tb ? g0.ldb : 1, 0, g0.K);
        for (int j = 0; j < nj; j++)
          for (int i = 0; i < mi; i++) {
            const int row = i0 + i, col = j0 + j;
            if (g0.tri == 1 && row > col) continue;
            if (g0.tri == 2 && row < col) continue;
            double* pc = C + row + (int64_t)col * g0.ldc;
            const double v = g0.alpha * c.at(i, j);
            *pc = g0.beta != 0.0 ? v + g0.beta * (*pc) : v;
          }
      }
  }
}

void k_skinny(void** a, bool tn, unsigned gx) {
  const SkinnyArgs g = arg<SkinnyArgs>(a, 0);
  const int64_t rows = std::min<int64_t>(g.M, (int64_t)gx * 64);
  auto A = [&](int64_t idx) -> double { return g.A32 ? (double)g.A32[idx] : g.A[idx]; };      // op(A) may live in fp32 (widened, same sums)
  if (tn) {
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; r++)
      for (int c = 0; c < g.N; c++) {
        const double* y = g.B + (int64_t)c * g.ldb;
        double s = 0.0;
        for (int64_t k = 0; k < g.K; k++) s += A(r * g.lda + k) * y[k];
        double* pc = g.C + r + (int64_t)c * g.ldc;
        *pc = g.beta != 0.0 ? g.alpha * s + g.beta * (*pc) : g.alpha * s;
      }
    return;
  }
  const int64_t RB = 512;                                   // row blocks: the operand is streamed once, M-contiguous
#pragma omp parallel for schedule(static)
  for (int64_t r0 = 0; r0 < rows; r0 += RB) {
    const int64_t nr = std::min(RB, rows - r0);
    std::vector<double> acc((size_t)nr * g.N, 0.0);
    for (int64_t k = 0; k < g.K; k++) {
      for (int c = 0; c < g.N; c++) { const double b = g.B[k + (int64_t)c * g.ldb]; double* ac = acc.data() + (size_t)c * nr; for (int64_t r = 0; r < nr; r++) ac[r] += A(r0 + r + k * g.lda) * b; }
    }
    for (int c = 0; c < g.N; c++)
      for (int64_t r = 0; r < nr; r++) { double* pc = g.C + r0 + r + (int64_t)c * g.ldc; const double v = g.alpha * acc[(size_t)c * nr + r]; *pc = g.beta != 0.0 ? v + g.beta * (*pc) : v; }
  }
}

void k_scale(void** a, unsigned gy) {
  double* C = arg<double*>(a, 0); const int64_t ldc = arg<int64_t>(a, 1), m = arg<int64_t>(a, 2), n = arg<int64_t>(a, 3);
  const double beta = arg<double>(a, 4); const int tri = arg<int>(a, 5);
  for (int64_t col = 0; col < std::min<int64_t>(n, gy); col++)
    for (int64_t row = 0; row < m; row++) {
      if (tri == 1 && row > col) continue;
      if (tri == 2 && row < col) continue;
      double* p = C + row + col * ldc; *p = beta == 0.0 ? 0.0 : beta * (*p);
    }
}

void k_splitk_reduce(void** a, unsigned gy) {
  double* C = arg<double*>(a, 0); const int64_t ldc = arg<int64_t>(a, 1); const double* P = arg<const double*>(a, 2);
  const int64_t slab = arg<int64_t>(a, 3); const int ks = arg<int>(a, 4); const int64_t m = arg<int64_t>(a, 5), n = arg<int64_t>(a, 6);
  const double beta = arg<double>(a, 7); const int tri = arg<int>(a, 8);
  for (int64_t col = 0; col < std::min<int64_t>(n, gy); col++)
    for (int64_t row = 0; row < m; row++) {
      if (tri == 1 && row > col) continue;
      if (tri == 2 && row < col) continue;
      double s = 0.0;
      for (int z = 0; z < ks; z++) s += P[(int64_t)z * slab + row + col * m];
      double* pc = C + row + col * ldc; *pc = beta == 0.0 ? s : s + beta * (*pc);
    }
}

// ------------------------------------------------------------------------------------------ aux.hip
constexpr uint64_t MASK48 = (1ull << 48) - 1, LCG_A = 0x5DEECE66Dull, LCG_C = 0xBull;
double drand48_of_seed(uint64_t seed) {
  const uint64_t x0 = ((seed & 0xFFFFFFFFull) << 16) | 0x330Eull, x1 = (LCG_A * x0 + LCG_C) & MASK48;
  return (double)x1 * (1.0 / 281474976710656.0);
}
uint64_t lcg_jump(uint64_t x0, uint64_t k) {
  uint64_t aa = LCG_A, c = LCG_C, ra = 1, rc = 0;
  while (k) { if (k & 1) { ra = (ra * aa) & MASK48; rc = (rc * aa + c) & MASK48; } c = (c * aa + c) & MASK48; aa = (aa * aa) & MASK48; k >>= 1; }
  return (ra * x0 + rc) & MASK48;
}
int64_t paddr(int packed, int64_t ld, int64_t r, int64_t c) { return packed ? (c * (c + 1) / 2 + r) : (r + c * ld); }

void k_fill_symmetric(void** a, unsigned gx, unsigned gy, unsigned bx) {
  double* out = arg<double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1), nl = arg<int64_t>(a, 2), n = arg<int64_t>(a, 3), x = arg<int64_t>(a, 4),
          y = arg<int64_t>(a, 5), d = arg<int64_t>(a, 6); const int dom = arg<int>(a, 7);
  for (int64_t jl = 0; jl < gy; jl++)
    for (int64_t il = 0; il < std::min<int64_t>(nl, (int64_t)gx * bx); il++) {
      const int64_t gyy = y + il * d, gxx = x + jl * d;
      double v = 0.0;
      if (gyy < n && gxx < n) { const int64_t hi = std::max(gxx, gyy), lo = std::min(gxx, gyy); v = drand48_of_seed((uint64_t)(hi + n * lo)); if (dom && gxx == gyy) v += (double)n; }
      out[il + jl * ld] = v;
    }
}
void k_fill_random(void** a) {
  double* out = arg<double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1), ml = arg<int64_t>(a, 2), nl = arg<int64_t>(a, 3), pad_y = arg<int64_t>(a, 4),
          pad_x = arg<int64_t>(a, 5); const uint64_t x0 = arg<uint64_t>(a, 6);
  for (int64_t jl = 0; jl < nl; jl++)
    for (int64_t il = 0; il < ml; il++)
      out[il + jl * ld] = (il < pad_y && jl < pad_x) ? (double)lcg_jump(x0, (uint64_t)(jl * pad_y + il) + 1) * (1.0 / 281474976710656.0) : 0.0;
}
void k_copy_window(void** a) {
  const double* src = arg<const double*>(a, 0); const int sp = arg<int>(a, 1); const int64_t sld = arg<int64_t>(a, 2), sr0 = arg<int64_t>(a, 3), sc0 = arg<int64_t>(a, 4);
  double* dst = arg<double*>(a, 5); const int dp = arg<int>(a, 6); const int64_t dld = arg<int64_t>(a, 7), dr0 = arg<int64_t>(a, 8), dc0 = arg<int64_t>(a, 9),
          rows = arg<int64_t>(a, 10), cols = arg<int64_t>(a, 11); const int tri = arg<int>(a, 12), zl = arg<int>(a, 13);
  for (int64_t c = 0; c < cols; c++)
    for (int64_t r = 0; r < rows; r++) {
      if (tri && r > c) { if (zl && !dp) dst[paddr(0, dld, dr0 + r, dc0 + c)] = 0.0; continue; }
      dst[paddr(dp, dld, dr0 + r, dc0 + c)] = src[paddr(sp, sld, sr0 + r, sc0 + c)];
    }
}
void k_copy_rect_v2(void** a) {
  const double* src = arg<const double*>(a, 0); const int64_t sld = arg<int64_t>(a, 1); double* dst = arg<double*>(a, 2);
  const int64_t dld = arg<int64_t>(a, 3), rows = arg<int64_t>(a, 4), cols = arg<int64_t>(a, 5);
  for (int64_t c = 0; c < cols; c++) memmove(dst + c * dld, src + c * sld, (size_t)(rows >> 1) * 16);
}
void k_zero_rect(void** a) {
  double* dst = arg<double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1), rows = arg<int64_t>(a, 2), cols = arg<int64_t>(a, 3);
  for (int64_t c = 0; c < cols; c++) memset(dst + c * ld, 0, (size_t)rows * 8);
}
void k_remove_triangle(void** a) {
  double* p = arg<double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1), rows = arg<int64_t>(a, 2), cols = arg<int64_t>(a, 3), x = arg<int64_t>(a, 4), y = arg<int64_t>(a, 5),
          d = arg<int64_t>(a, 6); const int upper = arg<int>(a, 7);
  for (int64_t c = 0; c < cols; c++)
    for (int64_t r = 0; r < rows; r++) { const int64_t gy = y + r * d, gx = x + c * d; if (upper ? (gy > gx) : (gy < gx)) p[r + c * ld] = 0.0; }
}
void k_sumsq(void** a) {
  const double* X = arg<const double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1), m = arg<int64_t>(a, 2), n = arg<int64_t>(a, 3);
  const int sub_id = arg<int>(a, 4), upper = arg<int>(a, 5); double* out = arg<double*>(a, 6);
  double s = 0.0;
  for (int64_t c = 0; c < n; c++)
    for (int64_t r = 0; r < m; r++) { if (upper && r > c) break; double v = X[r + c * ld]; if (sub_id && r == c) v -= 1.0; s += v * v; }
  *out += s;
}
void k_cyclic_piece(void** a) {
  double* piece = arg<double*>(a, 0); const int64_t ldp = arg<int64_t>(a, 1), rl = arg<int64_t>(a, 2), cl = arg<int64_t>(a, 3); double* dense = arg<double*>(a, 4);
  const int64_t ldd = arg<int64_t>(a, 5), m = arg<int64_t>(a, 6), n = arg<int64_t>(a, 7), x = arg<int64_t>(a, 8), y = arg<int64_t>(a, 9), dx = arg<int64_t>(a, 10),
          dy = arg<int64_t>(a, 11); const int to_dense = arg<int>(a, 12);
  for (int64_t c = 0; c < cl; c++)
    for (int64_t r = 0; r < rl; r++) {
      const int64_t gy = y + r * dy, gx = x + c * dx; const bool in = gy < m && gx < n;
      if (to_dense) { if (in) dense[gy + gx * ldd] = piece[r + c * ldp]; }
      else piece[r + c * ldp] = in ? dense[gy + gx * ldd] : 0.0;
    }
}

// ------------------------------------------------------------------------------------------ leaf.hip
// upper Cholesky of the n x n window S (column-major, ld), in place on the upper triangle; returns the 1-based first bad pivot or 0
int potrf_upper(double* S, int64_t ld, int n) {
  int bad = 0;
  for (int k = 0; k < n; k++) {
    double d = S[k + k * ld];
    for (int p = 0; p < k; p++) d -= S[p + k * ld] * S[p + k * ld];
    if (!(d > 0.0) && !bad) bad = k + 1;
    const double r = std::sqrt(d);
    S[k + k * ld] = r;
    for (int j = k + 1; j < n; j++) {
      double s = S[k + j * ld];
      for (int p = 0; p < k; p++) s -= S[p + k * ld] * S[p + j * ld];
      S[k + j * ld] = s / r;
    }
  }
  return bad;
}
// T (upper, ldt) = inverse of the upper triangle of R (ldr); nothing below the diagonal is read or written
void trtri_upper(const double* R, int64_t ldr, double* T, int64_t ldt, int n) {
  for (int c = 0; c < n; c++) {
    for (int i = c; i >= 0; i--) {
      double s = (i == c) ? 1.0 : 0.0;
      for (int p = i + 1; p <= c; p++) s -= R[i + p * ldr] * T[p + c * ldt];
      T[i + c * ldt] = s / R[i + i * ldr];
    }
  }
}
void cjob(const double* src, double* dst, int64_t ld, int cols) {
  for (int e = 0; e < 64 * cols; e++) dst[(e & 63) + (int64_t)(e >> 6) * ld] = src[e];
}
void k_leaf_cholinv(void** a) {
  double* A = arg<double*>(a, 0); const int64_t lda = arg<int64_t>(a, 1); double* Rinv = arg<double*>(a, 2); const int64_t ldr = arg<int64_t>(a, 3);
  const int n = arg<int>(a, 4), zero_lower = arg<int>(a, 5); int* info = arg<int*>(a, 6); const int info_base = arg<int>(a, 7);
  const double* cs = arg<const double*>(a, 8); double* cd = arg<double*>(a, 9); const int64_t cld = arg<int64_t>(a, 10); const int cc = arg<int>(a, 11);
  if (cc > 0) cjob(cs, cd, cld, cc);
  std::vector<double> S((size_t)n * n, 0.0), T((size_t)n * n, 0.0);
  for (int j = 0; j < n; j++) for (int i = 0; i <= j; i++) S[i + (size_t)j * n] = A[i + j * lda];
  const int bad = potrf_upper(S.data(), n, n);
  for (int j = 0; j < n; j++) for (int i = 0; i <= j; i++) A[i + j * lda] = S[i + (size_t)j * n];
  if (Rinv) {
    trtri_upper(S.data(), n, T.data(), n, n);
    for (int j = 0; j < n; j++) for (int i = 0; i < n; i++) { if (i <= j) Rinv[i + j * ldr] = T[i + (size_t)j * n]; else if (zero_lower) Rinv[i + j * ldr] = 0.0; }
  }
  if (bad && info && *info == 0) *info = info_base + bad;
}
void k_leaf_trtri(void** a) {
  const double* R = arg<const double*>(a, 0); const int64_t ldr = arg<int64_t>(a, 1); double* Ri = arg<double*>(a, 2); const int64_t ldi = arg<int64_t>(a, 3); const int n = arg<int>(a, 4);
  std::vector<double> S((size_t)n * n, 0.0), T((size_t)n * n, 0.0);
  for (int j = 0; j < n; j++) for (int i = 0; i <= j; i++) S[i + (size_t)j * n] = R[i + j * ldr];
  trtri_upper(S.data(), n, T.data(), n, n);
  for (int j = 0; j < n; j++) for (int i = 0; i <= j; i++) Ri[i + j * ldi] = T[i + (size_t)j * n];
}
// X (64 x 64) = Dinv^T B with Dinv upper (only its upper triangle is read)
void solve64(const double* Dinv, int64_t ldi, const double* B, int64_t ldb, double* X) {
  for (int c = 0; c < 64; c++)
    for (int p = 0; p < 64; p++) { double s = 0.0; for (int k = 0; k <= p; k++) s += Dinv[k + p * ldi] * B[k + c * ldb]; X[p + c * 64] = s; }
}
// the leaf of one 64 x 64 block: C upper <- chol, Dn (full 64 x 64, ldn) <- its inverse with a zero lower part; returns the bad pivot
int leaf64(double* C, int64_t ldc, double* Dn, int64_t ldn) {
  std::vector<double> S(64 * 64, 0.0), T(64 * 64, 0.0);
  for (int j = 0; j < 64; j++) for (int i = 0; i <= j; i++) S[i + j * 64] = C[i + j * ldc];
  const int bad = potrf_upper(S.data(), 64, 64);
  for (int j = 0; j < 64; j++) for (int i = 0; i <= j; i++) C[i + j * ldc] = S[i + j * 64];
  trtri_upper(S.data(), 64, T.data(), 64, 64);
  for (int j = 0; j < 64; j++) for (int i = 0; i < 64; i++) Dn[i + j * ldn] = i <= j ? T[i + j * 64] : 0.0;
  return bad;
}
void k_panel64(void** a, unsigned gx) {
  double* R = arg<double*>(a, 0); const int64_t ldr = arg<int64_t>(a, 1); const double* Dinv = arg<const double*>(a, 2); const int64_t ldi = arg<int64_t>(a, 3);
  const int i = arg<int>(a, 4), nblk = arg<int>(a, 5); double* Xs = arg<double*>(a, 6); const Panel64Fold f = arg<Panel64Fold>(a, 7);
  const int r = nblk - 1 - i;
  (void)gx;
  // every workgroup reads the UNSOLVED block row; the solved pieces go to scratch (or, single workgroup, in place) at the end
  std::vector<double> X((size_t)r * 64 * 64);
  for (int q = 0; q < r; q++) solve64(Dinv, ldi, R + (int64_t)i * 64 + (int64_t)(i + 1 + q) * 64 * ldr, ldr, X.data() + (size_t)q * 4096);
  if (f.cj_cols > 0) cjob(f.cj_src, f.cj_dst, f.cj_ld, f.cj_cols);
  for (int qb = 0; qb < r; qb++)
    for (int qa = 0; qa <= qb; qa++) {
      const double* Xa = X.data() + (size_t)qa * 4096; const double* Xb = X.data() + (size_t)qb * 4096;
      double* C = R + (int64_t)(i + 1 + qa) * 64 + (int64_t)(i + 1 + qb) * 64 * ldr;
      for (int col = 0; col < 64; col++)
        for (int row = 0; row < 64; row++) {
          if (qa == qb && row > col) continue;
          double s = 0.0;
          for (int k = 0; k < 64; k++) s += Xa[k + row * 64] * Xb[k + col * 64];
          C[row + (int64_t)col * ldr] -= s;
        }
    }
  for (int q = 0; q < r; q++) {
    if (f.direct) { double* dst = R + (int64_t)i * 64 + (int64_t)(i + 1 + q) * 64 * ldr; for (int e = 0; e < 4096; e++) dst[(e & 63) + (int64_t)(e >> 6) * ldr] = X[(size_t)q * 4096 + e]; }
    else memcpy(Xs + (int64_t)q * 4096, X.data() + (size_t)q * 4096, 4096 * 8);
  }
  if (f.Dnext) {
    const int bad = leaf64(R + (int64_t)(i + 1) * 64 * (ldr + 1), ldr, f.Dnext, f.ldn);
    if (bad && f.info && *f.info == 0) *f.info = f.info_base + bad;
  }
}
// one level of the inverse assembly on the pair at offset o: Ri12 = -Ri11 (R12 Ri22), with the kernels' 16-block K trimming
void merge_pair(const double* R, int64_t ldr, double* Ri, int64_t ldi, int64_t o, int H) {
  const double* R12 = R + o + (o + H) * ldr; const double* Ri11 = Ri + o + o * ldi;
  std::vector<double> W((size_t)H * H), Out((size_t)H * H);
  for (int c = 0; c < H; c++) {
    const int kmax = 16 * (c / 16 + 1);
    const double* Ri22c = Ri + (o + H) + (o + H + c) * ldi;
    for (int r = 0; r < H; r++) { double s = 0.0; for (int k = 0; k < kmax; k++) s += R12[r + (int64_t)k * ldr] * Ri22c[k]; W[r + (size_t)c * H] = s; }
  }
  for (int c = 0; c < H; c++)
    for (int r = 0; r < H; r++) {
      const int kb = 16 * ((r / 16) % 4);
      double s = 0.0;
      for (int k = kb; k < H; k++) s += Ri11[r + (int64_t)k * ldi] * W[k + (size_t)c * H];
      Out[r + (size_t)c * H] = -s;
    }
  for (int c = 0; c < H; c++) for (int r = 0; r < H; r++) Ri[o + r + (o + H + c) * ldi] = Out[r + (size_t)c * H];
}
void k_trinv_merge(const std::string& name, void** a, unsigned gy) {
  const double* R = arg<const double*>(a, 0); const int64_t ldr = arg<int64_t>(a, 1); double* Ri = arg<double*>(a, 2); const int64_t ldi = arg<int64_t>(a, 3);
  const int H = 64 * (int)tmpl(name, 0);
  for (unsigned z = 0; z < gy; z++) merge_pair(R, ldr, Ri, ldi, (int64_t)z * 2 * H, H);
}
void k_chain64(void** a) {
  const Chain64 g = arg<Chain64>(a, 0);
  if (g.recover) return;                                  // (no workgroup of the primary launch gives up here: nothing to recover)
  const int nblk = g.nblk; const int64_t n = (int64_t)nblk * 64;
  int bad = leaf64(g.R, g.ldr, g.Ri, g.ldi);
  if (bad && g.info && *g.info == 0) *g.info = g.info_base + bad;
  std::vector<double> X(4096);
  for (int i = 0; i + 1 < nblk; i++) {
    const double* Dinv = g.Ri + (int64_t)i * 64 * (g.ldi + 1);
    for (int b = i + 1; b < nblk; b++) {                  // phase S: the block row in place
      double* Aib = g.R + (int64_t)i * 64 + (int64_t)b * 64 * g.ldr;
      solve64(Dinv, g.ldi, Aib, g.ldr, X.data());
      for (int e = 0; e < 4096; e++) Aib[(e & 63) + (int64_t)(e >> 6) * g.ldr] = X[e];
    }
    for (int b = i + 1; b < nblk; b++)                    // phase U
      for (int aa = i + 1; aa <= b; aa++) {
        const double* Xa = g.R + (int64_t)i * 64 + (int64_t)aa * 64 * g.ldr; const double* Xb = g.R + (int64_t)i * 64 + (int64_t)b * 64 * g.ldr;
        double* C = g.R + (int64_t)aa * 64 + (int64_t)b * 64 * g.ldr;
        for (int col = 0; col < 64; col++)
          for (int row = 0; row < 64; row++) {
            if (aa == b && row > col) continue;
            double s = 0.0;
            for (int k = 0; k < 64; k++) s += Xa[k + (int64_t)row * g.ldr] * Xb[k + (int64_t)col * g.ldr];
            C[row + (int64_t)col * g.ldr] -= s;
          }
      }
    bad = leaf64(g.R + (int64_t)(i + 1) * 64 * (g.ldr + 1), g.ldr, g.Ri + (int64_t)(i + 1) * 64 * (g.ldi + 1), g.ldi);
    if (bad && g.info && *g.info == 0) *g.info = g.info_base + (i + 1) * 64 + bad;
  }
  for (int h = 64; h <= g.hmax && 2 * h <= n; h *= 2)
    for (int64_t o = 0; o + 2 * h <= n; o += 2 * h) merge_pair(g.R, g.ldr, g.Ri, g.ldi, o, h);
}


// ------------------------------------------------------------------------------------------ dist.hip / dist2d.hip / redist.hip / summa.hip / cacqr.hip
double sym_entry(int64_t n, int64_t grow, int64_t gcol, int dom) {
  const int64_t hi = std::max(gcol, grow), lo = std::min(gcol, grow);
  double v = drand48_of_seed((uint64_t)(hi + n * lo));
  if (dom && gcol == grow) v += (double)n;
  return v;
}
void k_fill_symmetric_bc(void** a) {
  double* out = arg<double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1), n = arg<int64_t>(a, 2), nb = arg<int64_t>(a, 3); const int P = arg<int>(a, 4), p = arg<int>(a, 5),
          dom = arg<int>(a, 6); const int64_t lc = arg<int64_t>(a, 7);
  for (int64_t l = 0; l < lc; l++) { const int64_t g = ((l / nb) * P + p) * nb + l % nb; for (int64_t r = 0; r < n; r++) out[r + l * ld] = sym_entry(n, r, g, dom); }
}
void k_pad_identity(void** a) {
  double* R = arg<double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1), n = arg<int64_t>(a, 2), nb = arg<int64_t>(a, 4); const int P = arg<int>(a, 5), p = arg<int>(a, 6);
  const int64_t row0 = arg<int64_t>(a, 7), col0 = arg<int64_t>(a, 8), rows = arg<int64_t>(a, 9), cols = arg<int64_t>(a, 10);
  for (int64_t l = col0; l < col0 + cols; l++) {
    const int64_t g = ((l / nb) * P + p) * nb + l % nb;
    for (int64_t r = row0; r < row0 + rows; r++) if (r >= n || g >= n) R[r + l * ld] = (r == g) ? 1.0 : 0.0;
  }
}
void k_export_upper_bc(void** a) {
  const double* R = arg<const double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1); double* out = arg<double*>(a, 2); const int64_t ldo = arg<int64_t>(a, 3), n = arg<int64_t>(a, 4),
          nb = arg<int64_t>(a, 5); const int P = arg<int>(a, 6), p = arg<int>(a, 7); const int64_t lc = arg<int64_t>(a, 8);
  for (int64_t l = 0; l < lc; l++) { const int64_t g = ((l / nb) * P + p) * nb + l % nb; for (int64_t r = 0; r < n; r++) out[r + l * ldo] = r <= g ? R[r + l * ld] : 0.0; }
}
void k_info_to_double(void** a) { const int* info = arg<const int*>(a, 0); double* out = arg<double*>(a, 1); *out = (double)*info; }
void k_identity_blocks_bc(void** a) {
  double* X = arg<double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1), nb = arg<int64_t>(a, 2); const int P = arg<int>(a, 3), p = arg<int>(a, 4); const int64_t lc = arg<int64_t>(a, 5);
  for (int64_t l = 0; l < lc; l++) X[((l / nb) * P + p) * nb + l % nb + l * ld] = 1.0;
}
struct G2 { int64_t nb; int Pr, Pc, pr, pc; int64_t grow(int64_t l) const { return ((l / nb) * Pr + pr) * nb + l % nb; } int64_t gcol(int64_t l) const { return ((l / nb) * Pc + pc) * nb + l % nb; } };
void k_fill_symmetric_bc2d(void** a) {
  double* out = arg<double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1), n = arg<int64_t>(a, 2); const G2 g{arg<int64_t>(a, 3), arg<int>(a, 4), arg<int>(a, 5), arg<int>(a, 6), arg<int>(a, 7)};
  const int dom = arg<int>(a, 8); const int64_t lr = arg<int64_t>(a, 9), lc = arg<int64_t>(a, 10);
  for (int64_t c = 0; c < lc; c++) for (int64_t r = 0; r < lr; r++) out[r + c * ld] = sym_entry(n, g.grow(r), g.gcol(c), dom);
}
void k_import_pad_2d(void** a) {
  const double* A = arg<const double*>(a, 0); const int64_t lda = arg<int64_t>(a, 1); double* R = arg<double*>(a, 2); const int64_t ld = arg<int64_t>(a, 3), n = arg<int64_t>(a, 4);
  const G2 g{arg<int64_t>(a, 5), arg<int>(a, 6), arg<int>(a, 7), arg<int>(a, 8), arg<int>(a, 9)}; const int64_t lr = arg<int64_t>(a, 10), lc = arg<int64_t>(a, 11);
  for (int64_t c = 0; c < lc; c++) for (int64_t r = 0; r < lr; r++) { const int64_t gr = g.grow(r), gc = g.gcol(c); R[r + c * ld] = (gr < n && gc < n) ? A[r + c * lda] : (gr == gc ? 1.0 : 0.0); }
}
void k_export_upper_2d(void** a) {
  const double* R = arg<const double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1); double* out = arg<double*>(a, 2); const int64_t ldo = arg<int64_t>(a, 3);
  const G2 g{arg<int64_t>(a, 4), arg<int>(a, 5), arg<int>(a, 6), arg<int>(a, 7), arg<int>(a, 8)}; const int64_t lr = arg<int64_t>(a, 9), lc = arg<int64_t>(a, 10);
  for (int64_t c = 0; c < lc; c++) for (int64_t r = 0; r < lr; r++) out[r + c * ldo] = g.grow(r) <= g.gcol(c) ? R[r + c * ld] : 0.0;
}
void k_identity_2d(void** a) {
  double* X = arg<double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1); const G2 g{arg<int64_t>(a, 2), arg<int>(a, 3), arg<int>(a, 4), arg<int>(a, 5), arg<int>(a, 6)}; const int64_t lc = arg<int64_t>(a, 7);
  for (int64_t l = 0; l < lc; l++) { const int64_t J = (l / g.nb) * g.Pc + g.pc; if ((int)(J % g.Pr) != g.pr) continue; X[(J / g.Pr) * g.nb + l % g.nb + l * ld] = 1.0; }
}
void k_zero_root_2d(void** a) {
  double* X = arg<double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1); const G2 g{arg<int64_t>(a, 2), arg<int>(a, 3), arg<int>(a, 4), arg<int>(a, 5), arg<int>(a, 6)};
  const int64_t lr = arg<int64_t>(a, 7), lc = arg<int64_t>(a, 8), n1 = arg<int64_t>(a, 9);
  for (int64_t c = 0; c < lc; c++) for (int64_t r = 0; r < lr; r++) if (g.grow(r) < n1 && g.gcol(c) >= n1) X[r + c * ld] = 0.0;
}
void k_redist(void** a, bool gather) {
  double* mat = arg<double*>(a, 0); const int64_t ld = arg<int64_t>(a, 1); const int64_t* ri = arg<const int64_t*>(a, 2); const int64_t nr = arg<int64_t>(a, 3);
  const int64_t* ci = arg<const int64_t*>(a, 4); const int64_t nc = arg<int64_t>(a, 5); double* buf = arg<double*>(a, 6);
  for (int64_t j = 0; j < nc; j++) for (int64_t i = 0; i < nr; i++) { if (gather) buf[i + j * nr] = mat[ri[i] + ci[j] * ld]; else mat[ri[i] + ci[j] * ld] = buf[i + j * nr]; }
}
void k_combine(void** a, unsigned gy) {
  double* C = arg<double*>(a, 0); const int64_t ldc = arg<int64_t>(a, 1); const double* acc = arg<const double*>(a, 2); const int64_t lda = arg<int64_t>(a, 3), m = arg<int64_t>(a, 4);
  const double beta = arg<double>(a, 6);
  for (int64_t c = 0; c < gy; c++) for (int64_t r = 0; r < m; r++) { double* p = C + r + c * ldc; const double v = acc[r + c * lda]; *p = beta == 0.0 ? v : beta * (*p) + v; }
}
void k_combine_packed(void** a, unsigned gy) {
  double* Cp = arg<double*>(a, 0); const double* acc = arg<const double*>(a, 1); const int64_t n = arg<int64_t>(a, 2); const double beta = arg<double>(a, 3);
  for (int64_t c = 0; c < gy; c++) for (int64_t r = 0; r <= c; r++) { double* q = Cp + c * (c + 1) / 2 + r; const double v = acc[r + c * n]; *q = beta == 0.0 ? v : beta * (*q) + v; }
}
void k_blocks_to_dense(void** a) {
  const double* blocks = arg<const double*>(a, 0); double* dense = arg<double*>(a, 1); const int64_t n = arg<int64_t>(a, 2), nl = arg<int64_t>(a, 3); const int c = arg<int>(a, 4);
  for (int64_t j = 0; j < n; j++) for (int64_t i = 0; i < n; i++) { const int64_t zi = i % c, aa = i / c, xj = j % c, b = j / c; dense[i + j * n] = blocks[(zi * c + xj) * nl * nl + aa + b * nl]; }
}
// ------------------------------------------------------------------------------------------ cqr_kernels.hip (n = 256)
void k_gram256(void** a, unsigned gx) {
  const GramArgs g = arg<GramArgs>(a, 0);
  for (unsigned b = 0; b < gx; b++) {
    const int64_t r0 = (int64_t)b * g.chunk, rows = std::max<int64_t>(0, std::min<int64_t>(g.chunk, g.m - r0)), nk = (rows / 16) * 16;
    double* P = g.P + (int64_t)b * 256 * 256;
    for (int c = 0; c < 256; c++)
      for (int r = 0; r <= c; r++) {
        const double* qr = g.Q + r0 + (int64_t)r * g.ld; const double* qc = g.Q + r0 + (int64_t)c * g.ld;
        double s = 0.0;
        for (int64_t k = 0; k < nk; k++) s += qr[k] * qc[k];
        P[r + (int64_t)c * 256] = s;
      }
  }
}
void k_gram256_reduce(void** a) {
  const double* P = arg<const double*>(a, 0); const int nslab = arg<int>(a, 1); double* G = arg<double*>(a, 2); const int64_t ldg = arg<int64_t>(a, 3);
  for (int c = 0; c < 256; c++)
    for (int r = 0; r < 256; r++) {
      double s = 0.0;
      if (r <= c) for (int z = 0; z < nslab; z++) s += P[(int64_t)z * 65536 + r + (int64_t)c * 256];
      G[r + c * ldg] = s;
    }
}
void k_qrapply256(void** a) {
  const ApplyArgs g = arg<ApplyArgs>(a, 0);
  const int64_t m = (int64_t)g.ntiles * 128;
  std::vector<double> col((size_t)m);
  std::vector<double> out((size_t)m * 256);
  for (int c = 0; c < 256; c++) {
    const int kmax = 16 * (c / 16 + 1);                   // Ri upper triangular: K stops at the diagonal 16-block
    std::fill(col.begin(), col.end(), 0.0);
    for (int k = 0; k < kmax; k++) { const double r = g.Ri[k + (int64_t)c * 256]; const double* q = g.Qin + (int64_t)k * g.ldin; for (int64_t i = 0; i < m; i++) col[i] += q[i] * r; }
    memcpy(out.data() + (size_t)c * m, col.data(), (size_t)m * 8);
  }
  for (int c = 0; c < 256; c++) memcpy(g.Qout + (int64_t)c * g.ldout, out.data() + (size_t)c * m, (size_t)m * 8);
}


// ------------------------------------------------------------------------------------------ mixed precision (mixed.hip, mixed_kernels.h, dist_mixed.hip)
typedef unsigned short bf16_t;
float bf2f(bf16_t h) { uint32_t u = (uint32_t)h << 16; float f; memcpy(&f, &u, 4); return f; }
bf16_t f2bf(float f) {                                   // round to nearest even, as the device cast does
  uint32_t u; memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);     // NaN stays NaN
  const uint32_t r = 0x7fffu + ((u >> 16) & 1u);
  return (bf16_t)((u + r) >> 16);
}
void k_f64_to_f32_upper(void** a) {
  const double* A = arg<const double*>(a, 0); const int64_t lda = arg<int64_t>(a, 1); float* R = arg<float*>(a, 2); const int64_t ldr = arg<int64_t>(a, 3), n = arg<int64_t>(a, 4);
  for (int64_t c = 0; c < n; c++) for (int64_t r = 0; r <= c; r++) R[r + c * ldr] = (float)A[r + c * lda];
}
void k_f32_to_f64(void** a) {
  const float* S = arg<const float*>(a, 0); const int64_t lds = arg<int64_t>(a, 1); double* D = arg<double*>(a, 2); const int64_t ldd = arg<int64_t>(a, 3), rows = arg<int64_t>(a, 4),
          cols = arg<int64_t>(a, 5); const int up = arg<int>(a, 6);
  for (int64_t c = 0; c < cols; c++) for (int64_t r = 0; r < rows; r++) D[r + c * ldd] = (!up || r <= c) ? (double)S[r + c * lds] : 0.0;
}
void k_f64_to_f32_bf16(void** a) {
  const double* S = arg<const double*>(a, 0); const int64_t lds = arg<int64_t>(a, 1); float* R = arg<float*>(a, 2); const int64_t ldr = arg<int64_t>(a, 3);
  bf16_t* P = arg<bf16_t*>(a, 4); const int64_t ldp = arg<int64_t>(a, 5), rows = arg<int64_t>(a, 6), cols = arg<int64_t>(a, 7); const int up = arg<int>(a, 8);
  for (int64_t c = 0; c < cols; c++) for (int64_t r = 0; r < rows; r++) {
    const double v = (!up || r <= c) ? S[r + c * lds] : 0.0;
    R[r + c * ldr] = (float)v;
    if (P) P[r + c * ldp] = f2bf((float)v);
  }
}
void bf_split(float x, bf16_t& hi, bf16_t& lo) { hi = f2bf(x); lo = f2bf(x - bf2f(hi)); }
void k_split3_row(void** a) {
  float* S = arg<float*>(a, 0); const int64_t lds = arg<int64_t>(a, 1); bf16_t* B3 = arg<bf16_t*>(a, 2); const int64_t rows = arg<int64_t>(a, 3), cols = arg<int64_t>(a, 4); const int z = arg<int>(a, 5);
  for (int64_t c = 0; c < cols; c++) { bf16_t* out = B3 + c * 3 * rows; for (int64_t r = 0; r < rows; r++) { bf16_t hi, lo; bf_split(S[r + c * lds], hi, lo); out[r] = hi; out[rows + r] = lo; out[2 * rows + r] = hi; if (z) S[r + c * lds] = 0.0f; } }
}
void k_split3_tri(void** a) {
  const double* D = arg<const double*>(a, 0); const int64_t ldd = arg<int64_t>(a, 1); bf16_t* A3 = arg<bf16_t*>(a, 2); const int64_t n = arg<int64_t>(a, 3);
  for (int64_t c = 0; c < n; c++) { bf16_t* out = A3 + c * 3 * n; for (int64_t r = 0; r < n; r++) { bf16_t hi, lo; bf_split(r <= c ? (float)D[r + c * ldd] : 0.0f, hi, lo); out[r] = hi; out[n + r] = hi; out[2 * n + r] = lo; } }
}
void k_f32_to_bf16(void** a) {
  const float* S = arg<const float*>(a, 0); const int64_t lds = arg<int64_t>(a, 1); bf16_t* P = arg<bf16_t*>(a, 2); const int64_t ldp = arg<int64_t>(a, 3), rows = arg<int64_t>(a, 4), cols = arg<int64_t>(a, 5);
  for (int64_t c = 0; c < cols; c++) for (int64_t r = 0; r < rows; r++) P[r + c * ldp] = f2bf(S[r + c * lds]);
}
void k_axpy_cols(void** a, unsigned gy) {
  double* X = arg<double*>(a, 0); const int64_t ldx = arg<int64_t>(a, 1); const double* D = arg<const double*>(a, 2); const int64_t ldd = arg<int64_t>(a, 3), rows = arg<int64_t>(a, 4);
  for (int64_t c = 0; c < gy; c++) for (int64_t r = 0; r < rows; r++) X[r + c * ldx] += D[r + c * ldd];
}
void k_import_f32_bc(void** a) {
  const double* A = arg<const double*>(a, 0); const int64_t lda = arg<int64_t>(a, 1); float* R = arg<float*>(a, 2); const int64_t ld = arg<int64_t>(a, 3), n = arg<int64_t>(a, 4), npad = arg<int64_t>(a, 5),
          nb = arg<int64_t>(a, 6); const int P = arg<int>(a, 7), p = arg<int>(a, 8); const int64_t lc = arg<int64_t>(a, 9);
  for (int64_t l = 0; l < lc; l++) { const int64_t g = ((l / nb) * P + p) * nb + l % nb; for (int64_t r = 0; r < npad; r++) R[r + l * ld] = (r < n && g < n) ? (float)A[r + l * lda] : (r == g ? 1.0f : 0.0f); }
}
void k_sub_from(void** a, unsigned gy) {
  double* T = arg<double*>(a, 0); const int64_t ldt = arg<int64_t>(a, 1); const double* V = arg<const double*>(a, 2); const int64_t ldv = arg<int64_t>(a, 3), rows = arg<int64_t>(a, 4), cols = arg<int64_t>(a, 5);
  for (int64_t c = 0; c < std::min<int64_t>(cols, gy); c++) for (int64_t r = 0; r < rows; r++) T[r + c * ldt] = V[r + c * ldv] - T[r + c * ldt];
}
void k_scatter_rows_bc(void** a, unsigned gy) {
  const double* Q = arg<const double*>(a, 0); const int64_t ldq = arg<int64_t>(a, 1); double* Out = arg<double*>(a, 2); const int64_t ldo = arg<int64_t>(a, 3), nb = arg<int64_t>(a, 4);
  const int P = arg<int>(a, 5), p = arg<int>(a, 6); const int64_t lc = arg<int64_t>(a, 7), cols = arg<int64_t>(a, 8);
  for (int64_t c = 0; c < std::min<int64_t>(cols, gy); c++) for (int64_t l = 0; l < lc; l++) Out[((l / nb) * P + p) * nb + l % nb + c * ldo] = Q[l + c * ldq];
}
// bf16_tn_kernel: C (fp32) += alpha A^T B on 128 x 128 tiles, the launch's blocks walked as the kernel walks them
void k_bf16_tn(void** a, unsigned gx, const int TB = 128, const int KG = 64) {
  const BfArgs g = arg<BfArgs>(a, 0);
  const bf16_t* A = reinterpret_cast<const bf16_t*>(g.A); const bf16_t* B = reinterpret_cast<const bf16_t*>(g.B);
  struct Tile { int ti, tj, gtj; const bf16_t* Abase; };
  std::vector<Tile> tiles;
  for (unsigned b = 0; b < gx; b++) {
    const int L = (b & 7) * g.chunk + (b >> 3);
    int ti, tj;
    if ((int)(b >> 3) >= g.chunk) continue;
    if (!g.stair && g.st > 0) {
      const int per = g.st * g.st, sl = L / per, q = L - sl * per;
      int si, sj;
      if (g.tri && g.tm == g.tn) {
        sj = (int)((sqrtf(8.0f * (float)sl + 1.0f) - 1.0f) * 0.5f);
        while ((sj + 1) * (sj + 2) / 2 <= sl) sj++;
        while (sj * (sj + 1) / 2 > sl) sj--;
        si = sl - sj * (sj + 1) / 2;
      } else { si = sl % g.nsm; sj = sl / g.nsm; }
      ti = si * g.st + q % g.st; tj = sj * g.st + q / g.st;
      if (ti >= g.tm || tj >= g.tn || (g.tri && ti > tj)) continue;
    } else if (!g.stair && g.tri && g.tm == g.tn) {
      tj = (int)((sqrtf(8.0f * (float)L + 1.0f) - 1.0f) * 0.5f);
      while ((tj + 1) * (tj + 2) / 2 <= L) tj++;
      while (tj * (tj + 1) / 2 > L) tj--;
      ti = L - tj * (tj + 1) / 2;
      if (tj >= g.tn) continue;
    } else {
      ti = L % g.tm; tj = L / g.tm;
      if (tj >= g.tn) continue;
      if (!g.stair && g.tri && ti > tj) continue;
    }
    int gtj = tj;
    const bf16_t* Abase = A + (int64_t)ti * TB * g.lda;
    if (g.stair) {
      const int J = g.sp + g.sP * (g.slb0 + tj / g.snbT);
      gtj = (J - g.sJ0) * g.snbT + tj % g.snbT;
      if (ti > gtj) continue;
      const int I = g.sJ0 + ti / g.snbT, r = I % g.sP, lb = I / g.sP - g.gstart[r];
      Abase = A + (int64_t)r * g.gpiece + ((int64_t)(lb * g.snbT + ti % g.snbT) * TB) * g.lda;
    }
    tiles.push_back(Tile{ti, tj, gtj, Abase});
  }
  const int64_t K = (g.K / KG) * KG;
#pragma omp parallel for schedule(dynamic)
  for (size_t t = 0; t < tiles.size(); t++) {          // (every tile of a launch has one writer)
    const int ti = tiles[t].ti, tj = tiles[t].tj, gtj = tiles[t].gtj; const bf16_t* Abase = tiles[t].Abase;
    std::vector<float> fa((size_t)TB * K), fb((size_t)TB * K);
    const int64_t i0 = (int64_t)ti * TB, j0 = (int64_t)tj * TB;
    for (int i = 0; i < TB; i++) for (int64_t k = 0; k < K; k++) fa[(size_t)i * K + k] = bf2f(Abase[(int64_t)i * g.lda + k]);
    for (int j = 0; j < TB; j++) for (int64_t k = 0; k < K; k++) fb[(size_t)j * K + k] = bf2f(B[(j0 + j) * g.ldb + k]);
    const bool diag = g.stair ? (ti == gtj) : (g.tri && ti == tj);
    for (int j = 0; j < TB; j++)
      for (int i = 0; i < TB; i++) {
        if (diag && i > j) continue;
        const float* x = fa.data() + (size_t)i * K; const float* y = fb.data() + (size_t)j * K;
        float s = 0.0f;
#pragma omp simd reduction(+ : s)
        for (int64_t k = 0; k < K; k++) s += x[k] * y[k];
        g.C[i0 + i + (j0 + j) * g.ldc] += g.alpha * s;
      }
  }
}

}  // namespace

#include <omp.h>
// OpenMP threads of the CALLING thread's kernel models (a rank thread of a multi-rank case takes its share of the host cores)
extern "C" void shim_set_threads(int n) { omp_set_num_threads(n < 1 ? 1 : n); }
static int dispatch(const char* mangled, void** args, unsigned gx, unsigned gy, unsigned gz, unsigned bx);
// SHIM_PROFILE=1: seconds per kernel model, printed when the process ends
#include <chrono>
#include <map>
#include <mutex>
static std::map<std::string, std::pair<double, long>>* prof = nullptr;
static std::mutex prof_mu;
static void prof_dump() { if (!prof) return; for (auto& kv : *prof) if (kv.second.first > 0.05) fprintf(stderr, "[shim profile] %8.2f s %7ld x %s\n", kv.second.first, kv.second.second, kv.first.substr(0, 90).c_str()); }
// -> 1: modelled and executed; 0: no model for this kernel (the caller reports it)
extern "C" int shim_cpu_kernel(const char* mangled, void** args, unsigned gx, unsigned gy, unsigned gz, unsigned bx) {
  static const bool on = getenv("SHIM_PROFILE") != nullptr;
  if (!on) return dispatch(mangled, args, gx, gy, gz, bx);
  const auto t0 = std::chrono::steady_clock::now();
  const int r = dispatch(mangled, args, gx, gy, gz, bx);
  const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::lock_guard<std::mutex> lk(prof_mu);
  if (!prof) { prof = new std::map<std::string, std::pair<double, long>>(); atexit(prof_dump); }
  auto& e = (*prof)[mangled]; e.first += dt; e.second++;
  return r;
}
static int dispatch(const char* mangled, void** args, unsigned gx, unsigned gy, unsigned gz, unsigned bx) {
  const std::string n(mangled);
  auto has = [&](const char* s) { return n.find(s) != std::string::npos; };
  if (has("dgemm_tn_dma_kernel")) { k_dgemm_dma(n, args, gx, gy); return 1; }
  if (has("dgemm_small_kernel")) { k_dgemm_small(n, args, gx, gy, gz); return 1; }
  if (has("dgemm_tn_skinny_kernel")) { k_skinny(args, true, gx); return 1; }
  if (has("dgemm_nn_skinny_kernel")) { k_skinny(args, false, gx); return 1; }
  if (has("dgemm_kernel")) { k_dgemm(n, args, gx, gy); return 1; }
  if (has("splitk_reduce_kernel")) { k_splitk_reduce(args, gy); return 1; }
  if (has("scale_kernel")) { k_scale(args, gy); return 1; }
  if (has("fill_symmetric_kernel")) { k_fill_symmetric(args, gx, gy, bx); return 1; }
  if (has("fill_random_kernel")) { k_fill_random(args); return 1; }
  if (has("copy_window_kernel")) { k_copy_window(args); return 1; }
  if (has("copy_rect_v2_kernel")) { k_copy_rect_v2(args); return 1; }
  if (has("zero_rect_kernel")) { k_zero_rect(args); return 1; }
  if (has("remove_triangle_kernel")) { k_remove_triangle(args); return 1; }
  if (has("sumsq_kernel")) { k_sumsq(args); return 1; }
  if (has("cyclic_piece_kernel")) { k_cyclic_piece(args); return 1; }
  if (has("leaf_cholinv_kernel")) { k_leaf_cholinv(args); return 1; }
  if (has("leaf_trtri_kernel")) { k_leaf_trtri(args); return 1; }
  if (has("panel64_solve_update_kernel")) { k_panel64(args, gx); return 1; }
  if (has("trinv_merge_kernel")) { k_trinv_merge(n, args, gy); return 1; }
  if (has("chain64_coop_kernel")) { k_chain64(args); return 1; }
  if (has("spin_kernel")) return 1;
  if (has("f64_to_f32_upper_kernel")) { k_f64_to_f32_upper(args); return 1; }
  if (has("f64_to_f32_bf16_kernel")) { k_f64_to_f32_bf16(args); return 1; }
  if (has("f32_to_f64_kernel")) { k_f32_to_f64(args); return 1; }
  if (has("split3_row_kernel")) { k_split3_row(args); return 1; }
  if (has("split3_tri_kernel")) { k_split3_tri(args); return 1; }
  if (has("f32_to_bf16_kernel")) { k_f32_to_bf16(args); return 1; }
  if (has("axpy_cols_kernel")) { k_axpy_cols(args, gy); return 1; }
  if (has("import_f32_bc_kernel")) { k_import_f32_bc(args); return 1; }
  if (has("sub_from_kernel")) { k_sub_from(args, gy); return 1; }
  if (has("scatter_rows_bc_kernel")) { k_scatter_rows_bc(args, gy); return 1; }
  if (has("bf16_tn_kernel")) { k_bf16_tn(args, gx); return 1; }
  if (has("bf16_tn3_kernel")) { k_bf16_tn(args, gx, 256, 32); return 1; }
  if (has("bf16_tn3w_kernel") || has("bf16_tn3x_kernel")) { k_bf16_tn(args, gx, 256, 64); return 1; }      // third generation: the same walk on 256 x 256 tiles, K in stages of 32
  if (has("fill_symmetric_bc2d_kernel")) { k_fill_symmetric_bc2d(args); return 1; }
  if (has("fill_symmetric_bc_kernel")) { k_fill_symmetric_bc(args); return 1; }
  if (has("pad_identity_kernel")) { k_pad_identity(args); return 1; }
  if (has("export_upper_bc_kernel")) { k_export_upper_bc(args); return 1; }
  if (has("info_to_double")) { k_info_to_double(args); return 1; }
  if (has("identity_blocks_bc_kernel")) { k_identity_blocks_bc(args); return 1; }
  if (has("import_pad_2d_kernel")) { k_import_pad_2d(args); return 1; }
  if (has("export_upper_2d_kernel")) { k_export_upper_2d(args); return 1; }
  if (has("identity_2d_kernel")) { k_identity_2d(args); return 1; }
  if (has("zero_root_2d_kernel")) { k_zero_root_2d(args); return 1; }
  if (has("redist_gather_kernel")) { k_redist(args, true); return 1; }
  if (has("redist_scatter_kernel")) { k_redist(args, false); return 1; }
  if (has("combine_packed_kernel")) { k_combine_packed(args, gy); return 1; }
  if (has("combine_kernel")) { k_combine(args, gy); return 1; }
  if (has("blocks_to_dense_kernel")) { k_blocks_to_dense(args); return 1; }
  if (has("gram256_reduce_kernel")) { k_gram256_reduce(args); return 1; }
  if (has("gram256_kernel")) { k_gram256(args, gx); return 1; }
  if (has("qrapply256_kernel")) { k_qrapply256(args); return 1; }
  return 0;
}
