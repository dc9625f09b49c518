// Replay communicator (measurement harness, NOT part of libcapital_amd.so): lets ONE GPU run the real schedule of rank `rank` of a
// P-rank 1 x P Cholesky plan (csrc/dist.hip, safe mode: one communicator, collectives in program order) at full speed.  The peers'
// contributions are not computed but COPIED out of a finished factor R of the same matrix (single-GPU plan), through the callback
// constructor of the product's communicator (cap_comm_create_callbacks, include/capital_amd.h):
//   broadcast k   (msg(k) = [ R(k-1,k) | Dinv(k) ], 2 nb^2 doubles, root = k % P)  foreign root: the two blocks are copied from R / the
//                 table of diagonal-block inverses, behind a spin of  lat_us + bytes / link_GBps
//   all-gather t  (the solved strip right of itself, q nb x cols per rank)          every peer's piece is gathered from R's rows of the
//                 strip, behind ONE spin of  lat_us + piece bytes / link_GBps (xGMI is point-to-point: each peer's piece has a link
//                 of its own); my own piece is the one I computed
//   all-reduce    not used by safe mode without the IPC exchange: identity
// What the replay measures is rank `rank`'s OWN timeline - its kernels at full speed, what its streams wait for - with peers that
// are never late beyond the link model and the owner's chain model (dist.hip option remote_chain_us).  It is a projection of the
// P-GPU run from one GPU, not a measurement of it: RCCL's kernels, their CU share and real arrival skew are absent.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include <algorithm>
#include <new>
#include <vector>

#include "../../../include/capital_amd.h"

namespace {
struct ReplayCtx {
  int rank, size;
  const double* R; int64_t ldr, n, nb, nblk; int strip;
  const double* Dinv;                         // nblk blocks of nb x nb (ld nb): inverses of R's diagonal blocks
  double link_GBps, lat_us; int channels;
  int64_t nbcast, ngather;                    // position in the factor call's sequence
  int64_t bytes_in; double model_us; int64_t calls;   // totals since the last reset (what the link model charged)
};

inline int64_t lbfirst(int64_t r, int64_t k, int64_t P) { return k >= r ? (k - r) / P + 1 : 0; }
inline int64_t nblocks_of(int64_t r, int64_t nblk, int64_t P) { return r < nblk ? (nblk - 1 - r) / P + 1 : 0; }

// the stand-in for a collective's kernel: `channels` workgroups of 512 threads holding 32 KiB of LDS each (what an RCCL channel occupies) for the
// modelled duration; one workgroup of 64 threads when channels <= 1 (copy engines / IPC pushes: no CU share)
__global__ void replay_spin_kernel(int us) {
  __shared__ char hold[32768];
  if (blockDim.x > 64) hold[threadIdx.x] = 0;
  const uint64_t t0 = wall_clock64();
  while (wall_clock64() - t0 < (uint64_t)us * 100ull) __builtin_amdgcn_s_sleep(32);
}
// dst (ld ldd) = src (ld lds): rows x cols, 2 doubles per lane
__global__ void replay_copy_kernel(const double* src, int64_t lds_, double* dst, int64_t ldd, int64_t rows, int64_t cols) {
  const int64_t r = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 2, c = blockIdx.y + (int64_t)blockIdx.z * 65535;
  if (r >= rows || c >= cols) return;
  if (r + 1 < rows) *reinterpret_cast<double2*>(dst + r + c * ldd) = *reinterpret_cast<const double2*>(src + r + c * lds_);
  else dst[r + c * ldd] = src[r + c * lds_];
}
// one peer's piece of a strip exchange: local blocks lb0 .. lb0 + nbl - 1 of rank r (global block lb P + r), rows [row0, row0 + ldS)
__global__ void replay_gather_kernel(const double* R, int64_t ldr, double* piece, int64_t ldS, int64_t row0, int64_t nb, int P, int r, int64_t lb0) {
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 2, c = blockIdx.y, lb = blockIdx.z;
  if (i >= ldS) return;
  const int64_t gcol = ((lb0 + lb) * P + r) * nb + c;
  *reinterpret_cast<double2*>(piece + i + (lb * nb + c) * ldS) = *reinterpret_cast<const double2*>(R + row0 + i + gcol * ldr);
}
__global__ void replay_fill1_kernel(double* recv, const double* send, int P) { if ((int)threadIdx.x < P) recv[threadIdx.x] = send[0]; }

int spin(ReplayCtx* c, double bytes, hipStream_t s) {
  const double us = c->lat_us + (c->link_GBps > 0 ? bytes / (c->link_GBps * 1e3) : 0.0);
  c->model_us += us; c->calls++;
  if (us >= 1.0) hipLaunchKernelGGL(replay_spin_kernel, dim3(c->channels > 1 ? c->channels : 1), dim3(c->channels > 1 ? 512 : 64), 0, s, (int)(us + 0.5));
  return 0;
}

int cb_bcast(void* ctx, double* buf, int64_t count, int root, void* stream) {
  ReplayCtx* c = (ReplayCtx*)ctx;
  hipStream_t s = (hipStream_t)stream;
  const int64_t nb2 = c->nb * c->nb;
  if (count != 2 * nb2) { fprintf(stderr, "replay: unexpected broadcast of %lld doubles\n", (long long)count); return 1; }
  const int64_t k = c->nbcast++;
  if (k >= c->nblk || root != (int)(k % c->size)) { fprintf(stderr, "replay: broadcast %lld out of sequence (root %d)\n", (long long)k, root); return 1; }
  if (root != c->rank) {
    spin(c, (double)count * 8.0, s);
    c->bytes_in += count * 8;
    const dim3 grid((unsigned)((c->nb / 2 + 255) / 256), (unsigned)c->nb);
    if (c->strip == 2 && (k & 1) && k >= 1)      // second block row of a strip: R(a, b) rides along
      hipLaunchKernelGGL(replay_copy_kernel, grid, dim3(256), 0, s, c->R + (k - 1) * c->nb + k * c->nb * c->ldr, c->ldr, buf, c->nb, c->nb, c->nb);
    hipLaunchKernelGGL(replay_copy_kernel, grid, dim3(256), 0, s, c->Dinv + k * nb2, c->nb, buf + nb2, c->nb, c->nb, c->nb);
  }
  if (c->nbcast == c->nblk) { c->nbcast = 0; c->ngather = 0; }      // the factor call is complete: the next one starts over
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

int cb_allgather(void* ctx, const double* send, double* recv, int64_t cpr, void* stream) {
  ReplayCtx* c = (ReplayCtx*)ctx;
  hipStream_t s = (hipStream_t)stream;
  const int P = c->size;
  if (cpr == 1) {                                // cap_dist_info: every peer reports what I report
    hipLaunchKernelGGL(replay_fill1_kernel, dim3(1), dim3(64), 0, s, recv, send, P);
    return hipGetLastError() == hipSuccess ? 0 : 1;
  }
  const int64_t t = c->ngather++;
  const int64_t a = t * c->strip, q = c->strip < c->nblk - a ? c->strip : c->nblk - a, b = a + q - 1, ldS = q * c->nb;
  int64_t nmax = 0;
  for (int r = 0; r < P; r++) { const int64_t w = (nblocks_of(r, c->nblk, P) - lbfirst(r, b, P)) * c->nb; if (w > nmax) nmax = w; }
  if (a >= c->nblk || cpr != ldS * nmax) { fprintf(stderr, "replay: all-gather %lld out of sequence (%lld doubles per rank, expected %lld)\n", (long long)t, (long long)cpr, (long long)(ldS * nmax)); return 1; }
  spin(c, (double)cpr * 8.0, s);                 // every peer's piece on its own link, all at once
  for (int r = 0; r < P; r++) {
    if (r == c->rank) continue;
    const int64_t lb0 = lbfirst(r, b, P), nbl = nblocks_of(r, c->nblk, P) - lb0;
    if (nbl <= 0) continue;
    c->bytes_in += nbl * c->nb * ldS * 8;
    hipLaunchKernelGGL(replay_gather_kernel, dim3((unsigned)((ldS / 2 + 255) / 256), (unsigned)c->nb, (unsigned)nbl), dim3(256), 0, s, c->R, c->ldr,
                       recv + (int64_t)r * cpr, ldS, a * c->nb, c->nb, P, r, lb0);
  }
  if (send != recv + (int64_t)c->rank * cpr) (void)hipMemcpyAsync(recv + (int64_t)c->rank * cpr, send, sizeof(double) * cpr, hipMemcpyDeviceToDevice, s);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}

int cb_allreduce(void*, double*, int64_t, void*) { return 0; }
}  // namespace

extern "C" {
// R: the finished n x n factor (column-major, ld ldr, n % nb == 0); Dinv: nblk blocks nb x nb, the inverses of R's diagonal blocks (upper).
// strip = the plan's block rows per strip (cap_dist_get_option "strip").  *ctx_out is owned by the caller (cap_replay_destroy).
int cap_replay_create(cap_comm** comm, void** ctx_out, int rank, int size, const double* R, int64_t ldr, int64_t n, int64_t nb, int strip,
                      const double* Dinv, double link_GBps, double lat_us) {
  if (!comm || !ctx_out || !R || !Dinv || size < 1 || size > 8 || rank < 0 || rank >= size || n <= 0 || nb <= 0 || n % nb || (strip != 1 && strip != 2)) return CAP_ERR_ARG;
  ReplayCtx* c = new (std::nothrow) ReplayCtx();
  if (!c) return CAP_ERR_ALLOC;
  c->rank = rank; c->size = size; c->R = R; c->ldr = ldr; c->n = n; c->nb = nb; c->nblk = n / nb; c->strip = strip; c->Dinv = Dinv;
  c->link_GBps = link_GBps; c->lat_us = lat_us; c->channels = 0; c->nbcast = c->ngather = 0; c->bytes_in = 0; c->model_us = 0; c->calls = 0;
  const int st = cap_comm_create_callbacks(comm, rank, size, cb_allgather, cb_bcast, cb_allreduce, c);
  if (st != CAP_OK) { delete c; return st; }
  *ctx_out = c;
  return CAP_OK;
}
// out3 = bytes the peers "sent" since the last reset, microseconds the link model charged, collectives it charged; resets the totals
int cap_replay_stats(void* ctx, double* out3) {
  ReplayCtx* c = (ReplayCtx*)ctx;
  if (!c || !out3) return CAP_ERR_ARG;
  out3[0] = (double)c->bytes_in; out3[1] = c->model_us; out3[2] = (double)c->calls;
  c->bytes_in = 0; c->model_us = 0; c->calls = 0;
  return CAP_OK;
}
// channels > 1: the link-time spin of every collective occupies that many workgroups (the CU share of an RCCL kernel); <= 1: one small workgroup
int cap_replay_set_channels(void* ctx, int channels) { ReplayCtx* c = (ReplayCtx*)ctx; if (!c || channels < 0 || channels > 64) return CAP_ERR_ARG; c->channels = channels; return CAP_OK; }
int cap_replay_set_strip(void* ctx, int strip) { ReplayCtx* c = (ReplayCtx*)ctx; if (!c || (strip != 1 && strip != 2)) return CAP_ERR_ARG; c->strip = strip; c->nbcast = c->ngather = 0; return CAP_OK; }
void cap_replay_destroy(void* ctx) { delete (ReplayCtx*)ctx; }
}

// ================================================================================================================================ 2D
// The same for the Pr x Pc block-cyclic plan (csrc/dist2d.hip, safe mode): three communicators - world (only cap_dist2d_info), my
// process row, my process column - whose broadcasts follow the plan's loop exactly, so the callbacks GENERATE that sequence and check
// every call against it:
//   row:     M(k) = msg(k) = [ R(a,b) | Dinv(k) ] for the block rows k of MY process row (root k % Pc), then per strip the A-operand
//            pieces A(t, m) of the contributors pcs = pr % Pr + m Pr (the strip's rows at THEIR columns J >= e, ld = q nb)
//   column:  C(k) = the solved block row k at MY columns J > k (nb x ncols, ld = nb), root k % Pr
// Foreign roots: the payload is gathered out of R (the table of inverses for Dinv) behind the link model; my own roots: nothing.
namespace {
struct Replay2D;
struct View2D { Replay2D* c; int kind; };      // 0 world, 1 row, 2 column
struct Ev2D { int type; int64_t k, t; int m; };  // type 0 M(k), 1 A(t, m), 2 C(k)
struct Replay2D {
  int rank, Pr, Pc, pr, pc; const double* R; int64_t ldr, n, nb, nblk; int strip; const double* Dinv; double link_GBps, lat_us;
  std::vector<Ev2D> row, col; size_t irow, icol;
  int64_t bytes_in; double model_us; int64_t calls;
  View2D view[3];
};
inline int64_t nlc_of(const Replay2D* c, int q) { return nblocks_of(q, c->nblk, c->Pc); }
void build_sequences(Replay2D* c) {
  c->row.clear(); c->col.clear(); c->irow = c->icol = 0;
  const int ncon = c->Pc / c->Pr;
  for (int64_t a = 0, t = 0; a < c->nblk; a += c->strip, t++) {
    const int64_t q = std::min<int64_t>(c->strip, c->nblk - a), b = a + q - 1, e = b + 1;
    for (int64_t r = 0; r < q; r++) {
      const int64_t k = a + r;
      if (c->pr == (int)(k % c->Pr) && c->Pc > 1) c->row.push_back(Ev2D{0, k, t, 0});
      const int64_t ncols = (nlc_of(c, c->pc) - lbfirst(c->pc, k, c->Pc)) * c->nb;
      if (ncols > 0 && c->Pr > 1) c->col.push_back(Ev2D{2, k, t, 0});
    }
    if (e >= c->nblk || c->Pc == 1) continue;
    for (int m = 0; m < ncon; m++) {
      const int pcs = c->pr % c->Pr + m * c->Pr;
      const int64_t cs = (nlc_of(c, pcs) - lbfirst(pcs, b, c->Pc)) * c->nb;
      if (cs > 0) c->row.push_back(Ev2D{1, a, t, m});
    }
  }
}
int spin2(Replay2D* c, double bytes, hipStream_t s) {
  const double us = c->lat_us + (c->link_GBps > 0 ? bytes / (c->link_GBps * 1e3) : 0.0);
  c->model_us += us; c->calls++;
  if (us >= 1.0) hipLaunchKernelGGL(replay_spin_kernel, dim3(1), dim3(64), 0, s, (int)(us + 0.5));
  return 0;
}
// piece (ld = ldp) = rows [row0, row0 + rows) of R at the columns of process column q's local blocks lb0 ..: global block lb Pc + q
__global__ void replay_gather2d_kernel(const double* R, int64_t ldr, double* piece, int64_t ldp, int64_t row0, int64_t rows, int64_t nb, int Pc, int q, int64_t lb0) {
  const int64_t i = (blockIdx.x * (int64_t)blockDim.x + threadIdx.x) * 2, c = blockIdx.y, lb = blockIdx.z;
  if (i >= rows) return;
  const int64_t gcol = ((lb0 + lb) * Pc + q) * nb + c;
  *reinterpret_cast<double2*>(piece + i + (lb * nb + c) * ldp) = *reinterpret_cast<const double2*>(R + row0 + i + gcol * ldr);
}
int cb2_bcast(void* vctx, double* buf, int64_t count, int root, void* stream) {
  View2D* v = (View2D*)vctx; Replay2D* c = v->c;
  hipStream_t s = (hipStream_t)stream;
  const int64_t nb = c->nb, nb2 = nb * nb;
  if (v->kind == 1) {
    if (c->irow >= c->row.size()) { fprintf(stderr, "replay2d: row broadcast beyond the sequence\n"); return 1; }
    const Ev2D ev = c->row[c->irow++];
    if (ev.type == 0) {
      const int64_t k = ev.k;
      if (count != 2 * nb2 || root != (int)(k % c->Pc)) { fprintf(stderr, "replay2d: msg(%lld) out of sequence (count %lld root %d)\n", (long long)k, (long long)count, root); return 1; }
      if (root != c->pc) {
        spin2(c, (double)count * 8.0, s); c->bytes_in += count * 8;
        const dim3 grid((unsigned)((nb / 2 + 255) / 256), (unsigned)nb);
        if (c->strip == 2 && (k & 1)) hipLaunchKernelGGL(replay_copy_kernel, grid, dim3(256), 0, s, c->R + (k - 1) * nb + k * nb * c->ldr, c->ldr, buf, nb, nb, nb);
        hipLaunchKernelGGL(replay_copy_kernel, grid, dim3(256), 0, s, c->Dinv + k * nb2, nb, buf + nb2, nb, nb, nb);
      }
    } else {
      const int64_t a = ev.k, q = std::min<int64_t>(c->strip, c->nblk - a), b = a + q - 1, ldS = q * nb;
      const int pcs = c->pr % c->Pr + ev.m * c->Pr;
      const int64_t lbs = lbfirst(pcs, b, c->Pc), nbl = nlc_of(c, pcs) - lbs;
      if (count != ldS * nbl * nb || root != pcs) { fprintf(stderr, "replay2d: A(%lld, %d) out of sequence (count %lld root %d)\n", (long long)ev.t, ev.m, (long long)count, root); return 1; }
      if (pcs != c->pc) {
        spin2(c, (double)count * 8.0, s); c->bytes_in += count * 8;
        hipLaunchKernelGGL(replay_gather2d_kernel, dim3((unsigned)((ldS / 2 + 255) / 256), (unsigned)nb, (unsigned)nbl), dim3(256), 0, s, c->R, c->ldr, buf, ldS, a * nb, ldS, nb,
                           c->Pc, pcs, lbs);
      }
    }
    if (c->irow == c->row.size() && c->icol == c->col.size()) { c->irow = c->icol = 0; }
  } else if (v->kind == 2) {
    if (c->icol >= c->col.size()) { fprintf(stderr, "replay2d: column broadcast beyond the sequence\n"); return 1; }
    const Ev2D ev = c->col[c->icol++];
    const int64_t k = ev.k, lbk = lbfirst(c->pc, k, c->Pc), nbl = nlc_of(c, c->pc) - lbk;
    if (count != nb * nbl * nb || root != (int)(k % c->Pr)) { fprintf(stderr, "replay2d: C(%lld) out of sequence (count %lld root %d)\n", (long long)k, (long long)count, root); return 1; }
    if (root != c->pr) {
      spin2(c, (double)count * 8.0, s); c->bytes_in += count * 8;
      hipLaunchKernelGGL(replay_gather2d_kernel, dim3((unsigned)((nb / 2 + 255) / 256), (unsigned)nb, (unsigned)nbl), dim3(256), 0, s, c->R, c->ldr, buf, nb, k * nb, nb, nb, c->Pc,
                         c->pc, lbk);
    }
    if (c->irow == c->row.size() && c->icol == c->col.size()) { c->irow = c->icol = 0; }
  } else {
    fprintf(stderr, "replay2d: unexpected broadcast on the world communicator\n"); return 1;
  }
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
int cb2_allgather(void* vctx, const double* send, double* recv, int64_t cpr, void* stream) {
  View2D* v = (View2D*)vctx;
  if (cpr != 1) { fprintf(stderr, "replay2d: unexpected all-gather of %lld doubles per rank\n", (long long)cpr); return 1; }
  const int np = v->kind == 0 ? v->c->Pr * v->c->Pc : (v->kind == 1 ? v->c->Pc : v->c->Pr);
  hipLaunchKernelGGL(replay_fill1_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, recv, send, np);
  return hipGetLastError() == hipSuccess ? 0 : 1;
}
int cb2_allreduce(void*, double*, int64_t, void*) { return 0; }
}  // namespace

extern "C" {
// three communicators of rank `rank` = pr Pc + pc of a Pr x Pc grid (Pr | Pc) for cap_dist2d_plan_create(world, Pr, row, col); one context
int cap_replay2d_create(cap_comm** world, cap_comm** row, cap_comm** col, void** ctx_out, int rank, int Pr, int Pc, const double* R, int64_t ldr, int64_t n, int64_t nb,
                        int strip, const double* Dinv, double link_GBps, double lat_us) {
  if (!world || !row || !col || !ctx_out || !R || !Dinv || Pr < 1 || Pc < 1 || Pr * Pc > 8 || Pc % Pr || rank < 0 || rank >= Pr * Pc || n <= 0 || nb <= 0 || n % nb ||
      (strip != 1 && strip != 2)) return CAP_ERR_ARG;
  Replay2D* c = new (std::nothrow) Replay2D();
  if (!c) return CAP_ERR_ALLOC;
  c->rank = rank; c->Pr = Pr; c->Pc = Pc; c->pr = rank / Pc; c->pc = rank % Pc; c->R = R; c->ldr = ldr; c->n = n; c->nb = nb; c->nblk = n / nb; c->strip = strip;
  c->Dinv = Dinv; c->link_GBps = link_GBps; c->lat_us = lat_us; c->bytes_in = 0; c->model_us = 0; c->calls = 0;
  for (int i = 0; i < 3; i++) { c->view[i].c = c; c->view[i].kind = i; }
  build_sequences(c);
  int st = cap_comm_create_callbacks(world, rank, Pr * Pc, cb2_allgather, cb2_bcast, cb2_allreduce, &c->view[0]);
  if (st == CAP_OK) st = cap_comm_create_callbacks(row, c->pc, Pc, cb2_allgather, cb2_bcast, cb2_allreduce, &c->view[1]);
  if (st == CAP_OK) st = cap_comm_create_callbacks(col, c->pr, Pr, cb2_allgather, cb2_bcast, cb2_allreduce, &c->view[2]);
  if (st != CAP_OK) { delete c; return st; }
  *ctx_out = c;
  return CAP_OK;
}
int cap_replay2d_stats(void* ctx, double* out3) {
  Replay2D* c = (Replay2D*)ctx;
  if (!c || !out3) return CAP_ERR_ARG;
  out3[0] = (double)c->bytes_in; out3[1] = c->model_us; out3[2] = (double)c->calls;
  c->bytes_in = 0; c->model_us = 0; c->calls = 0;
  return CAP_OK;
}
int cap_replay2d_set_strip(void* ctx, int strip) { Replay2D* c = (Replay2D*)ctx; if (!c || (strip != 1 && strip != 2)) return CAP_ERR_ARG; c->strip = strip; build_sequences(c); return CAP_OK; }
void cap_replay2d_destroy(void* ctx) { delete (Replay2D*)ctx; }
}
