// GPU-resident matrix descriptor + pinned host staging.
//
// Replaces matrix<T,U,Structure,Offload> (reference src/matrix/matrix.h:9-97): global dimensions, process-grid
// dimensions, local (ceil) dimensions, one data buffer and an ownership flag - the injection constructor
// (matrix.hpp:52-74) keeps the caller's buffer, the allocating ones (matrix.hpp:5-50, _register_ :141-155) own it
// and _destroy_ (:157-169) frees it.  The reference's scratch / pad buffers do not exist here (workspaces live in
// the plans).  Element-cyclic layout as upstream: local dims = ceil(global / grid), column-major, zero padded.
//
// Host <-> HBM traffic (upstream matrices live in host memory; north_star: "GPU-resident ... with pinned host
// staging"): import / export move a column-major host matrix through TWO pinned chunk buffers - while chunk i
// travels over PCIe on the copy stream, a few host threads fill (drain) chunk i + 1 - unless the host pointer is
// already pinned (hipHostMalloc / hipHostRegister), in which case the engine copies straight from it.
#include <algorithm>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "common.h"

struct cap_desc {
  int64_t gx, gy;        // global columns (X) / rows (Y): upstream's dimensionX / dimensionY naming (matrix.h:19,47)
  int64_t px, py;        // process grid (columns of processes = px, rows of processes = py)
  int64_t lx, ly, ld;    // local columns / rows, leading dimension (even: 16-byte aligned columns)
  // kind 0: element-cyclic (upstream, matrix.hpp:8-11): local row r of process row qy is global row qy + py r; the position
  //         (qx, qy) is not stored (upstream passes it to the generators again, matrix.h:65-68) unless given (-1 = unknown)
  // kind 1: block-cyclic with nb x nb blocks: global block I on process I mod p as local block I div p - the layout the multi-GPU
  //         plans run on (cap_dist_*: py = 1; cap_dist2d_*: py = Pr, px = Pc); local dims = the VALID rows / columns of position (qx, qy)
  int kind; int64_t nb, qx, qy;
  double* data; bool owns;
  // staging state (created on first use)
  double* pin[2]; int64_t pin_elems; hipStream_t s_copy; hipEvent_t ev[2]; hipEvent_t ev_done;
};

namespace {
constexpr int64_t CHUNK_BYTES = 64ll << 20;     // 64 MiB per pinned buffer: ~1.2 ms of PCIe Gen5, amortises the event round trip

int ensure_staging(cap_desc* d) {
  if (d->pin[0]) return CAP_OK;
  d->pin_elems = CHUNK_BYTES / 8;
  for (int i = 0; i < 2; i++) {
    CAP_HIP(hipHostMalloc((void**)&d->pin[i], CHUNK_BYTES, hipHostMallocDefault));
    CAP_HIP(hipEventCreateWithFlags(&d->ev[i], hipEventDisableTiming));
  }
  CAP_HIP(hipEventCreateWithFlags(&d->ev_done, hipEventDisableTiming));
  CAP_HIP(hipStreamCreateWithFlags(&d->s_copy, hipStreamNonBlocking));
  return CAP_OK;
}

bool host_is_pinned(const void* p) {
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}

// copy `cols` columns of `rows` doubles between two column-major images with a few host threads
void host_copy_cols(const double* src, int64_t lds, double* dst, int64_t ldd, int64_t rows, int64_t cols) {
  cap_acc_host(CAP_ACC_R, src, lds, rows, cols); cap_acc_host(CAP_ACC_W, dst, ldd, rows, cols);      // (only the pinned side is memory the checker knows)
  const int64_t bytes = rows * cols * 8;
  int nt = (int)std::min<int64_t>(8, std::max<int64_t>(1, bytes / (4 << 20)));
  nt = std::min<int>(nt, std::max(1u, std::thread::hardware_concurrency()));
  auto work = [&](int t) {
    for (int64_t c = cols * t / nt; c < cols * (t + 1) / nt; c++) memcpy(dst + c * ldd, src + c * lds, (size_t)rows * 8);
  };
  if (nt == 1) { work(0); return; }
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
}
// number of indices in [0, g) that process q of p owns, and the run structure of the map local -> global index:
// kind 0: runs of 1 with stride p (one run when p == 1); kind 1: runs of nb
int64_t owned(int kind, int64_t g, int64_t nb, int64_t p, int64_t q) {
  if (kind == 0) return q < g ? (g - q + p - 1) / p : 0;
  const int64_t nblk = cap_ceil_div(g, nb);
  int64_t cnt = 0;
  for (int64_t I = q; I < nblk; I += p) cnt += std::min(nb, g - I * nb);
  return cnt;
}
inline int64_t run_len(int kind, int64_t nb, int64_t p) { return kind == 0 ? (p == 1 ? (int64_t)1 << 62 : 1) : nb; }
// global index of local index l (l < owned)
inline int64_t to_global(int kind, int64_t nb, int64_t p, int64_t q, int64_t l) {
  if (kind == 0) return q + p * l;
  return ((l / nb) * p + q) * nb + l % nb;
}

// host GLOBAL matrix <-> packed image of local columns [c0, c0 + nc) (ld = local rows): the row runs of every column.  Element-cyclic
// pieces are ceil-sized (matrix.hpp:8-11): local rows / columns beyond what this position owns are padding - zero-filled on the way in,
// skipped on the way out, never looked up in the host matrix.
void host_pack_cols(const cap_desc* d, const double* hostg, int64_t ldh, double* img, int64_t c0, int64_t nc, bool to_image) {
  const int64_t rows = d->ly, rl = run_len(d->kind, d->nb, d->py);
  const int64_t vrows = owned(d->kind, d->gy, d->nb, d->py, d->qy), vcols = owned(d->kind, d->gx, d->nb, d->px, d->qx);
  cap_acc_host(to_image ? CAP_ACC_W : CAP_ACC_R, img, rows, rows, nc);
  const int64_t bytes = rows * nc * 8;
  int nt = (int)std::min<int64_t>(8, std::max<int64_t>(1, bytes / (4 << 20)));
  nt = std::min<int>(nt, std::max(1u, std::thread::hardware_concurrency()));
  auto work = [&](int t) {
    for (int64_t c = nc * t / nt; c < nc * (t + 1) / nt; c++) {
      double* icol = img + c * rows;
      if (c0 + c >= vcols) { if (to_image) memset(icol, 0, (size_t)rows * 8); continue; }
      const int64_t gc = to_global(d->kind, d->nb, d->px, d->qx, c0 + c);
      for (int64_t r = 0; r < vrows;) {
        const int64_t gr = to_global(d->kind, d->nb, d->py, d->qy, r);
        const int64_t len = std::min(vrows - r, rl - (d->kind == 1 ? r % d->nb : 0));
        double* hp = const_cast<double*>(hostg) + gr + gc * ldh;
        if (to_image) memcpy(icol + r, hp, (size_t)len * 8);
        else memcpy(hp, icol + r, (size_t)len * 8);
        r += len;
      }
      if (to_image && vrows < rows) memset(icol + vrows, 0, (size_t)(rows - vrows) * 8);
    }
  };
  if (nt == 1) { work(0); return; }
  std::vector<std::thread> th;
  for (int t = 1; t < nt; t++) th.emplace_back(work, t);
  work(0);
  for (auto& x : th) x.join();
}
}  // namespace

extern "C" {

static int desc_new(cap_desc** out, int64_t gx, int64_t gy, int64_t px, int64_t py, double* data, int64_t ld) {
  if (!out || gx <= 0 || gy <= 0 || px <= 0 || py <= 0) return CAP_ERR_ARG;
  cap_desc* d = new (std::nothrow) cap_desc();
  if (!d) return CAP_ERR_ALLOC;
  memset(d, 0, sizeof(*d));
  d->gx = gx; d->gy = gy; d->px = px; d->py = py;
  d->kind = 0; d->nb = 1; d->qx = d->qy = -1;
  d->lx = cap_ceil_div(gx, px); d->ly = cap_ceil_div(gy, py);      // matrix.hpp:8-11
  if (data) {
    if (ld < d->ly) { delete d; return CAP_ERR_ARG; }
    d->data = data; d->ld = ld; d->owns = false;
  } else {
    d->ld = cap_round_up(d->ly, 2);
    if (hipMalloc((void**)&d->data, sizeof(double) * d->ld * d->lx) != hipSuccess) { delete d; return CAP_ERR_ALLOC; }
    if (hipMemset(d->data, 0, sizeof(double) * d->ld * d->lx) != hipSuccess) { (void)hipFree(d->data); delete d; return CAP_ERR_HIP; }
    d->owns = true;
  }
  *out = d;
  return CAP_OK;
}

// matrix(globalDimensionX, globalDimensionY, globalPgridX, globalPgridY) - matrix.hpp:5-27: allocates the zero-filled piece
int cap_desc_create(cap_desc** desc, int64_t global_cols, int64_t global_rows, int64_t grid_x, int64_t grid_y) {
  return desc_new(desc, global_cols, global_rows, grid_x, grid_y, nullptr, 0);
}
// the injection constructor - matrix.hpp:52-74: wraps a caller-owned DEVICE buffer (never freed by the descriptor)
int cap_desc_create_view(cap_desc** desc, int64_t global_cols, int64_t global_rows, int64_t grid_x, int64_t grid_y, double* device_data,
                         int64_t ld) {
  if (!device_data) return CAP_ERR_ARG;
  return desc_new(desc, global_cols, global_rows, grid_x, grid_y, device_data, ld);
}

int cap_desc_destroy(cap_desc* d) {
  if (!d) return CAP_OK;
  if (d->pin[0]) {
    (void)hipStreamSynchronize(d->s_copy);
    for (int i = 0; i < 2; i++) { (void)hipHostFree(d->pin[i]); (void)hipEventDestroy(d->ev[i]); }
    (void)hipEventDestroy(d->ev_done);
    cap_stream_destroy(d->s_copy);
  }
  if (d->owns && d->data) (void)hipFree(d->data);
  delete d;
  return CAP_OK;
}

// Block-cyclic kind (north_star: "2D block-cyclic matrix descriptor"): nb x nb blocks, block (I, J) on process (I mod Pr, J mod Pc)
// as local block (I div Pr, J div Pc); this descriptor is the piece of process (pr, pc): the VALID local rows x columns, compact,
// column-major - exactly what cap_dist2d_factor / cap_dist2d_get_R take (Pr = 1: the block columns of cap_dist_factor / cap_cholinv_factor
// on a multi-rank plan).  device_data == NULL: allocated and zero-filled; else the injection constructor (matrix.hpp:52-74).
int cap_desc_create_bc(cap_desc** desc, int64_t global_cols, int64_t global_rows, int64_t nb, int Pr, int Pc, int pr, int pc,
                       double* device_data, int64_t ld) {
  if (!desc || global_cols <= 0 || global_rows <= 0 || nb <= 0 || Pr <= 0 || Pc <= 0 || pr < 0 || pr >= Pr || pc < 0 || pc >= Pc) return CAP_ERR_ARG;
  cap_desc* d = new (std::nothrow) cap_desc();
  if (!d) return CAP_ERR_ALLOC;
  memset(d, 0, sizeof(*d));
  d->gx = global_cols; d->gy = global_rows; d->px = Pc; d->py = Pr; d->kind = 1; d->nb = nb; d->qx = pc; d->qy = pr;
  d->lx = owned(1, global_cols, nb, Pc, pc); d->ly = owned(1, global_rows, nb, Pr, pr);
  if (device_data) {
    if (ld < std::max<int64_t>(d->ly, 1)) { delete d; return CAP_ERR_ARG; }
    d->data = device_data; d->ld = ld; d->owns = false;
  } else {
    d->ld = std::max<int64_t>(cap_round_up(d->ly, 2), 2);
    const size_t bytes = sizeof(double) * (size_t)d->ld * (size_t)std::max<int64_t>(d->lx, 1);
    if (hipMalloc((void**)&d->data, bytes) != hipSuccess) { delete d; return CAP_ERR_ALLOC; }
    if (hipMemset(d->data, 0, bytes) != hipSuccess) { (void)hipFree(d->data); delete d; return CAP_ERR_HIP; }
    d->owns = true;
  }
  *desc = d;
  return CAP_OK;
}

// element-cyclic descriptors do not know their grid position (upstream passes it to the generators again, matrix.h:65-68); the
// global import / export below need it
int cap_desc_set_position(cap_desc* d, int64_t x, int64_t y) {
  if (!d || d->kind != 0 || x < 0 || x >= d->px || y < 0 || y >= d->py) return CAP_ERR_ARG;
  d->qx = x; d->qy = y;
  return CAP_OK;
}

double* cap_desc_data(cap_desc* d) { return d ? d->data : nullptr; }
// field: 0 global cols, 1 global rows, 2 local cols, 3 local rows, 4 ld, 5 owns, 6 grid x, 7 grid y, 8 num_elems (local rows * cols),
//        9 kind (0 element-cyclic, 1 block-cyclic), 10 nb, 11 my grid column (pc / x, -1 unknown), 12 my grid row (pr / y)
int64_t cap_desc_get(const cap_desc* d, int field) {
  if (!d) return -1;
  const int64_t v[13] = {d->gx, d->gy, d->lx, d->ly, d->ld, d->owns ? 1 : 0, d->px, d->py, d->lx * d->ly, d->kind, d->nb, d->qx, d->qy};
  return (field >= 0 && field < 13) ? v[field] : -1;
}

// host (column-major local piece, ld_host >= local rows) -> HBM.  Ordered behind `stream`'s earlier work on the
// descriptor; returns after the last chunk has been handed to the copy engine and makes `stream` wait for it.
int cap_desc_import_host(cap_desc* d, const double* host, int64_t ld_host, void* stream) {
  if (!d || !host || ld_host < d->ly) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  const int64_t rows = d->ly, cols = d->lx;
  if (rows == 0 || cols == 0) return CAP_OK;            // (a block-cyclic piece can be empty)
  if (host_is_pinned(host)) {
    CAP_HIP(hipMemcpy2DAsync(d->data, d->ld * 8, host, ld_host * 8, rows * 8, cols, hipMemcpyHostToDevice, s));
    return CAP_OK;
  }
  CAP_TRY(ensure_staging(d));
  // a previous import may still have H2D copies in flight that READ the two pinned buffers (this call returns before they
  // finish, by design): drain the copy stream before the host refills them
  CAP_HIP(hipStreamSynchronize(d->s_copy));
  CAP_HIP(hipEventRecord(d->ev_done, s));                 // earlier users of the device buffer
  CAP_HIP(hipStreamWaitEvent(d->s_copy, d->ev_done, 0));
  const int64_t cpc = std::max<int64_t>(1, d->pin_elems / rows);   // columns per chunk (a column longer than the chunk is split below)
  if (rows > d->pin_elems) {
    // very tall columns: move each column in row segments
    for (int64_t c = 0, k = 0; c < cols; c++)
      for (int64_t r0 = 0; r0 < rows; r0 += d->pin_elems, k++) {
        const int b = (int)(k & 1); const int64_t len = std::min(d->pin_elems, rows - r0);
        if (k >= 2) CAP_HIP(hipEventSynchronize(d->ev[b]));
        host_copy_cols(host + c * ld_host + r0, len, d->pin[b], len, len, 1);
        CAP_HIP(hipMemcpyAsync(d->data + c * d->ld + r0, d->pin[b], len * 8, hipMemcpyHostToDevice, d->s_copy));
        CAP_HIP(hipEventRecord(d->ev[b], d->s_copy));
      }
  } else {
    for (int64_t c0 = 0, k = 0; c0 < cols; c0 += cpc, k++) {
      const int b = (int)(k & 1); const int64_t nc = std::min(cpc, cols - c0);
      if (k >= 2) CAP_HIP(hipEventSynchronize(d->ev[b]));   // the engine has drained this pinned buffer
      host_copy_cols(host + c0 * ld_host, ld_host, d->pin[b], rows, rows, nc);
      CAP_HIP(hipMemcpy2DAsync(d->data + c0 * d->ld, d->ld * 8, d->pin[b], rows * 8, rows * 8, nc, hipMemcpyHostToDevice, d->s_copy));
      CAP_HIP(hipEventRecord(d->ev[b], d->s_copy));
    }
  }
  CAP_HIP(hipEventRecord(d->ev_done, d->s_copy));
  CAP_HIP(hipStreamWaitEvent(s, d->ev_done, 0));
  return CAP_OK;
}

// HBM -> host.  Blocks until the host buffer is complete (the reference's construct_* return host matrices by value).
int cap_desc_export_host(cap_desc* d, double* host, int64_t ld_host, void* stream) {
  if (!d || !host || ld_host < d->ly) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  const int64_t rows = d->ly, cols = d->lx;
  if (rows == 0 || cols == 0) return CAP_OK;
  if (host_is_pinned(host)) {
    CAP_HIP(hipMemcpy2DAsync(host, ld_host * 8, d->data, d->ld * 8, rows * 8, cols, hipMemcpyDeviceToHost, s));
    CAP_HIP(hipStreamSynchronize(s));
    return CAP_OK;
  }
  CAP_TRY(ensure_staging(d));
  CAP_HIP(hipEventRecord(d->ev_done, s));
  CAP_HIP(hipStreamWaitEvent(d->s_copy, d->ev_done, 0));
  if (rows > d->pin_elems) {
    for (int64_t c = 0; c < cols; c++)
      for (int64_t r0 = 0; r0 < rows; r0 += d->pin_elems) {
        const int64_t len = std::min(d->pin_elems, rows - r0);
        CAP_HIP(hipMemcpyAsync(d->pin[0], d->data + c * d->ld + r0, len * 8, hipMemcpyDeviceToHost, d->s_copy));
        CAP_HIP(hipStreamSynchronize(d->s_copy));
        cap_acc_host(CAP_ACC_R, d->pin[0], len, len, 1);
        memcpy(host + c * ld_host + r0, d->pin[0], (size_t)len * 8);
      }
    return CAP_OK;
  }
  const int64_t cpc = std::max<int64_t>(1, d->pin_elems / rows);
  const int64_t nchunks = cap_ceil_div(cols, cpc);
  auto issue = [&](int64_t k) -> int {
    const int b = (int)(k & 1); const int64_t c0 = k * cpc, nc = std::min(cpc, cols - c0);
    CAP_HIP(hipMemcpy2DAsync(d->pin[b], rows * 8, d->data + c0 * d->ld, d->ld * 8, rows * 8, nc, hipMemcpyDeviceToHost, d->s_copy));
    CAP_HIP(hipEventRecord(d->ev[b], d->s_copy));
    return CAP_OK;
  };
  if (nchunks > 0) CAP_TRY(issue(0));
  for (int64_t k = 0; k < nchunks; k++) {
    if (k + 1 < nchunks) CAP_TRY(issue(k + 1));            // chunk k+1 flies while the host drains chunk k
    const int b = (int)(k & 1); const int64_t c0 = k * cpc, nc = std::min(cpc, cols - c0);
    CAP_HIP(hipEventSynchronize(d->ev[b]));
    host_copy_cols(d->pin[b], rows, host + c0 * ld_host, ld_host, rows, nc);
  }
  return CAP_OK;
}

// The caller holds the GLOBAL matrix in host memory (column-major, ld_host >= global rows - what upstream's matrix<> would be on one
// rank): import picks this process's rows / columns out of it (block-cyclic: nb x nb blocks; element-cyclic: every py-th row of every
// px-th column, position set with cap_desc_set_position), export writes them back to their global places and touches nothing else.
// Same pinned double buffer as the local import / export: a few host threads pack (unpack) the image of a range of local columns
// while the previous range is on the PCIe link.  import is ordered on `stream`, export blocks until the host matrix holds the piece.
int cap_desc_import_host_global(cap_desc* d, const double* host_global, int64_t ld_host, void* stream) {
  if (!d || !host_global || ld_host < d->gy || d->qx < 0 || d->qy < 0) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  const int64_t rows = d->ly, cols = d->lx;
  if (rows == 0 || cols == 0) return CAP_OK;
  if (rows > CHUNK_BYTES / 8) return CAP_ERR_UNSUPPORTED;            // (a local column longer than a chunk: 8 Mi rows)
  CAP_TRY(ensure_staging(d));
  CAP_HIP(hipStreamSynchronize(d->s_copy));
  CAP_HIP(hipEventRecord(d->ev_done, s));
  CAP_HIP(hipStreamWaitEvent(d->s_copy, d->ev_done, 0));
  const int64_t cpc = std::max<int64_t>(1, d->pin_elems / rows);
  for (int64_t c0 = 0, k = 0; c0 < cols; c0 += cpc, k++) {
    const int b = (int)(k & 1); const int64_t nc = std::min(cpc, cols - c0);
    if (k >= 2) CAP_HIP(hipEventSynchronize(d->ev[b]));
    host_pack_cols(d, host_global, ld_host, d->pin[b], c0, nc, true);
    CAP_HIP(hipMemcpy2DAsync(d->data + c0 * d->ld, d->ld * 8, d->pin[b], rows * 8, rows * 8, nc, hipMemcpyHostToDevice, d->s_copy));
    CAP_HIP(hipEventRecord(d->ev[b], d->s_copy));
  }
  CAP_HIP(hipEventRecord(d->ev_done, d->s_copy));
  CAP_HIP(hipStreamWaitEvent(s, d->ev_done, 0));
  return CAP_OK;
}

int cap_desc_export_host_global(cap_desc* d, double* host_global, int64_t ld_host, void* stream) {
  if (!d || !host_global || ld_host < d->gy || d->qx < 0 || d->qy < 0) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  const int64_t rows = d->ly, cols = d->lx;
  if (rows == 0 || cols == 0) return CAP_OK;
  if (rows > CHUNK_BYTES / 8) return CAP_ERR_UNSUPPORTED;
  CAP_TRY(ensure_staging(d));
  CAP_HIP(hipStreamSynchronize(d->s_copy));
  CAP_HIP(hipEventRecord(d->ev_done, s));
  CAP_HIP(hipStreamWaitEvent(d->s_copy, d->ev_done, 0));
  const int64_t cpc = std::max<int64_t>(1, d->pin_elems / rows);
  const int64_t nchunks = cap_ceil_div(cols, cpc);
  auto issue = [&](int64_t k) -> int {
    const int b = (int)(k & 1); const int64_t c0 = k * cpc, nc = std::min(cpc, cols - c0);
    CAP_HIP(hipMemcpy2DAsync(d->pin[b], rows * 8, d->data + c0 * d->ld, d->ld * 8, rows * 8, nc, hipMemcpyDeviceToHost, d->s_copy));
    CAP_HIP(hipEventRecord(d->ev[b], d->s_copy));
    return CAP_OK;
  };
  if (nchunks > 0) CAP_TRY(issue(0));
  for (int64_t k = 0; k < nchunks; k++) {
    if (k + 1 < nchunks) CAP_TRY(issue(k + 1));
    const int b = (int)(k & 1); const int64_t c0 = k * cpc, nc = std::min(cpc, cols - c0);
    CAP_HIP(hipEventSynchronize(d->ev[b]));
    host_pack_cols(d, host_global, ld_host, d->pin[b], c0, nc, false);
  }
  return CAP_OK;
}

}  // extern "C"
