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
  int64_t px, py;        // process grid
  int64_t lx, ly, ld;    // local columns / rows, leading dimension (even: 16-byte aligned columns)
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
}  // namespace

extern "C" {

static int desc_new(cap_desc** out, int64_t gx, int64_t gy, int64_t px, int64_t py, double* data, int64_t ld) {
  if (!out || gx <= 0 || gy <= 0 || px <= 0 || py <= 0) return CAP_ERR_ARG;
  cap_desc* d = new (std::nothrow) cap_desc();
  if (!d) return CAP_ERR_ALLOC;
  memset(d, 0, sizeof(*d));
  d->gx = gx; d->gy = gy; d->px = px; d->py = py;
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
    (void)hipStreamDestroy(d->s_copy);
  }
  if (d->owns && d->data) (void)hipFree(d->data);
  delete d;
  return CAP_OK;
}

double* cap_desc_data(cap_desc* d) { return d ? d->data : nullptr; }
// field: 0 global cols, 1 global rows, 2 local cols, 3 local rows, 4 ld, 5 owns, 6 grid x, 7 grid y, 8 num_elems (local rows * cols)
int64_t cap_desc_get(const cap_desc* d, int field) {
  if (!d) return -1;
  const int64_t v[9] = {d->gx, d->gy, d->lx, d->ly, d->ld, d->owns ? 1 : 0, d->px, d->py, d->lx * d->ly};
  return (field >= 0 && field < 9) ? v[field] : -1;
}

// host (column-major local piece, ld_host >= local rows) -> HBM.  Ordered behind `stream`'s earlier work on the
// descriptor; returns after the last chunk has been handed to the copy engine and makes `stream` wait for it.
int cap_desc_import_host(cap_desc* d, const double* host, int64_t ld_host, void* stream) {
  if (!d || !host || ld_host < d->ly) return CAP_ERR_ARG;
  hipStream_t s = cap_stream(stream);
  const int64_t rows = d->ly, cols = d->lx;
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

}  // extern "C"
