// TEST INFRASTRUCTURE - a recording stand-in for the HIP runtime and RCCL entry points libcapital_amd.so imports.
//
// tests/hipshim/build_shim.py links the library's OWN object files (capital_amd/lib/obj/*.o, the ones the product .so is made of)
// against this file instead of libamdhip64 / librccl.  The C++ host side - plan creation, argument checks, index arithmetic, the
// multi-stream schedules with their event edges - then runs on a machine WITHOUT a GPU: "device" memory is zeroed host memory, a
// kernel launch / copy / collective is a line in a trace, nothing is computed.  tests/test_schedule_structure.py replays the trace
// with vector clocks and checks what can be checked without knowing what a kernel touches:
//   * every stream a call put work on is joined into the caller's stream (or waited for by the host) before the call returns;
//   * no wait names an event that was never recorded;
//   * every copy / memset stays inside ONE allocation (host index arithmetic at ragged sizes).
// Access notes (capital_amd/csrc/common.h, cap_access_hook): the library declares, next to every launch, the windows the kernel reads and
// writes; the stand-in attaches them to the launch's trace line ("A" lines: allocation serial number, offset, pitch, row bytes, columns,
// triangle, element size) - as it does itself for copies, memsets and collectives - and trace_check.py looks for two operations that
// touch the same bytes, one of them writing, without being ordered.  A note that leaves its allocation is an out-of-range finding.
// Never linked into the product; nothing here computes anything.
#include <hip/hip_runtime_api.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

namespace {
struct Alloc { size_t bytes; int kind; std::string where; long long id; };   // kind 0 device, 1 pinned host, 2 mapping of a peer's buffer; where: the last MARK before the allocation; id: serial number
// IPC: hipIpcGetMemHandle writes {pointer, magic, peer tag = 0, bytes} into the handle; the test's all-gather callback tags the copy it
// puts into slot q with q + 1 (run_scenarios.py), and hipIpcOpenMemHandle of a TAGGED handle returns a mapping of its own (a fresh
// range of the same size, "ALIAS" line: peer, the exporter's export number, offset) - the joint replay of all ranks' traces
// (trace_check.check_joint) resolves it to the peer's own allocation.  An untagged handle maps to the buffer itself, as before.
constexpr unsigned long long IPC_MAGIC = 0x4c444e4148435049ull;      // "IPCHANDL"
struct IpcHandle { void* ptr; unsigned long long magic; int tag; int export_no; unsigned long long bytes; long long off; };
static_assert(sizeof(IpcHandle) <= sizeof(hipIpcMemHandle_t), "fits the 64-byte handle");
long long next_export = 0;
struct Stream { int id; unsigned flags; bool alive; };
struct Event { int id; };
struct Rec { std::string text; };

std::mutex mu;
std::map<uintptr_t, Alloc> allocs;                   // base -> allocation
std::map<const void*, std::string> kernels;          // host stub -> device name
std::vector<std::string> trace;
std::string last_mark = "(start)";
int next_stream = 1, next_event = 1;
long long total_alloc = 0, oob = 0, next_alloc = 1;
struct Note { int mode; const void* base; long long pitch, row_bytes, cols; int tri, elem; };
thread_local std::vector<Note> pending;                  // access notes waiting for the launch / collective they describe
hipError_t last_error = hipSuccess;
constexpr size_t TOUCH_LIMIT = 1 << 16;              // payloads up to this size are really copied / set (info words, handles); larger ones only traced
// compute mode (shim_set_compute, run_compute.py): every copy and memset is carried out, fresh allocations are filled with NaN patterns
// (device memory is NOT zero after hipMalloc) and every launch runs the kernel's CPU model (kernels_cpu.cpp) at enqueue time
int compute = getenv("SHIM_COMPUTE") ? atoi(getenv("SHIM_COMPUTE")) : 0;      // (a plain C program linked against the stand-in: examples/*.c)
long long unmodelled = 0;

struct PendingCfg { dim3 grid, block; size_t shmem; hipStream_t stream; };
thread_local std::vector<PendingCfg> cfg_stack;

int sid(hipStream_t s) { return s ? reinterpret_cast<Stream*>(s)->id : 0; }
int eid(hipEvent_t e) { return e ? reinterpret_cast<Event*>(e)->id : 0; }

// which allocation does [p, p + bytes) fall into?  0 = not a tracked pointer (plain host memory), 1 = inside one, -1 = crosses its end
int range_check(const void* p, size_t bytes, int* kind = nullptr) {
  const uintptr_t a = (uintptr_t)p;
  auto it = allocs.upper_bound(a);
  if (it == allocs.begin()) return 0;
  --it;
  if (a >= it->first + it->second.bytes) return 0;
  if (kind) *kind = it->second.kind;
  return a + bytes <= it->first + it->second.bytes ? 1 : -1;
}
void note(const char* fmt, ...) __attribute__((format(printf, 1, 2)));
void note(const char* fmt, ...) {
  char buf[1024];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof(buf), fmt, ap); va_end(ap);
  trace.emplace_back(buf);
}
void check_range(const char* what, const void* p, size_t bytes) {
  if (bytes == 0) return;
  if (range_check(p, bytes) < 0) { oob++; note("OOB %s %p %zu", what, p, bytes); }
}
// one access line behind the operation it belongs to; memory the stand-in did not allocate (plain host arrays of a caller) is skipped
void access_line(int mode, const void* base, long long pitch, long long row_bytes, long long cols, int tri, int elem, const char* what) {
  if (mode == 0 || !base || row_bytes <= 0 || cols <= 0) return;
  const uintptr_t a = (uintptr_t)base;
  auto it = allocs.upper_bound(a);
  if (it == allocs.begin()) return;
  --it;
  if (a >= it->first + it->second.bytes) return;
  const unsigned long long span = (unsigned long long)(cols - 1) * (unsigned long long)(pitch > 0 ? pitch : 0) + (unsigned long long)row_bytes;
  if (a + span > it->first + it->second.bytes) { oob++; note("OOB access of %s: %lld columns of %lld bytes, pitch %lld, leave the allocation of %zu bytes by %llu", what, cols, row_bytes, pitch, it->second.bytes, (unsigned long long)(a + span - it->first - it->second.bytes)); }
  note("A %d %lld %llu %lld %lld %lld %d %d", mode, it->second.id, (unsigned long long)(a - it->first), pitch, row_bytes, cols, tri, elem);
}
void flush_pending(const char* what) {
  for (const Note& n : pending) access_line(n.mode, n.base, n.pitch, n.row_bytes, n.cols, n.tri, n.elem, what);
  pending.clear();
}
void orphan_check(const char* what) {
  if (!pending.empty()) { note("ORPHAN %zu access notes in front of %s", pending.size(), what); pending.clear(); }
}
}  // namespace

extern "C" int shim_cpu_kernel(const char* mangled, void** args, unsigned gx, unsigned gy, unsigned gz, unsigned bx);      // kernels_cpu.cpp

extern "C" {

// ---------------------------------------------------------------- the trace, for the test
void shim_reset() { std::lock_guard<std::mutex> lk(mu); trace.clear(); oob = 0; next_export = 0; }
void shim_mark(const char* what) { std::lock_guard<std::mutex> lk(mu); note("MARK %s", what); last_mark = what; }
long long shim_oob() { return oob; }
long long shim_live_allocations() { std::lock_guard<std::mutex> lk(mu); return (long long)allocs.size(); }
// sizes of the live allocations, for leak hunting: writes up to cap entries, returns the number of live allocations
long long shim_live_sizes(long long* out, long long cap) {
  std::lock_guard<std::mutex> lk(mu);
  long long i = 0;
  for (const auto& a : allocs) { if (i < cap) out[i] = (long long)a.second.bytes; i++; }
  return i;
}
int shim_live_report(const char* path) {
  std::lock_guard<std::mutex> lk(mu);
  FILE* f = fopen(path, "w");
  if (!f) return 1;
  for (const auto& a : allocs) fprintf(f, "%zu bytes, allocated after: %s\n", a.second.bytes, a.second.where.c_str());
  fclose(f);
  return 0;
}
int shim_dump(const char* path) {
  std::lock_guard<std::mutex> lk(mu);
  FILE* f = fopen(path, "w");
  if (!f) return 1;
  for (const auto& l : trace) fprintf(f, "%s\n", l.c_str());
  fclose(f);
  return 0;
}
void shim_set_compute(int on) { compute = on; }
long long shim_unmodelled() { return unmodelled; }
// an op of the test's own making on a stream (the stand-in collectives of a callback communicator); its access notes come first
void shim_note_op(const char* name, void* stream) { std::lock_guard<std::mutex> lk(mu); note("OP %d %s", sid((hipStream_t)stream), name); flush_pending(name); }
// the library's access hook (cap_access_hook) and the tests' own way to describe a collective: the NEXT launch / op touches this window
void shim_access_note(int mode, const void* base, long long pitch, long long row_bytes, long long cols, int tri, int elem) {
  if (mode & 16) {       // the host itself touches the window, now ("HA" line: an access of the host thread, not of a stream operation)
    std::lock_guard<std::mutex> lk(mu);
    const size_t before = trace.size();
    access_line(mode & 15, base, pitch, row_bytes, cols, tri, elem, "host access");
    for (size_t i = before; i < trace.size(); i++) if (trace[i].compare(0, 2, "A ") == 0) trace[i] = "H" + trace[i];
    return;
  }
  pending.push_back(Note{mode, base, pitch, row_bytes, cols, tri, elem});
}

// ---------------------------------------------------------------- kernel registration / launch
void** __hipRegisterFatBinary(const void*) { static void* handle[1]; return handle; }
void __hipUnregisterFatBinary(void**) {}
void __hipRegisterFunction(void**, const void* hostFunction, char*, const char* deviceName, unsigned, void*, void*, void*, void*, int*) {
  std::lock_guard<std::mutex> lk(mu);
  kernels[hostFunction] = deviceName ? deviceName : "?";
}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t stream) {
  cfg_stack.push_back(PendingCfg{grid, block, shmem, stream});
  return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3* grid, dim3* block, size_t* shmem, hipStream_t* stream) {
  if (cfg_stack.empty()) return hipErrorInvalidValue;
  const PendingCfg c = cfg_stack.back(); cfg_stack.pop_back();
  *grid = c.grid; *block = c.block; *shmem = c.shmem; *stream = c.stream;
  return hipSuccess;
}
hipError_t hipLaunchKernel(const void* f, dim3 grid, dim3 block, void** args, size_t shmem, hipStream_t s) {
  std::lock_guard<std::mutex> lk(mu);
  auto it = kernels.find(f);
  // a launch the real runtime would refuse: empty grid / block, more than 64 KiB of LDS without the attribute is NOT modelled
  if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024) {
    note("BADLAUNCH %d %s grid %u %u %u block %u %u %u", sid(s), it == kernels.end() ? "?" : it->second.c_str(), grid.x, grid.y, grid.z, block.x, block.y, block.z);
    last_error = hipErrorInvalidConfiguration;
    return last_error;
  }
  note("K %d %s %u %u %u %zu %zu", sid(s), it == kernels.end() ? "?" : it->second.c_str(), grid.x, grid.y, grid.z, shmem, pending.size());
  flush_pending(it == kernels.end() ? "?" : it->second.c_str());
  if (compute && it != kernels.end() && !shim_cpu_kernel(it->second.c_str(), args, grid.x, grid.y, grid.z, block.x)) {
    unmodelled++; note("UNMODELLED %s", it->second.c_str());
  }
  return hipSuccess;
}
hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }
hipError_t hipOccupancyMaxActiveBlocksPerMultiprocessor(int* n, const void*, int, size_t) { *n = 2; return hipSuccess; }

// ---------------------------------------------------------------- device
// SHIM_DEVICES=N: N visible devices (default 1); the current device is per thread, as in HIP
static int shim_device_count() { static const int n = getenv("SHIM_DEVICES") ? atoi(getenv("SHIM_DEVICES")) : 1; return n < 1 ? 1 : n; }
static thread_local int current_device = 0;
hipError_t hipGetDevice(int* d) { *d = current_device; return hipSuccess; }
hipError_t hipSetDevice(int d) { if (d < 0 || d >= shim_device_count()) return hipErrorInvalidDevice; current_device = d; return hipSuccess; }
hipError_t hipGetDeviceCount(int* n) { *n = shim_device_count(); return hipSuccess; }
hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t a, int) {
  *v = a == hipDeviceAttributeMultiprocessorCount ? 256 : 0;
  return hipSuccess;
}
hipError_t hipDeviceGetStreamPriorityRange(int* lo, int* hi) { *lo = 0; *hi = -1; return hipSuccess; }
hipError_t hipGetDevicePropertiesR0600(hipDeviceProp_tR0600* p, int) {
  memset(p, 0, sizeof(*p));
  snprintf(p->name, sizeof(p->name), "recording stand-in");
  snprintf(p->gcnArchName, sizeof(p->gcnArchName), "gfx950");
  p->multiProcessorCount = 256; p->totalGlobalMem = (size_t)288 << 30;
  return hipSuccess;
}
const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : "error (stand-in)"; }
hipError_t hipGetLastError() { const hipError_t e = last_error; last_error = hipSuccess; return e; }

// ---------------------------------------------------------------- memory
static hipError_t alloc_(void** p, size_t bytes, int kind) {
  if (!p) return hipErrorInvalidValue;
  const size_t b = bytes ? bytes : 1;
  void* q = nullptr;
  // untouched pages of a calloc-sized mapping cost nothing: N = 65536 plans fit a small box as long as nobody writes the payload
  q = calloc((b + 255) / 256, 256);
  if (!q) { last_error = hipErrorOutOfMemory; return last_error; }
  if (compute) memset(q, 0xff, b);
  std::lock_guard<std::mutex> lk(mu);
  allocs[(uintptr_t)q] = Alloc{b, kind, last_mark, next_alloc++};
  total_alloc += (long long)b;
  *p = q;
  return hipSuccess;
}
static hipError_t free_(void* p) {
  if (!p) return hipSuccess;
  std::lock_guard<std::mutex> lk(mu);
  auto it = allocs.find((uintptr_t)p);
  if (it == allocs.end()) { note("BADFREE %p", p); last_error = hipErrorInvalidValue; return last_error; }
  note("FREE %lld", it->second.id);
  allocs.erase(it);
  free(p);
  return hipSuccess;
}
hipError_t hipMalloc(void** p, size_t bytes) { return alloc_(p, bytes, 0); }
hipError_t hipMallocAsync(void** p, size_t bytes, hipStream_t) { return alloc_(p, bytes, 0); }
hipError_t hipHostMalloc(void** p, size_t bytes, unsigned) { return alloc_(p, bytes, 1); }
hipError_t hipFree(void* p) { { std::lock_guard<std::mutex> lk(mu); note("HOSTSYNC device (hipFree)"); } return free_(p); }
hipError_t hipFreeAsync(void* p, hipStream_t s) {
  if (p) {     // stream-ordered: the allocation is "written" by the free - whoever still touches it must be ordered in front of this point
    std::lock_guard<std::mutex> lk(mu);
    orphan_check("hipFreeAsync");
    auto it = allocs.find((uintptr_t)p);
    if (it != allocs.end()) { note("OP %d hipFreeAsync", sid(s)); access_line(2, p, 0, (long long)it->second.bytes, 1, 0, 1, "hipFreeAsync"); }
  }
  return free_(p);
}
hipError_t hipHostFree(void* p) { return free_(p); }
hipError_t hipPointerGetAttributes(hipPointerAttribute_t* a, const void* p) {
  std::lock_guard<std::mutex> lk(mu);
  int kind = 0;
  if (range_check(p, 1, &kind) == 0) { last_error = hipErrorInvalidValue; return last_error; }
  memset(a, 0, sizeof(*a));
  a->type = kind == 1 ? hipMemoryTypeHost : hipMemoryTypeDevice;
  a->isManaged = 0;
  return hipSuccess;
}
static void copy_(void* dst, const void* src, size_t bytes, int stream, const char* what) {
  std::lock_guard<std::mutex> lk(mu);
  orphan_check(what);
  check_range("copy dst", dst, bytes); check_range("copy src", src, bytes);
  note("%s %d %p %p %zu", what, stream, dst, src, bytes);
  if (bytes && range_check(dst, bytes) > 0) access_line(2, dst, 0, (long long)bytes, 1, 0, 1, what);
  if (bytes && range_check(src, bytes) > 0) access_line(1, src, 0, (long long)bytes, 1, 0, 1, what);
  if (bytes && (bytes <= TOUCH_LIMIT || compute) && range_check(dst, bytes) >= 0 && range_check(src, bytes) >= 0) memmove(dst, src, bytes);
}
hipError_t hipMemcpyAsync(void* dst, const void* src, size_t bytes, hipMemcpyKind, hipStream_t s) { copy_(dst, src, bytes, sid(s), "COPY"); return hipSuccess; }
hipError_t hipMemcpy(void* dst, const void* src, size_t bytes, hipMemcpyKind) {
  copy_(dst, src, bytes, 0, "COPY");
  std::lock_guard<std::mutex> lk(mu); note("HOSTSYNC stream 0");
  return hipSuccess;
}
hipError_t hipMemcpy2DAsync(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind, hipStream_t s) {
  std::lock_guard<std::mutex> lk(mu);
  if (width && height) {
    if (width > dpitch || width > spitch) { oob++; note("OOB copy2d width %zu over pitch %zu / %zu", width, dpitch, spitch); }
    check_range("copy2d dst", dst, (height - 1) * dpitch + width); check_range("copy2d src", src, (height - 1) * spitch + width);
  }
  orphan_check("COPY2D");
  note("COPY2D %d %p %p %zu %zu", sid(s), dst, src, width, height);
  if (width && height) {
    if (range_check(dst, (height - 1) * dpitch + width) > 0) access_line(2, dst, (long long)dpitch, (long long)width, (long long)height, 0, 1, "COPY2D");
    if (range_check(src, (height - 1) * spitch + width) > 0) access_line(1, src, (long long)spitch, (long long)width, (long long)height, 0, 1, "COPY2D");
  }
  if (width * height && (width * height <= TOUCH_LIMIT || compute) && range_check(dst, (height - 1) * dpitch + width) >= 0 && range_check(src, (height - 1) * spitch + width) >= 0)
    for (size_t r = 0; r < height; r++) memmove((char*)dst + r * dpitch, (const char*)src + r * spitch, width);
  return hipSuccess;
}
hipError_t hipMemcpy2D(void* dst, size_t dpitch, const void* src, size_t spitch, size_t width, size_t height, hipMemcpyKind k) {
  (void)hipMemcpy2DAsync(dst, dpitch, src, spitch, width, height, k, nullptr);
  std::lock_guard<std::mutex> lk(mu); note("HOSTSYNC stream 0");
  return hipSuccess;
}
static void set_(void* p, int v, size_t bytes, int stream) {
  std::lock_guard<std::mutex> lk(mu);
  orphan_check("memset");
  check_range("memset", p, bytes);
  note("SET %d %p %zu", stream, p, bytes);
  if (bytes && range_check(p, bytes) > 0) access_line(2, p, 0, (long long)bytes, 1, 0, 1, "memset");
  if (bytes && (bytes <= TOUCH_LIMIT || compute) && range_check(p, bytes) >= 0) memset(p, v, bytes);
}
hipError_t hipMemsetAsync(void* p, int v, size_t bytes, hipStream_t s) { set_(p, v, bytes, sid(s)); return hipSuccess; }
// (HIP's synchronous memset waits for its own fill command - unlike CUDA's, which may return early on device memory)
hipError_t hipMemset(void* p, int v, size_t bytes) {
  set_(p, v, bytes, 0);
  std::lock_guard<std::mutex> lk(mu); note("HOSTSYNC stream 0");
  return hipSuccess;
}
hipError_t hipIpcGetMemHandle(hipIpcMemHandle_t* h, void* p) {
  memset(h, 0, sizeof(*h));
  std::lock_guard<std::mutex> lk(mu);
  const uintptr_t a = (uintptr_t)p;
  auto it = allocs.upper_bound(a);
  if (it == allocs.begin()) { last_error = hipErrorInvalidValue; return last_error; }
  --it;
  if (a >= it->first + it->second.bytes) { last_error = hipErrorInvalidValue; return last_error; }
  IpcHandle q{p, IPC_MAGIC, 0, (int)next_export++, (unsigned long long)it->second.bytes, (long long)(a - it->first)};
  memcpy(h, &q, sizeof(q));
  note("IPCGET %d %lld %lld", q.export_no, it->second.id, q.off);
  return hipSuccess;
}
hipError_t hipIpcOpenMemHandle(void** p, hipIpcMemHandle_t h, unsigned) {
  IpcHandle q; memcpy(&q, &h, sizeof(q));
  if (!q.ptr) return hipErrorInvalidValue;
  if (q.magic != IPC_MAGIC || q.tag == 0) { *p = q.ptr; return hipSuccess; }       // untagged: the buffer itself
  void* m = calloc((q.bytes - q.off + 255) / 256 + 1, 256);
  if (!m) return hipErrorOutOfMemory;
  std::lock_guard<std::mutex> lk(mu);
  allocs[(uintptr_t)m] = Alloc{(size_t)(q.bytes - q.off), 2, last_mark, next_alloc++};
  note("ALIAS %lld %d %d %lld", allocs[(uintptr_t)m].id, q.tag - 1, q.export_no, q.off);
  *p = m;
  return hipSuccess;
}
hipError_t hipIpcCloseMemHandle(void* p) {
  std::lock_guard<std::mutex> lk(mu);
  auto it = allocs.find((uintptr_t)p);
  if (it != allocs.end() && it->second.kind == 2) { note("FREE %lld", it->second.id); allocs.erase(it); free(p); }
  return hipSuccess;
}

// ---------------------------------------------------------------- streams and events
static hipError_t stream_(hipStream_t* s, unsigned flags, const char* how) {
  if (!s) return hipErrorInvalidValue;
  std::lock_guard<std::mutex> lk(mu);
  Stream* q = new Stream{next_stream++, flags, true};
  note("STREAM %d %s %s", q->id, (flags & hipStreamNonBlocking) ? "nonblocking" : "blocking", how);
  *s = reinterpret_cast<hipStream_t>(q);
  return hipSuccess;
}
hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned flags) { return stream_(s, flags, "flags"); }
hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned flags, int) { return stream_(s, flags, "priority"); }
hipError_t hipExtStreamCreateWithCUMask(hipStream_t* s, uint32_t, const uint32_t*) { return stream_(s, 0, "cumask"); }
hipError_t hipStreamDestroy(hipStream_t s) {
  std::lock_guard<std::mutex> lk(mu);
  if (!s || !reinterpret_cast<Stream*>(s)->alive) { note("BADSTREAMDESTROY"); last_error = hipErrorInvalidHandle; return last_error; }
  reinterpret_cast<Stream*>(s)->alive = false;      // (kept allocated: a later use of the handle is reported, not a crash)
  note("STREAMDESTROY %d", sid(s));
  return hipSuccess;
}
hipError_t hipStreamSynchronize(hipStream_t s) { std::lock_guard<std::mutex> lk(mu); note("HOSTSYNC stream %d", sid(s)); return hipSuccess; }
hipError_t hipDeviceSynchronize() { std::lock_guard<std::mutex> lk(mu); note("HOSTSYNC device"); return hipSuccess; }
static hipError_t event_(hipEvent_t* e) {
  if (!e) return hipErrorInvalidValue;
  std::lock_guard<std::mutex> lk(mu);
  *e = reinterpret_cast<hipEvent_t>(new Event{next_event++});
  return hipSuccess;
}
hipError_t hipEventCreate(hipEvent_t* e) { return event_(e); }
hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return event_(e); }
hipError_t hipEventDestroy(hipEvent_t e) { std::lock_guard<std::mutex> lk(mu); note("EVENTDESTROY %d", eid(e)); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { std::lock_guard<std::mutex> lk(mu); orphan_check("hipEventRecord"); note("RECORD %d %d", sid(s), eid(e)); return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned) { std::lock_guard<std::mutex> lk(mu); orphan_check("hipStreamWaitEvent"); note("WAIT %d %d", sid(s), eid(e)); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { std::lock_guard<std::mutex> lk(mu); note("HOSTSYNC event %d", eid(e)); return hipSuccess; }
// (the stand-in runs everything at once, so a query always finds the event complete - and a successful query IS host knowledge of that: what the
//  host enqueues afterwards is ordered behind the event, exactly like after hipEventSynchronize)
hipError_t hipEventQuery(hipEvent_t e) { std::lock_guard<std::mutex> lk(mu); note("HOSTSYNC event %d", eid(e)); return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 1.0f; return hipSuccess; }

// ---------------------------------------------------------------- RCCL: one-rank communicators only; every collective is a traced op
struct ShimComm { int rank, size; };
typedef ShimComm* ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
int ncclGetUniqueId(ncclUniqueId* id) { memset(id, 7, sizeof(*id)); return 0; }
int ncclCommInitRank(ncclComm_t* c, int size, ncclUniqueId, int rank) { *c = new ShimComm{rank, size}; return 0; }
int ncclCommSplit(ncclComm_t c, int color, int, ncclComm_t* out, void*) { *out = color < 0 ? nullptr : new ShimComm{0, 1}; (void)c; return 0; }
int ncclCommDestroy(ncclComm_t c) { delete c; return 0; }
int ncclCommCount(ncclComm_t c, int* n) { *n = c->size; return 0; }
int ncclCommUserRank(ncclComm_t c, int* r) { *r = c->rank; return 0; }
int ncclCommCuDevice(ncclComm_t, int* d) { *d = 0; return 0; }
const char* ncclGetErrorString(int) { return "stand-in"; }
int ncclGroupStart() { return 0; }
int ncclGroupEnd() { return 0; }
#define SHIM_COLL(name) { std::lock_guard<std::mutex> lk(mu); note("OP %d " name, sid(s)); return 0; }
int ncclAllGather(const void*, void*, size_t, int, ncclComm_t, hipStream_t s) SHIM_COLL("ncclAllGather")
int ncclAllReduce(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t s) SHIM_COLL("ncclAllReduce")
int ncclBroadcast(const void*, void*, size_t, int, int, ncclComm_t, hipStream_t s) SHIM_COLL("ncclBroadcast")
int ncclReduce(const void*, void*, size_t, int, int, int, ncclComm_t, hipStream_t s) SHIM_COLL("ncclReduce")
int ncclSend(const void*, size_t, int, int, ncclComm_t, hipStream_t s) SHIM_COLL("ncclSend")
int ncclRecv(void*, size_t, int, int, ncclComm_t, hipStream_t s) SHIM_COLL("ncclRecv")

}  // extern "C"
