// Probe (not product): how do hipExtStreamCreateWithCUMask bits map onto XCDs / CUs on MI355X?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <set>
#include <map>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n",hipGetErrorString(e),__LINE__); exit(1);} }while(0)
__global__ void who(unsigned* out, int spin) {
  unsigned xcc, hwid;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
  // keep the block alive a little so the grid spreads over every enabled CU
  unsigned long long t0 = wall_clock64();
  while (wall_clock64() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = xcc; out[2 * blockIdx.x + 1] = hwid; }
}
static void run(hipStream_t s, const char* tag) {
  const int nb = 4096;
  unsigned* d; CK(hipMalloc(&d, nb * 8)); CK(hipMemset(d, 0xff, nb * 8));
  who<<<nb, 256, 0, s>>>(d, 2000); CK(hipStreamSynchronize(s));
  std::vector<unsigned> h(2 * nb); CK(hipMemcpy(h.data(), d, nb * 8, hipMemcpyDeviceToHost));
  std::map<unsigned, std::set<unsigned>> cus;
  for (int i = 0; i < nb; i++) {
    unsigned xcc = h[2 * i] & 0xf, hw = h[2 * i + 1];
    unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 0x1, se = (hw >> 13) & 0x7;   // gfx9 HW_ID: CU_ID[11:8] SH_ID[12] SE_ID[15:13]
    cus[xcc].insert((se << 8) | (sh << 4) | cu);
  }
  printf("%s: CUs seen per XCC:", tag);
  int tot = 0;
  for (auto& kv : cus) { printf(" x%u=%zu", kv.first, kv.second.size()); tot += kv.second.size(); }
  printf("  total=%d", tot);
  // is block b still dispatched to XCD b % 8 (what the XCD-aware tile maps assume)?
  int agree = 0; unsigned first[8];
  for (int r = 0; r < 8; r++) first[r] = h[2 * r] & 0xf;
  for (int i = 0; i < nb; i++) agree += ((h[2 * i] & 0xf) == first[i & 7]);
  printf("  blocks on XCD of (b %% 8): %d / %d\n", agree, nb);
  CK(hipFree(d));
}
int main() {
  hipStream_t s0; CK(hipStreamCreate(&s0)); run(s0, "no mask       ");
  auto masked = [&](const char* tag, std::vector<int> off) {
    uint32_t m[8]; memset(m, 0xff, sizeof(m));
    for (int b : off) m[b / 32] &= ~(1u << (b % 32));
    hipStream_t s; CK(hipExtStreamCreateWithCUMask(&s, 8, m)); run(s, tag); CK(hipStreamDestroy(s));
  };
  masked("bits 0-7 off  ", {0,1,2,3,4,5,6,7});
  masked("bits 0-15 off ", {0,1,2,3,4,5,6,7,8,9,10,11,12,13,14,15});
  masked("bits 0,32,..  ", {0,32,64,96,128,160,192,224});
  masked("bits 0,8,16.. ", {0,8,16,24,32,40,48,56});
  masked("bits 248-255  ", {248,249,250,251,252,253,254,255});
  return 0;
}
