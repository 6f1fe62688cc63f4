// micro-benchmark (round 5): how fast does a launch get its workgroups started? Every workgroup's thread 0 stamps wall_clock64 (100 MHz)
// at its first instruction; the spread last start - first start over the grid, for grids of 48 .. 768 workgroups, 256 / 512 / 1024
// threads, few or many (180) VGPRs, no or 5 KB of LDS. Input to: the decoder's short attention launches (384 workgroups of 256 threads
// at 64 clips) reach their LAST workgroup's first instruction ~6 us after the first one's (profiles/r05_attn_stamps_b64.txt).
//   hipcc -O3 --offload-arch=gfx950 wg_dispatch_rate.cpp -o wg_dispatch_rate && ./wg_dispatch_rate
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
template <int T, bool BIGV, bool LDS>
__global__ __launch_bounds__(T) void k(unsigned long long* st, float* sink) {
  __shared__ float pad[LDS ? 1280 : 1];
  if (threadIdx.x == 0) st[blockIdx.x] = wall_clock64();
  if (BIGV) asm volatile("" ::: "v180");
  if (LDS) pad[threadIdx.x % 1280] = 1.f;
  // ~2 us of dependent work, as a short attention workgroup has
  float v = threadIdx.x;
  for (int i = 0; i < 400; ++i) v = __builtin_fmaf(v, 1.0001f, 0.5f);
  if (v == 12345.f) sink[0] = v + (LDS ? pad[0] : 0.f);
}
template <int T, bool BIGV, bool LDS>
static void run(const char* name, hipStream_t s, unsigned long long* d, float* sink) {
  printf("%-38s", name);
  for (int wg : {48, 96, 192, 256, 384, 512, 768}) {
    double best = 1e9;
    for (int rep = 0; rep < 8; ++rep) {
      hipLaunchKernelGGL((k<T, BIGV, LDS>), dim3(wg), dim3(T), 0, s, d, sink);
      CK(hipStreamSynchronize(s));
      std::vector<unsigned long long> h(wg);
      CK(hipMemcpy(h.data(), d, wg * 8, hipMemcpyDeviceToHost));
      const auto mm = std::minmax_element(h.begin(), h.end());
      best = std::min(best, (double)(*mm.second - *mm.first) * 0.01);
    }
    printf(" %4d: %5.2f", wg, best);
  }
  printf("   (us, last start - first start)\n");
}
int main() {
  unsigned long long* d; float* sink;
  CK(hipMalloc((void**)&d, 1024 * 8)); CK(hipMalloc((void**)&sink, 16));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  run<256, false, false>("256 threads, few VGPRs, no LDS", s, d, sink);
  run<256, true, true>("256 threads, 181 VGPRs, 5 KB LDS", s, d, sink);
  run<512, false, false>("512 threads, few VGPRs, no LDS", s, d, sink);
  run<512, true, true>("512 threads, 181 VGPRs, 5 KB LDS", s, d, sink);
  run<1024, false, false>("1024 threads, few VGPRs, no LDS", s, d, sink);
  return 0;
}
