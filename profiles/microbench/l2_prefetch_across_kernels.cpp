// micro-benchmark (round 5): does a weight slice that kernel A pulls into an XCD's L2 still sit there when the NEXT kernel of the
// stream starts, and what does the first dependent load of that next kernel cost cold (HBM), from the memory-side cache (the
// line was read a kernel ago by ANOTHER XCD) and from its own XCD's L2? Input to: should a decode GEMM launch prefetch the next
// launch's weight rows (the batched decoder step is a chain of ~87 dependent launches of ~6.5 us, each of which starts with one
// memory round trip for its weight rows).
// Kernel B: 192 workgroups x 512 threads, workgroup i reads slice i (24 KB: three 16-byte loads per thread), thread 0 records
// wall_clock64 ticks (100 MHz) from before the first load is issued to after the last has landed. Kernel A: same grid, workgroup
// i touches every 128-byte line of slice (i + shift) with one dword load. Between trials a scrub kernel streams 2 GB.
//   hipcc -O3 --offload-arch=gfx950 l2_prefetch_across_kernels.cpp -o l2_prefetch_across_kernels && ./l2_prefetch_across_kernels
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

constexpr int WG = 192, T = 512, SLICE = 3 * T * 16;  // 24 KB per workgroup
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(T) void consume(const char* w, long long* ticks, unsigned* sink) {
  const u32x4* p = reinterpret_cast<const u32x4*>(w + (long)blockIdx.x * SLICE) + threadIdx.x;
  __syncthreads();
  const long long t0 = wall_clock64();
  u32x4 a = p[0], b = p[T], c = p[2 * T];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  unsigned s = a[0] ^ b[1] ^ c[2];
  asm volatile("" : "+v"(s));
  __syncthreads();
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0) ticks[blockIdx.x] = t1 - t0;
  if (s == 0x12345678u) sink[0] = s;
}
__global__ __launch_bounds__(T) void touch(const char* w, int shift, unsigned* sink) {
  const int j = (blockIdx.x + shift) % WG;
  const unsigned* p = reinterpret_cast<const unsigned*>(w + (long)j * SLICE);
  unsigned s = 0;
  for (int line = threadIdx.x; line < SLICE / 128; line += T) s ^= p[line * 32];
  if (s == 0x12345678u) sink[0] = s;
}
__global__ void scrub(const u32x4* p, long n, unsigned* sink) {
  unsigned s = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) s ^= p[i][0];
  if (s == 0x12345678u) sink[0] = s;
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  char *w, *big; long long* ticks; unsigned* sink;
  const long big_bytes = 2L << 30;
  CK(hipMalloc((void**)&w, (size_t)WG * SLICE)); CK(hipMalloc((void**)&big, big_bytes)); CK(hipMalloc((void**)&ticks, WG * 8)); CK(hipMalloc((void**)&sink, 16));
  CK(hipMemset(w, 1, (size_t)WG * SLICE)); CK(hipMemset(big, 2, big_bytes));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto trial = [&](const char* name, int mode, int shift) {
    std::vector<double> med;
    for (int rep = 0; rep < 7; ++rep) {
      hipLaunchKernelGGL(scrub, dim3(2048), dim3(256), 0, s, reinterpret_cast<const u32x4*>(big), big_bytes / 16, sink);
      if (mode == 1) hipLaunchKernelGGL(touch, dim3(WG), dim3(T), 0, s, w, shift, sink);
      if (mode == 2) {  // a kernel in between that does not touch the slices (is it the boundary or the time that evicts?)
        hipLaunchKernelGGL(touch, dim3(WG), dim3(T), 0, s, w, shift, sink);
        hipLaunchKernelGGL(scrub, dim3(256), dim3(256), 0, s, reinterpret_cast<const u32x4*>(big), (long)(4 << 20) / 16, sink);
      }
      hipLaunchKernelGGL(consume, dim3(WG), dim3(T), 0, s, w, ticks, sink);
      CK(hipStreamSynchronize(s));
      std::vector<long long> h(WG); CK(hipMemcpy(h.data(), ticks, WG * 8, hipMemcpyDeviceToHost));
      std::sort(h.begin(), h.end());
      med.push_back(h[WG / 2] * 0.01);
      if (rep == 6) printf("%-64s median %.2f us   (min %.2f, max %.2f over workgroups; medians of 7 runs: ", name, h[WG / 2] * 0.01, h[0] * 0.01, h[WG - 1] * 0.01);
    }
    std::sort(med.begin(), med.end());
    printf("%.2f .. %.2f)\n", med.front(), med.back());
  };
  printf("%s: first-load latency of a 24 KB weight slice per workgroup (192 workgroups)\n", pr.gcnArchName);
  trial("cold (2 GB streamed since the slices were last read)", 0, 0);
  trial("touched by the previous kernel, same workgroup index (same XCD)", 1, 0);
  trial("touched by the previous kernel, index + 1 (another XCD)", 1, 1);
  trial("touched by the previous kernel, index + 8 (same XCD, another CU)", 1, 8);
  trial("touched two kernels ago (a 4 MB kernel in between), same index", 2, 0);
  return 0;
}
