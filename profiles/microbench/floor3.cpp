// micro-benchmark 3: does straight-line code size (cold I-cache at every kernel boundary?) cost time in a dependent chain
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
template <int N, int SALT>
__global__ void kcode(const float* in, float* out) {
  int i = blockIdx.x * 256 + threadIdx.x;
  float v = in[i];
#pragma unroll
  for (int k = 0; k < N; ++k) v = fmaf(v, 1.0001f + (float)(k + SALT) * 1e-7f, 0.5f + (float)k);  // N distinct 12-byte instructions... roughly
  out[i] = v;
}
template <int N>
__global__ void kloop(const float* in, float* out, int n) {
  int i = blockIdx.x * 256 + threadIdx.x;
  float v = in[i];
#pragma unroll 1
  for (int k = 0; k < n; ++k) v = fmaf(v, 1.0001f, 0.5f);
  out[i] = v;
}
template <typename F>
int timeit(const char* name, hipStream_t s, F enqueue) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  enqueue();
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
  CK(hipEventRecord(e0, s));
  for (int r = 0; r < 50; ++r) CK(hipGraphLaunch(ge, s));
  CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  printf("%s: %.3f us/kernel\n", name, ms * 1e3 / (100 * 50));
  return 0;
}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int n = 1 << 20;
  float *a, *b; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4));
  const int grid = 96;
#define RUN1(K, NAME) timeit(NAME, s, [&]() { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((K), dim3(grid), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b); });
  RUN1((kcode<16, 0>), "straight 16 fma, same kernel");
  RUN1((kcode<500, 0>), "straight 500 fma, same kernel");
  RUN1((kcode<2000, 0>), "straight 2000 fma, same kernel");
  timeit("loop 500 fma", s, [&]() { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((kloop<0>), dim3(grid), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 500); });
  timeit("loop 2000 fma", s, [&]() { for (int i = 0; i < 100; ++i) hipLaunchKernelGGL((kloop<0>), dim3(grid), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, 2000); });
  timeit("straight 500 fma, 4 different kernels alternating", s, [&]() {
    for (int i = 0; i < 100; ++i) {
      const float* in = (i & 1) ? b : a; float* out = (i & 1) ? a : b;
      switch (i & 3) {
        case 0: hipLaunchKernelGGL((kcode<500, 1>), dim3(grid), dim3(256), 0, s, in, out); break;
        case 1: hipLaunchKernelGGL((kcode<500, 2>), dim3(grid), dim3(256), 0, s, in, out); break;
        case 2: hipLaunchKernelGGL((kcode<500, 3>), dim3(grid), dim3(256), 0, s, in, out); break;
        case 3: hipLaunchKernelGGL((kcode<500, 4>), dim3(grid), dim3(256), 0, s, in, out); break;
      }
    }
  });
  return 0;
}
