// micro-benchmark 2: cost of a dependent kernel in a 100-node graph as the body grows
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
struct Big { const float* in; float* out; const int* idx; float pad[56]; int n; };
__global__ void k0(Big p) { if (p.n < 0) p.out[0] = 1; }
__global__ void k1(Big p) { int i = blockIdx.x * 256 + threadIdx.x; p.out[i] = p.in[i] + 1.f; }
__global__ void k2(Big p) { int i = blockIdx.x * 256 + threadIdx.x; int j = p.idx[i]; p.out[i] = p.in[j] + 1.f; }
__global__ void k3(Big p) {
  extern __shared__ float sm[];
  int i = blockIdx.x * 256 + threadIdx.x; int j = p.idx[i];
  sm[threadIdx.x] = p.in[j]; __syncthreads();
  float v = sm[threadIdx.x ^ 1];
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  sm[threadIdx.x] = v; __syncthreads();
  p.out[i] = sm[(threadIdx.x + 64) & 255] + 1.f;
}
__global__ void k4(Big p) {  // 3-level dependent chain
  int i = blockIdx.x * 256 + threadIdx.x; int j = p.idx[i]; int k = p.idx[j]; p.out[i] = p.in[k] + 1.f;
}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const int n = 1 << 20;
  float *a, *b; int* idx; CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&idx, n * 4));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(b, 0, n * 4)); CK(hipMemset(idx, 0, n * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 100, R = 50;
  for (int variant = 0; variant < 5; ++variant) for (int grid : {96, 384}) {
    auto enqueue = [&]() {
      for (int i = 0; i < N; ++i) {
        Big p{}; p.in = (i & 1) ? b : a; p.out = (i & 1) ? a : b; p.idx = idx; p.n = n;
        switch (variant) {
          case 0: hipLaunchKernelGGL(k0, dim3(grid), dim3(256), 0, s, p); break;
          case 1: hipLaunchKernelGGL(k1, dim3(grid), dim3(256), 0, s, p); break;
          case 2: hipLaunchKernelGGL(k2, dim3(grid), dim3(256), 0, s, p); break;
          case 3: hipLaunchKernelGGL(k3, dim3(grid), dim3(256), 3072, s, p); break;
          case 4: hipLaunchKernelGGL(k4, dim3(grid), dim3(256), 0, s, p); break;
        }
      }
    };
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    enqueue();
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("variant %d grid %d: %.3f us/kernel\n", variant, grid, ms * 1e3 / (N * R));
  }
  return 0;
}
