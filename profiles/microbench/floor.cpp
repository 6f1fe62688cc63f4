// micro-benchmark: per-kernel cost of a chain of dependent launches (eager vs hipGraph), trivial and "one dependent load" bodies
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__global__ void k_empty(int* p) { if (p == nullptr) p[0] = 1; }
__global__ void k_chain(const int* __restrict__ in, int* __restrict__ out, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + 1;
}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  int *a, *b; CK(hipMalloc(&a, 1 << 20)); CK(hipMalloc(&b, 1 << 20)); CK(hipMemset(a, 0, 1 << 20)); CK(hipMemset(b, 0, 1 << 20));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 100, R = 50;
  for (int variant = 0; variant < 4; ++variant) {
    int grid = (variant & 1) ? 256 : 1;
    bool chain = variant >= 2;
    auto enqueue = [&]() {
      for (int i = 0; i < N; ++i) {
        if (chain) hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, s, (i & 1) ? b : a, (i & 1) ? a : b, grid * 256);
        else hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s, a);
      }
    };
    // eager
    enqueue(); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < R; ++r) enqueue();
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("variant grid=%d chain=%d eager: %.3f us/kernel\n", grid, (int)chain, ms * 1e3 / (N * R));
    // graph
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    enqueue();
    CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(e0, s));
    for (int r = 0; r < R; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("variant grid=%d chain=%d graph: %.3f us/kernel\n", grid, (int)chain, ms * 1e3 / (N * R));
  }
  return 0;
}
