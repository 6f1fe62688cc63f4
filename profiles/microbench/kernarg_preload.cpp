// micro-benchmark (round 5): what does a dependent launch of a SHORT kernel pay for reading its kernel arguments, and does the
// gfx940+ kernarg preload (arguments delivered in SGPRs at wave launch: -mllvm -amdgpu-kernarg-preload-count=N, scalar leading
// arguments only — a by-value struct is never preloaded) take it away? A chain of 200 dependent launches in a hipGraph; each kernel
// (48 workgroups x 512 threads, like an N = 768 clip-block GEMM of the decoder) loads 16 bytes per thread through a pointer
// argument, adds, stores. Variant S: one by-value struct of 25 pointers / ints (the engine's style). Variant P: the pointers it
// uses as leading scalar arguments + the struct. Build both ways:
//   hipcc -O3 --offload-arch=gfx950 kernarg_preload.cpp -o kernarg_plain
//   hipcc -O3 --offload-arch=gfx950 -mllvm -amdgpu-kernarg-preload-count=8 kernarg_preload.cpp -o kernarg_preload
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
struct Params { const float4* in; float4* out; const float* bias; int n, k, batch; const void* pad[20]; };
__global__ __launch_bounds__(512) void k_struct(Params p) {
  const int i = blockIdx.x * 512 + threadIdx.x;
  float4 v = p.in[i];
  const float b = p.bias[threadIdx.x & 15];
  v.x += b; v.y += 1.f;
  p.out[i] = v;
}
__global__ __launch_bounds__(512) void k_scalar(const float4* in, float4* out, const float* bias, int n, Params p) {
  const int i = blockIdx.x * 512 + threadIdx.x;
  float4 v = in[i];
  const float b = bias[threadIdx.x & 15];
  v.x += b; v.y += 1.f;
  out[i] = v;
}
int main() {
  const int WG = 48, N = WG * 512, CHAIN = 200;
  float4 *a, *b; float* bias;
  CK(hipMalloc((void**)&a, N * 16)); CK(hipMalloc((void**)&b, N * 16)); CK(hipMalloc((void**)&bias, 64));
  CK(hipMemset(a, 0, N * 16)); CK(hipMemset(b, 0, N * 16)); CK(hipMemset(bias, 0, 64));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  for (int variant = 0; variant < 2; ++variant) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < CHAIN; ++i) {
      Params p{}; p.in = (i & 1) ? b : a; p.out = (i & 1) ? a : b; p.bias = bias; p.n = N;
      if (variant == 0) hipLaunchKernelGGL(k_struct, dim3(WG), dim3(512), 0, s, p);
      else hipLaunchKernelGGL(k_scalar, dim3(WG), dim3(512), 0, s, p.in, p.out, p.bias, N, p);
    }
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f;
    for (int rep = 0; rep < 6; ++rep) {
      CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
      CK(hipEventRecord(e0, s));
      for (int r = 0; r < 5; ++r) CK(hipGraphLaunch(ge, s));
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (ms < best) best = ms;
    }
    printf("%s: %.3f us per dependent launch (chain of %d, 48 workgroups x 512 threads)\n", variant == 0 ? "by-value struct       " : "leading scalar args   ",
           best * 1e3 / (5 * CHAIN), CHAIN);
  }
  return 0;
}
