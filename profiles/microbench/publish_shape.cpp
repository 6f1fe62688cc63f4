// micro-benchmark 6: how does the shape of the PUBLISH side change an all-to-all hand-off of 768 granules to 256
// polling workgroups? Same chain as allgather_phase.cpp (gather -> trivial compute -> publish), 8 waves per workgroup;
// only `nprod` workgroups publish, each `768 / nprod` rows, either one store instruction per wave (2 adjacent lanes,
// 16 B: the persistent decoder's rows) or ONE wave storing all the workgroup's rows contiguously (values passed
// through LDS first). Prints microseconds per phase and the gather share.
//   hipcc -O3 --offload-arch=gfx950 publish_shape.cpp -o publish_shape && ./publish_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);}}while(0)
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
constexpr int D = 768, WAVES = 8, NT = WAVES * 64;

__global__ __launch_bounds__(NT) void chain(u64* bufs, int nphase, int nprod, int one_store, unsigned* tmo, long long* stamps) {
  __shared__ float vec[D];
  __shared__ float outv[64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wg = blockIdx.x;
  const int rows = D / nprod;  // per producer (<= 64)
  long long tg = 0;
  for (int p = 0; p < nphase; ++p) {
    const unsigned epoch = p + 1;
    gu64* in = (gu64*)(bufs + (long)(p & 1) * D);
    const long long t0 = wall_clock64();
    int failed = 0;
    for (int i = tid; i < D; i += NT) {
      u64 x; unsigned spins = 0;
      while (true) {
        x = __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if ((unsigned)(x >> 32) == epoch) break;
        if (++spins > 4000000u) { *tmo = 1; failed = 1; break; }
      }
      vec[i] = __uint_as_float((unsigned)x);
    }
    if (__syncthreads_or(failed)) return;
    tg += wall_clock64() - t0;
    gu64* out = (gu64*)(bufs + (long)((p + 1) & 1) * D);
    if (wg < nprod) {
      const int r0 = wg * rows;
      // every row: a cheap function of the whole vector (one lane per row here; the timing of interest is the publish)
      if (tid < rows) {
        float acc = 0.f;
        for (int i = 0; i < 16; ++i) acc += vec[(r0 + tid + 37 * i) % D] * 0.0625f;
        outv[tid] = acc * 0.9f + 0.01f;
      }
      __syncthreads();
      if (one_store) {
        if (tid < rows) __hip_atomic_store(out + r0 + tid, ((u64)(epoch + 1) << 32) | __float_as_uint(outv[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      } else {  // rows spread over the waves, 2 adjacent rows per wave instruction (as many instructions as needed)
        for (int r = wave * 2; r < rows; r += WAVES * 2)
          if (lane < 2 && r + lane < rows)
            __hip_atomic_store(out + r0 + r + lane, ((u64)(epoch + 1) << 32) | __float_as_uint(outv[r + lane]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
  }
  if (wg == 0 && tid == 0) stamps[0] = tg;
}

int main() {
  const int nphase = 2000, P = 256;
  u64* bufs; CK(hipMalloc(&bufs, (size_t)2 * D * 8));
  unsigned* tmo; CK(hipMalloc(&tmo, 16));
  long long* stamps; CK(hipMalloc(&stamps, 64));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<u64> init((size_t)2 * D, 0);
  for (int i = 0; i < D; ++i) { float v = sinf(0.37f * i); unsigned u; memcpy(&u, &v, 4); init[i] = (1ull << 32) | u; }
  for (int nprod : {12, 24, 48, 96, 192}) {
    for (int one : {0, 1}) {
      float best = 1e30f; long long hs = 0; unsigned ht = 0;
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemcpy(bufs, init.data(), init.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(tmo, 0, 16));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(chain, dim3(P), dim3(NT), 0, s, bufs, nphase, nprod, one, tmo, stamps);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; CK(hipMemcpy(&hs, stamps, 8, hipMemcpyDeviceToHost)); }
        CK(hipMemcpy(&ht, tmo, 4, hipMemcpyDeviceToHost));
      }
      printf("producers %3d x %2d rows, %s: %.3f us/phase (gather %.2f) timeout %u\n", nprod, D / nprod,
             one ? "ONE store instruction per producer " : "one store instruction per 2 rows   ", best * 1e3 / nphase, hs * 0.01 / nphase, ht);
    }
  }
  return 0;
}
