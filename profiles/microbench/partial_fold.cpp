// micro-benchmark: what would folding the attention output projection into the self-attention owners buy the one-clip
// persistent decode launch (round 5, verdict r4 #3)?  One launch, 256 resident workgroups of 8 polling waves, data-tagged
// 8-byte granules, 16-byte sc1 pair polls, one full-line store instruction per 16 granules — the kernel's own transport.
//   variant A (today):  H owners publish 64 granules each (the head's attention vector) -> NP_D workgroups gather all
//                       D = 64 H of them, (compute 16 rows), publish 16 granules each -> ALL workgroups gather D granules
//   variant B (fold):   H owners publish D partial-sum granules each -> ALL workgroups gather H x D granules and add
//                       them in head order
//   variant C (fold, two-level): H owners publish D partials each -> only NP_D workgroups gather H x D, add, and publish
//                       16 summed granules each -> the others gather D (off the critical path in the real kernel; here it
//                       is on it, so C bounds the fold from above)
// Every iteration is one dependent cycle (the owners wait for the cycle's last gather before they publish again); time per
// cycle over 2000 cycles.   hipcc -O3 --offload-arch=gfx950 partial_fold.cpp -o partial_fold && ./partial_fold
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

typedef unsigned long long u64;
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) u64 gu64;
constexpr int T = 512;  // 8 polling waves

__device__ __forceinline__ void gput(u64* g, unsigned tag, float v) {
  __hip_atomic_store((gu64*)g, ((u64)tag << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
// lane gathers NP pairs (granule index pidx(k), even; < 0: none); returns the sum of all values; spins bounded
template <int NP, typename IDX>
__device__ __forceinline__ float gather_pairs(__amdgpu_buffer_rsrc_t rs, unsigned tag, IDX pidx, unsigned* tmo) {
  bool ok[NP]; int ix[NP]; float acc[NP];
#pragma unroll
  for (int k = 0; k < NP; ++k) { ix[k] = pidx(k); ok[k] = ix[k] < 0; acc[k] = 0.f; }
  for (unsigned spins = 0;; ++spins) {
    bool all = true;
    u32x4 x[NP];
#pragma unroll
    for (int k = 0; k < NP; ++k) if (!ok[k]) x[k] = __builtin_amdgcn_raw_buffer_load_b128(rs, ix[k] * 8, 0, 16);
#pragma unroll
    for (int k = 0; k < NP; ++k)
      if (!ok[k]) {  // >=: a consumer that publishes nothing may be lapped by a cycle (the real kernel's consumers all publish once per layer)
        if (x[k][1] >= tag && x[k][3] >= tag) { acc[k] = __uint_as_float(x[k][0]) + __uint_as_float(x[k][2]); ok[k] = true; } else all = false;
      }
    if (all) break;
    if (spins > 1000000u) { *tmo = 1; break; }
    if ((spins & 1023u) == 1023u && __hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;  // somebody gave up: drain
  }
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NP; ++k) s += acc[k];  // fixed order
  return s;
}

template <int H, int VARIANT>
__global__ __launch_bounds__(T) void cycle_kernel(u64* G, int gbytes, int cycles, int np_d, unsigned* tmo, float* sink, long long* ticks) {
  constexpr int D = 64 * H;
  extern __shared__ float lds[];
  const int tid = threadIdx.x, wg = blockIdx.x, P = gridDim.x;
  const __amdgpu_buffer_rsrc_t GR = __builtin_amdgcn_make_buffer_rsrc((void*)G, 0, gbytes, 0x27000);
  // buffers (granules): ATT [D] | Y1 [D] | PART [H][D]
  constexpr int O_ATT = 0, O_Y1 = D, O_PART = 2 * D;
  const bool owner = wg >= P - H;         // the last H workgroups own a head
  const int head = wg - (P - H);
  const bool in_o = wg < np_d;
  float keep = 0.f;
  const long long t0 = wall_clock64();
  for (int c = 0; c < cycles; ++c) {
    const unsigned tag = c + 1;
    if (__hip_atomic_load(tmo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) break;
    if (owner) {
      if constexpr (VARIANT == 0) {
        if (tid < 64) gput(G + O_ATT + head * 64 + tid, tag, keep + 1.f);                 // one store of 4 full lines
      } else {
        for (int r = tid; r < D; r += T) gput(G + O_PART + head * D + r, tag, keep + 1.f);  // every wave instruction = 4 full lines
      }
    }
    if constexpr (VARIANT == 0) {
      if (in_o) {
        constexpr int NP = (D / 2 + T - 1) / T;
        float s = gather_pairs<NP>(GR, tag, [&](int j) { const int pr = tid + j * T; return 2 * pr < D ? O_ATT + 2 * pr : -1; }, tmo);
        lds[tid] = s;
        __syncthreads();
        const int rows = D / np_d;  // 16
        if (tid < rows) gput(G + O_Y1 + wg * rows + tid, tag, lds[tid] * 1e-3f);
      }
      constexpr int NP = (D / 2 + T - 1) / T;
      keep = gather_pairs<NP>(GR, tag, [&](int j) { const int pr = tid + j * T; return 2 * pr < D ? O_Y1 + 2 * pr : -1; }, tmo) * 1e-3f;
    } else if constexpr (VARIANT == 1) {
      constexpr int NPD = (D / 2 + T - 1) / T;
      keep = gather_pairs<NPD * H>(GR, tag, [&](int j) { const int pr = tid + (j / H) * T; return 2 * pr < D ? O_PART + (j % H) * D + 2 * pr : -1; }, tmo) * 1e-3f;
    } else {
      constexpr int NPD = (D / 2 + T - 1) / T;
      if (in_o) {
        float s = gather_pairs<NPD * H>(GR, tag, [&](int j) { const int pr = tid + (j / H) * T; return 2 * pr < D ? O_PART + (j % H) * D + 2 * pr : -1; }, tmo);
        lds[tid] = s;
        __syncthreads();
        const int rows = D / np_d;
        if (tid < rows) gput(G + O_Y1 + wg * rows + tid, tag, lds[tid] * 1e-3f);
        keep = s * 1e-3f;
      }
      keep += gather_pairs<NPD>(GR, tag, [&](int j) { const int pr = tid + j * T; return 2 * pr < D ? O_Y1 + 2 * pr : -1; }, tmo) * 1e-3f;
    }
    __syncthreads();
  }
  if (tid == 0) { ticks[wg] = wall_clock64() - t0; sink[wg] = keep; }
}

template <int H, int V>
static void run(const char* name, int P, int np_d, int cycles) {
  constexpr int D = 64 * H;
  const int gran = (2 + H) * D + 64;
  u64* G; unsigned* tmo; float* sink; long long* ticks;
  CK(hipMalloc((void**)&G, (size_t)gran * 8)); CK(hipMalloc((void**)&tmo, 16)); CK(hipMalloc((void**)&sink, P * 4)); CK(hipMalloc((void**)&ticks, P * 8));
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemset(G, 0, (size_t)gran * 8)); CK(hipMemset(tmo, 0, 16));
    CK(hipDeviceSynchronize());
    auto k = cycle_kernel<H, V>;
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024));  // one workgroup per CU
    hipLaunchKernelGGL(k, dim3(P), dim3(T), 100 * 1024, 0, G, gran * 8, cycles, np_d, tmo, sink, ticks);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(P); CK(hipMemcpy(h.data(), ticks, P * 8, hipMemcpyDeviceToHost));
    best = std::min(best, (float)(*std::max_element(h.begin(), h.end()) * 0.01 / cycles));
  }
  unsigned h_tmo; CK(hipMemcpy(&h_tmo, tmo, 4, hipMemcpyDeviceToHost));
  printf("%-44s H=%2d D=%4d  %6.2f us per cycle%s\n", name, H, D, best, h_tmo ? "  (TIMED OUT)" : "");
  CK(hipFree(G)); CK(hipFree(tmo)); CK(hipFree(sink)); CK(hipFree(ticks));
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  const int P = std::min(256, pr.multiProcessorCount), cycles = 2000;
  setvbuf(stdout, nullptr, _IOLBF, 0);
  printf("%s, %d workgroups of %d threads, %d cycles\n", pr.gcnArchName, P, T, cycles);
  for (int rep = 0; rep < 2; ++rep) {
    run<12, 0>("A: vector -> 48 row producers -> all", P, 48, cycles);
    run<12, 1>("B: 12 x 768 partials -> all", P, 48, cycles);
    run<12, 2>("C: partials -> 48 summers -> all (upper bound)", P, 48, cycles);
    run<6, 0>("A (d 384)", P, 24, cycles);
    run<6, 1>("B (d 384)", P, 24, cycles);
    run<8, 0>("A (d 512)", P, 32, cycles);
    run<8, 1>("B (d 512)", P, 32, cycles);
  }
  return 0;
}
