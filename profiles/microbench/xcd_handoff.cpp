// micro-benchmark 5: how much cheaper is an all-to-all hand-off that stays inside ONE XCD (shared L2) than one that
// crosses XCDs? Input to the next design step of the persistent batch-1 decoder: tensor-parallel layers
// over the 8 XCDs would turn 5 of a layer's 8 hand-offs into intra-XCD ones.
// Every workgroup reads its XCC_ID, takes a rank inside its XCD with an atomic, and the 32 workgroups of an XCD then
// run a chain of dependent phases among themselves: gather a 768-value vector from granules, compute 24 rows each,
// publish. Protocol A: sc1 (write-through) stores + sc1 loads (correct for any placement). Protocol B: plain stores
// (the line stays dirty in the XCD's L2) + sc1 loads (bypass L1, served by that L2) — valid ONLY inside one XCD.
//   hipcc -O3 --offload-arch=gfx950 xcd_handoff.cpp -o xcd_handoff && ./xcd_handoff
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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

constexpr int D = 768, WAVES = 8, NT = WAVES * 64;

// mode 0: one chip-wide group (all workgroups), sc1 stores; mode 1: per-XCD groups, sc1 stores; mode 2: per-XCD groups, plain stores
__global__ __launch_bounds__(NT) void chain(u64* bufs, int* rank_cnt, int* group_size, int nphase, int mode, unsigned* tmo, int* xcc_out) {
  __shared__ float vec[D];
  __shared__ int s_rank, s_gsz;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int xcc = mode == 0 ? 0 : (int)(__builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11)) & 7);
  if (tid == 0) {
    s_rank = atomicAdd(&rank_cnt[xcc], 1);
    xcc_out[blockIdx.x] = xcc;
  }
  __syncthreads();
  const int rank = s_rank;
  // wait until every workgroup has registered (group sizes are then final)
  if (tid == 0) {
    atomicAdd(&rank_cnt[8], 1);
    int spins = 0;
    while (atomicAdd(&rank_cnt[8], 0) < (int)gridDim.x && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(4);
    s_gsz = atomicAdd(&rank_cnt[xcc], 0);
  }
  __syncthreads();
  const int gsz = s_gsz;
  if (blockIdx.x == 0 && tid == 0) group_size[0] = gsz;
  u64* gb = bufs + (long)xcc * 2 * D;  // this group's double buffer
  const int rpw = (D + gsz - 1) / gsz, r0 = rank * rpw, r1 = min(D, r0 + rpw);
  for (int p = 0; p < nphase; ++p) {
    const unsigned epoch = p + 1;
    gu64* in = (gu64*)(gb + (p & 1) * D);
    int failed = 0;
    for (int i = tid; i < D; i += NT) {
      u64 x;
      unsigned spins = 0;
      while (true) {
        x = __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // sc1: bypasses L1
        if ((unsigned)(x >> 32) == epoch) break;
        if (++spins > 4000000u) { *tmo = 1; failed = 1; break; }
      }
      vec[i] = __uint_as_float((unsigned)x);
    }
    if (__syncthreads_or(failed)) return;
    u64* out = gb + ((p + 1) & 1) * D;
    for (int r = r0 + wave; r < r1; r += WAVES) {
      float acc = 0.f;
      for (int i = lane; i < D; i += 64) acc += vec[i] * (0.001f * (float)((r * 7 + i) % 13 - 6));
      acc = wave_sum(acc) * 0.05f + 0.01f * (float)(r % 5);
      if (lane == 0) {
        const u64 g = ((u64)(epoch + 1) << 32) | __float_as_uint(acc);
        if (mode == 2) *(volatile u64*)(out + r) = g;  // plain store: stays in this XCD's L2
        else __hip_atomic_store((gu64*)(out + r), g, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
  }
}

int main() {
  const int nphase = 2000, P = 256;
  u64* bufs; CK(hipMalloc(&bufs, (size_t)8 * 2 * D * 8));
  int *rank_cnt, *gsz, *xcc; CK(hipMalloc(&rank_cnt, 64)); CK(hipMalloc(&gsz, 16)); CK(hipMalloc(&xcc, P * 4));
  unsigned* tmo; CK(hipMalloc(&tmo, 16));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<u64> init((size_t)8 * 2 * D, 0);
  for (int g = 0; g < 8; ++g)
    for (int i = 0; i < D; ++i) { float v = sinf(0.37f * i); unsigned u; memcpy(&u, &v, 4); init[(size_t)g * 2 * D + i] = (1ull << 32) | u; }
  const char* names[3] = {"chip-wide group of 256, sc1 stores", "per-XCD groups, sc1 stores", "per-XCD groups, plain stores (L2-resident)"};
  for (int mode = 0; mode < 3; ++mode) {
    float best = 1e30f;
    int h_gsz = 0;
    unsigned h_tmo = 0;
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipMemcpy(bufs, init.data(), init.size() * 8, hipMemcpyHostToDevice));
      CK(hipMemset(rank_cnt, 0, 64)); CK(hipMemset(tmo, 0, 16));
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(e0, s));
      hipLaunchKernelGGL(chain, dim3(P), dim3(NT), 0, s, bufs, rank_cnt, gsz, nphase, mode, tmo, xcc);
      CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      best = std::min(best, ms);
      CK(hipMemcpy(&h_gsz, gsz, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&h_tmo, tmo, 4, hipMemcpyDeviceToHost));
    }
    std::vector<int> hx(P); CK(hipMemcpy(hx.data(), xcc, P * 4, hipMemcpyDeviceToHost));
    int cnt[8] = {0};
    for (int v : hx) cnt[v & 7]++;
    std::vector<u64> fin(init.size()); CK(hipMemcpy(fin.data(), bufs, fin.size() * 8, hipMemcpyDeviceToHost));
    float v0; unsigned u = (unsigned)fin[(size_t)(nphase & 1) * D]; memcpy(&v0, &u, 4);
    printf("%-44s: %.3f us/phase (group of workgroup 0: %d; per-XCD counts %d %d %d %d %d %d %d %d; timeout %u; v0 %.5f)\n", names[mode],
           best * 1e3 / nphase, h_gsz, cnt[0], cnt[1], cnt[2], cnt[3], cnt[4], cnt[5], cnt[6], cnt[7], h_tmo, v0);
  }
  return 0;
}
