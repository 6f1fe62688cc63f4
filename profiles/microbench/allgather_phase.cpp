// micro-benchmark 4: what does one all-to-all phase of a PERSISTENT batch-1 decoder cost on MI355X?
// A chain of dependent "GEMV phases" inside one launch: every workgroup gathers the whole input vector from 8-byte
// {tag, value} granules (sc1 stores / sc1 loads, the data is the flag), LayerNorms it, multiplies its own weight rows
// (prefetched before the wait) and publishes its outputs as granules for the next phase. Compared with the
// ~4.3 us a dependent hipGraph node of the launch-per-phase decoder costs.
//   hipcc -O3 --offload-arch=gfx950 allgather_phase.cpp -o allgather_phase && ./allgather_phase
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <algorithm>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);}}while(0)

typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef __attribute__((address_space(1))) unsigned gu32;

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// phases alternate between (Din=a -> Dout=b) and (Din=b -> Dout=a)
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64) void chain(u64* bufs, int bufstride, const unsigned short* W, long wstride, int nw,
                                                    int a, int b, int nphase, unsigned* tmo, int sleep_n, long long* stamps, int late_w, int R, long rep_stride) {
  extern __shared__ float vec[];  // [max(a,b)]
  __shared__ float red[WAVES * 2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int P = gridDim.x, wg = blockIdx.x;
  long long acc_t[4] = {0, 0, 0, 0};
  for (int p = 0; p < nphase; ++p) {
    const long long t0 = wall_clock64();
    const int din = (p & 1) ? b : a, dout = (p & 1) ? a : b;
    const unsigned epoch = p + 1;
    const unsigned short* Wp = W + (long)(p % nw) * wstride;
    // rows of this workgroup: contiguous block; one wave per row, round-robin
    const int rpw = (dout + P - 1) / P;
    const int r0 = wg * rpw, r1 = min(dout, r0 + rpw);
    // ---- prefetch this phase's weights (independent of the activations)
    constexpr int MAXR = 2;    // rows per wave
    constexpr int MAXC = 12;   // 4-element chunks per lane per row (din <= 3072)
    uint2 wreg[MAXR][MAXC];
    const int nch = din / 256;
#pragma unroll
    for (int rr = 0; rr < MAXR; ++rr) {
      const int r = r0 + wave + rr * WAVES;
#pragma unroll
      for (int c = 0; c < MAXC; ++c)
        if (!late_w && r < r1 && c < nch) wreg[rr][c] = *reinterpret_cast<const uint2*>(Wp + (long)r * din + c * 256 + lane * 4);
    }
    // ---- gather
    gu64* in = (gu64*)(bufs + (long)(wg % R) * rep_stride + (long)(p & 1) * bufstride);
    int failed = 0;
    {
      const int per = (din + WAVES * 64 - 1) / (WAVES * 64);
      for (int k = 0; k < per; ++k) {
        const int i = tid + k * WAVES * 64;
        if (i < din) {
          u64 x;
          unsigned spins = 0;
          while (true) {
            x = __hip_atomic_load(in + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if ((unsigned)(x >> 32) == epoch) break;
            if (++spins > 4000000u) { *tmo = 1; failed = 1; break; }
            if (sleep_n) __builtin_amdgcn_s_sleep(1);
          }
          vec[i] = __uint_as_float((unsigned)x);
        }
      }
    }
    if (__syncthreads_or(failed)) return;  // uniform give-up
    if (late_w) {
#pragma unroll
      for (int rr = 0; rr < MAXR; ++rr) {
        const int r = r0 + wave + rr * WAVES;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (r < r1 && c < nch) wreg[rr][c] = *reinterpret_cast<const uint2*>(Wp + (long)r * din + c * 256 + lane * 4);
      }
    }
    const long long t1 = wall_clock64();
    // ---- LayerNorm statistics (every wave redundantly) + rows
    float s1 = 0.f, s2 = 0.f;
    for (int i = lane; i < din; i += 64) { const float t = vec[i]; s1 += t; s2 += t * t; }
    s1 = wave_sum(s1); s2 = wave_sum(s2);
    const float mean = s1 / din, rstd = rsqrtf(fmaxf(s2 / din - mean * mean, 0.f) + 1e-5f);
    gu64* out = (gu64*)(bufs + (long)((p + 1) & 1) * bufstride);
    const long long t2 = wall_clock64();
    for (int rr = 0; r0 + wave + rr * WAVES < r1; ++rr) {
      const int r = r0 + wave + rr * WAVES;
      {
        float acc = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
          if (c < nch) {
            const float4 v = *reinterpret_cast<const float4*>(vec + c * 256 + lane * 4);
            uint2 wv;
            if (rr == 0) wv = wreg[0][c]; else if (rr == 1) wv = wreg[1][c];
            else wv = *reinterpret_cast<const uint2*>(Wp + (long)r * din + c * 256 + lane * 4);
            const unsigned u0 = wv.x, u1 = wv.y;
            acc = fmaf(__uint_as_float(u0 << 16), (v.x - mean) * rstd, acc);
            acc = fmaf(__uint_as_float(u0 & 0xffff0000u), (v.y - mean) * rstd, acc);
            acc = fmaf(__uint_as_float(u1 << 16), (v.z - mean) * rstd, acc);
            acc = fmaf(__uint_as_float(u1 & 0xffff0000u), (v.w - mean) * rstd, acc);
          }
        acc = wave_sum(acc);
        if (lane < R)
          __hip_atomic_store(out + (long)lane * rep_stride + r, ((u64)(epoch + 1) << 32) | __float_as_uint(acc), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    const long long t3 = wall_clock64();
    __syncthreads();  // vec is rewritten by the next gather
    acc_t[0] += t1 - t0; acc_t[1] += t2 - t1; acc_t[2] += t3 - t2; acc_t[3] += wall_clock64() - t3;
  }
  if (tid == 0 && (wg == 0 || wg == P - 1)) for (int i = 0; i < 4; ++i) stamps[(wg ? 4 : 0) + i] = acc_t[i];
}

static unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)((u + 0x7fff + ((u >> 16) & 1)) >> 16); }

static long long* stamps;
template <int WAVES>
static void run(int P, int a, int b, int nphase, int nw, int sleep_n, int late_w, int R, const unsigned short* dW, long wstride, u64* bufs, int bufstride, unsigned* tmo) {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const long rep_stride = 2 * bufstride + 512 + 64;  // odd number of 512-byte units apart
  std::vector<u64> init(rep_stride * 32, 0);
  for (int r = 0; r < 32; ++r) for (int i = 0; i < a; ++i) { float v = sinf(0.37f * i); unsigned u; memcpy(&u, &v, 4); init[r * rep_stride + i] = (1ull << 32) | u; }
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipMemcpy(bufs, init.data(), init.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(tmo, 0, 16));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL((chain<WAVES>), dim3(P), dim3(WAVES * 64), (size_t)std::max(a, b) * 4, s, bufs, bufstride, dW, wstride, nw, a, b, nphase, tmo, sleep_n, stamps, late_w, R, rep_stride);
    CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    best = std::min(best, ms);
  }
  long long hs[8]; CK(hipMemcpy(hs, stamps, 64, hipMemcpyDeviceToHost));
  printf("   wg0: gather %.2f ln %.2f rows %.2f tail %.2f | wgLast: gather %.2f ln %.2f rows %.2f tail %.2f  (us/phase, 100 MHz clock)\n",
         hs[0] * 0.01 / nphase, hs[1] * 0.01 / nphase, hs[2] * 0.01 / nphase, hs[3] * 0.01 / nphase, hs[4] * 0.01 / nphase, hs[5] * 0.01 / nphase, hs[6] * 0.01 / nphase, hs[7] * 0.01 / nphase);
  unsigned h_tmo; CK(hipMemcpy(&h_tmo, tmo, 4, hipMemcpyDeviceToHost));
  std::vector<u64> fin(rep_stride * 32);
  CK(hipMemcpy(fin.data(), bufs, fin.size() * 8, hipMemcpyDeviceToHost));
  const u64 g = fin[(size_t)(nphase & 1) * bufstride];
  float v0; unsigned u = (unsigned)g; memcpy(&v0, &u, 4);
  printf("P=%3d waves=%2d %4d->%4d nw=%2d late_w=%d R=%2d sleep=%d: %.3f us/phase  (timeout=%u, final tag %u, v0 %.5f)\n", P, WAVES, a, b, nw, late_w, R, sleep_n,
         best * 1e3 / nphase, h_tmo, (unsigned)(g >> 32), v0);
  CK(hipStreamDestroy(s));
}

int main() {
  const int DMAX = 3072, NW = 72;
  const long wstride = (long)DMAX * 768;
  std::vector<unsigned short> hW((size_t)NW * wstride);
  srand(1);
  for (auto& w : hW) w = f2bf(((rand() & 0xffff) / 65536.f - 0.5f) * 0.12f);
  unsigned short* dW; CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
  u64* bufs; CK(hipMalloc(&bufs, (size_t)(2 * DMAX + 576) * 32 * 8));
  unsigned* tmo; CK(hipMalloc(&tmo, 16));
  CK(hipMalloc(&stamps, 64));
  const int nphase = 2000;
  for (int R : {1, 2, 4, 8, 16, 32}) {
    run<8>(256, 768, 768, nphase, 72, 0, 0, R, dW, wstride, bufs, DMAX, tmo);
    run<8>(256, 768, 3072, nphase, 72, 0, 0, R, dW, wstride, bufs, DMAX, tmo);
  }
  run<16>(256, 768, 768, nphase, 72, 0, 0, 16, dW, wstride, bufs, DMAX, tmo);
  run<16>(256, 768, 3072, nphase, 72, 0, 0, 16, dW, wstride, bufs, DMAX, tmo);
  return 0;
}
