// micro-benchmark 7: does a SECOND poll in flight, half a round trip behind the first, shorten an all-to-all hand-off?
// Same chain as publish_shape.cpp (256 workgroups x 8 waves gather 768 granules, 48 producers publish 16 rows each with
// one store instruction). A lane polls one PAIR of adjacent granules. Modes:
//   0  compiler-generated loop: one 16-byte sc1 load, wait, compare, repeat (what decode_persistent.hip does)
//   1  the same loop written in inline asm with two 8-byte loads per poll (control for the asm form)
//   2  inline asm, two polls in flight (A, B), B issued `s_sleep` later; each is re-issued as soon as it has been
//      looked at, so the memory is sampled every half round trip instead of every round trip
//   hipcc -O3 --offload-arch=gfx950 poll_stagger.cpp -o poll_stagger && ./poll_stagger
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); exit(1);}}while(0)
typedef unsigned long long u64;
typedef __attribute__((address_space(1))) u64 gu64;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr int D = 768, WAVES = 8, NT = WAVES * 64;

// polls the pair at `addr` until both granules carry a tag >= tag64 >> 32; returns the lanes (exec mask) still waiting
template <bool STAGGER>
__device__ __forceinline__ u64 poll_pair_asm(const u64* addr, u64 tag64, u64& g0, u64& g1) {
  u64 xa0, xa1, xb0, xb1, left, tmp;
  unsigned cnt;
  if constexpr (STAGGER) {
    asm volatile(
        "v_mov_b64 %[o0], 0\n\tv_mov_b64 %[o1], 0\n\t"
        "global_load_dwordx2 %[xa0], %[addr], off sc1\n\t"
        "global_load_dwordx2 %[xa1], %[addr], off offset:8 sc1\n\t"
        "s_sleep 5\n\t"
        "global_load_dwordx2 %[xb0], %[addr], off sc1\n\t"
        "global_load_dwordx2 %[xb1], %[addr], off offset:8 sc1\n\t"
        "s_movk_i32 %[cnt], 0x4000\n"
        "1:\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_cmp_le_u64 vcc, %[tag], %[xa0]\n\t"
        "s_and_saveexec_b64 %[tmp], vcc\n\t"
        "v_mov_b64 %[o0], %[xa0]\n\t"
        "s_mov_b64 exec, %[tmp]\n\t"
        "v_cmp_le_u64 vcc, %[tag], %[xa1]\n\t"
        "s_and_saveexec_b64 %[tmp], vcc\n\t"
        "v_mov_b64 %[o1], %[xa1]\n\t"
        "s_mov_b64 exec, %[tmp]\n\t"
        "v_cmp_le_u64 vcc, %[tag], %[o0]\n\t"
        "v_cmp_le_u64 %[left], %[tag], %[o1]\n\t"
        "s_and_b64 %[left], %[left], vcc\n\t"
        "s_andn2_b64 %[left], exec, %[left]\n\t"
        "s_cbranch_scc0 2f\n\t"
        "global_load_dwordx2 %[xa0], %[addr], off sc1\n\t"
        "global_load_dwordx2 %[xa1], %[addr], off offset:8 sc1\n\t"
        "s_waitcnt vmcnt(2)\n\t"
        "v_cmp_le_u64 vcc, %[tag], %[xb0]\n\t"
        "s_and_saveexec_b64 %[tmp], vcc\n\t"
        "v_mov_b64 %[o0], %[xb0]\n\t"
        "s_mov_b64 exec, %[tmp]\n\t"
        "v_cmp_le_u64 vcc, %[tag], %[xb1]\n\t"
        "s_and_saveexec_b64 %[tmp], vcc\n\t"
        "v_mov_b64 %[o1], %[xb1]\n\t"
        "s_mov_b64 exec, %[tmp]\n\t"
        "v_cmp_le_u64 vcc, %[tag], %[o0]\n\t"
        "v_cmp_le_u64 %[left], %[tag], %[o1]\n\t"
        "s_and_b64 %[left], %[left], vcc\n\t"
        "s_andn2_b64 %[left], exec, %[left]\n\t"
        "s_cbranch_scc0 2f\n\t"
        "global_load_dwordx2 %[xb0], %[addr], off sc1\n\t"
        "global_load_dwordx2 %[xb1], %[addr], off offset:8 sc1\n\t"
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"
        "s_cmp_lg_u32 %[cnt], 0\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n\t"
        "s_waitcnt vmcnt(0)"
        : [o0] "=&v"(g0), [o1] "=&v"(g1), [xa0] "=&v"(xa0), [xa1] "=&v"(xa1), [xb0] "=&v"(xb0), [xb1] "=&v"(xb1),
          [left] "=&s"(left), [tmp] "=&s"(tmp), [cnt] "=&s"(cnt)
        : [addr] "v"(addr), [tag] "s"(tag64)
        : "vcc", "scc", "memory");
  } else {
    asm volatile(
        "v_mov_b64 %[o0], 0\n\tv_mov_b64 %[o1], 0\n\t"
        "s_movk_i32 %[cnt], 0x4000\n"
        "1:\n\t"
        "global_load_dwordx2 %[xa0], %[addr], off sc1\n\t"
        "global_load_dwordx2 %[xa1], %[addr], off offset:8 sc1\n\t"
        "s_waitcnt vmcnt(0)\n\t"
        "v_cmp_le_u64 vcc, %[tag], %[xa0]\n\t"
        "s_and_saveexec_b64 %[tmp], vcc\n\t"
        "v_mov_b64 %[o0], %[xa0]\n\t"
        "s_mov_b64 exec, %[tmp]\n\t"
        "v_cmp_le_u64 vcc, %[tag], %[xa1]\n\t"
        "s_and_saveexec_b64 %[tmp], vcc\n\t"
        "v_mov_b64 %[o1], %[xa1]\n\t"
        "s_mov_b64 exec, %[tmp]\n\t"
        "v_cmp_le_u64 vcc, %[tag], %[o0]\n\t"
        "v_cmp_le_u64 %[left], %[tag], %[o1]\n\t"
        "s_and_b64 %[left], %[left], vcc\n\t"
        "s_andn2_b64 %[left], exec, %[left]\n\t"
        "s_cbranch_scc0 2f\n\t"
        "s_sub_u32 %[cnt], %[cnt], 1\n\t"
        "s_cmp_lg_u32 %[cnt], 0\n\t"
        "s_cbranch_scc1 1b\n"
        "2:\n\t"
        "s_waitcnt vmcnt(0)"
        : [o0] "=&v"(g0), [o1] "=&v"(g1), [xa0] "=&v"(xa0), [xa1] "=&v"(xa1), [left] "=&s"(left), [tmp] "=&s"(tmp), [cnt] "=&s"(cnt)
        : [addr] "v"(addr), [tag] "s"(tag64)
        : "vcc", "scc", "memory");
    (void)xb0; (void)xb1;
  }
  return left;
}

__global__ __launch_bounds__(NT) void chain(u64* bufs, int nphase, int nprod, int mode, unsigned* tmo, long long* stamps) {
  __shared__ float vec[D];
  __shared__ float outv[64];
  const int tid = threadIdx.x, wg = blockIdx.x;
  const int rows = D / nprod;  // per producer (<= 64)
  const __amdgpu_buffer_rsrc_t GR = __builtin_amdgcn_make_buffer_rsrc((void*)bufs, 0, 2 * D * 8, 0x27000);
  long long tg = 0;
  for (int p = 0; p < nphase; ++p) {
    const unsigned epoch = p + 1;
    const u64* in = bufs + (long)(p & 1) * D;
    const long long t0 = wall_clock64();
    int failed = 0;
    if (tid < 6 * 64) {  // 384 pairs: waves 0-5 poll, one pair per lane
      unsigned v0 = 0, v1 = 0;
      if (mode == 0) {
        unsigned spins = 0;
        while (true) {
          const u32x4 x = __builtin_amdgcn_raw_buffer_load_b128(GR, ((p & 1) * D + 2 * tid) * 8, 0, 16);
          if (x[1] == epoch && x[3] == epoch) { v0 = x[0]; v1 = x[2]; break; }
          if (++spins > 4000000u) { *tmo = 1; failed = 1; break; }
        }
      } else {
        u64 g0, g1;
        const u64 left = mode == 2 ? poll_pair_asm<true>(in + 2 * tid, (u64)epoch << 32, g0, g1)
                                   : poll_pair_asm<false>(in + 2 * tid, (u64)epoch << 32, g0, g1);
        if (left) { *tmo = 1; failed = 1; }
        v0 = (unsigned)g0; v1 = (unsigned)g1;
      }
      vec[2 * tid] = __uint_as_float(v0);
      vec[2 * tid + 1] = __uint_as_float(v1);
    }
    if (__syncthreads_or(failed)) return;
    tg += wall_clock64() - t0;
    gu64* out = (gu64*)(bufs + (long)((p + 1) & 1) * D);
    if (wg < nprod) {
      const int r0 = wg * rows;
      if (tid < rows) {
        float acc = 0.f;
        for (int i = 0; i < 16; ++i) acc += vec[(r0 + tid + 37 * i) % D] * 0.0625f;
        outv[tid] = acc * 0.9f + 0.01f;
      }
      __syncthreads();
      if (tid < rows) __hip_atomic_store(out + r0 + tid, ((u64)(epoch + 1) << 32) | __float_as_uint(outv[tid]), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  if (wg == 0 && tid == 0) stamps[0] = tg;
}

int main() {
  const int nphase = 4000, P = 256;
  u64* bufs; CK(hipMalloc(&bufs, (size_t)2 * D * 8));
  unsigned* tmo; CK(hipMalloc(&tmo, 16));
  long long* stamps; CK(hipMalloc(&stamps, 64));
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<u64> init((size_t)2 * D, 0);
  for (int i = 0; i < D; ++i) { float v = sinf(0.37f * i); unsigned u; memcpy(&u, &v, 4); init[i] = (1ull << 32) | u; }
  std::vector<u64> ref;
  for (int nprod : {48, 192}) {
    for (int mode : {0, 1, 2}) {
      float best = 1e30f; long long hs = 0; unsigned ht = 0;
      std::vector<u64> fin((size_t)2 * D);
      for (int rep = 0; rep < 3; ++rep) {
        CK(hipMemcpy(bufs, init.data(), init.size() * 8, hipMemcpyHostToDevice));
        CK(hipMemset(tmo, 0, 16));
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s));
        hipLaunchKernelGGL(chain, dim3(P), dim3(NT), 0, s, bufs, nphase, nprod, mode, tmo, stamps);
        CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; CK(hipMemcpy(&hs, stamps, 8, hipMemcpyDeviceToHost)); }
        CK(hipMemcpy(&ht, tmo, 4, hipMemcpyDeviceToHost));
      }
      CK(hipMemcpy(fin.data(), bufs, fin.size() * 8, hipMemcpyDeviceToHost));
      if (mode == 0) ref = fin;
      const bool same = fin == ref;  // the chain's final vector: every mode must have consumed exactly the same values
      printf("producers %3d, mode %d: %.3f us/phase (gather %.2f) timeout %u  final vector %s\n", nprod, mode, best * 1e3 / nphase,
             hs * 0.01 / nphase, ht, same ? "identical" : "DIFFERS");
    }
  }
  return 0;
}
