// cu_mask_probe.cpp — do CU masks of a stream (hipExtStreamCreateWithCUMask) hold for (a) eager launches, (b) a captured
// single-branch graph launched on that stream; which mask bits map to which XCD; and what a latency-bound dependent chain of
// small kernels costs on one half of the chip while a bandwidth-saturating kernel streams (i) on every CU, (ii) on the other half.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 cu_mask_probe.cpp -o cu_mask_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

__global__ void census(unsigned* out, long long spin_ticks) {
  unsigned hw, xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = hw; out[2 * blockIdx.x + 1] = xcc; }
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < spin_ticks) {}
}

// a bandwidth-saturating stream: every workgroup reads its slice of a big buffer `rounds` times
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void streamer(const float4* __restrict__ src0, float* sink, long n_vec, int rounds) {
  const f4* src = reinterpret_cast<const f4*>(src0);
  float acc = 0.f;
  for (int r = 0; r < rounds; ++r)
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n_vec; i += (long)gridDim.x * 256) {
      const f4 v = __builtin_nontemporal_load(src + i);
      acc += v[0] + v[1] + v[2] + v[3];
    }
  if (acc == 123.456f) sink[0] = acc;
}

// one link of a latency-bound chain: 96 workgroups, each reads 16 KB that the previous link wrote, reduces, writes 16 KB
__global__ __launch_bounds__(512) void link(const float* __restrict__ in, float* __restrict__ out, const float* __restrict__ w) {
  __shared__ float red[512];
  const int t = threadIdx.x, b = blockIdx.x;
  float a = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) a += in[(b * 8 + i) * 512 + t] * w[(b * 8 + i) * 512 + t];
  red[t] = a;
  __syncthreads();
  float s = red[t] + red[(t + 1) & 511];
#pragma unroll
  for (int i = 0; i < 8; ++i) out[(b * 8 + i) * 512 + t] = s + (float)i;
}

static int count_cus(const std::vector<unsigned>& h, int n, int* per_xcc) {
  std::set<unsigned> s;
  for (int i = 0; i < 8; ++i) per_xcc[i] = 0;
  std::set<unsigned> seen[8];
  for (int i = 0; i < n; ++i) {
    const unsigned hw = h[2 * i], xcc = h[2 * i + 1] & 0xf;
    const unsigned cu = (hw >> 8) & 0xf, sh = (hw >> 12) & 1, se = (hw >> 13) & 7;
    const unsigned key = (xcc << 12) | (se << 8) | (sh << 4) | cu;
    s.insert(key);
    seen[xcc & 7].insert(key);
  }
  for (int i = 0; i < 8; ++i) per_xcc[i] = (int)seen[i].size();
  return (int)s.size();
}

int main(int argc, char** argv) {
  int dev = 0, ncu = 0;
  CK(hipSetDevice(dev));
  CK(hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev));
  printf("CUs: %d\n", ncu);
  const int words = (ncu + 31) / 32;
  const int NB = 4096;
  unsigned* d_out;
  CK(hipMalloc(&d_out, NB * 8));
  std::vector<unsigned> h(2 * NB);
  auto report = [&](const char* what) {
    CK(hipMemcpy(h.data(), d_out, NB * 8, hipMemcpyDeviceToHost));
    int px[8];
    const int n = count_cus(h, NB, px);
    printf("%-58s distinct CUs %3d | per XCC:", what, n);
    for (int i = 0; i < 8; ++i) printf(" %2d", px[i]);
    printf("\n");
  };
  struct MaskCase { const char* name; std::vector<uint32_t> m; };
  std::vector<MaskCase> cases;
  { MaskCase c{"bits 0..127", std::vector<uint32_t>(words, 0)}; for (int i = 0; i < 128; ++i) c.m[i / 32] |= 1u << (i % 32); cases.push_back(c); }
  { MaskCase c{"bits 128..255", std::vector<uint32_t>(words, 0)}; for (int i = 128; i < 256 && i < ncu; ++i) c.m[i / 32] |= 1u << (i % 32); cases.push_back(c); }
  { MaskCase c{"even bits", std::vector<uint32_t>(words, 0x55555555u)}; cases.push_back(c); }
  { MaskCase c{"bits with (i % 8) < 4", std::vector<uint32_t>(words, 0x0f0f0f0fu)}; cases.push_back(c); }
  { MaskCase c{"bits 0..31", std::vector<uint32_t>(words, 0)}; c.m[0] = 0xffffffffu; cases.push_back(c); }
  hipStream_t plain;
  CK(hipStreamCreateWithFlags(&plain, hipStreamNonBlocking));
  hipLaunchKernelGGL(census, dim3(NB), dim3(64), 0, plain, d_out, 2000);
  CK(hipStreamSynchronize(plain));
  report("no mask, eager");
  std::vector<hipStream_t> masked;
  for (auto& c : cases) {
    hipStream_t s;
    CK(hipExtStreamCreateWithCUMask(&s, (uint32_t)words, c.m.data()));
    masked.push_back(s);
    CK(hipMemset(d_out, 0, NB * 8));
    hipLaunchKernelGGL(census, dim3(NB), dim3(64), 0, s, d_out, 2000);
    CK(hipStreamSynchronize(s));
    char buf[128];
    snprintf(buf, sizeof buf, "mask %s, eager", c.name);
    report(buf);
    // the same launch captured into a graph, the graph launched on the masked stream
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    hipLaunchKernelGGL(census, dim3(NB), dim3(64), 0, s, d_out, 2000);
    hipLaunchKernelGGL(census, dim3(NB), dim3(64), 0, s, d_out, 2000);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipMemset(d_out, 0, NB * 8));
    CK(hipGraphLaunch(ge, s));
    CK(hipStreamSynchronize(s));
    snprintf(buf, sizeof buf, "mask %s, captured graph launched on that stream", c.name);
    report(buf);
    CK(hipMemset(d_out, 0, NB * 8));
    CK(hipGraphLaunch(ge, plain));
    CK(hipStreamSynchronize(plain));
    snprintf(buf, sizeof buf, "mask %s, the same graph launched on an UNMASKED stream", c.name);
    report(buf);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }

  // ---- a latency-bound chain beside a bandwidth-saturating stream
  const long n_vec = (1L << 30) / 16;  // 1 GiB
  float4* big; float* sink; float *ca, *cb, *cw;
  CK(hipMalloc(&big, n_vec * 16)); CK(hipMemset(big, 1, n_vec * 16));
  CK(hipMalloc(&sink, 64));
  const size_t cn = 96 * 8 * 512;
  CK(hipMalloc(&ca, cn * 4)); CK(hipMalloc(&cb, cn * 4)); CK(hipMalloc(&cw, cn * 4));
  CK(hipMemset(ca, 0, cn * 4)); CK(hipMemset(cb, 0, cn * 4)); CK(hipMemset(cw, 0, cn * 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  // 200 dependent links as ONE captured graph per chain stream (eager launches of ~2 us kernels are host-bound), replayed on cs
  // while the streamer runs on ss (ss == nullptr: alone)
  auto make_chain = [&](hipStream_t cs) {
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < 100; ++i) {
      hipLaunchKernelGGL(link, dim3(96), dim3(512), 0, cs, cb, ca, cw);
      hipLaunchKernelGGL(link, dim3(96), dim3(512), 0, cs, ca, cb, cw);
    }
    CK(hipStreamEndCapture(cs, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphDestroy(g));
    CK(hipGraphLaunch(ge, cs));  // first launch of a fresh exec
    CK(hipStreamSynchronize(cs));
    return ge;
  };
  auto chain_us = [&](hipGraphExec_t ge, hipStream_t cs, hipStream_t ss, int stream_grid) {
    if (ss) hipLaunchKernelGGL(streamer, dim3(stream_grid), dim3(256), 0, ss, big, sink, n_vec, 6);
    hipLaunchKernelGGL(link, dim3(96), dim3(512), 0, cs, ca, cb, cw);  // (the streamer ramps up meanwhile)
    CK(hipEventRecord(e0, cs));
    CK(hipGraphLaunch(ge, cs));
    CK(hipEventRecord(e1, cs));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (ss) CK(hipStreamSynchronize(ss));
    return ms * 1000.f / 200.f;
  };
  auto stream_ms = [&](hipStream_t ss, int grid) {
    CK(hipEventRecord(e0, ss));
    hipLaunchKernelGGL(streamer, dim3(grid), dim3(256), 0, ss, big, sink, n_vec, 2);
    CK(hipEventRecord(e1, ss));
    CK(hipEventSynchronize(e1));
    float ms = 0.f;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms;
  };
  hipStream_t lo = masked[0], hi = masked[1], plain2;
  CK(hipStreamCreateWithFlags(&plain2, hipStreamNonBlocking));
  hipGraphExec_t g_plain = make_chain(plain), g_lo = make_chain(lo), g_hi = make_chain(hi), g_plain2 = make_chain(plain2);
  {  // two chains side by side: both on plain streams, both on masked streams (what a CU-split decoder step does), with per-replay events
    auto pair_us = [&](hipGraphExec_t ga, hipStream_t sa, hipGraphExec_t gb, hipStream_t sb, int reps) {
      hipEvent_t ef, ej;
      CK(hipEventCreateWithFlags(&ef, hipEventDisableTiming)); CK(hipEventCreateWithFlags(&ej, hipEventDisableTiming));
      CK(hipStreamSynchronize(sa)); CK(hipStreamSynchronize(sb));
      CK(hipEventRecord(e0, sa));
      for (int r = 0; r < reps; ++r) {
        CK(hipEventRecord(ef, sa));
        CK(hipStreamWaitEvent(sb, ef, 0));
        CK(hipGraphLaunch(ga, sa));
        CK(hipGraphLaunch(gb, sb));
        CK(hipEventRecord(ej, sb));
        CK(hipStreamWaitEvent(sa, ej, 0));
      }
      CK(hipEventRecord(e1, sa));
      CK(hipEventSynchronize(e1));
      float ms = 0.f;
      CK(hipEventElapsedTime(&ms, e0, e1));
      return ms * 1000.f / (200.f * reps);
    };
    printf("two 200-link chains side by side, fork/join events per replay, 5 replays: plain + plain %.2f us per link-pair | masked A + masked B %.2f | plain + masked B %.2f\n",
           pair_us(g_plain, plain, g_plain2, plain2, 5), pair_us(g_lo, lo, g_hi, hi, 5), pair_us(g_plain, plain, g_hi, hi, 5));
  }
  for (int rep = 0; rep < 2; ++rep) {
    printf("streamer alone: whole chip %.2f TB/s | half (mask 0..127) %.2f TB/s | half (mask 128..255) %.2f TB/s\n",
           2.0 * n_vec * 16 / stream_ms(plain, 2048) / 1e9, 2.0 * n_vec * 16 / stream_ms(lo, 1024) / 1e9, 2.0 * n_vec * 16 / stream_ms(hi, 1024) / 1e9);
    printf("chain link (96 workgroups, dependent, graph replay): alone, unmasked %.2f us | alone, on half A %.2f us\n", chain_us(g_plain, plain, nullptr, 0), chain_us(g_lo, lo, nullptr, 0));
    printf("chain beside a streamer on EVERY CU (both unmasked)          %.2f us per link\n", chain_us(g_plain, plain, plain2, 2048));
    printf("chain on half A (mask 0..127), streamer on half B (128..255)  %.2f us per link\n", chain_us(g_lo, lo, hi, 1024));
    printf("chain unmasked, streamer on half B                            %.2f us per link\n", chain_us(g_plain, plain, hi, 1024));
    printf("chain on half A, streamer unmasked (every CU)                 %.2f us per link\n", chain_us(g_lo, lo, plain2, 2048));
  }
  return 0;
}
