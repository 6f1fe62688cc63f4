// Stand-alone timing + spot check of the encoder GEMM (csrc/gemm.hip is compiled into this file unchanged) at the
// encoder's own shapes (64 clips x 1500 rows) and at 4096^3 for comparison with published tile structures.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../include -I../../whisper.axera_amd/csrc gemm_shapes.cpp -o gemm_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <vector>
#include <random>
#include "../../whisper.axera_amd/csrc/gemm.hip"

using namespace axw;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

static float bf16_to_f(unsigned short u) { unsigned v = (unsigned)u << 16; float f; memcpy(&f, &v, 4); return f; }
static unsigned short f_to_bf16(float f) { unsigned v; memcpy(&v, &f, 4); v += 0x7fff + ((v >> 16) & 1); return (unsigned short)(v >> 16); }

struct Shape { const char* name; int M, batch, N, K, epi; };

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 10;
  gemm_force_tile = argc > 2 ? atoi(argv[2]) : 0;  // 0 auto, 1: 128x128, 2: 256x128, 5: 256x256 stream per CU
  printf("tile selection %d\n", gemm_force_tile);
  const Shape shapes[] = {
      {"attn.out / cq  (N=768  K=768,  resid f32)", 1500, 64, 768, 768, EPI_RESID_F32},
      {"qkv-like       (N=2304 K=768,  bias bf16)", 1500, 64, 2304, 768, EPI_BIAS_BF16},
      {"mlp.0          (N=3072 K=768,  gelu bf16)", 1500, 64, 3072, 768, EPI_BIAS_GELU_BF16},
      {"mlp.2          (N=768  K=3072, resid f32)", 1500, 64, 768, 3072, EPI_RESID_F32},
      {"4096^3         (bias bf16)", 4096, 1, 4096, 4096, EPI_BIAS_BF16},
      {"one clip mlp.0 (N=3072 K=768,  gelu bf16)", 1500, 1, 3072, 768, EPI_BIAS_GELU_BF16},
      {"q,k,v          (N=2304 K=768,  Q K V^T    )", 1500, 64, 2304, 768, EPI_QKV},
  };
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> U(-1.f, 1.f);
  for (const Shape& sh : shapes) {
    const long rows = (long)sh.M * sh.batch;
    std::vector<unsigned short> hA((size_t)rows * sh.K), hW((size_t)sh.N * sh.K);
    for (auto& v : hA) v = f_to_bf16(U(rng));
    for (auto& v : hW) v = f_to_bf16(U(rng) * 0.05f);
    std::vector<float> hb(sh.N);
    for (auto& v : hb) v = U(rng);
    h16 *dA, *dW; float* db; void* dC;
    const bool f32out = sh.epi == EPI_RESID_F32;
    const bool qkv = sh.epi == EPI_QKV;  // Q [B][M][d] | K [B][M][d] | V^T [B][d][t_pad] in one buffer
    const int d = 768, t_pad = (sh.M + 63) / 64 * 64;
    const size_t q_elems = (size_t)rows * d, vt_elems = (size_t)sh.batch * d * t_pad;
    const size_t cbytes = qkv ? (2 * q_elems + vt_elems) * 2 : (size_t)rows * sh.N * (f32out ? 4 : 2);
    CK(hipMalloc(&dA, hA.size() * 2)); CK(hipMalloc(&dW, hW.size() * 2)); CK(hipMalloc(&db, sh.N * 4)); CK(hipMalloc(&dC, cbytes));
    CK(hipMemcpy(dA, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(dW, hW.data(), hW.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), sh.N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(dC, 0, cbytes));
    GemmParams p{};
    p.A = dA; p.lda = sh.K; p.a_batch_stride = (long)sh.M * sh.K; p.W = dW; p.bias = db;
    p.C = dC; p.ldc = sh.N; p.c_batch_stride = (long)sh.M * sh.N;
    p.M = sh.M; p.N = sh.N; p.K = sh.K; p.batch = sh.batch; p.d_model = 768; p.epilogue = sh.epi;
    if (qkv) {
      h16* base = reinterpret_cast<h16*>(dC);
      p.C = base; p.c_batch_stride = (long)sh.M * d;
      p.C2 = base + q_elems; p.c2_batch_stride = (long)sh.M * d;
      p.C3 = base + 2 * q_elems; p.c3_batch_stride = (long)d * t_pad;
      p.t_pad = t_pad;
    }
    hipStream_t s; CK(hipStreamCreate(&s));
    launch_gemm(p, s);  // first launch: checked below (resid: C was 0)
    CK(hipStreamSynchronize(s));
    std::vector<unsigned char> hC(cbytes);
    CK(hipMemcpy(hC.data(), dC, cbytes, hipMemcpyDeviceToHost));
    double max_err = 0;
    for (int t = 0; t < 4000; ++t) {
      const long m = (long)(rng() % rows); const int n = (int)(rng() % sh.N);
      double acc = hb[n];
      for (int k = 0; k < sh.K; ++k) acc += (double)bf16_to_f(hA[m * sh.K + k]) * bf16_to_f(hW[(size_t)n * sh.K + k]);
      if (sh.epi == EPI_BIAS_GELU_BF16) acc = 0.5 * acc * (1 + erf(acc * 0.7071067811865476));
      double got;
      if (qkv) {  // V^T: frames of a 16-group are stored in the order [0-3, 8-11, 4-7, 12-15] (gemm.hip, encoder_attn.hip)
        const unsigned short* o = reinterpret_cast<unsigned short*>(hC.data());
        const long b = m / sh.M, mm = m % sh.M;
        const long mp = (mm & ~15L) | ((mm & 4) << 1) | ((mm & 8) >> 1) | (mm & 3);
        got = n < d ? bf16_to_f(o[m * d + n]) : n < 2 * d ? bf16_to_f(o[q_elems + m * d + (n - d)])
                                                          : bf16_to_f(o[2 * q_elems + (b * d + (n - 2 * d)) * t_pad + mp]);
      } else {
        got = f32out ? (double)reinterpret_cast<float*>(hC.data())[m * sh.N + n] : (double)bf16_to_f(reinterpret_cast<unsigned short*>(hC.data())[m * sh.N + n]);
      }
      const double err = fabs(got - acc) / (f32out ? 1.0 : fmax(1.0, fabs(acc)) * 4.0);  // bf16 output: within half an ulp-ish
      if (err > max_err) max_err = err;
      if (qkv && err > 4e-3 && getenv("GEMM_SHAPES_DEBUG")) fprintf(stderr, "  qkv mismatch m=%ld (clip %ld row %ld) n=%d got %.5f want %.5f\n", m, m / sh.M, m % sh.M, n, got, acc);
    }
#ifdef AXW_GEMM_TIMING  // one stamped launch of the stream kernel: where a tile's time goes (100 MHz ticks -> us)
    if (sh.batch > 1) {
      const size_t n_st = (size_t)256 * 32 * 2 * 5;
      unsigned long long* d_st; CK(hipMalloc(&d_st, n_st * 8)); CK(hipMemset(d_st, 0, n_st * 8));
      GemmParams q = p; q.part = reinterpret_cast<float*>(d_st);
      launch_gemm(q, s); CK(hipStreamSynchronize(s));
      std::vector<unsigned long long> st(n_st);
      CK(hipMemcpy(st.data(), d_st, n_st * 8, hipMemcpyDeviceToHost));
      double sum[2][6] = {}; long cnt[2] = {};
      for (int wg = 0; wg < 256; ++wg)
        for (int t = 0; t + 1 < 32; ++t)
          for (int g = 0; g < 2; ++g) {
            const unsigned long long* a = &st[(((size_t)wg * 32 + t) * 2 + g) * 5];
            const unsigned long long* b = a + 10;
            if (!a[0] || !a[4] || b[0] <= a[4]) continue;
            sum[g][0] += (double)(a[1] - a[0]); sum[g][1] += (double)(a[2] - a[1]); sum[g][2] += (double)(a[3] - a[2]);
            sum[g][3] += (double)(a[4] - a[3]); sum[g][4] += (double)(b[0] - a[4]); sum[g][5] += (double)(b[0] - a[0]);
            ++cnt[g];
          }
      for (int g = 0; g < 2; ++g)
        if (cnt[g])
          printf("    wave group %d, %ld tiles: first 2 k-tiles %.2f us | rest of the k-loop %.2f (%d k-tiles) | wait for the other group %.2f | epilogue %.2f | to the next tile %.2f | tile %.2f\n",
                 g, cnt[g], sum[g][0] / cnt[g] / 100, sum[g][1] / cnt[g] / 100, sh.K / 64 - 2, sum[g][2] / cnt[g] / 100, sum[g][3] / cnt[g] / 100,
                 sum[g][4] / cnt[g] / 100, sum[g][5] / cnt[g] / 100);
      (void)hipFree(d_st);
    }
#endif
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 2; ++i) launch_gemm(p, s);
    CK(hipEventRecord(e0, s));
    for (int i = 0; i < iters; ++i) launch_gemm(p, s);
    CK(hipEventRecord(e1, s));
    CK(hipStreamSynchronize(s));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double tf = 2.0 * rows * sh.N * sh.K / (ms / iters * 1e-3) / 1e12;
    printf("%-46s %8.3f ms  %7.1f TFLOP/s  (%.3f of 2500)   spot err %.2e %s\n", sh.name, ms / iters, tf, tf / 2500, max_err,
           max_err < (f32out ? 2e-3 : 4e-3) ? "ok" : "MISMATCH");
    fflush(stdout);
    (void)hipFree(dA); (void)hipFree(dW); (void)hipFree(db); (void)hipFree(dC); (void)hipStreamDestroy(s);
  }
  return 0;
}
