// frontend.hip — log-mel front-end on gfx950 (K1/K2 of SURVEY §8a).
//
// Replaces librosa::Feature::melspectrogram (cpp/src/librosa/librosa.h:46-155) and the
// clamp/normalise/pad of Whisper::preprocess (cpp/src/Whisper.cpp:151-184):
//   reflect pad 200 | periodic Hann | 400-pt DFT bins 0..200 | re^2+im^2 | Slaney mel GEMM |
//   log10(max(.,1e-10)) | global max over ALL frames | max(., gmax-8) | (.+4)/4 | zero-fill to 3000
// "ALL frames" means all frames of the input however long it is (Whisper.cpp:158-172 takes the maximum before it
// truncates to 3000 frames): the grid covers every frame of the longest clip, frames past 3000 only feed the maximum.
//
// Kernel 1 (stft_mel_kernel): one workgroup = 32 consecutive frames of one clip. The windowed
// frames are staged in LDS ([32][400] f32, 51 KB), the 400-point DFT is evaluated directly with
// a 400-entry twiddle table in LDS (each lane owns up to 4 bins x 8 frames = 64 accumulators),
// the power spectrum goes back to LDS and the mel projection + log10 + clip maximum are fused in.
// Kernel 2 (mel_normalize_kernel): clamp/scale/zero-fill and layout: time-major h16 rows for the
// encoder's conv-as-GEMM (and the reference's [n_mels][3000] f32 layout when a caller asks for it).
// HBM traffic per clip: 1.92 MB PCM in, 0.96 MB log-mel scratch out+in, 0.48 MB h16 out.
#include "common.hpp"

namespace axw {
inline namespace AXW_NS {

constexpr int FR = 32;        // frames per workgroup
constexpr int PW_LD = 208;    // power row stride in LDS

__device__ __forceinline__ unsigned float_to_ordered(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(256) void stft_mel_kernel(FrontendParams p, const float* __restrict__ basis_t /*[201][n_mels]*/) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xw = reinterpret_cast<float*>(smem);                 // [FR][400] windowed frames; later power [FR][PW_LD]
  float2* tw = reinterpret_cast<float2*>(smem + FR * kNFFT * 4);  // [400] (cos, sin)
  __shared__ float red[4];

  const int b = blockIdx.y;
  const int n_real = p.n_samples[b];
  // openai mode: the clip is (virtually) zero-padded or trimmed to 480000 samples before the STFT and frame 3000 of
  // its 3001 frames is dropped (upstream pad_or_trim + stft[..., :-1]; call site generate_data.py:162-176)
  const int n = p.openai ? kFramesOut * kHop : n_real;
  const int n_frames = p.openai ? kFramesOut : 1 + n / kHop;  // 1 + (n + 400 - 400) / 160   (librosa.h:87)
  const int f0 = blockIdx.x * FR;
  if (f0 >= n_frames) return;  // uniform per workgroup
  const float* x = p.pcm + (long)b * p.stride;
  const int tid = threadIdx.x;

  for (int i = tid; i < kNFFT; i += 256) tw[i] = make_float2(p.twiddle[2 * i], p.twiddle[2 * i + 1]);
  for (int i = tid; i < FR * kNFFT; i += 256) {
    int f = i / kNFFT, k = i - f * kNFFT;
    float v = 0.f;
    if (f0 + f < n_frames) {
      int j = (f0 + f) * kHop + k - kNFFT / 2;       // index into the un-padded signal
      if (j < 0) j = -j;                              // librosa.h:51  x[left - i]
      if (j >= n) j = 2 * n - 2 - j;                  // librosa.h:54  x[size - 2 - i + left]
      j = min(max(j, 0), n - 1);                      // clips shorter than the pad: stay in bounds
      float smp = 0.f;                                // librosa.h:92 (openai mode: zeros behind the clip's end)
      if (j < n_real) smp = j < p.stride ? x[j] : p.overflow[p.over_off[b] + (j - p.stride)];  // clips beyond the staging row
      v = smp * p.window[k];
    }
    xw[i] = v;
  }
  __syncthreads();

  // ---- DFT: lane -> bins {l, l+64, l+128, l+192}, wave -> frames [8w, 8w+8)
  const int lane = tid & 63, w = tid >> 6;
  float re[4][8], im[4][8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int f = 0; f < 8; ++f) { re[j][f] = 0.f; im[j][f] = 0.f; }
  int idx[4] = {0, 0, 0, 0};
  const int kk[4] = {lane, lane + 64, lane + 128, lane + 192};
  const float* xf = xw + (w * 8) * kNFFT;
  for (int t = 0; t < kNFFT; ++t) {
    float xs[8];
#pragma unroll
    for (int f = 0; f < 8; ++f) xs[f] = xf[f * kNFFT + t];   // wave-uniform address: LDS broadcast
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float2 c = tw[idx[j]];
#pragma unroll
      for (int f = 0; f < 8; ++f) { re[j][f] = fmaf(xs[f], c.x, re[j][f]); im[j][f] = fmaf(-xs[f], c.y, im[j][f]); }
      idx[j] += kk[j];
      if (idx[j] >= kNFFT) idx[j] -= kNFFT;
    }
  }
  __syncthreads();  // all waves are done reading xw
  float* pw = xw;   // power [FR][PW_LD]
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (kk[j] < kBins) {
#pragma unroll
      for (int f = 0; f < 8; ++f) pw[(w * 8 + f) * PW_LD + kk[j]] = re[j][f] * re[j][f] + im[j][f] * im[j][f];  // librosa.h:98-100
    }
  }
  __syncthreads();

  // ---- mel projection (librosa.h:153) + log10 (Whisper.cpp:160) + clip maximum (:162-164)
  const int nm = p.n_mels;
  float lmax = -3.402823466e38f;
  for (int o = tid; o < FR * nm; o += 256) {
    int f = o / nm, m = o - f * nm;
    if (f0 + f >= n_frames) continue;
    const float* pr = pw + f * PW_LD;
    float acc = 0.f;
    for (int k = 0; k < kBins; ++k) acc = fmaf(basis_t[k * nm + m], pr[k], acc);
    float v = log10f(fmaxf(acc, 1e-10f));
    lmax = fmaxf(lmax, v);
    if (f0 + f < kFramesOut) p.logmel[((long)b * kFramesOut + f0 + f) * nm + m] = v;
  }
  lmax = wave_max(lmax);
  if (lane == 0) red[w] = lmax;
  __syncthreads();
  if (tid == 0) atomicMax(&p.gmax[b], float_to_ordered(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
}

// Whisper.cpp:169-181: max(., mmax-8), (.+4)/4, rows resized to 3000 with zero fill.
__global__ __launch_bounds__(256) void mel_normalize_kernel(FrontendParams p) {
  const int b = blockIdx.y;
  const int nm = p.n_mels;
  const int n_frames = p.openai ? kFramesOut : min(1 + p.n_samples[b] / kHop, kFramesOut);
  const float floor_v = ordered_to_float(p.gmax[b]) - 8.0f;
  const long total = (long)kFramesOut * nm;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int f = (int)(i / nm), m = (int)(i - (long)f * nm);
    float v = 0.f;
    if (f < n_frames) v = (fmaxf(p.logmel[(long)b * total + i], floor_v) + 4.0f) * 0.25f;
    if (p.mel_ref) p.mel_ref[((long)b * nm + m) * kFramesOut + f] = v;
    if (p.mel_tm) p.mel_tm[((long)b * p.mel_rows + f + 1) * nm + m] = (h16)v;  // row 0 = conv left pad
  }
}

// host-supplied mel [B][n_mels][3000] f32 -> encoder input layout (used by AX_WHISPER_EncodeMel)
__global__ __launch_bounds__(256) void mel_to_tm_kernel(const float* __restrict__ mel_ref, h16* __restrict__ mel_tm, int n_mels,
                                                        int mel_rows) {
  const int b = blockIdx.y;
  const long total = (long)kFramesOut * n_mels;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    int f = (int)(i / n_mels), m = (int)(i - (long)f * n_mels);
    mel_tm[((long)b * mel_rows + f + 1) * n_mels + m] = (h16)mel_ref[((long)b * n_mels + m) * kFramesOut + f];
  }
}

__global__ void gmax_reset_kernel(unsigned* gmax, int batch) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < batch) gmax[i] = 0u;  // below every ordered-encoded float
}

void launch_frontend(const FrontendParams& p, hipStream_t s) {
  hipLaunchKernelGGL(gmax_reset_kernel, dim3((p.batch + 63) / 64), dim3(64), 0, s, p.gmax, p.batch);
  const int lds = FR * kNFFT * 4 + kNFFT * 8;
  dim3 grid((p.max_frames + FR - 1) / FR, p.batch);
  // basis is passed transposed ([201][n_mels]) by the engine in p.mel_basis
  hipLaunchKernelGGL(stft_mel_kernel, grid, dim3(256), lds, s, p, p.mel_basis);
  hipLaunchKernelGGL(mel_normalize_kernel, dim3(64, p.batch), dim3(256), 0, s, p);
}

void launch_mel_to_tm(const float* mel_ref, h16* mel_tm, int batch, int n_mels, int mel_rows, hipStream_t s) {
  hipLaunchKernelGGL(mel_to_tm_kernel, dim3(64, batch), dim3(256), 0, s, mel_ref, mel_tm, n_mels, mel_rows);
}

}  // inline namespace AXW_NS
}  // namespace axw
