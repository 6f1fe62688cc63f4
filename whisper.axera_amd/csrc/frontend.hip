// frontend.hip — log-mel front-end on gfx950 (K1/K2 of SURVEY §8a).
//
// Replaces librosa::Feature::melspectrogram (cpp/src/librosa/librosa.h:46-155) and the
// clamp/normalise/pad of Whisper::preprocess (cpp/src/Whisper.cpp:151-184):
//   reflect pad 200 | periodic Hann | 400-pt DFT bins 0..200 | re^2+im^2 | Slaney mel GEMM |
//   log10(max(.,1e-10)) | global max over ALL frames | max(., gmax-8) | (.+4)/4 | zero-fill to 3000
// "ALL frames" means all frames of the input however long it is (Whisper.cpp:158-172 takes the maximum before it
// truncates to 3000 frames): the grid covers every frame of the longest clip, frames past 3000 only feed the maximum.
//
// Kernel 1 (stft_mel_kernel): one workgroup = 32 consecutive frames of one clip. The windowed frames are staged in
// LDS ([32][401] f32, 51 KB) and the 400-point DFT runs on the matrix cores as an exact-fp32 GEMM
//   [32 frames x 400 samples] . [400 x (cos | sin) of 224 bins]     (v_mfma_f32_32x32x2_f32, fp32 in, fp32 accumulate:
// the arithmetic class of the FMA loop it replaces — rounds 1-3 evaluated the DFT with 64 scalar accumulators per lane at
// 0.18 of the fp32 vector peak); the twiddle operand is never materialised: a lane walks a 400-entry (cos, sin) table in
// LDS with its own stride (k * bin mod 400). The power spectrum goes back to LDS and the mel projection + log10 + clip
// maximum are fused in.
// Kernel 2 (mel_normalize_kernel): clamp/scale/zero-fill and layout: time-major h16 rows for the
// encoder's conv-as-GEMM (and the reference's [n_mels][3000] f32 layout when a caller asks for it).
// HBM traffic per clip: 1.92 MB PCM in, 0.96 MB log-mel scratch out+in, 0.48 MB h16 out.
#include "common.hpp"

namespace axw {
inline namespace AXW_NS {

constexpr int FR = 32;        // frames per workgroup = rows of one MFMA tile
constexpr int PW_LD = 209;    // power row stride in LDS: odd (column reads of the mel GEMM), >= 202 (bins padded to an even count)
constexpr int XS = kNFFT + 1; // row stride of the staged frames: odd, so a column read (32 frames, one sample) hits 32 banks
constexpr int kBinGroups = (kBins + 31) / 32;  // 7 groups of 32 bins

__device__ __forceinline__ unsigned float_to_ordered(float f) {
  unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ordered_to_float(unsigned u) {
  return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}

__global__ __launch_bounds__(256) void stft_mel_kernel(FrontendParams p, const float* __restrict__ basis_t /*[201][n_mels]*/) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* xw = reinterpret_cast<float*>(smem);                 // [FR][XS] windowed frames; later power [FR][PW_LD]
  float2* tw = reinterpret_cast<float2*>(smem + FR * XS * 4);  // [400] (cos, sin)
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
  // Stage the 32 windowed frames. Interior workgroups (no reflection at either end of the clip, every sample inside the
  // staging row) take the plain path: k walks the frame, f the frames, no division and no per-sample tests, so the loads
  // of an iteration are independent and go out together.
  const int j_first = f0 * kHop - kNFFT / 2, j_last = (f0 + FR - 1) * kHop + kNFFT - 1 - kNFFT / 2;
  if (j_first >= 0 && j_last < min(n_real, p.stride) && f0 + FR <= n_frames) {  // uniform per workgroup
    const float* xs = x + j_first;
    for (int k = tid; k < kNFFT; k += 256) {
      const float wk = p.window[k];
#pragma unroll 8
      for (int f = 0; f < FR; ++f) xw[f * XS + k] = xs[f * kHop + k] * wk;
    }
  } else {
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
      xw[f * XS + k] = v;
    }
  }
  __syncthreads();

  // ---- DFT on the matrix cores. One v_mfma_f32_32x32x2_f32 multiplies A = 32 frames x 2 samples (lane: frame lane % 32,
  // sample 2s + lane / 32) by B = 2 samples x 32 bins (lane: bin lane % 32) into a 32 x 32 fp32 tile whose lanes run along
  // the bins and whose registers run along the frames; the cos and the sin tile of the same 32 bins share lane and
  // register, so re^2 + im^2 is register-local. 201 bins = 7 groups of 32: wave w takes groups w and w + 4.
  // B is read from the twiddle table: bin n at sample k needs entry (k * n) mod 400, and k advances by 2 per step.
  const int lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int n_groups = (w + 4 < kBinGroups) ? 2 : 1;  // wave-uniform
  f32x16 acc[2][2];  // [group][cos | sin]
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[g][c][e] = 0.f;
  const int bin0 = w * 32 + r, bin1 = (w + 4) * 32 + r;
  int idx0 = (h * bin0) % kNFFT, idx1 = (h * bin1) % kNFFT;
  const int inc0 = (2 * bin0) % kNFFT, inc1 = (2 * bin1) % kNFFT;
  const float* xa = xw + r * XS + h;
  if (n_groups == 2) {
#pragma unroll 4
    for (int s2 = 0; s2 < kNFFT / 2; ++s2) {
      const float a = xa[2 * s2];
      const float2 t0 = tw[idx0], t1 = tw[idx1];
      idx0 += inc0; idx0 -= idx0 >= kNFFT ? kNFFT : 0;
      idx1 += inc1; idx1 -= idx1 >= kNFFT ? kNFFT : 0;
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, t0.x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, t0.y, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, t1.x, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, t1.y, acc[1][1], 0, 0, 0);
    }
  } else {
#pragma unroll 4
    for (int s2 = 0; s2 < kNFFT / 2; ++s2) {
      const float a = xa[2 * s2];
      const float2 t0 = tw[idx0];
      idx0 += inc0; idx0 -= idx0 >= kNFFT ? kNFFT : 0;
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, t0.x, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, t0.y, acc[0][1], 0, 0, 0);
    }
  }
  __syncthreads();  // all waves are done reading xw
  float* pw = xw;   // power [FR][PW_LD]
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int bin = g ? bin1 : bin0;
    if (g < n_groups && bin <= kBins) {  // bin 201 (of the last group) is the zero that pads the bins to an even count
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int f = (e & 3) + 8 * (e >> 2) + 4 * h;
        const float pv = acc[g][0][e] * acc[g][0][e] + acc[g][1][e] * acc[g][1][e];  // librosa.h:98-100
        pw[f * PW_LD + bin] = bin < kBins ? pv : 0.f;
      }
    }
  }
  __syncthreads();

  // ---- mel projection (librosa.h:153) + log10 (Whisper.cpp:160) + clip maximum (:162-164), also on the matrix cores:
  // [32 frames x 202 bins] . [202 x 32 mels] per wave (wave w: mels 32w .. 32w + 31), the filterbank operand straight from
  // global memory (64-103 KB shared by every workgroup: L1 / L2 hits), lanes along the mels = the contiguous axis of logmel.
  const int nm = p.n_mels;
  float lmax = -3.402823466e38f;
  if (w * 32 < nm) {
    const int m = w * 32 + r;
    const bool m_ok = m < nm;
    f32x16 macc;
#pragma unroll
    for (int e = 0; e < 16; ++e) macc[e] = 0.f;
    const float* pa = pw + r * PW_LD + h;
    const float* bb = basis_t + (long)h * nm + (m_ok ? m : 0);
#pragma unroll 4
    for (int s2 = 0; s2 < (kBins + 1) / 2; ++s2) {
      const int k = 2 * s2 + h;
      const float a = pa[2 * s2];                                  // bin 201 of every row is zero (written above)
      const float bv = (m_ok && k < kBins) ? bb[(long)2 * s2 * nm] : 0.f;
      macc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bv, macc, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int f = (e & 3) + 8 * (e >> 2) + 4 * h;
      if (!m_ok || f0 + f >= n_frames) continue;
      const float v = log10f(fmaxf(macc[e], 1e-10f));
      lmax = fmaxf(lmax, v);
      if (f0 + f < kFramesOut) p.logmel[((long)b * kFramesOut + f0 + f) * nm + m] = v;
    }
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
  const int lds = FR * XS * 4 + kNFFT * 8;
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
