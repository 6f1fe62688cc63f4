// ref_frontend_driver.cpp — builds the REFERENCE's own log-mel front-end into oracle/_ref/.
//
// TEST INFRASTRUCTURE ONLY. This file is ours; the reference sources it compiles against
// (cpp/src/librosa/librosa.h + vendored Eigen/kissfft) stay where they lie under
// /root/reference and are never copied into this repository. Built only in the container
// that has /root/reference (oracle/Makefile target `ref`); outputs go to oracle/_ref/ only.
//
// librosa::Feature::melspectrogram is called verbatim with the arguments of
// Whisper::preprocess (cpp/src/Whisper.cpp:153). Whisper.cpp itself cannot be compiled
// (it includes the closed AXera BSP headers), so its 20 post-processing lines
// (cpp/src/Whisper.cpp:157-181) are restated below.
#include <librosa/librosa.h>

#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

extern "C" int ref_log_mel(const float* pcm, int n_samples, int n_mels, float* out /*[n_mels*3000]*/,
                           float* mmax_out) {
  std::vector<float> audio(pcm, pcm + n_samples);
  auto mel = librosa::Feature::melspectrogram(audio, 16000, 400, 160, "hann", true, "reflect", 2.0f,
                                              n_mels, 0.0f, 16000 / 2.0f);
  int n_len = (int)mel[0].size();
  int n_frames = n_len;
  float mmax = -std::numeric_limits<float>::max();
  for (int i = 0; i < n_mels; i++)
    for (int n = 0; n < n_len; n++) {
      mel[i][n] = std::log10(std::max(mel[i][n], 1e-10f));
      if (mel[i][n] > mmax) mmax = mel[i][n];
    }
  for (int i = 0; i < n_mels; i++) {
    for (int n = 0; n < n_len && n < 3000; n++)
      mel[i][n] = (std::max(mel[i][n], (float)(mmax - 8.0)) + 4.0) / 4.0;
    mel[i].resize(3000);  // zero fill (or truncate) exactly as Whisper.cpp:172
  }
  for (int i = 0; i < n_mels; i++) std::memcpy(out + (size_t)i * 3000, mel[i].data(), sizeof(float) * 3000);
  if (mmax_out) *mmax_out = mmax;
  return n_frames;
}

// raw mel filterbank of librosa.h:102-144 for a direct comparison with the restatement
extern "C" void ref_mel_filterbank(int n_mels, float* out /*[n_mels*201]*/) {
  librosa::Matrixf w = librosa::internal::melfilter(16000, 400, n_mels, 0, 8000);
  for (int i = 0; i < n_mels; ++i)
    for (int k = 0; k < 201; ++k) out[i * 201 + k] = w(i, k);
}
