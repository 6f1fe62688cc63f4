// whisper_cli.cpp — command-line entry point, MI355X build.
//
// Keeps the reference CLI's contract (cpp/whisper_cli.cpp:19-110): flags --wav/-w (required),
// --model_type/-t (default "turbo"), --model_path/-p, --language (no short flag), and the stdout
// lines "wav_file:", "model_path:", "model_type:", "language:", "Init whisper success, take
// %.4fseconds", "Result: %s", "RTF: %.4f" where RTF = wall time of AX_WHISPER_RunFile / true clip
// duration (:76,93-103). The AX_SYS_Init / AX_ENGINE_Init block (:37-61) is gone: the library
// initialises the GPU itself. Only the C ABI is used.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "ax_whisper_api.h"

static void usage(const char* prog) {
  fprintf(stderr,
          "usage: %s --wav=string [options] ...\noptions:\n"
          "  -w, --wav           wav file (string)\n"
          "  -t, --model_type    tiny, base, small, turbo, large (string [=turbo])\n"
          "  -p, --model_path    model path which contains tiny/ base/ small/ turbo/ (string [=../models-mi355x])\n"
          "      --language      en, zh (string [=zh])\n"
          "  -?, --help          print this message\n",
          prog);
}

// frame count of a RIFF/WAVE or FORM/AIFF file (the reference loads the file with AudioFile only to get
// duration = n_samples / 16000, whisper_cli.cpp:68-76)
static bool wav_frames(const char* path, long* frames) {
  FILE* f = fopen(path, "rb");
  if (!f) return false;
  unsigned char h[12];
  if (fread(h, 1, 12, f) != 12) { fclose(f); return false; }
  if (!memcmp(h, "FORM", 4) && (!memcmp(h + 8, "AIFF", 4) || !memcmp(h + 8, "AIFC", 4))) {  // AIFF: frames from the COMM chunk
    for (;;) {
      unsigned char c[8];
      if (fread(c, 1, 8, f) != 8) { fclose(f); return false; }
      const unsigned len = ((unsigned)c[4] << 24) | (c[5] << 16) | (c[6] << 8) | c[7];
      if (!memcmp(c, "COMM", 4)) {
        unsigned char cm[6];
        if (len < 18 || fread(cm, 1, 6, f) != 6) { fclose(f); return false; }
        fclose(f);
        *frames = (long)(((unsigned)cm[2] << 24) | (cm[3] << 16) | (cm[4] << 8) | cm[5]);
        return true;
      }
      fseek(f, (long)(len + (len & 1)), SEEK_CUR);
    }
  }
  if (memcmp(h, "RIFF", 4) || memcmp(h + 8, "WAVE", 4)) { fclose(f); return false; }
  int ch = 0, bits = 0;
  for (;;) {
    unsigned char c[8];
    if (fread(c, 1, 8, f) != 8) { fclose(f); return false; }
    unsigned len = c[4] | (c[5] << 8) | (c[6] << 16) | ((unsigned)c[7] << 24);
    if (!memcmp(c, "fmt ", 4)) {
      unsigned char fm[16];
      if (len < 16 || fread(fm, 1, 16, f) != 16) { fclose(f); return false; }
      ch = fm[2] | (fm[3] << 8);
      bits = fm[14] | (fm[15] << 8);
      fseek(f, (long)(len - 16 + (len & 1)), SEEK_CUR);
    } else if (!memcmp(c, "data", 4)) {
      fclose(f);
      const int frame_bytes = ch * (bits / 8);  // 0 for a bit depth below 8: not a format load_wav accepts either
      if (frame_bytes <= 0) return false;
      *frames = (long)len / frame_bytes;
      return true;
    } else {
      fseek(f, (long)(len + (len & 1)), SEEK_CUR);
    }
  }
}

int main(int argc, char** argv) {
  std::string wav, model_type = "turbo", model_path = "../models-mi355x", language = "zh";
  for (int i = 1; i < argc; ++i) {
    std::string a = argv[i];
    auto val = [&](const char* lng, const char* sht, std::string& dst) -> bool {
      std::string l = std::string("--") + lng;
      if (a.rfind(l + "=", 0) == 0) { dst = a.substr(l.size() + 1); return true; }
      if (a == l || (sht && a == sht)) {
        if (i + 1 >= argc) { fprintf(stderr, "option needs value: %s\n", a.c_str()); usage(argv[0]); exit(1); }
        dst = argv[++i];
        return true;
      }
      return false;
    };
    if (val("wav", "-w", wav) || val("model_type", "-t", model_type) || val("model_path", "-p", model_path) ||
        val("language", nullptr, language))
      continue;
    if (a == "--help" || a == "-?") { usage(argv[0]); return 0; }
    fprintf(stderr, "undefined option: %s\n", a.c_str());
    usage(argv[0]);
    return 1;
  }
  if (wav.empty()) { fprintf(stderr, "need option: --wav\n"); usage(argv[0]); return 1; }

  printf("wav_file: %s\n", wav.c_str());
  printf("model_path: %s\n", model_path.c_str());
  printf("model_type: %s\n", model_type.c_str());
  printf("language: %s\n", language.c_str());

  long frames = 0;
  if (!wav_frames(wav.c_str(), &frames) || frames <= 0) { printf("load wav failed!\n"); return -1; }
  const float duration = frames * 1.f / 16000;

  auto t0 = std::chrono::steady_clock::now();
  AX_WHISPER_HANDLE handle = AX_WHISPER_Init(model_type.c_str(), model_path.c_str(), language.c_str());
  auto t1 = std::chrono::steady_clock::now();
  if (!handle) { printf("AX_WHISPER_Init failed!\n"); return -1; }
  printf("Init whisper success, take %.4fseconds\n", std::chrono::duration<double>(t1 - t0).count());

  t0 = std::chrono::steady_clock::now();
  char* result = nullptr;
  if (0 != AX_WHISPER_RunFile(handle, wav.c_str(), &result)) {
    printf("AX_WHISPER_Run failed!\n");
    AX_WHISPER_Uninit(handle);
    return -1;
  }
  t1 = std::chrono::steady_clock::now();
  printf("Result: %s\n", result);
  printf("RTF: %.4f\n", std::chrono::duration<double>(t1 - t0).count() / duration);
  free(result);
  AX_WHISPER_Uninit(handle);
  return 0;
}
