// engine_impl.hpp — what the translation units of axw::Engine share besides engine.hpp (engine.cpp: construction, weights,
// front-end, encoder, the entry points; engine_decode.cpp: the decode paths; engine_stream.cpp: utterance slots, bench hooks).
#pragma once
#include "engine.hpp"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>

namespace axw {
inline namespace AXW_NS {


#define HIP_CHECK(expr)                                                                                  \
  do {                                                                                                   \
    hipError_t _e = (expr);                                                                              \
    if (_e != hipSuccess)                                                                                \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " #expr);       \
  } while (0)

// Launch-per-row-block form of the batched vocabulary projection (used where the register-resident form does not fit:
// d_model 1280 beyond 48 clips): weight-row tiles of 16 rows per workgroup, two per wave (1 / 2 / 4 measured alike).
static int logits_rt() { return 2; }

}  // inline namespace AXW_NS
}  // namespace axw
