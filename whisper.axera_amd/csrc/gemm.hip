// gemm.hip — h16 MFMA GEMM with fused epilogues for the encoder (K3, K4, K6, K8 of SURVEY §8a).
//
// C[M,N] = A[M,K] * W[N,K]^T, A and W h16 with K contiguous (nn.Linear weights are [out][in],
// so no transposition is ever needed), fp32 accumulation on the matrix cores.
//
// Replaces, inside the reference's opaque encoder blob (cpp/src/ax_model_runner/
// ax_model_runner.cpp:95-108 running export_onnx.py:153-213): conv1/conv2 (as GEMMs over
// overlapping time-major rows: A row t = 3 consecutive input frames, lda = stride*C_in),
// every nn.Linear of the encoder blocks and the per-decoder-layer cross K/V projections.
//
// Kernels, chosen per launch by how many tiles it has (launch_one):
//   gemm_bf16_kernel       128x128, 4 waves, two LDS buffers (this comment)        few tiles: one to four clips
//   gemm256_bf16_kernel    256x128, 8 waves, 3-stage LDS-DMA ring                  mid-size launches
//   gemm256ps_bf16_kernel  256x256, phased 16x16x32 k-loop, one k-tile STREAM per CU the batched encoder (>= 256 tiles)
// (Two more 256x256 kernels of rounds 1-2 — a two-stage 32x32x16 loop and the phased loop with one tile per workgroup —
// served only odd k-tile counts and operands beyond 4 GB; those corner cases now take the 256x128 kernel.)
// All share the swizzled LDS image, the tile order and the epilogues (epilogue_rows).
// Tiling of the smallest (wave64): 128x128 output tile, BK = 64, 256 threads = 4 waves in 2x2, each wave a
// 64x64 sub-tile = 2x2 MFMA 32x32 tiles (64 accumulator VGPRs). A and W tiles are staged by LDS-DMA
// (global_load_lds, no VGPR round trip) with an XOR swizzle of the 16-byte chunks (chunk ^ ((row>>1)&7)),
// applied on the source address, which makes every ds_read_b128 fragment read conflict-free; k-tile t+1
// streams into the other LDS buffer while k-tile t is multiplied (one drain + barrier per k-tile). Epilogues write
// the layouts the consumers want (V^T for the encoder attention, the blocked K / row-major V of the decoder's
// cross-attention) so no transposition kernel exists; where the consumer wants M contiguous the MFMA operands are
// swapped so lanes run along M. Row-major outputs leave through LDS (gemm_epilogue): 16-byte row accesses.
#include "common.hpp"

#include <atomic>
#include <type_traits>

namespace axw {
inline namespace AXW_NS {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KB per operand tile

__device__ __forceinline__ int swz(int row, int chunk) { return row * (BK * 2) + 16 * (chunk ^ ((row >> 1) & 7)); }
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gptr_t;

// Second half of the LDS epilogue: the wave's parked 64x64 fp32 sub-tile (lw[row * 64 + column], row = the output's
// strided axis) leaves as whole rows. Shared by the 32x32 and the 16x16 accumulator layouts (they differ in the parking).
// NT = row steps of 4 rows: 16 for a 64-row sub-tile, 4 for the 16-row slabs of the persistent kernel.
template <int EPI, bool SWAPPED, int NT = 16>
__device__ __forceinline__ void epilogue_rows(const GemmParams& p, const float* lw, int mb, int nb, int bz, int lane) {
  const int d = p.d_model;
  // read back rows: lane -> row 4t + (lane>>4), columns 4*(lane&15) .. +3
  const int rr = lane >> 4, cc = (lane & 15) * 4;
  if constexpr (!SWAPPED) {  // rows = m, columns = n
    const int n = nb + cc;
    f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
    if (p.bias) bias4 = *reinterpret_cast<const f32x4*>(p.bias + n);
    if constexpr (EPI == EPI_PARTIAL_F32) {
      float* slab = p.part + (long)blockIdx.y * p.part_stride + (long)bz * p.M * p.N;
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int row = 4 * t + rr, m = mb + row;
        if (m < p.M) *reinterpret_cast<f32x4*>(slab + (long)m * p.N + n) = *reinterpret_cast<const f32x4*>(lw + row * 64 + cc);
      }
    } else if constexpr (EPI == EPI_RESID_F32) {
      f32x4 cur[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int m = min(mb + 4 * t + rr, p.M - 1);
        cur[t] = *reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(p.C) + (long)bz * p.c_batch_stride + (long)m * p.ldc + n);
      }
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int row = 4 * t + rr, m = mb + row;
        const f32x4 v = *reinterpret_cast<const f32x4*>(lw + row * 64 + cc) + bias4 + cur[t];
        if (m < p.M) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)bz * p.c_batch_stride + (long)m * p.ldc + n) = v;
      }
    } else if constexpr (EPI == EPI_GELU_POS_F32) {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int row = 4 * t + rr, m = mb + row;
        if (m >= p.M) continue;
        f32x4 v = *reinterpret_cast<const f32x4*>(lw + row * 64 + cc) + bias4;
        const f32x4 pos = *reinterpret_cast<const f32x4*>(p.aux + (long)m * p.N + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = gelu_erf(v[e]) + pos[e];
        *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.C) + (long)bz * p.c_batch_stride + (long)m * p.ldc + n) = v;
      }
    } else {
      // h16 outputs: lane -> row 8t + (lane>>3), 8 consecutive columns = one 16-byte store (a wave instruction writes
      // 8 whole 128-byte row segments; with 8-byte stores the epilogue's tail is bound by store issue, guide T21)
      const int r8 = lane >> 3, c8 = (lane & 7) * 8;
      const int n8 = nb + c8;
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) {
        b0 = *reinterpret_cast<const f32x4*>(p.bias + n8);
        b1 = *reinterpret_cast<const f32x4*>(p.bias + n8 + 4);
      }
#pragma unroll
      for (int t = 0; t < NT / 2; ++t) {
        const int row = 8 * t + r8, m = mb + row;
        if (m >= p.M) continue;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(lw + row * 64 + c8) + b0;
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(lw + row * 64 + c8 + 4) + b1;
        h16x8 o;
        if constexpr (EPI == EPI_BIAS_GELU_BF16) {
          const f32x2_t g0 = gelu_erf_fast2(f32x2_t{v0[0], v0[1]}), g1 = gelu_erf_fast2(f32x2_t{v0[2], v0[3]});
          const f32x2_t g2 = gelu_erf_fast2(f32x2_t{v1[0], v1[1]}), g3 = gelu_erf_fast2(f32x2_t{v1[2], v1[3]});
          o[0] = (h16)g0[0]; o[1] = (h16)g0[1]; o[2] = (h16)g1[0]; o[3] = (h16)g1[1];
          o[4] = (h16)g2[0]; o[5] = (h16)g2[1]; o[6] = (h16)g3[0]; o[7] = (h16)g3[1];
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) { o[e] = (h16)v0[e]; o[4 + e] = (h16)v1[e]; }
        }
        h16* dst;
        if constexpr (EPI == EPI_QKV) {  // Q (n < d) or K (d <= n < 2d), both h16 [m][d]
          dst = n8 < d ? reinterpret_cast<h16*>(p.C) + (long)bz * p.c_batch_stride + (long)m * d + n8
                       : reinterpret_cast<h16*>(p.C2) + (long)bz * p.c2_batch_stride + (long)m * d + (n8 - d);
        } else if constexpr (EPI == EPI_CROSS_KV) {  // V rows [n_layer*d, 2*n_layer*d): [l][slot][head][t_pad][64]
          const int nv = n8 - p.n_layer * d;
          const int l = nv / d, c = nv - l * d;
          const long slot = ((long)l * p.n_batch_total + (p.kv_slot_map ? p.kv_slot_map[bz] : bz)) * (d >> 6) + (c >> 6);
          dst = reinterpret_cast<h16*>(p.C2) + (slot * p.t_pad + m) * 64 + (c & 63);
        } else {
          dst = reinterpret_cast<h16*>(p.C) + (long)bz * p.c_batch_stride + (long)m * p.ldc + n8;
        }
        *reinterpret_cast<h16x8*>(dst) = o;
      }
    }
  } else {  // EPI_QKV swapped: rows = n (V^T [head][64][t_pad]: row c = n - 2d), columns = m
    static_assert(EPI == EPI_QKV, "swapped row epilogue: V^T only");
    // The frames of a 16-group are stored in the order [0-3, 8-11, 4-7, 12-15]: a lane of the attention kernel then finds
    // the 8 keys of its P^T operand fragment in ONE 16-byte chunk (encoder_attn.hip). A lane here takes such a chunk:
    // frames mA..mA+3 and mA+8..mA+11 of its 16-group -> positions 8*hs .. 8*hs+7 (M % 4 == 0, checked by launch_gemm;
    // mb % 16 == 0); a group of frames at or beyond M is not written (the padding stays zero).
    const int r8 = lane >> 3, g16 = ((lane & 7) >> 1) * 16, hs = lane & 1;
    const int cA = g16 + hs * 4, mA = mb + cA;
    if (mA < p.M) {
#pragma unroll
      for (int t = 0; t < NT / 2; ++t) {
        const int row = 8 * t + r8, n = nb + row;
        const float bias = p.bias ? p.bias[n] : 0.f;
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(lw + row * 64 + cA);
        const f32x4 v1 = *reinterpret_cast<const f32x4*>(lw + row * 64 + cA + 8);
        h16x8 o;
#pragma unroll
        for (int e = 0; e < 4; ++e) { o[e] = (h16)(v0[e] + bias); o[4 + e] = (h16)(v1[e] + bias); }
        h16* dst = reinterpret_cast<h16*>(p.C3) + (long)bz * p.c3_batch_stride + (long)(n - 2 * d) * p.t_pad + mb + g16 + hs * 8;
        if (mA + 8 < p.M) {
          *reinterpret_cast<h16x8*>(dst) = o;
        } else {
          h16x4 lo;
#pragma unroll
          for (int e = 0; e < 4; ++e) lo[e] = o[e];
          *reinterpret_cast<h16x4*>(dst) = lo;
        }
      }
    }
  }
}

// Epilogue of one wave's 64x64 sub-tile (2x2 accumulator tiles), shared by both tile shapes.
//   normal:  acc[i][j][e] = C[m = mb + i*32 + row(e,h)][n = nb + j*32 + r]
//   swapped: acc[i][j][e] = C[m = mb + i*32 + r][n = nb + j*32 + row(e,h)],  row(e,h) = (e&3) + 8*(e>>2) + 4*h
// In both orientations a lane holds single elements of 16 different output rows, so writing from the registers
// costs 64 narrow store instructions per wave (and, for the in-place residual add, 64 dependent load -> add -> store
// round trips: 33 us per 256x128 tile against 16 us for its whole k-loop). Instead every wave parks its sub-tile
// as fp32 in its own 16 KB of the staging buffers (idle once the k-loop is done) with the lanes along the output's
// CONTIGUOUS axis, and reads it back as rows: one 16-byte ds_read per lane = 4 consecutive outputs, a wave
// instruction = 4 full rows of the sub-tile, every global access on whole 128/256-byte row segments, all loads of
// the residual add issued before the first store. The one layout that is contiguous along neither axis of the
// accumulator tile (the decoder's blocked cross K) keeps the register path: its lanes already store 8 contiguous
// bytes each, 512 contiguous bytes per wave instruction.
// NI = 32-row accumulator tiles per wave along M, I0 = the first of the two that this call writes.
template <int EPI, bool SWAPPED, int NI = 2, int I0 = 0>
__device__ __forceinline__ void gemm_epilogue(const GemmParams& p, f32x16 (&accw)[NI][2], float* lw, int mb, int nb, int bz, int lane) {
  f32x16 (&acc)[2][2] = *reinterpret_cast<f32x16 (*)[2][2]>(&accw[I0]);
  const int r = lane & 31, h = lane >> 5;
  const int d = p.d_model;
  if constexpr (EPI == EPI_CROSS_KV && SWAPPED) {  // K rows [0, n_layer*d): blocked [l][slot][head][m/64][dd/8][m%64][8]
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = mb + i * 32 + r;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int n = nb + j * 32 + 8 * q + 4 * h;
          const int l = n / d, c = n - l * d;
          const int head = c >> 6, dd = c & 63;
          h16x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[e] = (h16)(acc[i][j][4 * q + e] + (p.bias ? p.bias[n + e] : 0.f));
          const long slot = ((long)l * p.n_batch_total + (p.kv_slot_map ? p.kv_slot_map[bz] : bz)) * (d >> 6) + head;
          h16* dst = reinterpret_cast<h16*>(p.C) + slot * p.t_pad * 64 + (long)(m >> 6) * 4096 + (dd >> 3) * 512 + (m & 63) * 8 + (dd & 7);
          *reinterpret_cast<h16x4*>(dst) = pk;
        }
      }
    }
  } else {
    // park: LDS row = index along the accumulator's register axis, LDS column = lane axis (conflict-free b32 stores)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) {
          const int reg_ax = (SWAPPED ? j : i) * 32 + (e & 3) + 8 * (e >> 2) + 4 * h;
          const int lane_ax = (SWAPPED ? i : j) * 32 + r;
          lw[reg_ax * 64 + lane_ax] = acc[i][j][e];
        }
    epilogue_rows<EPI, SWAPPED>(p, lw, mb, nb, bz, lane);
  }
}

template <int EPI, bool SWAPPED>
__global__ __launch_bounds__(256, 2) void gemm_bf16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* As = smem;                     // [2][TILE_BYTES]
  char* Ws = smem + 2 * TILE_BYTES;    // [2][TILE_BYTES]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  // XCD-aware tile order (1-D grid): workgroups are dealt round-robin over the 8 XCDs (each with a private L2), so
  // give every XCD a CONTIGUOUS chunk of the (clip, row-tile, column-tile) space with the column tile fastest: the
  // N/128 workgroups that share one 128-row A panel then run on one XCD and hit its L2 (bijective for any grid size).
  int n0, m0, bz;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    const int nt = p.n_tiles, mt = (p.M + BM - 1) / BM;
    n0 = p.n_begin + (wg % nt) * BN;
    const int rest = wg / nt;
    m0 = (rest % mt) * BM;
    bz = rest / mt;
  }

  const h16* A = p.A + (long)bz * p.a_batch_stride;
  const h16* W = p.W;

  // Staging: LDS-DMA (global_load_lds, 16 B per lane). One wave-instruction lands 1 KiB = 8 tile rows x 128 B
  // LINEARLY in LDS (hardware: wave-uniform base + lane*16), so the bank-conflict swizzle is applied on the SOURCE
  // side: LDS chunk position c of row `row` receives global chunk c ^ ((row>>1)&7), and fragment reads look for
  // global chunk g at position g ^ ((row>>1)&7) (the same involution; swz()). No VGPR staging, no ds_write.
  const int ld_row = tid >> 3, ld_c = tid & 7;   // lane l of wave w: row 8w + l/8 (+32 i), position l%8
  const h16* a_src[4];
  const h16* w_src[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = ld_row + 32 * i;
    const int gc = ld_c ^ ((row >> 1) & 7);
    const int ra = min(m0 + row, p.M - 1);
    a_src[i] = A + (long)ra * p.lda + gc * 8;
    w_src[i] = W + (long)(n0 + row) * p.K + gc * 8;
  }
  auto stage = [&](int buf, int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int off = buf * TILE_BYTES + (32 * i + 8 * wave) * (BK * 2);  // wave-uniform LDS base of this 1 KiB piece
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + kt * BK), (lds_ptr_t)(As + off), 16, 0, 0);
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + kt * BK), (lds_ptr_t)(Ws + off), 16, 0, 0);
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // split-K (EPI_PARTIAL_F32, grid.y slices): this workgroup multiplies k-tiles [k_first, k_first + nk)
  const int nk = EPI == EPI_PARTIAL_F32 ? p.K / BK / p.ksplit : p.K / BK;
  const int k_first = EPI == EPI_PARTIAL_F32 ? (int)blockIdx.y * nk : 0;
  stage(0, k_first);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // One k-tile of MFMAs out of LDS buffer `cur`: the fragment reads of k-step s+1 are issued before the MFMAs of
  // k-step s (two fragment register sets), so LDS latency hides under the matrix pipe.
  auto compute = [&](int cur) {
    const char* Ab = As + cur * TILE_BYTES;
    const char* Wb = Ws + cur * TILE_BYTES;
    h16x8 af[2][2], wf[2][2];
    auto frags = [&](int s, int set) {
#pragma unroll
      for (int i = 0; i < 2; ++i) af[set][i] = *reinterpret_cast<const h16x8*>(Ab + swz(wm * 64 + i * 32 + r, 2 * s + h));
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[set][j] = *reinterpret_cast<const h16x8*>(Wb + swz(wn * 64 + j * 32 + r, 2 * s + h));
    };
    frags(0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) frags(s + 1, (s + 1) & 1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = SWAPPED ? AXW_MFMA_32x32x16(wf[s & 1][j], af[s & 1][i], acc[i][j])
                              : AXW_MFMA_32x32x16(af[s & 1][i], wf[s & 1][j], acc[i][j]);
    }
  };
  // k-tile kt+1 streams into the other LDS buffer while k-tile kt is multiplied; one drain + barrier per k-tile.
  for (int kt = 0; kt + 1 < nk; ++kt) {
    const int cur = kt & 1;
    stage(cur ^ 1, k_first + kt + 1);
    compute(cur);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  compute((nk - 1) & 1);

  __syncthreads();  // every wave is done with the staging buffers: they now hold the parked sub-tiles
  gemm_epilogue<EPI, SWAPPED>(p, acc, reinterpret_cast<float*>(smem) + wave * 4096, m0 + wm * 64, n0 + wn * 64, bz, lane);
}

// ---------------------------------------------------------------------------- 256 x 128 tile, 3-stage LDS-DMA ring
// Used when the launch has enough tiles to fill the chip (batched encoder). Same per-wave 64x64 sub-tile and the
// same swizzled LDS image as above, but 8 waves (4 along M x 2 along N) share a 256x128 output tile, so a k-tile
// costs 48 KB of staging for twice the MFMAs (1.33x the arithmetic intensity), and the ring is three k-tiles deep:
// two k-tiles (96 KB) are in flight while one is multiplied. The waits are COUNTED (`s_waitcnt vmcnt(6)` leaves the
// youngest k-tile's 6 LDS-DMA pieces per wave in flight) and the barrier is a raw `s_barrier`: `__syncthreads()`
// would drain the LDS-DMA queue. A staged buffer is read only after [own vmcnt wait -> barrier], and refilled only
// after the barrier that follows its last read (cdna guide: "read a staged buffer one phase after the wait").
constexpr int BM2 = 256;
constexpr int A2_BYTES = BM2 * BK * 2;   // 32 KB
constexpr int W2_BYTES = BN * BK * 2;    // 16 KB
constexpr int STAGE2_BYTES = A2_BYTES + W2_BYTES;

template <int EPI, bool SWAPPED>
__global__ __launch_bounds__(512) void gemm256_bf16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [3][A 32 KB | W 16 KB]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int r = lane & 31, h = lane >> 5;
  int n0, m0, bz;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, qq = nwg >> 3, rr = nwg & 7;
    const int wg = (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (orig >> 3);
    const int nt = p.n_tiles, mt = (p.M + BM2 - 1) / BM2;
    n0 = p.n_begin + (wg % nt) * BN;
    const int rest = wg / nt;
    m0 = (rest % mt) * BM2;
    bz = rest / mt;
  }
  const h16* A = p.A + (long)bz * p.a_batch_stride;
  const h16* W = p.W;

  const int ld_row = tid >> 3, ld_c = tid & 7;  // lane l of wave w: tile row 8w + l/8 (+64 i), chunk position l%8
  const h16* a_src[4];
  const h16* w_src[2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = ld_row + 64 * i;
    a_src[i] = A + (long)min(m0 + row, p.M - 1) * p.lda + (ld_c ^ ((row >> 1) & 7)) * 8;
  }
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = ld_row + 64 * i;
    w_src[i] = W + (long)(n0 + row) * p.K + (ld_c ^ ((row >> 1) & 7)) * 8;
  }
  auto stage = [&](int buf, int kt) {  // 6 LDS-DMA pieces (1 KiB each) per wave
    char* base = smem + buf * STAGE2_BYTES;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(a_src[i] + kt * BK), (lds_ptr_t)(base + (64 * i + 8 * wave) * (BK * 2)), 16, 0, 0);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      __builtin_amdgcn_global_load_lds((gptr_t)(w_src[i] + kt * BK), (lds_ptr_t)(base + A2_BYTES + (64 * i + 8 * wave) * (BK * 2)), 16, 0, 0);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto compute = [&](int buf) {
    const char* Ab = smem + buf * STAGE2_BYTES;
    const char* Wb = Ab + A2_BYTES;
    h16x8 af[2][2], wf[2][2];
    auto frags = [&](int s, int set) {
#pragma unroll
      for (int i = 0; i < 2; ++i) af[set][i] = *reinterpret_cast<const h16x8*>(Ab + swz(wm * 64 + i * 32 + r, 2 * s + h));
#pragma unroll
      for (int j = 0; j < 2; ++j) wf[set][j] = *reinterpret_cast<const h16x8*>(Wb + swz(wn * 64 + j * 32 + r, 2 * s + h));
    };
    frags(0, 0);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if (s < 3) frags(s + 1, (s + 1) & 1);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = SWAPPED ? AXW_MFMA_32x32x16(wf[s & 1][j], af[s & 1][i], acc[i][j])
                              : AXW_MFMA_32x32x16(af[s & 1][i], wf[s & 1][j], acc[i][j]);
    }
  };

  const int nk = p.K / BK;
  stage(0, 0);
  if (nk > 1) stage(1, 1);
  for (int kt = 0; kt < nk; ++kt) {
    // k-tile kt has landed once at most the younger k-tile's 6 pieces are still outstanding
    if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();  // every wave's pieces of k-tile kt are in LDS; every wave is done reading k-tile kt-1
    if (kt + 2 < nk) stage((kt + 2) % 3, kt + 2);  // refills the buffer k-tile kt-1 was read from
    compute(kt % 3);
  }
  __syncthreads();  // every wave is done with the ring: it now holds the parked sub-tiles (8 x 16 KB)
  gemm_epilogue<EPI, SWAPPED>(p, acc, reinterpret_cast<float*>(smem) + wave * 4096, m0 + wm * 64, n0 + wn * 64, bz, lane);
}

// ---------------------------------------------------------------------------- 256 x 256 tile
// For launches with enough 256-square tiles to fill the chip: 8 waves as 2 (M) x 4 (N), each a 128x64 sub-tile; a k-tile
// costs 64 KB of staging for 128 FLOP per staged byte (256x128: 85) — the k-loop of these kernels is bound by how many
// staged bytes can be in flight, not by the matrix pipe.
constexpr int BM3 = 256, BN3 = 256;

// ---------------------------------------------------------------------------- 256 x 256 tile, phased k-loop
// The batched encoder's kernel (launches with at least one 256-square tile per CU and an even number of k-tiles).
// Same tile, same 128x64 per wave and the same swizzled LDS rows as above; what changes is the k-loop (the "8-phase"
// schedule of the CDNA4 guide, section 5): v_mfma_f32_16x16x32 instead of 32x32x16, a k-tile cut into FOUR phases of 16
// MFMAs per wave (one 64x32 quadrant of the wave's sub-tile), every phase = [fragment reads + one 16 KB half-tile of
// LDS-DMA prefetch] -> barrier -> [16 MFMAs] -> barrier, and the two wave groups (rows 0-127 / 128-255 of the tile:
// one wave of each per SIMD) run ONE BARRIER APART, so a SIMD's matrix pipe always has one group's MFMA cluster while
// the other group's reads and DMA issues run in its shadow. The LDS-DMA is never drained inside the loop: one counted
// wait per k-tile (vmcnt(6) = three half-tiles stay in flight across the barriers).
//
// An operand's k-tile (256 rows x 64 k) is staged as two half-tiles of 128 rows, cut by what a PHASE reads, not by tile
// half: A half h = rows {wr*128 + h*64 + 0..63}, W half h = rows {wc*64 + h*32 + 0..31} (wr, wc: wave row / column), so
//   phase 1 reads W half 0 (4 ds_read_b128, kept for phase 4) then A half 0 (8)   -> quadrant (0,0)
//   phase 2 reads W half 1 (4)                                                   -> quadrant (0,1)
//   phase 3 reads A half 1 (8, over A half 0's registers)                        -> quadrant (1,1)
//   phase 4 reads nothing                                                        -> quadrant (1,0)
// and a half-tile's LDS is free for the k-tile after next as soon as its phase is over. Prefetch issue order, k-tile t:
//   phase 1: A1(t+1)   phase 2: W0(t+2)   phase 3: A0(t+2)   phase 4: W1(t+2), then vmcnt(6): all of k-tile t+1 landed.
// Hazards (guide, "read a staged buffer one phase AFTER the wait that retires it"):
//   RAW  both groups' counted waits sit before the first barrier of phase 4; the first read of k-tile t+1 (group 0,
//        phase 1) comes after the second barrier of phase 4, which group 1 reaches only past its own wait.
//   WAR  W0 is re-staged one phase after its reads, which `lgkmcnt(8)` retires before phase 1's first barrier (the W
//        reads are issued first); A0, W1, A1 are re-staged two phases after their reads (retired by the lgkmcnt(0)
//        in front of that phase's MFMAs).
constexpr int HALF_BYTES = 128 * BK * 2;    // 16 KB
constexpr int KT4_BYTES = 4 * HALF_BYTES;   // one k-tile: A0 | A1 | W0 | W1

template <int EPI, bool SWAPPED>
__device__ __forceinline__ void gemm_epilogue16(const GemmParams& p, f32x4 (&acc)[2][4][2], float* lw, int mb, int nb, int bz, int lane) {
  // 16x16 accumulator tiles of one 64-row half of the wave's sub-tile: acc[nj][it][jt][e] =
  //   normal  C[m = mb + it*16 + fq*4 + e][n = nb + nj*32 + jt*16 + fr]
  //   swapped C[m = mb + it*16 + fr][n = nb + nj*32 + jt*16 + fq*4 + e]            fr = lane & 15, fq = lane >> 4
  const int fr = lane & 15, fq = lane >> 4;
  const int d = p.d_model;
  if constexpr (EPI == EPI_CROSS_KV && SWAPPED) {  // blocked K [l][slot][head][m/64][dd/8][m%64][8]: 8 contiguous bytes per lane
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int m = mb + it * 16 + fr;
      if (m >= p.M) continue;
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
          const int n = nb + nj * 32 + jt * 16 + fq * 4;
          const int l = n / d, c = n - l * d;
          const int head = c >> 6, dd = c & 63;
          h16x4 pk;
#pragma unroll
          for (int e = 0; e < 4; ++e) pk[e] = (h16)(acc[nj][it][jt][e] + (p.bias ? p.bias[n + e] : 0.f));
          const long slot = ((long)l * p.n_batch_total + (p.kv_slot_map ? p.kv_slot_map[bz] : bz)) * (d >> 6) + head;
          h16* dst = reinterpret_cast<h16*>(p.C) + slot * p.t_pad * 64 + (long)(m >> 6) * 4096 + (dd >> 3) * 512 + (m & 63) * 8 + (dd & 7);
          *reinterpret_cast<h16x4*>(dst) = pk;
        }
    }
  } else {
#pragma unroll
    for (int nj = 0; nj < 2; ++nj)
#pragma unroll
      for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int reg_ax = (SWAPPED ? nj * 32 + jt * 16 : it * 16) + fq * 4 + e;
            const int lane_ax = (SWAPPED ? it * 16 : nj * 32 + jt * 16) + fr;
            lw[reg_ax * 64 + lane_ax] = acc[nj][it][jt][e];
          }
    epilogue_rows<EPI, SWAPPED>(p, lw, mb, nb, bz, lane);
  }
}

// ---------------------------------------------------------------------------- 256 x 256 tiles, one k-tile STREAM per CU
// The phased k-loop above, run by a workgroup per CU that walks its share of the tiles, so that the LDS-DMA
// pipeline never drains at a tile boundary: the last two k-tiles of a tile prefetch the first two of the NEXT tile
// exactly where the steady state would prefetch k-tiles t+1 / t+2, and the epilogue runs with three half-tiles in
// flight instead of being followed by a cold prologue (first loads of a 256-row A panel from HBM: 2-3 us of a ~30 us
// tile at K = 768). For that the epilogue may not touch the two staging buffers: every wave parks 16-row slabs
// (4 KB) in its own piece of the 32 KB that 160 KB of LDS leave beside them, 8 slabs per wave instead of 2 halves.
// The two wave groups fall back in step before the epilogue (otherwise each group's epilogue would wait for the
// other's at the phase barriers) and part again behind it.
// Tile order: XCD x (workgroups with blockIdx % 8 == x) owns a contiguous chunk of the tile space and its 32
// workgroups take 32 consecutive tiles of it per round: the same L2 sharing as the one-tile-per-workgroup order.
constexpr int SLAB_FLOATS = 16 * 64;

template <int EPI, bool SWAPPED>
__global__ __launch_bounds__(512) void gemm256ps_bf16_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];  // [2 k-tiles][A0 | A1 | W0 | W1] | 8 slabs x 4 KB

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid) >> 6;
  const int wr = wave >> 2, wc = wave & 3;

  const int nt = p.n_tiles, mt = (p.M + BM3 - 1) / BM3;
  const int tiles = nt * mt * p.batch;
  // this workgroup's tiles: first, first + step, ... < last
  int t_cur, t_last, t_step;
  {
    const int nwg = gridDim.x, orig = blockIdx.x;
    const int xcd = orig & 7, j = orig >> 3;
    const int q = tiles >> 3, r = tiles & 7;
    const int start = xcd * q + min(xcd, r);
    t_last = start + q + (xcd < r ? 1 : 0);
    t_step = (nwg >> 3) + (xcd < (nwg & 7) ? 1 : 0);
    t_cur = start + j;
  }
  if (t_cur >= t_last) return;  // (whole workgroup)

  auto decode = [&](int wg, int& n0, int& m0, int& bz) {
    // Wide launches (the cross-K/V projection: 36-40 column tiles, 14-26 MB of weights) walk the columns in groups of
    // 8 tiles: an XCD sweeps its A panels against ONE group (3 MB of weights, which stay in its 4 MB L2) before it moves
    // to the next, instead of streaming all the weights past every panel (5.7 GB fetched per launch at 64 clips).
    constexpr int G = 8;
    int nl, rest;
    if (nt <= G) {
      nl = wg % nt;
      rest = wg / nt;
    } else {
      const int panels = mt * p.batch, full = nt / G, in_full = full * G * panels;
      if (wg < in_full) {
        const int g = wg / (G * panels), within = wg % (G * panels);
        rest = within / G;
        nl = g * G + within % G;
      } else {
        const int gl = nt - full * G, w2 = wg - in_full;
        rest = w2 / gl;
        nl = full * G + w2 % gl;
      }
    }
    n0 = p.n_begin + nl * BN3;
    m0 = (rest % mt) * BM3;
    bz = rest / mt;
  };
  // LDS-DMA sources as byte offsets from A / W (launch_one keeps both operands under 4 GB for this kernel): piece q of
  // a half-tile = its rows q*64 + 8*wave + lane/8, chunk position lane%8 holding global chunk position ^ ((row>>1)&7)
  const char* Ab = reinterpret_cast<const char*>(p.A);
  const char* Wb = reinterpret_cast<const char*>(p.W);
  // (W rows are never clamped: their offsets are a per-lane constant plus the tile's n0 * K, kept as a scalar; A rows
  // are clamped to M - 1, so their offsets are recomputed per tile)
  auto offsets = [&](int m0, int bz, unsigned (&ao)[2][2]) {
    int ln = lane;
    asm volatile("" : "+v"(ln));  // recomputed per tile: hoisted out of the tile loop these terms cost registers in the k-loop
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int lr = q * 64 + 8 * wave + (ln >> 3);
      const int gc = ((ln & 7) ^ ((lr >> 1) & 7)) * 8;
#pragma unroll
      for (int hh = 0; hh < 2; ++hh) {
        const int arow = (lr >> 6) * 128 + hh * 64 + (lr & 63);
        ao[hh][q] = (unsigned)(((long)bz * p.a_batch_stride + (long)min(m0 + arow, p.M - 1) * p.lda + gc) * 2);
      }
    }
  };
  unsigned wo[2][2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int lr = q * 64 + 8 * wave + (lane >> 3);
    const int gc = ((lane & 7) ^ ((lr >> 1) & 7)) * 8;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh) wo[hh][q] = (unsigned)((((lr >> 5) * 64 + hh * 32 + (lr & 31)) * p.K + gc) * 2);
  }
  unsigned ao_cur[2][2], ao_nxt[2][2];
  int n0, m0, bz;
  decode(t_cur, n0, m0, bz);
  offsets(m0, bz, ao_cur);
  unsigned wn_cur = (unsigned)n0 * (unsigned)p.K * 2u, wn_nxt = 0;  // byte offset of W row n0

  // which: 0 A0, 1 A1, 2 W0, 3 W1. Buffer form of the LDS-DMA load: scalar resource of the operand + this lane's 32-bit
  // offset + a scalar offset (k-tile, and the tile's first weight row) — no 64-bit address arithmetic in the k-loop
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc((void*)Ab, 0, 0xffffffff, 0x27000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, 0xffffffff, 0x27000);
  auto stage = [&](auto WHICH, auto BUF, const unsigned (&ao)[2][2], unsigned wn, int kt) {
    constexpr int which = decltype(WHICH)::value, buf = decltype(BUF)::value;
    char* base = smem + buf * KT4_BYTES + which * HALF_BYTES + 8 * wave * (BK * 2);
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if constexpr (which < 2)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr_t)(base + q * 64 * (BK * 2)), 16, (int)ao[which & 1][q], kt * (BK * 2), 0, 0);
      else
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr_t)(base + q * 64 * (BK * 2)), 16, (int)wo[which & 1][q], (int)wn + kt * (BK * 2), 0, 0);
    }
  };
  using I0_ = std::integral_constant<int, 0>;
  using I1_ = std::integral_constant<int, 1>;
  using I2_ = std::integral_constant<int, 2>;
  using I3_ = std::integral_constant<int, 3>;

  const int swz_c = ((lane >> 4) ^ ((lane & 15) >> 1)) * 16;
  const int a_off = (wr * 64 + (lane & 15)) * (BK * 2) + swz_c;
  const int w_off = (wc * 32 + (lane & 15)) * (BK * 2) + swz_c;

  f32x4 acc[2][2][4][2];  // [mi][nj][it][jt]
  h16x8 af[4][2], wf0[2][2], wf1[2][2];
  auto read_a = [&](const char* half) {
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
      for (int s = 0; s < 2; ++s) af[it][s] = *reinterpret_cast<const h16x8*>(half + ((a_off + it * 2048) ^ (s * 64)));
  };
  auto read_w = [&](const char* half, h16x8 (&wf)[2][2]) {
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
      for (int s = 0; s < 2; ++s) wf[jt][s] = *reinterpret_cast<const h16x8*>(half + ((w_off + jt * 2048) ^ (s * 64)));
  };
  auto quadrant = [&](f32x4 (&c)[4][2], const h16x8 (&wf)[2][2]) {
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
          c[it][jt] = SWAPPED ? AXW_MFMA_16x16x32(wf[jt][s], af[it][s], c[it][jt]) : AXW_MFMA_16x16x32(af[it][s], wf[jt][s], c[it][jt]);
  };
#define AXW_PHASE_MFMA_BEGIN()                             \
  __builtin_amdgcn_sched_barrier(0);                       \
  __builtin_amdgcn_s_barrier();                            \
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
  __builtin_amdgcn_sched_barrier(0);                       \
  __builtin_amdgcn_s_setprio(1)
#define AXW_PHASE_MFMA_END()                               \
  __builtin_amdgcn_s_setprio(0);                           \
  __builtin_amdgcn_sched_barrier(0);                       \
  __builtin_amdgcn_s_barrier();                            \
  __builtin_amdgcn_sched_barrier(0)

  // One k-tile out of LDS buffer BUF (phases and hazards: the comment above HALF_BYTES). ISSUE1: phase 1 prefetches A1 of the
  // stream's next k-tile = (ao1, kt1); ISSUE2: phases 2-4 prefetch W0, A0, W1 of the one after = (ao2 / wo2, kt2).
  auto ktile = [&](auto BUF, auto ISSUE1, auto ISSUE2, const unsigned (&ao1)[2][2], unsigned wo1, int kt1,
                   const unsigned (&ao2)[2][2], unsigned wo2, int kt2) {
    constexpr int buf = decltype(BUF)::value;
    constexpr bool issue1 = decltype(ISSUE1)::value != 0, issue2 = decltype(ISSUE2)::value != 0;
    using OTHER = std::integral_constant<int, buf ^ 1>;
    const char* base = smem + buf * KT4_BYTES;
    // phase 1
    read_w(base + 2 * HALF_BYTES, wf0);
    __builtin_amdgcn_sched_barrier(0);
    read_a(base);
    if constexpr (issue1) stage(I1_{}, OTHER{}, ao1, wo1, kt1);
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
    AXW_PHASE_MFMA_BEGIN();
    quadrant(acc[0][0], wf0);
    AXW_PHASE_MFMA_END();
    // phase 2
    read_w(base + 3 * HALF_BYTES, wf1);
    if constexpr (issue2) stage(I2_{}, BUF, ao2, wo2, kt2);
    AXW_PHASE_MFMA_BEGIN();
    quadrant(acc[0][1], wf1);
    AXW_PHASE_MFMA_END();
    // phase 3
    read_a(base + HALF_BYTES);
    if constexpr (issue2) stage(I0_{}, BUF, ao2, wo2, kt2);
    AXW_PHASE_MFMA_BEGIN();
    quadrant(acc[1][1], wf1);
    AXW_PHASE_MFMA_END();
    // phase 4
    if constexpr (issue2) {
      stage(I3_{}, BUF, ao2, wo2, kt2);
      asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    } else if constexpr (issue1) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    AXW_PHASE_MFMA_BEGIN();
    quadrant(acc[1][0], wf0);
    AXW_PHASE_MFMA_END();
  };

  const int nk = p.K / BK;  // even, >= 4 (launch_one)
  float* lw = reinterpret_cast<float*>(smem + 2 * KT4_BYTES) + wave * SLAB_FLOATS;
#ifdef AXW_GEMM_TIMING  // profiles/microbench/gemm_shapes.cpp: p.part = [workgroup][32 tiles][2 wave groups][5 stamps], 100 MHz ticks
  int stamp_tile = 0;
#define AXW_STAMP(i)                                                                                                   \
  do {                                                                                                                 \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
    if ((wave & 3) == 0 && lane == 0 && stamp_tile < 32 && p.part)                                                     \
      reinterpret_cast<unsigned long long*>(p.part)[(((long)blockIdx.x * 32 + stamp_tile) * 2 + wr) * 5 + (i)] = wall_clock64(); \
    if ((i) == 4) ++stamp_tile;                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                                 \
  } while (0)
#else
#define AXW_STAMP(i) do {} while (0)
#endif

  // the stream's first k-tile and three half-tiles of its second
  stage(I2_{}, I0_{}, ao_cur, wn_cur, 0);
  stage(I0_{}, I0_{}, ao_cur, wn_cur, 0);
  stage(I3_{}, I0_{}, ao_cur, wn_cur, 0);
  stage(I1_{}, I0_{}, ao_cur, wn_cur, 0);
  stage(I2_{}, I1_{}, ao_cur, wn_cur, 1);
  stage(I0_{}, I1_{}, ao_cur, wn_cur, 1);
  stage(I3_{}, I1_{}, ao_cur, wn_cur, 1);
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_sched_barrier(0);

  for (;;) {
    const int t_nxt = t_cur + t_step;
    const bool has_next = t_nxt < t_last;  // workgroup-uniform
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int nj = 0; nj < 2; ++nj)
#pragma unroll
        for (int it = 0; it < 4; ++it)
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) acc[mi][nj][it][jt] = f32x4{0.f, 0.f, 0.f, 0.f};
    __builtin_amdgcn_sched_barrier(0);
    if (wr == 1) __builtin_amdgcn_s_barrier();  // the second wave group runs one barrier behind the first inside a tile
    __builtin_amdgcn_sched_barrier(0);
    AXW_STAMP(0);
    int kt = 0;
    for (; kt + 2 < nk; kt += 2) {
      ktile(I0_{}, I1_{}, I1_{}, ao_cur, wn_cur, kt + 1, ao_cur, wn_cur, kt + 2);
      ktile(I1_{}, I1_{}, I1_{}, ao_cur, wn_cur, kt + 2, ao_cur, wn_cur, kt + 3);
      if (kt == 0) { AXW_STAMP(1); }
    }
    // the last two k-tiles prefetch the next tile's first two; the last tile "prefetches" its own again (one code
    // path; 112 KB of L2 reads per workgroup, drained at the end)
    int n0n, m0n, bzn;
    decode(has_next ? t_nxt : t_cur, n0n, m0n, bzn);
    offsets(m0n, bzn, ao_nxt);
    wn_nxt = (unsigned)n0n * (unsigned)p.K * 2u;
    __builtin_amdgcn_sched_barrier(0);
    ktile(I0_{}, I1_{}, I1_{}, ao_cur, wn_cur, kt + 1, ao_nxt, wn_nxt, 0);
    ktile(I1_{}, I1_{}, I1_{}, ao_nxt, wn_nxt, 0, ao_nxt, wn_nxt, 1);
    AXW_STAMP(2);
    if (wr == 0) __builtin_amdgcn_s_barrier();  // back in step for the epilogue
    __builtin_amdgcn_sched_barrier(0);
    AXW_STAMP(3);

    // epilogue: 16-row slabs through this wave's 4 KB (the staging buffers belong to the next tile already)
    const int mbw = m0 + wr * 128, nbw = n0 + wc * 64;
    int le = lane;
    asm volatile("" : "+v"(le));  // as in offsets(): the epilogue's per-lane terms are not worth registers in the k-loop
    const int fr = le & 15, fq = le >> 4;
    if constexpr (EPI == EPI_CROSS_KV && SWAPPED) {
      gemm_epilogue16<EPI, SWAPPED>(p, acc[0], lw, mbw, nbw, bz, le);  // register path, no LDS
      gemm_epilogue16<EPI, SWAPPED>(p, acc[1], lw, mbw + 64, nbw, bz, le);
    } else if constexpr (!SWAPPED) {  // slab = 16 rows (m) x 64 columns (n): acc[mi][.][it][.]
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int it = 0; it < 4; ++it) {
#pragma unroll
          for (int nj = 0; nj < 2; ++nj)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
              for (int e = 0; e < 4; ++e) lw[(fq * 4 + e) * 64 + nj * 32 + jt * 16 + fr] = acc[mi][nj][it][jt][e];
          epilogue_rows<EPI, SWAPPED, 4>(p, lw, mbw + mi * 64 + it * 16, nbw, bz, le);
          __builtin_amdgcn_sched_barrier(0);  // slab by slab: interleaving all eight costs more registers than there are
        }
    } else {  // swapped: slab = 16 rows (n) x 64 columns (m): acc[mi][nj][.][jt]
#pragma unroll
      for (int mi = 0; mi < 2; ++mi)
#pragma unroll
        for (int nj = 0; nj < 2; ++nj)
#pragma unroll
          for (int jt = 0; jt < 2; ++jt) {
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
              for (int e = 0; e < 4; ++e) lw[(fq * 4 + e) * 64 + it * 16 + fr] = acc[mi][nj][it][jt][e];
            epilogue_rows<EPI, SWAPPED, 4>(p, lw, mbw + mi * 64, nbw + nj * 32 + jt * 16, bz, le);
            __builtin_amdgcn_sched_barrier(0);
          }
    }
    AXW_STAMP(4);
    if (!has_next) break;
    t_cur = t_nxt;
    n0 = n0n; m0 = m0n; bz = bzn;
    wn_cur = wn_nxt;
#pragma unroll
    for (int hh = 0; hh < 2; ++hh)
#pragma unroll
      for (int q = 0; q < 2; ++q) ao_cur[hh][q] = ao_nxt[hh][q];
  }
#undef AXW_PHASE_MFMA_BEGIN
#undef AXW_PHASE_MFMA_END
#undef AXW_STAMP
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // no LDS-DMA may outlive the workgroup's LDS allocation
}

int gemm_force_tile = 0;  // 0: by tile count; 1: 128x128, 2: 256x128, 5: 256x256 stream per CU (tests and microbenchmarks)

template <int EPI, bool SW>
static void launch_one(GemmParams p, int n_begin, int n_end, hipStream_t s) {
  if (n_end <= n_begin) return;
  p.n_begin = n_begin;
  const int mt256 = (p.M + BM2 - 1) / BM2;
  if ((n_end - n_begin) % BN3 == 0) {
    const int tiles_sq = (n_end - n_begin) / BN3 * mt256 * p.batch;
    // one workgroup per CU: a launch runs in ceil(tiles / 256) rounds; the square tile is ~1.15x faster per flop
    auto fill = [](int tiles) { return (double)tiles / (double)((tiles + 255) / 256 * 256); };
    const bool sq_pays = tiles_sq >= 256 && 1.15 * fill(tiles_sq) >= fill(2 * tiles_sq);
    const int nk = p.K / BK;
    const bool small_ops = ((long)p.batch * p.a_batch_stride + (long)p.M * p.lda) * 2 < (1L << 32) && (long)p.N * p.K * 2 < (1L << 32);
    if ((gemm_force_tile == 5 || (gemm_force_tile == 0 && sq_pays)) && nk >= 4 && nk % 2 == 0 && small_ops) {
      constexpr int lds = 2 * KT4_BYTES + 8 * SLAB_FLOATS * 4;  // all 160 KB of a CU: opt-in per kernel and device
      static std::atomic<unsigned long long> attr_done{0};     // (per instantiation) bit = device ordinal
      static std::atomic<int> n_cus{0};
      int dev = 0;
      (void)hipGetDevice(&dev);
      if (!((attr_done.load(std::memory_order_acquire) >> (dev & 63)) & 1)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm256ps_bf16_kernel<EPI, SW>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        int cu = 0;
        if (e == hipSuccess) e = hipDeviceGetAttribute(&cu, hipDeviceAttributeMultiprocessorCount, dev);
        if (e != hipSuccess || cu < 1) { fprintf(stderr, "[ax_whisper] launch_gemm: %s\n", hipGetErrorString(e)); abort(); }
        n_cus.store(cu, std::memory_order_relaxed);  // the devices of one node are alike
        attr_done.fetch_or(1ull << (dev & 63), std::memory_order_release);
      }
      const int cus = n_cus.load(std::memory_order_relaxed);
      p.n_tiles = (n_end - n_begin) / BN3;
      hipLaunchKernelGGL((gemm256ps_bf16_kernel<EPI, SW>), dim3(tiles_sq < cus ? tiles_sq : cus), dim3(512), lds, s, p);
      return;
    }
  }
  p.n_tiles = (n_end - n_begin) / BN;
  const int tiles256 = p.n_tiles * mt256 * p.batch;
  if (gemm_force_tile == 2 || (gemm_force_tile == 0 && tiles256 >= 256 && p.K >= 2 * BK)) {  // enough 256-row tiles for every CU: deep-ring kernel
    hipLaunchKernelGGL((gemm256_bf16_kernel<EPI, SW>), dim3(tiles256), dim3(512), 3 * STAGE2_BYTES, s, p);
    return;
  }
  dim3 grid(p.n_tiles * ((p.M + BM - 1) / BM) * p.batch);
  hipLaunchKernelGGL((gemm_bf16_kernel<EPI, SW>), grid, dim3(256), 4 * TILE_BYTES, s, p);
}

void launch_gemm(const GemmParams& p, hipStream_t s) {
  switch (p.epilogue) {
    case EPI_BIAS_BF16: launch_one<EPI_BIAS_BF16, false>(p, 0, p.N, s); break;
    case EPI_BIAS_GELU_BF16: launch_one<EPI_BIAS_GELU_BF16, false>(p, 0, p.N, s); break;
    case EPI_GELU_POS_F32: launch_one<EPI_GELU_POS_F32, false>(p, 0, p.N, s); break;
    case EPI_RESID_F32: launch_one<EPI_RESID_F32, false>(p, 0, p.N, s); break;
    case EPI_PARTIAL_F32: {  // few-tile launches only: always the 128x128 kernel, K slices along grid.y
      if (p.ksplit < 1 || (p.K / BK) % p.ksplit != 0 || p.N % BN != 0) { fprintf(stderr, "[ax_whisper] launch_gemm: bad split-K (K=%d, ksplit=%d)\n", p.K, p.ksplit); abort(); }
      GemmParams q = p;
      q.n_begin = 0;
      q.n_tiles = p.N / BN;
      dim3 grid(q.n_tiles * ((p.M + BM - 1) / BM) * p.batch, p.ksplit);
      hipLaunchKernelGGL((gemm_bf16_kernel<EPI_PARTIAL_F32, false>), grid, dim3(256), 4 * TILE_BYTES, s, q);
      break;
    }
    case EPI_QKV:  // Q,K rows normal; V rows with swapped operands (V^T output, four consecutive frames per lane)
      if (p.M % 4 != 0) { fprintf(stderr, "[ax_whisper] launch_gemm: EPI_QKV needs M %% 4 == 0 (M=%d)\n", p.M); abort(); }
      if (p.qkv_part != 2) launch_one<EPI_QKV, false>(p, 0, 2 * p.d_model, s);
      if (p.qkv_part != 1) launch_one<EPI_QKV, true>(p, 2 * p.d_model, 3 * p.d_model, s);
      break;
    case EPI_CROSS_KV:  // weight rows: all layers' K first (swapped, blocked layout), then all layers' V
      launch_one<EPI_CROSS_KV, true>(p, 0, p.n_layer * p.d_model, s);
      launch_one<EPI_CROSS_KV, false>(p, p.n_layer * p.d_model, 2 * p.n_layer * p.d_model, s);
      break;
  }
}

// ---------------------------------------------------------------------------- LayerNorm
// fp32 rows -> h16 rows, eps 1e-5, biased variance (nn.LayerNorm [upstream]); one wave per row.
// n_part > 0: the row first takes in the split-K partials of the GEMM before it (x += bias + part[0] + part[1] + ...,
// always in that order: deterministic) and is written back — the residual add of an EPI_PARTIAL_F32 launch.
__global__ __launch_bounds__(256) void layernorm_bf16_kernel(float* __restrict__ x, const float* __restrict__ g,
                                                             const float* __restrict__ b, h16* __restrict__ y, long rows, int d,
                                                             const float* __restrict__ part, int n_part, long part_stride,
                                                             const float* __restrict__ part_bias) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float4* xr = reinterpret_cast<float4*>(x + row * d);
  const int nv = d >> 2;  // float4 per row
  float4 v[8];            // d <= 2048
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int c = lane + 64 * i;
    if (c < nv) {
      v[i] = xr[c];
      if (n_part > 0) {
        const float4 bb = reinterpret_cast<const float4*>(part_bias)[c];
        v[i].x += bb.x; v[i].y += bb.y; v[i].z += bb.z; v[i].w += bb.w;
        for (int q = 0; q < n_part; ++q) {
          const float4 pp = reinterpret_cast<const float4*>(part + q * part_stride + row * d)[c];
          v[i].x += pp.x; v[i].y += pp.y; v[i].z += pp.z; v[i].w += pp.w;
        }
        xr[c] = v[i];
      }
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
  }
  const float mean = wave_sum(s) / d;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int c = lane + 64 * i;
    if (c < nv) {
      float a0 = v[i].x - mean, a1 = v[i].y - mean, a2 = v[i].z - mean, a3 = v[i].w - mean;
      q += (a0 * a0 + a1 * a1) + (a2 * a2 + a3 * a3);
    }
  }
  const float rstd = rsqrtf(wave_sum(q) / d + 1e-5f);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    int c = lane + 64 * i;
    if (c < nv) {
      float4 gg = reinterpret_cast<const float4*>(g)[c], bb = reinterpret_cast<const float4*>(b)[c];
      h16x4 o;
      o[0] = (h16)((v[i].x - mean) * rstd * gg.x + bb.x);
      o[1] = (h16)((v[i].y - mean) * rstd * gg.y + bb.y);
      o[2] = (h16)((v[i].z - mean) * rstd * gg.z + bb.z);
      o[3] = (h16)((v[i].w - mean) * rstd * gg.w + bb.w);
      reinterpret_cast<h16x4*>(y + row * d)[c] = o;
    }
  }
}

void launch_layernorm_bf16(float* x, const float* g, const float* b, h16* y, long rows, int d, hipStream_t s, const float* part,
                           int n_part, long part_stride, const float* part_bias) {
  hipLaunchKernelGGL(layernorm_bf16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, g, b, y, rows, d, part, n_part,
                     part_stride, part_bias);
}

}  // inline namespace AXW_NS
}  // namespace axw
