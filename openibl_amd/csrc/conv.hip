// VGG16 conv1_1..conv5_3 on gfx950: NHWC activations, 3x3 convolutions as implicit-GEMM on the
// shared MFMA core (gemm_core.h), bias + ReLU + 2x2 max-pool fused into the epilogue.
// Reference behaviour: ibl/models/vgg.py:40-42 (layer list), :61-70 (forward).
#include "conv_halo.h"
#include "conv_halo4.h"
#include "conv_ring.h"
#include "gemm_core.h"

namespace oibl {

// ---------------------------------------------------------------------------------------------
// weight re-pack: [Cout][Cin][3][3] fp32 -> [tap][Cout][Cin] T
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_conv3x3_kernel(const float* __restrict__ w, T* __restrict__ packed, int cout,
                                    int cin) {
  const size_t total = (size_t)9 * cout * cin;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int ci = (int)(i % cin);
    const size_t t = i / cin;
    const int co = (int)(t % cout);
    const int tap = (int)(t / cout);
    const float v = w[((size_t)co * cin + ci) * 9 + tap];
    if constexpr (std::is_same<T, bf16x3_t>::value)
      x3_store(reinterpret_cast<char*>(packed) + (i - ci) * 4, ci, v);  // row (tap, co): cin x 4 bytes
    else
      Elem<T>::store(packed + i, v);
  }
}

// fp32 rows <-> bf16x3 rows (groups of [32 hi | 32 lo]); C % 32 == 0
__global__ void x3_split_rows_kernel(const float* __restrict__ src, char* __restrict__ dst, size_t n,
                                     int C) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / C, c = i - row * C;
    x3_store(dst + row * C * 4, c, src[i]);
  }
}
__global__ void x3_join_rows_kernel(const char* __restrict__ src, float* __restrict__ dst, size_t n,
                                    int C) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    const size_t row = i / C, c = i - row * C;
    dst[i] = x3_load(src + row * C * 4, c);
  }
}

// ---- f16mx (common.h) ------------------------------------------------------------------------
// weights: one thread per (tap, cout, 32-channel group) line of the packed tensor [tap][Cout][Cin]
__global__ void pack_conv3x3_mx_kernel(const float* __restrict__ w, char* __restrict__ packed, int cout,
                                       int cin) {
  const int groups = cin >> 5;
  const size_t total = (size_t)9 * cout * groups;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total;
       i += (size_t)gridDim.x * blockDim.x) {
    const int g = (int)(i % groups);
    const size_t t = i / groups;
    const int co = (int)(t % cout);
    const int tap = (int)(t / cout);
    float v[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) v[e] = w[((size_t)co * cin + g * 32 + e) * 9 + tap];
    uint4 line[8];
    mx_pack_line(v, line);
    uint4* dst = reinterpret_cast<uint4*>(packed + i * 128);
#pragma unroll
    for (int k = 0; k < 8; ++k) dst[k] = line[k];
  }
}
// rows: SRC = 0 fp32 rows [rows][C] -> f16mx lines; SRC = 1 bf16x3 lines -> f16mx lines (may be in
// place: a thread reads its whole line before it writes it).  One thread per line.
template <int SRC>
__global__ void mx_pack_rows_kernel(const char* __restrict__ src, char* __restrict__ dst, size_t lines,
                                    unsigned* range_flag) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < lines;
       i += (size_t)gridDim.x * blockDim.x) {
    float v[32];
    const uint4* sp = reinterpret_cast<const uint4*>(src + i * 128);
    if constexpr (SRC == 0) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const uint4 t = sp[k];
        v[4 * k] = __builtin_bit_cast(float, t.x);
        v[4 * k + 1] = __builtin_bit_cast(float, t.y);
        v[4 * k + 2] = __builtin_bit_cast(float, t.z);
        v[4 * k + 3] = __builtin_bit_cast(float, t.w);
      }
    } else {
      uint4 hi[4], lo[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        hi[k] = sp[k];
        lo[k] = sp[4 + k];
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const unsigned h[4] = {hi[k].x, hi[k].y, hi[k].z, hi[k].w}, l[4] = {lo[k].x, lo[k].y, lo[k].z, lo[k].w};
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          v[8 * k + 2 * d] = __builtin_bit_cast(float, h[d] << 16) + __builtin_bit_cast(float, l[d] << 16);
          v[8 * k + 2 * d + 1] = __builtin_bit_cast(float, h[d] & 0xffff0000u) + __builtin_bit_cast(float, l[d] & 0xffff0000u);
        }
      }
    }
    uint4 line[8];
    mx_pack_line(v, line, range_flag);
    uint4* dp = reinterpret_cast<uint4*>(dst + i * 128);
#pragma unroll
    for (int k = 0; k < 8; ++k) dp[k] = line[k];
  }
}
// f16mx lines -> fp32 rows as the kernels see the values: WHICH = 0 hi + q6(lo) (the stored value to
// ~2^-15), 1 hi alone, 2 q6(hi), 3 q6(lo)   (tests)
__global__ void mx_join_rows_kernel(const char* __restrict__ src, float* __restrict__ dst, size_t n, int which) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (size_t)gridDim.x * blockDim.x) {
    float hi, hi6, lo6;
    mx_line_decode(src + (i >> 5) * 128, (int)(i & 31), hi, hi6, lo6);
    dst[i] = which == 0 ? hi + lo6 : which == 1 ? hi : which == 2 ? hi6 : lo6;
  }
}

// ---------------------------------------------------------------------------------------------
// conv1_1: x [N][3][H][W] fp32 -> out [N][H][W][64] T.   Cin = 3 gives K = 27: no MFMA shape fits
// without an explicit im2col pass, and the layer is 0.56 % of the backbone FLOPs, so it runs on
// the vector ALU in exact fp32:  lane = output channel (its 27 weights live in registers), a wave
// walks a strip of 8 pixels; the strip's input window is wave-uniform (scalar/broadcast loads) and
// every store is one full NHWC line (64 channels contiguous).
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void conv1_1_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ w,
                                                      const float* __restrict__ bias,
                                                      T* __restrict__ out, int N, int H, int W) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int spr = (W + 7) >> 3;  // strips per row
  const long nstrips = (long)N * H * spr;
  const long strip = (long)blockIdx.x * 4 + wave;
  if (strip >= nstrips) return;
  const int n = (int)(strip / ((long)H * spr));
  const int rem = (int)(strip - (long)n * H * spr);
  const int y = rem / spr;
  const int x0 = (rem - y * spr) * 8;

  float wr[27];
#pragma unroll
  for (int k = 0; k < 27; ++k) wr[k] = w[lane * 27 + k];
  const float b = bias[lane];
  float acc[8];
#pragma unroll
  for (int p = 0; p < 8; ++p) acc[p] = b;

#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int yy = y + ky - 1;
      if (yy < 0 || yy >= H) continue;  // wave-uniform
      const float* row = x + (((size_t)n * 3 + c) * H + yy) * W;
      float v[10];
#pragma unroll
      for (int i = 0; i < 10; ++i) {
        const int xx = x0 - 1 + i;
        v[i] = (xx >= 0 && xx < W) ? row[xx] : 0.f;
      }
#pragma unroll
      for (int kx = 0; kx < 3; ++kx)
#pragma unroll
        for (int p = 0; p < 8; ++p) acc[p] = fmaf(v[p + kx], wr[c * 9 + ky * 3 + kx], acc[p]);
    }
  }
  T* o = out + (((size_t)n * H + y) * W + x0) * 64 + lane;
#pragma unroll
  for (int p = 0; p < 8; ++p)
    if (x0 + p < W) Elem<T>::store(o + (size_t)p * 64, fmaxf(acc[p], 0.f));
}

// ---------------------------------------------------------------------------------------------
// conv1_1 on the matrix cores (bf16 precision only): K = 27 is padded to 32 = two k-steps of
// v_mfma_f32_32x32x16_bf16.  The GEMM is run transposed (D[cout][pixel] = W . X^T): the weights
// are the A operand (kept in registers for the whole kernel) and 32 pixels of an image row are
// the B operand, so that each lane ends up with 4 CONSECUTIVE output channels of one pixel per
// register quad -> 8-byte packed bf16 writes into an LDS staging tile and full 128-byte NHWC
// lines on the way out.  The layer is bound by writing its 64-channel output (39 MB / image in
// bf16), not by the MFMA work.
//   workgroup = 4 waves = 128 consecutive pixels of one image row; persistent over row segments;
//   input patch (3 channels x 3 rows x 130 columns, zero padded) staged in LDS as fp32, rounded
//   to bf16 when the fragments are built.
// ---------------------------------------------------------------------------------------------
OIBL_HOOK(int, g_conv11_valu, 0);  // test hook: force the vector-ALU conv1_1 in bf16 mode too
constexpr int C11_TW = 128;
constexpr int C11_PITCH = 132;
constexpr int C11_ZERO = 9 * C11_PITCH;  // index of a zero float (k >= 27)

typedef __attribute__((ext_vector_type(2))) float f32x2_t;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
typedef __attribute__((ext_vector_type(2))) short i16x2_t;
// two floats -> one dword of two bf16 (lo in bits 0-15): a single v_cvt_pk_bf16_f32
__device__ static inline uint32_t pack_bf16x2(float lo, float hi) {
  const f32x2_t v = {lo, hi};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_t));
}
// ReLU on two packed bf16: as signed 16-bit integers every negative value (sign bit set, -0
// included) is < 0, so max(x, 0) per half is exactly relu — one v_pk_max_i16 for two values, and
// rounding first / clamping second gives the same bits as clamping first.
__device__ static inline uint32_t relu_bf16x2(uint32_t packed) {
  const i16x2_t z = {0, 0};
  return __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(i16x2_t, packed), z));
}

// X3 = bf16x3: weights and pixels are split into (hi, lo) fragments, 3 MFMAs per k-step, and the
// output is written as groups of [32 hi | 32 lo] (256 B per pixel).
template <bool X3>
__global__ __launch_bounds__(256) void conv1_1_mfma_kernel(const float* __restrict__ x,
                                                           const float* __restrict__ w,
                                                           const float* __restrict__ bias,
                                                           char* __restrict__ out, int N, int H,
                                                           int W, int tiles_per_row, long ntiles) {
  constexpr int PX_BYTES = X3 ? 256 : 128, OPITCH = PX_BYTES + 16;
  __shared__ __attribute__((aligned(16))) float patch[9 * C11_PITCH + 4];
  __shared__ __attribute__((aligned(16))) char ostage_all[4 * 32 * OPITCH];  // per wave: 32 px
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;

  // A operand: weights.  wf[t][s] element e  <->  cout = 32 t + l31,  k = 16 s + 8 half + e
  bf16x8_t wf[2][2], wl[2][2];
  int koff[2][8];  // LDS float offset of input element k (relative to the pixel's column)
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 16 * s + 8 * half + e;
      koff[s][e] = k < 27 ? (k / 3) * C11_PITCH + (k % 3) : -1;
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float v = k < 27 ? w[(32 * t + l31) * 27 + k] : 0.f;
        uint16_t hi, lo;
        x3_split(v, hi, lo);
        wf[t][s][e] = (short)hi;
        wl[t][s][e] = (short)lo;
      }
    }
  float bb[2][16];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) bb[t][r] = bias[32 * t + acc_row(r, lane)];
  if (threadIdx.x < 4) patch[C11_ZERO + threadIdx.x] = 0.f;

  for (long tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int tx = (int)(tile % tiles_per_row);
    const long ny = tile / tiles_per_row;
    const int y = (int)(ny % H), n = (int)(ny / H);
    const int x0 = tx * C11_TW;
    __syncthreads();  // the previous tile's fragment reads are done
    for (int i = threadIdx.x; i < 9 * 130; i += 256) {
      const int r = i / 130, col = i - r * 130;
      const int yy = y + (r % 3) - 1, xx = x0 - 1 + col;
      float v = 0.f;
      if (yy >= 0 && yy < H && xx >= 0 && xx < W)
        v = x[(((size_t)n * 3 + r / 3) * H + yy) * W + xx];
      patch[r * C11_PITCH + col] = v;
    }
    __syncthreads();

    const int px = wave * 32 + l31;  // pixel column inside the tile
    bf16x8_t xf[2], xl[2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float v = koff[s][e] >= 0 ? patch[koff[s][e] + px] : 0.f;
        uint16_t hi, lo;
        x3_split(v, hi, lo);
        xf[s][e] = (short)hi;
        xl[s][e] = (short)lo;
      }
    f32x16_t acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][r] = bb[t][r];
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        if constexpr (X3) {
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[t][s], xf[s], acc[t], 0, 0, 0);
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][s], xl[s], acc[t], 0, 0, 0);
        }
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][s], xf[s], acc[t], 0, 0, 0);
      }
    }
    // D[row = cout][col = pixel]: registers 4g..4g+3 = couts 32t + 8g + 4*half + 0..3 of pixel l31
    char* ost = ostage_all + wave * 32 * OPITCH;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if constexpr (X3) {
          uint2 hi, lo;
          ring_split4(fmaxf(acc[t][4 * g], 0.f), fmaxf(acc[t][4 * g + 1], 0.f),
                      fmaxf(acc[t][4 * g + 2], 0.f), fmaxf(acc[t][4 * g + 3], 0.f), hi, lo);
          char* q = ost + l31 * OPITCH + t * 128 + (8 * g + 4 * half) * 2;
          *reinterpret_cast<uint2*>(q) = hi;
          *reinterpret_cast<uint2*>(q + 64) = lo;
        } else {
          uint2 v;
          v.x = relu_bf16x2(pack_bf16x2(acc[t][4 * g], acc[t][4 * g + 1]));
          v.y = relu_bf16x2(pack_bf16x2(acc[t][4 * g + 2], acc[t][4 * g + 3]));
          *reinterpret_cast<uint2*>(ost + l31 * OPITCH + (32 * t + 8 * g + 4 * half) * 2) = v;
        }
      }
    __builtin_amdgcn_wave_barrier();  // same-wave exchange through LDS: DS ops retire in order
    char* orow = out + (((size_t)n * H + y) * W + x0 + wave * 32) * PX_BYTES;
    constexpr int PARTS = PX_BYTES / 16;
#pragma unroll
    for (int it = 0; it < 32 * PARTS / 64; ++it) {
      const int idx = it * 64 + lane, p = idx / PARTS, part = idx % PARTS;
      const uint4 v = *reinterpret_cast<const uint4*>(ost + p * OPITCH + part * 16);
      if (x0 + wave * 32 + p < W)
        *reinterpret_cast<uint4*>(orow + (size_t)p * PX_BYTES + part * 16) = v;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// implicit-GEMM 3x3 convolution
//   M = output pixels, N = Cout, K = 9 * Cin ordered (tap, cin).  A K-step is one tap x one
//   128-byte run of input channels of each pixel, fetched straight from the NHWC tensor (or from a
//   zero line when the tap falls outside the image).
//   POOL: pixels are enumerated quad-major (m = 4*quad + 2*dy + dx over the floor(H/2) x floor(W/2)
//   pooled grid), so that the four members of a 2x2 window are 4 consecutive GEMM rows = registers
//   4g..4g+3 of one lane in the 32x32 accumulator layout: the pool is an in-register max and the
//   kernel writes the pooled NHWC tensor directly.
// ---------------------------------------------------------------------------------------------
struct ConvParams {
  const void* in;
  const void* w;
  const float* bias;
  void* out;
  const void* zero;
  int N, H, W, cin, cout;
  long m_total;   // GEMM rows (pixels enumerated, 4 per pooled output when POOL)
  long out_rows;  // rows of the output tensor ( = m_total or m_total / 4)
  int tiles_n;
  int relu;
  int ablate;  // timing experiments only (wrong results): 1 = A loads only at tap 0, 2 = B loads
               // only at the first step, 3 = both
  int out_f32;  // bf16x3 only: the output is written as plain fp32 NHWC (the layer feeding the head)
  int korder;   // K order of the implicit GEMM (test hook, see ConvRingALoader::begin_tile)
  // split-K (layers with too few tiles to fill the chip, e.g. conv5 of a single image: 40 tiles):
  // gridDim.y = ksplit workgroups share a tile, each contracts steps/ksplit K-steps from zero and
  // dumps its fp32 accumulators to partial[ks][m_total][cout]; conv_splitk_reduce_kernel adds them in
  // a fixed order, then bias / ReLU / pool / store.
  int ksplit;
  float* partial;
  unsigned* range_flag;  // f16mx: the pass's range flag (common.h, mx_raise_range_flag); may be null
  float bias_mul = 1.f, out_mul = 1.f;   // f16mx backbone: activation scale (g_mx_act_shift; conv_ring.h, RingParams)
};

// one output element into the staged tile row (row-major [BN] of T; bf16x3: (hi, lo) groups or fp32)
template <typename T>
__device__ static inline void conv_stage_store(char* row, int col, float v, int out_f32) {
  if constexpr (std::is_same<T, bf16x3_t>::value) {
    if (out_f32)
      reinterpret_cast<float*>(row)[col] = v;
    else
      x3_store(row, col, v);
  } else {
    Elem<T>::store(reinterpret_cast<T*>(row) + col, v);
  }
}

template <typename Cfg, bool POOL>
struct ConvALoader {
  const char* base[Cfg::A_LOADS];
  unsigned mask[Cfg::A_LOADS];
  const char* zero;
  long tap_off;
  int tap, cc, cchunks, W, ablate, korder;
  long pix_bytes;

  // step0: first K-step of this workgroup (split-K), in the loop's own order
  __device__ inline void init(const WaveCoord& c, const ConvParams& p, long m0, int step0) {
    using T = typename Cfg::T;
    const int piece = load_piece_bytes<Cfg>(c);
    pix_bytes = (long)p.cin * sizeof(T);
    W = p.W;
    ablate = p.ablate;
    korder = p.korder;
    cchunks = p.cin / Cfg::BK;
    zero = reinterpret_cast<const char*>(p.zero) + (c.lane & 7) * 16;
    const int Hq = POOL ? (p.H >> 1) : p.H, Wq = POOL ? (p.W >> 1) : p.W;
#pragma unroll
    for (int j = 0; j < Cfg::A_LOADS; ++j) {
      // 32-bit index math (the host checks m_total < 2^31): 64-bit divides cost hundreds of cycles
      const unsigned m = (unsigned)m0 + (unsigned)load_row<Cfg>(c, j);
      unsigned mk = 0;
      long off = 0;
      if (m < (unsigned)p.m_total) {
        const unsigned q = POOL ? (m >> 2) : m;
        const unsigned sub = POOL ? (m & 3u) : 0u;
        const unsigned hw = (unsigned)Hq * (unsigned)Wq;
        const unsigned n = q / hw;
        const unsigned rem = q - n * hw;
        unsigned yq = rem / (unsigned)Wq;
        int y = (int)yq, x = (int)(rem - yq * (unsigned)Wq);
        if (POOL) {
          y = 2 * y + (int)(sub >> 1);
          x = 2 * x + (int)(sub & 1);
        }
        const bool y0 = y > 0, y2 = y + 1 < p.H, x0 = x > 0, x2 = x + 1 < p.W;
        mk = (y0 && x0 ? 1u : 0u) | (y0 ? 2u : 0u) | (y0 && x2 ? 4u : 0u) | (x0 ? 8u : 0u) | 16u |
             (x2 ? 32u : 0u) | (y2 && x0 ? 64u : 0u) | (y2 ? 128u : 0u) | (y2 && x2 ? 256u : 0u);
        off = (((long)n * p.H + y) * p.W + x) * pix_bytes;
      }
      mask[j] = mk;
      base[j] = reinterpret_cast<const char*>(p.in) + off + piece;
    }
    tap = korder ? step0 % 9 : step0 / cchunks;
    cc = korder ? step0 / 9 : step0 % cchunks;
    tap_off = (long)((tap / 3 - 1) * W + (tap % 3 - 1)) * pix_bytes;
  }
  __device__ inline const char* src(int j) const {
    return ((mask[j] >> tap) & 1u) ? base[j] + tap_off + cc * 128 : zero;
  }
  __device__ inline bool active() const { return !((ablate & 1) && tap != 0); }
  // K order as in ConvRingALoader::begin_tile: 0 = (tap, chunk), 1 = (chunk, tap; test hook)
  __device__ inline void next() {
    if (korder == 0) {
      if (++cc == cchunks) {
        cc = 0;
        ++tap;
        tap_off = (long)((tap / 3 - 1) * W + (tap % 3 - 1)) * pix_bytes;
      }
    } else {
      if (++tap == 9) {
        tap = 0;
        ++cc;
      }
      tap_off = (long)((tap / 3 - 1) * W + (tap % 3 - 1)) * pix_bytes;
    }
  }
};

template <typename Cfg>
struct ConvBLoader {
  const char* p0[Cfg::B_LOADS];  // row pointers at (tap 0, channel chunk 0)
  long tap_stride, off;
  int tap, cc, cchunks, ablate, step, korder;
  __device__ inline bool active() const { return !((ablate & 2) && step != 0); }
  __device__ inline void init(const WaveCoord& c, const ConvParams& prm, long n0, int step0) {
    using T = typename Cfg::T;
    const int piece = load_piece_bytes<Cfg>(c);
    cchunks = prm.cin / Cfg::BK;
    ablate = prm.ablate;
    korder = prm.korder;
    tap = korder ? step0 % 9 : step0 / cchunks;
    cc = korder ? step0 / 9 : step0 % cchunks;
    step = 0;
    tap_stride = (long)prm.cin * sizeof(T) * prm.cout;
    off = tap * tap_stride + cc * 128;
#pragma unroll
    for (int j = 0; j < Cfg::B_LOADS; ++j) {
      const long n = n0 + load_row<Cfg>(c, j);
      p0[j] = reinterpret_cast<const char*>(prm.w) + n * prm.cin * (long)sizeof(T) + piece;
    }
  }
  __device__ inline const char* src(int j) const { return p0[j] + off; }
  __device__ inline void next() {
    ++step;
    if (korder == 0) {
      if (++cc == cchunks) {
        cc = 0;
        ++tap;
      }
    } else if (++tap == 9) {
      tap = 0;
      ++cc;
    }
    off = tap * tap_stride + cc * 128;
  }
};

template <typename Cfg, bool POOL>
constexpr int conv_lds_bytes() {
  constexpr int rows = POOL ? Cfg::BM / 4 : Cfg::BM;
  constexpr int epi = rows * (Cfg::BN * (int)sizeof(typename Cfg::T) + 16);
  return epi > Cfg::MAIN_LDS_BYTES ? epi : Cfg::MAIN_LDS_BYTES;
}

template <typename Cfg, bool POOL, bool GLDS, bool SPLITK = false>
__global__ __launch_bounds__(Cfg::NTHREADS) void conv3x3_igemm_kernel(ConvParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = typename Cfg::T;
  constexpr int TM = Cfg::TM, TN = Cfg::TN;
  const WaveCoord c = wave_coord<Cfg>();
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
  const long m0 = (long)tm * Cfg::BM, n0 = (long)tn * Cfg::BN;
  const int steps_all = 9 * (p.cin / Cfg::BK);
  const int nsteps = SPLITK ? steps_all / p.ksplit : steps_all;
  const int step0 = SPLITK ? (int)blockIdx.y * nsteps : 0;

  ConvALoader<Cfg, POOL> la;
  ConvBLoader<Cfg> lb;
  la.init(c, p, m0, step0);
  lb.init(c, p, n0, step0);

  // accumulators start at the bias (every kernel of this file orders the sum that way, so that
  // all variants of a layer produce identical bits); split-K parts start at zero, the bias is the
  // first term of the reduction
  f32x16_t acc[TM][TN];
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    float b = SPLITK ? 0.f : p.bias[n0 + (c.wn * TN + j) * 32 + (c.lane & 31)];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        asm volatile("" : "+v"(b));  // distinct registers, not aliases of one value
        acc[i][j][r] = b;
      }
  }

  gemm_nt_mainloop<Cfg, GLDS>(acc, smem, c, la, lb, nsteps);
  // (the main loop ends on a workgroup barrier: the staging LDS is free for the epilogue)

  if constexpr (SPLITK) {
    float* part = p.partial + (size_t)blockIdx.y * (size_t)p.m_total * p.cout;
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const long n = n0 + (c.wn * TN + j) * 32 + (c.lane & 31);
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const long m = m0 + (c.wm * TM + i) * 32 + acc_row(r, c.lane);
          if (m < p.m_total) part[m * p.cout + n] = acc[i][j][r];
        }
    }
    return;
  }

  constexpr int PITCH = Cfg::BN * (int)sizeof(T) + 16;
  constexpr int OUT_ROWS = POOL ? Cfg::BM / 4 : Cfg::BM;
#pragma unroll
  for (int j = 0; j < TN; ++j) {
    const int col = (c.wn * TN + j) * 32 + (c.lane & 31);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if constexpr (POOL) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v = fmaxf(fmaxf(acc[i][j][4 * g], acc[i][j][4 * g + 1]),
                          fmaxf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]));
          if (p.relu) v = fmaxf(v, 0.f);
          const int row = (c.wm * TM + i) * 8 + 2 * g + (c.lane >> 5);
          conv_stage_store<T>(smem + row * PITCH, col, v, p.out_f32);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r];
          if (p.relu) v = fmaxf(v, 0.f);
          const int row = (c.wm * TM + i) * 32 + acc_row(r, c.lane);
          conv_stage_store<T>(smem + row * PITCH, col, v, p.out_f32);
        }
      }
    }
  }
  __syncthreads();
  // coalesced copy-out: every output row is BN * sizeof(T) contiguous bytes of the NHWC tensor
  constexpr int CPR = Cfg::BN * (int)sizeof(T) / 16;  // 16-byte chunks per row
  const long row0 = POOL ? (m0 >> 2) : m0;
  char* obase = reinterpret_cast<char*>(p.out) + n0 * (long)sizeof(T);
  const long orow_bytes = (long)p.cout * sizeof(T);
  for (int idx = threadIdx.x; idx < OUT_ROWS * CPR; idx += Cfg::NTHREADS) {
    const int row = idx / CPR, ch = idx - row * CPR;
    const long grow = row0 + row;
    if (grow < p.out_rows)
      *reinterpret_cast<uint4*>(obase + grow * orow_bytes + ch * 16) =
          *reinterpret_cast<const uint4*>(smem + row * PITCH + ch * 16);
  }
}

// out[row][ch] = act(bias[ch] + sum_ks partial[ks][m][ch])  (POOL: max over the 4 GEMM rows of a quad
// first) — the bias leads the sum like in the one-pass kernels, parts are added in ks order.
template <typename T, bool POOL>
__global__ void conv_splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                          char* __restrict__ out, long out_rows, long m_total, int cout,
                                          int ksplit, int relu, int out_f32) {
  const long total = out_rows * cout;
  const size_t part = (size_t)m_total * cout;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const long row = i / cout;
    const int ch = (int)(i - row * cout);
    float v = -INFINITY;
#pragma unroll
    for (int q = 0; q < (POOL ? 4 : 1); ++q) {
      const size_t m = POOL ? (size_t)row * 4 + q : (size_t)row;
      float a = bias[ch];
      for (int ks = 0; ks < ksplit; ++ks) a += partial[ks * part + m * cout + ch];
      v = fmaxf(v, a);
    }
    if (relu) v = fmaxf(v, 0.f);
    conv_stage_store<T>(out + (size_t)row * cout * sizeof(T), ch, v, out_f32);
  }
}

// split factor for a layer whose tiling leaves most CUs idle (0 = none); shared by the launch and
// by the workspace query.  128-row tiles; parts of >= 6 K-steps; up to 512 workgroups.
static int conv_splitk_factor(long m_total, int cout, int steps) {
  const long tiles = ((m_total + 127) / 128) * (cout / (cout % 128 == 0 ? 128 : 64));
  if (tiles > 192 || steps < 12) return 0;
  const int cands[] = {8, 6, 4, 3, 2};
  for (int s : cands)
    if (steps % s == 0 && steps / s >= 6 && tiles * s <= 512) return s;
  return 0;
}

template <typename Cfg, bool POOL, bool GLDS>
static int launch_conv_kernel(const ConvParams& q, long grid, hipStream_t st) {
  constexpr int lds = conv_lds_bytes<Cfg, POOL>();
  auto kern = conv3x3_igemm_kernel<Cfg, POOL, GLDS>;
  if (lds > 64 * 1024) OIBL_SET_MAX_LDS(kern, lds);  // opt in to more than 64 KiB of dynamic LDS
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cfg::NTHREADS), lds, st, q);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

template <typename Cfg, bool POOL>
static int launch_conv_cfg(const ConvParams& p, hipStream_t st) {
  ConvParams q = p;
  q.tiles_n = p.cout / Cfg::BN;
  const long tiles_m = (p.m_total + Cfg::BM - 1) / Cfg::BM;
  const long grid = tiles_m * q.tiles_n;
  if (grid <= 0 || grid > 0x7fffffffL) {
    set_error("conv3x3: grid %ld out of range", grid);
    return OIBL_E_INVALID;
  }
  return g_regstage ? launch_conv_kernel<Cfg, POOL, false>(q, grid, st)
                    : launch_conv_kernel<Cfg, POOL, true>(q, grid, st);
}

// split-K launch: the partial pass of Cfg (128-row tiles) + the reduction
template <typename Cfg, bool POOL>
static int launch_conv_splitk(const ConvParams& p, hipStream_t st) {
  using T = typename Cfg::T;
  ConvParams q = p;
  q.tiles_n = p.cout / Cfg::BN;
  const long grid = ((p.m_total + Cfg::BM - 1) / Cfg::BM) * q.tiles_n;
  auto launch = [&](auto kern) {
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)p.ksplit), dim3(Cfg::NTHREADS),
                       Cfg::MAIN_LDS_BYTES, st, q);
  };
  if (g_regstage)
    launch(conv3x3_igemm_kernel<Cfg, POOL, false, true>);
  else
    launch(conv3x3_igemm_kernel<Cfg, POOL, true, true>);
  OIBL_LAUNCH_CHECK();
  const long total = p.out_rows * p.cout;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((conv_splitk_reduce_kernel<T, POOL>), dim3(blocks), dim3(256), 0, st, p.partial, p.bias,
                     (char*)p.out, p.out_rows, p.m_total, p.cout, p.ksplit, p.relu, p.out_f32);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

OIBL_HOOK(int, g_ring_raster, 0);                     // test hook: xcd_tile() mode of the ring kernels
// K order of the implicit-GEMM convolutions: 0 = (tap, channel chunk), 1 = (channel chunk, tap), -1 = per
// layer (default).  Order 1 cuts the fetched bytes 3-8x and is slower on every layer except
// the one whose re-fetches run at HBM-class bandwidth: bf16x3 conv2_2 (128 -> 128 at 240 x 320: 11.3 GB
// fetched per launch at 6.6 TB/s in order 0; 1.66 -> 1.54 ms in order 1).  The hook forces one order.
OIBL_HOOK(int, g_conv_korder, -1);
static int conv_korder_for(int precision, int cin, int cout) {
  if (g_conv_korder >= 0) return g_conv_korder;
  return (precision == OIBL_BF16X3 || precision == OIBL_F16MX) && cin == 128 && cout == 128 ? 1 : 0;
}
OIBL_HOOK(unsigned long long*, g_prof_buf, nullptr);  // test hook: phase profile of block 0
OIBL_HOOK(int, g_ring_ablate, 0);                     // test hook: see RingParams::ablate

// ring-schedule kernel (conv_ring.h): bf16, Cin % 64 == 0; WM = 2: 256 x 256 tile (Cout % 256 == 0),
// WM = 4: 512 x 128 tile (Cout % 128 == 0)
// BAR1: the one-barrier-per-phase schedule (ring_core.h).  g_ring_bar1 (test hook): 0 = two barriers per
// phase (rounds 1-3), 1 = one (the default).  Measured, bit-identical and race-free both times: with the
// stagger group tested at run time inside the loop 3-8 % SLOWER (profiles/r04_a_bar1_ab.txt); with one kernel
// body per group 1-4 % faster per layer, 11 % on conv2_1 — f16mx ring + halo layers 7.47 -> 7.25 ms, bf16
// 4.26 -> 4.19, bf16x3 11.04 -> 10.90 (profiles/r04_b_bar1_ab.txt).
OIBL_HOOK(int, g_ring_bar1, 1);
OIBL_HOOK(int, g_ring_stagger, 0);   // experiment: phase groups of the first round (conv_ring.h, RingParams::stagger)
// a launch over a row sub-range and / or a K split of the layer (conv_ring.h, RingParams; f16mx split-K)
struct RingSub {
  int tiles_m;       // M tiles of this launch, starting at GEMM row m_base
  long m_base;
  int parts;         // 0: one pass; else gridDim.y = parts split-K workgroups per tile
  int nsteps_part, outer_step;
  void* out;         // parts: the partial tensor [parts][out_rows][cout] fp32
  long out_rows;
  size_t part_stride;
};
template <int WM, bool POOL, bool ODD, int P = RING_BF16, bool OUTMX = (P >= RING_MX), bool BAR1 = false>
static int launch_conv_ring_impl(const ConvParams& p, hipStream_t st, const RingSub* sub = nullptr) {
  using G = RingGeo<WM>;
  constexpr bool X3 = P != RING_BF16;  // 4-byte elements
  RingParams q;
  q.in = p.in;
  q.w = p.w;
  q.bias = p.bias;
  q.out = p.out;
  q.in_bytes = (unsigned)((size_t)p.N * p.H * p.W * p.cin * (X3 ? 4 : 2));
  q.w_bytes = (unsigned)((size_t)9 * p.cout * p.cin * (X3 ? 4 : 2));
  q.out_f32 = X3 ? p.out_f32 : 0;
  q.N = p.N;
  q.H = p.H;
  q.W = p.W;
  q.cin = p.cin;
  q.cout = p.cout;
  q.m_total = (int)p.m_total;
  q.out_rows = (int)p.out_rows;
  q.tiles_n = p.cout / G::BN;
  q.relu = p.relu;
  q.prof = g_prof_buf;
  q.ablate = g_ring_ablate;
  {
    const unsigned hq = POOL ? p.H >> 1 : p.H, wq = POOL ? p.W >> 1 : p.W;
    ring_magic_u31(hq * wq ? hq * wq : 1u, &q.hw_mul, &q.hw_sh);
    ring_magic_u31(wq ? wq : 1u, &q.w_mul, &q.w_sh);
  }
  long tiles_m = (p.m_total + G::BM - 1) / G::BM;
  q.m_base = 0;
  q.nsteps_part = q.k_outer_step = 0;
  q.part_stride = 0;
  unsigned parts = 1;
  if (sub) {
    tiles_m = sub->tiles_m;
    q.m_base = (int)sub->m_base;
    if (sub->parts) {   // split-K: raw fp32 accumulators into the partial tensor
      parts = (unsigned)sub->parts;
      q.nsteps_part = sub->nsteps_part;
      q.k_outer_step = sub->outer_step;
      q.part_stride = sub->part_stride;
      q.out = sub->out;
      q.out_rows = (int)sub->out_rows;
      q.out_f32 = 1;
      q.relu = 0;
    }
  }
  const long grid = tiles_m * q.tiles_n;
  q.tiles_m = (int)tiles_m;
  q.raster = g_ring_raster;
  q.korder = p.korder;
  q.range_flag = p.range_flag;
  q.bias_mul = p.bias_mul;
  q.out_mul = (sub && sub->parts) ? 1.f : p.out_mul;   // (split-K partials are raw sums: the reduction scales)
  q.stagger = g_ring_stagger;
  constexpr int lds = ring_lds_bytes<WM, POOL, P, OUTMX>();
  auto kern = conv3x3_ring_kernel<WM, POOL, ODD, P, OUTMX, BAR1>;
  OIBL_SET_MAX_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3((unsigned)grid, parts), dim3(512), lds, st, q);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

template <int WM, bool POOL, int P = RING_BF16>
static int launch_conv_ring(const ConvParams& p, hipStream_t st, const RingSub* sub = nullptr) {
  // an odd number of K-tiles happens only for Cin = 64 in bf16 (the 4-byte element types have twice
  // the K-tiles), which only the 512 x 128 variant serves
  if constexpr (WM == 4 && P == RING_BF16) {
    if ((9 * (p.cin / 64)) & 1)
      return g_ring_bar1 ? launch_conv_ring_impl<WM, POOL, true, P, false, true>(p, st)
                         : launch_conv_ring_impl<WM, POOL, true>(p, st);
  }
  if constexpr (P == RING_BF16 || P == RING_X3 || P == RING_MX_EARLY || P == RING_MX) {
    if (g_ring_bar1) return launch_conv_ring_impl<WM, POOL, false, P, (P >= RING_MX), true>(p, st, sub);
  }
  return launch_conv_ring_impl<WM, POOL, false, P>(p, st, sub);
}

// 0 = not eligible, else the wave-row count of the instantiation to use (es = bytes per element)
static int ring_variant(const ConvParams& p, int es = 2) {
  if (p.cin % 64 != 0 || (size_t)p.N * p.H * p.W * p.cin * es >= (size_t)0xE0000000u ||
      p.m_total >= 0x7fffff00L)
    return 0;
  if (p.cout % 256 == 0 && p.cin % 128 == 0) return 2;
  if (p.cout % 128 == 0) return 4;
  return 0;
}

// Tile selection.  The implicit GEMM is bound by L2 -> LDS traffic before it is bound by the
// matrix cores: a 128 x 128 x 64 step needs 32 KB per 2.1 MFLOP (64 flop/B, i.e. ~64 B/clk/CU at
// the bf16 MFMA peak), a 256 x 256 step half of that.  So the largest tile that still gives every
// CU a couple of workgroups wins; small problems (conv5 at small batch) fall back to 128-row tiles.
// g_conv_tile (test hook): 0 = auto, 1 = 128x{128,64}, 2 = 256x{128,64}, 3 = 256x256 where legal,
// 4 = ring schedule where legal.  Auto prefers the ring kernel whenever its 256x256 tiling gives
// every CU at least one workgroup.
OIBL_HOOK(int, g_conv_tile, 0);
OIBL_HOOK(long, g_ring_min_tiles, 256);
OIBL_HOOK(int, g_conv_ablate, 0);

template <typename T>
static int launch_conv(const ConvParams& p, int pool, hipStream_t st) {
  using C128x128 = GemmCfg<T, 2, 2, 2, 2>;
  using C128x64 = GemmCfg<T, 2, 2, 2, 1>;
#define OIBL_CONV_DISPATCH(CFG) \
  return pool ? launch_conv_cfg<CFG, true>(p, st) : launch_conv_cfg<CFG, false>(p, st)
  if (p.ksplit > 1 && p.partial && !p.ablate && g_conv_tile == 0) {
    if (p.cout % 128 == 0)
      return pool ? launch_conv_splitk<C128x128, true>(p, st) : launch_conv_splitk<C128x128, false>(p, st);
    return pool ? launch_conv_splitk<C128x64, true>(p, st) : launch_conv_splitk<C128x64, false>(p, st);
  }
  if constexpr (std::is_same<T, bf16x3_t>::value) {
    // ring kernels as in bf16 (pooled layers never write fp32); generic tiles whose staged output
    // (4 bytes per element) fits next to nothing else in LDS: 256 x {128, 64}, 128 x {128, 64}
    using C256x128 = GemmCfg<T, 4, 2, 2, 2>;
    using C256x64 = GemmCfg<T, 4, 2, 2, 1>;
    using C512x64 = GemmCfg<T, 8, 1, 2, 2>;  // Cout = 64 (conv1_2): 64 x 64 per wave instead of 64 x 32
    using C256x64w4 = GemmCfg<T, 4, 1, 2, 2>;  // the same wave tile with 4 waves: 80 KB, two workgroups per CU
    const int rv = (g_regstage || p.ablate || (pool && p.out_f32)) ? 0 : ring_variant(p, 4);
    const long t256 = (p.m_total + 255) / 256;
    const long ring_tiles = rv == 2 ? t256 * (p.cout / 256) : ((p.m_total + 511) / 512) * (p.cout / 128);
    int mode = g_conv_tile;
    if (rv && (mode == 4 || (mode == 0 && ring_tiles >= g_ring_min_tiles))) {
      if (rv == 2)
        return pool ? launch_conv_ring<2, true, RING_X3>(p, st) : launch_conv_ring<2, false, RING_X3>(p, st);
      return pool ? launch_conv_ring<4, true, RING_X3>(p, st) : launch_conv_ring<4, false, RING_X3>(p, st);
    }
    if (mode == 4) mode = 0;
    // (the 512 x 64 tile — 64 x 64 per wave — is kept behind the hook: 2.73 ms vs 2.48 ms for the
    //  256 x 64 tile on conv1_2 at batch 32)
    if (mode == 0) mode = t256 * (p.cout / (p.cout % 128 == 0 ? 128 : 64)) >= 512 ? 2 : 1;
    if (mode == 3 && p.cout % 128 != 0) { OIBL_CONV_DISPATCH(C512x64); }
    if (mode == 5 && p.cout % 128 != 0) { OIBL_CONV_DISPATCH(C256x64w4); }
    if (mode == 5) mode = 2;
    if (mode >= 2) {
      if (p.cout % 128 == 0) { OIBL_CONV_DISPATCH(C256x128); }
      OIBL_CONV_DISPATCH(C256x64);
    }
  } else if constexpr (sizeof(T) == 2) {
    using C256x256 = GemmCfg<T, 2, 4, 4, 2>;
    using C256x128 = GemmCfg<T, 4, 2, 2, 2>;
    using C256x64 = GemmCfg<T, 4, 2, 2, 1>;
    const long t256 = (p.m_total + 255) / 256;
    int mode = g_conv_tile;
    const int rv = (g_regstage || p.ablate) ? 0 : ring_variant(p);
    const long ring_tiles = rv == 2 ? t256 * (p.cout / 256) : ((p.m_total + 511) / 512) * (p.cout / 128);
    if (rv && (mode == 4 || (mode == 0 && ring_tiles >= g_ring_min_tiles))) {
      if (rv == 2) return pool ? launch_conv_ring<2, true>(p, st) : launch_conv_ring<2, false>(p, st);
      return pool ? launch_conv_ring<4, true>(p, st) : launch_conv_ring<4, false>(p, st);
    }
    if (mode == 4) mode = 0;
    if (mode == 0) {
      if (p.cout % 256 == 0 && t256 * (p.cout / 256) >= 512) mode = 3;
      else if (t256 * (p.cout / (p.cout % 128 == 0 ? 128 : 64)) >= 512) mode = 2;
      else mode = 1;
    }
    if (mode == 3 && p.cout % 256 == 0) { OIBL_CONV_DISPATCH(C256x256); }
    if (mode >= 2) {
      if (p.cout % 128 == 0) { OIBL_CONV_DISPATCH(C256x128); }
      OIBL_CONV_DISPATCH(C256x64);
    }
  }
  if (p.cout % 128 == 0) { OIBL_CONV_DISPATCH(C128x128); }
  OIBL_CONV_DISPATCH(C128x64);
#undef OIBL_CONV_DISPATCH
}

// f16mx: the ring kernels are the only implementation (Cin % 64 == 0, Cout % 128 == 0 — every layer
// of the backbone behind the stem)
// Patch of the halo kernel (conv_halo.h) for an Hn x Wn map: PH, PW even, PH * PW <= 256 pixels,
// (PH + 2) * (PW + 2) <= 344 halo lines; fewest patches first, then the smaller halo.
static void halo_patch(int Hn, int Wn, int* PH, int* PW) {
  long best = -1;
  int bh = 2, bw = 2, bhalo = 0;
  for (int ph = 2; ph <= 128; ph += 2) {
    int pwmax = 256 / ph;
    const int hcap = HALO_MAX_POS / (ph + 2) - 2;
    if (hcap < pwmax) pwmax = hcap;
    pwmax &= ~1;
    if (pwmax < 2) continue;
    const int tx = (Wn + pwmax - 1) / pwmax;
    int pw = ((Wn + tx - 1) / tx + 1) & ~1;  // the narrowest even width that still needs tx patches
    if (pw > pwmax) pw = pwmax;
    const long cost = (long)((Hn + ph - 1) / ph) * tx;
    const int halo = (ph + 2) * (pw + 2);
    if (best < 0 || cost < best || (cost == best && halo < bhalo)) {
      best = cost;
      bh = ph;
      bw = pw;
      bhalo = halo;
    }
  }
  *PH = bh;
  *PW = bw;
}

OIBL_HOOK(int, g_halo_var, 0);  // experiment: 3 = the waits that count the halo instructions (rarely wrong: conv_halo.h)
template <bool POOL>
static int launch_conv_halo(const ConvParams& p, hipStream_t st) {
  using G = RingGeo<2>;
  HaloParams q = {};
  q.in = p.in;
  q.w = p.w;
  q.bias = p.bias;
  q.out = p.out;
  q.in_bytes = (unsigned)((size_t)p.N * p.H * p.W * p.cin * 4);
  q.w_bytes = (unsigned)((size_t)9 * p.cout * p.cin * 4);
  q.N = p.N;
  q.H = p.H;
  q.W = p.W;
  q.cin = p.cin;
  q.cout = p.cout;
  const int Hn = POOL ? (p.H / 2) * 2 : p.H, Wn = POOL ? (p.W / 2) * 2 : p.W;
  halo_patch(Hn, Wn, &q.PH, &q.PW);
  q.tiles_y = (Hn + q.PH - 1) / q.PH;
  q.tiles_x = (Wn + q.PW - 1) / q.PW;
  const long tiles_m = (long)p.N * q.tiles_y * q.tiles_x;
  q.tiles_n = p.cout / G::BN;
  OIBL_REQUIRE(tiles_m * q.tiles_n <= 0x7fffffffL, "conv3x3 (halo): grid out of range");
  q.tiles_m = (int)tiles_m;
  q.raster = g_ring_raster;
  ring_magic_u31((unsigned)(q.tiles_y * q.tiles_x), &q.img_mul, &q.img_sh);
  ring_magic_u31((unsigned)q.tiles_x, &q.tx_mul, &q.tx_sh);
  ring_magic_u31((unsigned)(POOL ? q.PW / 2 : q.PW), &q.pw_mul, &q.pw_sh);
  ring_magic_u31((unsigned)(q.PW + 2), &q.hp_mul, &q.hp_sh);
  q.relu = p.relu;
  q.out_f32 = p.out_f32;
  q.range_flag = p.range_flag;
  q.bias_mul = p.bias_mul;
  q.out_mul = p.out_mul;
  const dim3 grid((unsigned)(tiles_m * q.tiles_n));
  if (g_halo_var == 3) {
    auto kern = conv3x3_halo_kernel<POOL, RING_MX_EARLY, 3>;
    OIBL_SET_MAX_LDS(kern, HALO_LDS);
    hipLaunchKernelGGL(kern, grid, dim3(512), HALO_LDS, st, q);
  } else if (g_ring_bar1) {
    auto kern = conv3x3_halo_kernel<POOL, RING_MX_EARLY, 0, true>;
    OIBL_SET_MAX_LDS(kern, HALO_LDS);
    hipLaunchKernelGGL(kern, grid, dim3(512), HALO_LDS, st, q);
  } else {
    auto kern = conv3x3_halo_kernel<POOL, RING_MX_EARLY>;
    OIBL_SET_MAX_LDS(kern, HALO_LDS);
    hipLaunchKernelGGL(kern, grid, dim3(512), HALO_LDS, st, q);
  }
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

// The 128-output-channel layers (conv2_1, conv2_2): conv_halo4.h — 256 pixels x 128 channels, 4 waves, two
// workgroups per CU.
template <bool POOL>
static int launch_conv_halo4(const ConvParams& p, hipStream_t st) {
  HaloParams q = {};
  q.in = p.in;
  q.w = p.w;
  q.bias = p.bias;
  q.out = p.out;
  q.in_bytes = (unsigned)((size_t)p.N * p.H * p.W * p.cin * 4);
  q.w_bytes = (unsigned)((size_t)9 * p.cout * p.cin * 4);
  q.N = p.N;
  q.H = p.H;
  q.W = p.W;
  q.cin = p.cin;
  q.cout = p.cout;
  const int Hn = POOL ? (p.H / 2) * 2 : p.H, Wn = POOL ? (p.W / 2) * 2 : p.W;
  halo_patch(Hn, Wn, &q.PH, &q.PW);
  q.tiles_y = (Hn + q.PH - 1) / q.PH;
  q.tiles_x = (Wn + q.PW - 1) / q.PW;
  const long tiles_m = (long)p.N * q.tiles_y * q.tiles_x;
  q.tiles_n = p.cout / H4_BN;
  OIBL_REQUIRE(tiles_m * q.tiles_n <= 0x7fffffffL, "conv3x3 (halo4): grid out of range");
  q.tiles_m = (int)tiles_m;
  q.raster = g_ring_raster & 255;
  ring_magic_u31((unsigned)(q.tiles_y * q.tiles_x), &q.img_mul, &q.img_sh);
  ring_magic_u31((unsigned)q.tiles_x, &q.tx_mul, &q.tx_sh);
  ring_magic_u31((unsigned)(POOL ? q.PW / 2 : q.PW), &q.pw_mul, &q.pw_sh);
  ring_magic_u31((unsigned)(q.PW + 2), &q.hp_mul, &q.hp_sh);
  q.relu = p.relu;
  q.out_f32 = p.out_f32;
  q.range_flag = p.range_flag;
  q.bias_mul = p.bias_mul;
  q.out_mul = p.out_mul;
  q.prof = g_prof_buf;
  auto kern = conv3x3_halo4_kernel<POOL>;
  OIBL_SET_MAX_LDS(kern, H4_LDS);
  hipLaunchKernelGGL(kern, dim3((unsigned)(tiles_m * q.tiles_n)), dim3(H4_THREADS), H4_LDS, st, q);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

// ---- f16mx: row sub-ranges + split-K on the ring kernels ------------------------------------------
// The ring tiles are 256 x 256 / 512 x 128 outputs and a workgroup owns a CU: a layer with T tiles runs
// ceil(T / 256) rounds and the last one may be nearly empty — conv5_x at batch 32: 300 tiles = one full
// round + 44 workgroups for a second; a single 480x640 image: 10 tiles on 256 CUs.  The plan: the FULL
// rounds run as they are; the REMAINDER tiles are contracted by s workgroups each (s = 3 or 9 in K order
// (tap, chunk): a part is 3 taps / one tap of all channel chunks; s = 2 in order (chunk, tap)), every part
// from zero accumulators into an fp32 partial tensor, and conv_mx_splitk_reduce_kernel adds bias + parts in
// a fixed order, applies ReLU (and the 2x2 max-pool), packs the f16mx lines.  Deterministic; the sums are
// those of the one-pass kernel up to fp32 association.  A pooled layer is split only as a whole (its
// remainder is not a contiguous range of the pooling kernel's quad-major rows).
constexpr int MX_CUS = 256;
struct MxSplitPlan {
  int wm;           // wave rows of the ring instantiation (2: 256 x 256 tiles, 4: 512 x 128)
  int tm_main;      // M tiles of the unsplit part (0: none)
  int tm_rem;       // M tiles of the split part (0: the layer runs in one pass)
  int s;            // parts per remainder tile
  int nsteps_part;  // K-tiles per part
  int outer_step;   // outer K indices per part
  long m_base;      // first GEMM row of the split part
  long rows_part;   // GEMM rows of the split part
};
OIBL_HOOK(int, g_mx_act_shift, 3);   // test hook: log2 of the f16mx backbone's activation down-scale (vgg_forward_impl); 0 = none
OIBL_HOOK(int, g_mx_splitk, 1);   // test hook: 0 = never split; 2 = split, reduced by conv_mx_splitk_reduce_kernel
OIBL_HOOK(int, g_mx_variant, 0);  // test hook: kernel choice of the f16mx layers (launch_conv_mx)
// the 128-output-channel layers run on conv_halo4.h (256-pixel tiles, two workgroups per CU): no ring rounds to balance
static bool mx_halo4_layer(int cin, int cout) {
  if (g_mx_variant == 13) return cout % 128 == 0 && cin % 64 == 0;   // experiment: every layer on conv_halo4.h
  return (g_mx_variant == 0 || g_mx_variant == 3) && cout == 128 && cin % 64 == 0;
}
static MxSplitPlan mx_split_plan(long m_plain, int cin, int cout, int pool, int korder, int wm) {
  MxSplitPlan pl = {};
  pl.wm = wm;
  if (wm == 0 || m_plain <= 0) return pl;   // no ring tiling for this layer (Cout = 64: the stem's conv1_2)
  if (mx_halo4_layer(cin, cout)) {          // one pass, whatever the batch
    pl.tm_main = (int)((m_plain + wm * 128 - 1) / (wm * 128));
    return pl;
  }
  const int bm = wm * 128, tiles_n = cout / (wm == 2 ? 256 : 128);
  const long tm = (m_plain + bm - 1) / bm;
  const long T = tm * tiles_n;
  const int cchunks = cin >> 5, nsteps = 9 * cchunks;
  pl.tm_main = (int)tm;
  if (!g_mx_splitk) return pl;
  // full rounds of whole M-tile columns stay unsplit
  long tm_main = (T / MX_CUS) * MX_CUS / tiles_n;
  if (pool && tm_main != 0) return pl;
  const long r = (tm - tm_main) * tiles_n;
  if (r == 0) return pl;
  // cost in K-tile times: a workgroup pays ~12 K-tiles of prologue + epilogue on top of its contraction
  auto rounds = [](long wgs) { return (wgs + MX_CUS - 1) / MX_CUS; };
  const long base = rounds(r) * (nsteps + 12);
  long best = base;
  int best_s = 1;
  const int cands0[] = {3, 9}, cands1[] = {2, 4};
  for (int ci = 0; ci < 2; ++ci) {
    const int s = korder == 0 ? cands0[ci] : cands1[ci];
    const int outer = korder == 0 ? 9 : cchunks;
    if (outer % s != 0) continue;
    const int part = nsteps / s;
    if (part < 4 || (part & 1)) continue;            // the ring loop wants an even number >= 4 of K-tiles
    const long c = rounds(r * s) * (part + 12) + 4;    // + the reduction pass
    if (c < best) {
      best = c;
      best_s = s;
    }
  }
  // worth it from 10 % of the layer on (two more launches)
  const long whole = (tm_main * tiles_n / MX_CUS) * (nsteps + 12) + base;
  if (best_s == 1 || (base - best) * 10 < whole) return pl;
  pl.tm_main = (int)tm_main;
  pl.tm_rem = (int)(tm - tm_main);
  pl.s = best_s;
  pl.nsteps_part = nsteps / best_s;
  pl.outer_step = (korder == 0 ? 9 : cchunks) / best_s;
  pl.m_base = tm_main * bm;
  pl.rows_part = m_plain - pl.m_base;
  return pl;
}
static int mx_ring_wm(int cin, int cout) {
  if (cin % 64 != 0) return 0;
  if (cout % 256 == 0 && cin % 128 == 0) return 2;
  return cout % 128 == 0 ? 4 : 0;
}
static size_t mx_split_bytes(long m_plain, int cin, int cout, int pool, int korder) {
  const MxSplitPlan pl = mx_split_plan(m_plain, cin, cout, pool, korder, mx_ring_wm(cin, cout));
  return pl.tm_rem ? align_up((size_t)pl.s * pl.rows_part * cout * sizeof(float), 256) : 0;
}

// partial [s][rows_part][cout] fp32 -> out rows (f16mx lines, or fp32 when out_f32): one thread per
// (output row, 32-channel group).  POOL: an output row is a pooled pixel, its four sources are GEMM rows
// (n, 2 yo + dy, 2 xo + dx) in plain pixel order; rows_part then covers the whole layer (m_base = 0).
template <bool POOL>
__global__ void conv_mx_splitk_reduce_kernel(const float* __restrict__ partial, const float* __restrict__ bias,
                                             char* __restrict__ out, long out_row0, long out_rows_here,
                                             long rows_part, int cout, int s, int relu, int out_f32, int H,
                                             int W, unsigned* range_flag, float bias_mul, float out_mul) {
  const int groups = cout >> 5;
  const long items = out_rows_here * groups;
  const size_t part = (size_t)rows_part * cout;
  const int Ho = H >> 1, Wo = W >> 1;
  for (long it = (long)blockIdx.x * blockDim.x + threadIdx.x; it < items; it += (long)gridDim.x * blockDim.x) {
    const long r = it / groups;
    const int g = (int)(it - r * groups);
    float v[32];
#pragma unroll
    for (int e = 0; e < 32; ++e) v[e] = -INFINITY;
#pragma unroll
    for (int q = 0; q < (POOL ? 4 : 1); ++q) {
      long src = r;
      if constexpr (POOL) {
        const long n = r / ((long)Ho * Wo), rem = r - n * (long)Ho * Wo;
        const int yo = (int)(rem / Wo), xo = (int)(rem - (long)yo * Wo);
        src = (n * H + 2 * yo + (q >> 1)) * W + 2 * xo + (q & 1);
      }
      float a[32];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float4 b = *reinterpret_cast<const float4*>(bias + g * 32 + 4 * k);
        a[4 * k] = b.x * bias_mul;
        a[4 * k + 1] = b.y * bias_mul;
        a[4 * k + 2] = b.z * bias_mul;
        a[4 * k + 3] = b.w * bias_mul;
      }
      for (int ks = 0; ks < s; ++ks) {
        const float4* pp = reinterpret_cast<const float4*>(partial + ks * part + (size_t)src * cout + g * 32);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          const float4 t = pp[k];
          a[4 * k] += t.x;
          a[4 * k + 1] += t.y;
          a[4 * k + 2] += t.z;
          a[4 * k + 3] += t.w;
        }
      }
#pragma unroll
      for (int e = 0; e < 32; ++e) v[e] = fmaxf(v[e], a[e]);
    }
    if (relu) {
#pragma unroll
      for (int e = 0; e < 32; ++e) v[e] = fmaxf(v[e], 0.f);
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) v[e] *= out_mul;
    char* dst = out + ((size_t)(out_row0 + r) * cout + g * 32) * 4;
    if (out_f32) {
#pragma unroll
      for (int k = 0; k < 8; ++k)
        reinterpret_cast<float4*>(dst)[k] = make_float4(v[4 * k], v[4 * k + 1], v[4 * k + 2], v[4 * k + 3]);
    } else {
      uint4 line[8];
      mx_pack_line(v, line, range_flag);
#pragma unroll
      for (int k = 0; k < 8; ++k) reinterpret_cast<uint4*>(dst)[k] = line[k];
    }
  }
}

// The same reduction with EIGHT threads per (output row, 32-channel group), four channels each: the sums are
// formed by 8x as many threads as above with fully coalesced 16-byte loads (a single image's conv5_x has 19200
// lines — 75 workgroups of one-thread-per-line, each thread a chain of 9 x 8 loads: the reduce kernels were 22 %
// of a single image's forward, profiles/r04_j_single_image_kernels.md), then handed over through LDS to one
// thread per line that packs and stores it (fp32 output: stored by the eight threads directly).  Same
// operations in the same order per element as the kernel above: bit-identical (tests/test_gpu_splitk.py).
constexpr int RED_LINES = 32;    // lines per workgroup pass = 256 threads / 8
constexpr int RED_PITCH = 36;    // floats per staged line (16-byte aligned rows; a line's reader meets 4-way conflicts)
template <bool POOL>
__global__ __launch_bounds__(256) void conv_mx_splitk_reduce8_kernel(
    const float* __restrict__ partial, const float* __restrict__ bias, char* __restrict__ out, long out_row0,
    long out_rows_here, long rows_part, int cout, int s, int relu, int out_f32, int H, int W, unsigned* range_flag,
    float bias_mul, float out_mul) {
  __shared__ __attribute__((aligned(16))) float stage[RED_LINES * RED_PITCH];
  const int groups = cout >> 5;
  const long items = out_rows_here * groups;
  const size_t part = (size_t)rows_part * cout;
  const int Ho = H >> 1, Wo = W >> 1;
  const int ln = threadIdx.x >> 3, k = threadIdx.x & 7;
  for (long base = (long)blockIdx.x * RED_LINES; base < items; base += (long)gridDim.x * RED_LINES) {
    const long it = base + ln;
    const bool live = it < items;
    const long r = live ? it / groups : 0;
    const int g = live ? (int)(it - r * groups) : 0;
    float4 v = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    if (live) {
      float4 b = *reinterpret_cast<const float4*>(bias + g * 32 + 4 * k);
      b = make_float4(b.x * bias_mul, b.y * bias_mul, b.z * bias_mul, b.w * bias_mul);
#pragma unroll
      for (int q = 0; q < (POOL ? 4 : 1); ++q) {
        long src = r;
        if constexpr (POOL) {
          const long n = r / ((long)Ho * Wo), rem = r - n * (long)Ho * Wo;
          const int yo = (int)(rem / Wo), xo = (int)(rem - (long)yo * Wo);
          src = (n * H + 2 * yo + (q >> 1)) * W + 2 * xo + (q & 1);
        }
        float4 a = b;
        const float* pp = partial + (size_t)src * cout + g * 32 + 4 * k;
        for (int ks = 0; ks < s; ++ks) {
          const float4 t = *reinterpret_cast<const float4*>(pp + ks * part);
          a.x += t.x;
          a.y += t.y;
          a.z += t.z;
          a.w += t.w;
        }
        v = make_float4(fmaxf(v.x, a.x), fmaxf(v.y, a.y), fmaxf(v.z, a.z), fmaxf(v.w, a.w));
      }
      if (relu) v = make_float4(fmaxf(v.x, 0.f), fmaxf(v.y, 0.f), fmaxf(v.z, 0.f), fmaxf(v.w, 0.f));
      v = make_float4(v.x * out_mul, v.y * out_mul, v.z * out_mul, v.w * out_mul);
    }
    if (out_f32) {   // (uniform)
      if (live) *reinterpret_cast<float4*>(out + ((size_t)(out_row0 + r) * cout + g * 32 + 4 * k) * 4) = v;
      continue;
    }
    *reinterpret_cast<float4*>(&stage[ln * RED_PITCH + 4 * k]) = v;
    __syncthreads();
    if (threadIdx.x < RED_LINES && base + threadIdx.x < items) {
      const long it2 = base + threadIdx.x;
      const long r2 = it2 / groups;
      const int g2 = (int)(it2 - r2 * groups);
      float w[32];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float4 t = *reinterpret_cast<const float4*>(&stage[threadIdx.x * RED_PITCH + 4 * j]);
        w[4 * j] = t.x;
        w[4 * j + 1] = t.y;
        w[4 * j + 2] = t.z;
        w[4 * j + 3] = t.w;
      }
      uint4 line[8];
      mx_pack_line(w, line, range_flag);
      uint4* dst = reinterpret_cast<uint4*>(out + ((size_t)(out_row0 + r2) * cout + g2 * 32) * 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) dst[j] = line[j];
    }
    __syncthreads();
  }
}

// f16mx: ring kernels (Cin % 64 == 0, Cout % 128 == 0 — every layer of the backbone behind the stem).
// g_mx_variant (test hook): 0 = that; 3 = the halo kernel (conv_halo.h) for the 256-channel-tile layers —
// 0.58x the LDS-DMA bytes, +5 % on conv3_x, -5 % on conv4_x / conv5_x since the ring's K cursor left its
// LOAD segments (profiles/r03_*): kept as the tested alternative; 2 = ring kernels with the LDS-DMA issue
// inside COMPUTE (RING_MX); 4..8 = timing experiments (wrong results) / stamps.
static int launch_conv_mx_split(const ConvParams& p, int pool, const MxSplitPlan& pl, hipStream_t st) {
  // 1. the full rounds, unsplit, straight into the output (never pooled: see mx_split_plan)
  int rc;
  if (pl.tm_main) {
    RingSub main = {};
    main.tiles_m = pl.tm_main;
    rc = pl.wm == 2 ? launch_conv_ring<2, false, RING_MX_EARLY>(p, st, &main)
                    : launch_conv_ring<4, false, RING_MX_EARLY>(p, st, &main);
    if (rc) return rc;
  }
  // 2. the remainder tiles, s parts each, plain pixel order, raw fp32 accumulators
  ConvParams q = p;
  const long m_plain = (long)p.N * p.H * p.W;
  q.m_total = m_plain;
  q.out_rows = m_plain;
  RingSub rem = {};
  rem.tiles_m = pl.tm_rem;
  rem.m_base = pl.m_base;
  rem.parts = pl.s;
  rem.nsteps_part = pl.nsteps_part;
  rem.outer_step = pl.outer_step;
  rem.out = p.partial;
  rem.out_rows = pl.rows_part;
  rem.part_stride = (size_t)pl.rows_part * p.cout * sizeof(float);
  rc = pl.wm == 2 ? launch_conv_ring<2, false, RING_MX_EARLY>(q, st, &rem)
                  : launch_conv_ring<4, false, RING_MX_EARLY>(q, st, &rem);
  if (rc) return rc;
  // 3. bias + parts in order, ReLU, pool, pack
  const long out_row0 = pool ? 0 : pl.m_base;
  const long out_rows_here = pool ? p.out_rows : pl.rows_part;
  const long items = out_rows_here * (p.cout / 32);
  if (g_mx_splitk != 2) {   // (2 = test hook: the one-thread-per-line reduction below)
    unsigned blocks8 = (unsigned)((items + RED_LINES - 1) / RED_LINES);
    if (blocks8 > 16384) blocks8 = 16384;
    if (pool)
      hipLaunchKernelGGL(conv_mx_splitk_reduce8_kernel<true>, dim3(blocks8), dim3(256), 0, st, p.partial, p.bias,
                         (char*)p.out, out_row0, out_rows_here, pl.rows_part, p.cout, pl.s, p.relu, p.out_f32, p.H,
                         p.W, p.range_flag, p.bias_mul, p.out_mul);
    else
      hipLaunchKernelGGL(conv_mx_splitk_reduce8_kernel<false>, dim3(blocks8), dim3(256), 0, st, p.partial, p.bias,
                         (char*)p.out, out_row0, out_rows_here, pl.rows_part, p.cout, pl.s, p.relu, p.out_f32, p.H,
                         p.W, p.range_flag, p.bias_mul, p.out_mul);
    OIBL_LAUNCH_CHECK();
    return OIBL_OK;
  }
  unsigned blocks = (unsigned)((items + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (pool)
    hipLaunchKernelGGL(conv_mx_splitk_reduce_kernel<true>, dim3(blocks), dim3(256), 0, st, p.partial, p.bias,
                       (char*)p.out, out_row0, out_rows_here, pl.rows_part, p.cout, pl.s, p.relu, p.out_f32, p.H, p.W,
                       p.range_flag, p.bias_mul, p.out_mul);
  else
    hipLaunchKernelGGL(conv_mx_splitk_reduce_kernel<false>, dim3(blocks), dim3(256), 0, st, p.partial, p.bias,
                       (char*)p.out, out_row0, out_rows_here, pl.rows_part, p.cout, pl.s, p.relu, p.out_f32, p.H, p.W,
                       p.range_flag, p.bias_mul, p.out_mul);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

static int launch_conv_mx(const ConvParams& p, int pool, hipStream_t st) {
  // row sub-ranges + split-K where the tiling leaves a nearly empty round (needs the partial scratch)
  if (p.partial && g_mx_variant <= 1 && !(pool && p.out_f32) && ring_variant(p, 4)) {
    const MxSplitPlan pl = mx_split_plan((long)p.N * p.H * p.W, p.cin, p.cout, pool, p.korder, mx_ring_wm(p.cin, p.cout));
    if (pl.tm_rem) return launch_conv_mx_split(p, pool, pl, st);
  }
  const int rv = (pool && p.out_f32) ? 0 : ring_variant(p, 4);
  // the halo kernel (conv_halo.h) where it is the faster one: the 256-output-channel layers at 120 x 160
  // (conv3_1..conv3_3: 0.57 / 1.00 / 0.95 ms against 0.60 / 1.04 / 0.97 on the ring; deeper layers lose 5-10 %).
  // Hook: 1 = ring kernels only, 3 = halo kernel wherever it applies.
  const bool halo_ok = rv == 2 && ((p.cin >> 5) & 1) == 0;
  if (halo_ok && (g_mx_variant == 3 || (g_mx_variant == 0 && p.cout == 256)))
    return pool ? launch_conv_halo<true>(p, st) : launch_conv_halo<false>(p, st);
  // the 4-wave halo kernel (conv_halo4.h) for the 128-output-channel layers (rv == 4: conv2_1 / conv2_2): a third
  // of the ring's L2 -> LDS bytes per K-tile.  Hook: 1 = ring kernels; 13 = EVERY layer on it (experiment: conv3_x
  // ties with the 8-wave halo kernel, conv4_x / conv5_x lose 12-16 %, profiles/r05_*_timing.txt).
  if ((rv == 4 || (g_mx_variant == 13 && rv == 2)) && mx_halo4_layer(p.cin, p.cout))
    return pool ? launch_conv_halo4<true>(p, st) : launch_conv_halo4<false>(p, st);
  if (g_mx_variant == 2) {
    if (rv == 2) return pool ? launch_conv_ring<2, true, RING_MX>(p, st) : launch_conv_ring<2, false, RING_MX>(p, st);
    if (rv == 4) return pool ? launch_conv_ring<4, true, RING_MX>(p, st) : launch_conv_ring<4, false, RING_MX>(p, st);
  }
  if (g_mx_variant == 8 && rv == 2 && !pool) return launch_conv_ring<2, false, RING_MX_PROF>(p, st);
  if (g_mx_variant == 8 && rv == 4 && !pool) return launch_conv_ring<4, false, RING_MX_PROF>(p, st);
  if (g_mx_variant >= 4 && g_mx_variant <= 7 && rv == 2 && !pool) {   // timing experiments (wrong results)
    switch (g_mx_variant) {
      case 4: return launch_conv_ring<2, false, RING_MX_NOMFMA>(p, st);
      case 5: return launch_conv_ring<2, false, RING_MX_NODMA>(p, st);
      case 6: return launch_conv_ring<2, false, RING_MX_NOREAD>(p, st);
      default: return launch_conv_ring<2, false, RING_MX_NOBAR>(p, st);
    }
  }
  if (rv == 2) return pool ? launch_conv_ring<2, true, RING_MX_EARLY>(p, st) : launch_conv_ring<2, false, RING_MX_EARLY>(p, st);
  if (rv == 4) return pool ? launch_conv_ring<4, true, RING_MX_EARLY>(p, st) : launch_conv_ring<4, false, RING_MX_EARLY>(p, st);
  set_error("conv3x3 (f16mx): unsupported layer cin=%d cout=%d at N=%d H=%d W=%d (needs Cin %% 64 == 0, "
            "Cout %% 128 == 0 and an input below 3.5 GB)", p.cin, p.cout, p.N, p.H, p.W);
  return OIBL_E_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// Cin = 64 convolutions (conv1_2, conv2_1) — "resident weights + LDS halo" kernel, bf16.
// With K = 9 * 64 the generic implicit GEMM has only nine K-steps per tile: its cost is the
// per-tile prologue / epilogue and the L2 -> LDS traffic of re-fetching every input pixel once per
// tap, not the matrix cores.  Here a persistent workgroup
//   * keeps ALL weights of its 64-output-channel slice in LDS (9 taps x 64 x 64 bf16 = 72 KiB,
//     fetched once),
//   * stages the (8+2) x (32+2) pixel halo of an 8 x 32 output tile ONCE (42.5 KiB instead of
//     9 x 32 KiB), double-buffered so the next tile's halo streams in (global_load_lds) while the
//     current one is multiplied,
//   * runs the 9 taps x 4 k-steps = 144 MFMAs per wave straight out of LDS with no barrier,
//   * reuses the consumed halo buffer as the staging area of the coalesced NHWC store.
// LDS: 72 KiB + 2 x 43 KiB = 158 KiB of the CU's 160 KiB -> one workgroup (4 waves) per CU.
// The halo image is XOR-swizzled by f(hy, hx) = ((hx >> 1) & 7) ^ ((hy & 1) << 2) (halo rows are
// 34 x 128 B = 17 bank rows, so banks depend on hx only; the hy term separates the two image rows
// a pooled quad spans): every ds_read_b128 lane group is conflict-free for all nine taps in both
// the linear and the quad-major (pooling) pixel order (checked exhaustively).
// ---------------------------------------------------------------------------------------------
constexpr int C64_HW = 34;                         // halo width  (32 + 2)
constexpr int C64_HALO_ROWS = 344;                 // 10 x 34 = 340 halo pixels, padded to 43 x 8
constexpr int C64_HALO_BYTES = C64_HALO_ROWS * 128;
constexpr int C64_W_BYTES = 9 * 64 * 128;
constexpr int C64_LDS_BYTES = C64_W_BYTES + 2 * C64_HALO_BYTES;
constexpr int C64_HALO_LOADS = 11;                 // ceil(43 wave-instructions / 4 waves)
constexpr int C64_WAVE_REGION = 10880;             // per-wave epilogue staging inside a halo buffer
OIBL_HOOK(int, g_conv_c64, 1);

struct C64Params {
  const char* in;
  const char* w;
  const float* bias;
  char* out;
  const char* zero;
  int N, H, W, cout, relu;
  int tiles_x, tiles_y;
  int ntiles;
  unsigned long long* prof;  // optional (test hook): per-phase shader-clock totals of block 0 wave 0
};

__device__ static inline int c64_swz(int hy, int hx) { return ((hx >> 1) & 7) ^ ((hy & 1) << 2); }

template <bool POOL>
__global__ __launch_bounds__(256, 1) void conv3x3_c64_kernel(C64Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wl = smem;
  char* const hb = smem + C64_W_BYTES;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int co0 = blockIdx.y * 64;

  // weights of this 64-channel slice -> LDS rows r = tap * 64 + c (same swizzle as the GEMM core)
  {
    const int piece = ((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) * 16;
#pragma unroll
    for (int j = 0; j < 18; ++j) {
      const int q = j * 4 + wave;
      const int r = q * 8 + (lane >> 3);
      const int tap = r >> 6, c = r & 63;
      glds16(p.w + ((long)(tap * p.cout + co0 + c) * 64) * 2 + piece, wl + q * 1024);
    }
  }
  // halo loader: instruction j of this wave fills halo pixels r = (4j + wave) * 8 + (lane >> 3)
  int h_rel[C64_HALO_LOADS];    // hy << 8 | hx   (hy >= 10: padding row, always zero)
  int h_piece[C64_HALO_LOADS];
#pragma unroll
  for (int j = 0; j < C64_HALO_LOADS; ++j) {
    const int r = (j * 4 + wave) * 8 + (lane >> 3);
    const int hy = r / C64_HW, hx = r - hy * C64_HW;
    h_rel[j] = (hy << 8) | hx;
    h_piece[j] = ((lane & 7) ^ c64_swz(hy, hx)) * 16;
  }
  const char* const zsrc = p.zero + (lane & 7) * 16;
  const long row_bytes = (long)p.W * 128, img_bytes = (long)p.H * row_bytes;

  auto issue_halo = [&](int tile, char* buf) {
    const unsigned r2 = (unsigned)tile / (unsigned)p.tiles_x;
    const int tx = tile - (int)r2 * p.tiles_x;
    const int n = (int)(r2 / (unsigned)p.tiles_y), ty = (int)r2 - n * p.tiles_y;
    const int y0 = ty * 8 - 1, x0 = tx * 32 - 1;
    const char* base = p.in + n * img_bytes;
#pragma unroll
    for (int j = 0; j < C64_HALO_LOADS; ++j) {
      const int q = j * 4 + wave;
      if (q < C64_HALO_ROWS / 8) {  // wave-uniform
        const int hy = h_rel[j] >> 8, hx = h_rel[j] & 255;
        const int y = y0 + hy, x = x0 + hx;
        const bool ok = hy < 10 && y >= 0 && y < p.H && x >= 0 && x < p.W;
        glds16(ok ? base + y * row_bytes + (long)x * 128 + h_piece[j] : zsrc, buf + q * 1024);
      }
    }
  };

  // lane geometry inside the wave's 2 tile rows x 32 columns
  int lhy[2], lhx[2];  // halo coordinates (tap 0,0) of this lane's pixel in M-tile i
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (POOL) {
      lhy[i] = 2 * wave + ((l31 >> 1) & 1);
      lhx[i] = 16 * i + 2 * (l31 >> 2) + (l31 & 1);
    } else {
      lhy[i] = 2 * wave + i;
      lhx[i] = l31;
    }
  }
  int w_off[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) w_off[kk] = l31 * 128 + (((2 * kk + half) ^ ((l31 >> 1) & 7)) << 4);
  float bvals[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) bvals[tn] = p.bias[co0 + tn * 32 + l31];

  int tile = blockIdx.x;
  if (tile < p.ntiles) issue_halo(tile, hb);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  const bool prof = p.prof != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0;
  unsigned long long pt[6] = {0, 0, 0, 0, 0, 0};
#define C64_TICK(i)                                        \
  if (prof) {                                              \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    pt[i] += now_ - t_prev;                                \
    t_prev = now_;                                         \
  }
  for (int it = 0; tile < p.ntiles; tile += gridDim.x, ++it) {
    unsigned long long t_prev = prof ? __builtin_amdgcn_s_memtime() : 0;
    char* const cur = hb + (it & 1) * C64_HALO_BYTES;
    const int nxt = tile + (int)gridDim.x;
    if (nxt < p.ntiles) issue_halo(nxt, hb + ((it & 1) ^ 1) * C64_HALO_BYTES);
    C64_TICK(0)

    f32x16_t acc[2][2];  // start at the bias, like every convolution kernel here
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][tn][r] = bvals[tn];

    // 36 steps (tap-major, 4 k-steps per tap), fragment reads software-pipelined one step ahead:
    // with one wave per SIMD nothing else hides the LDS latency.
    bf16x8_t fa[2][2], fb[2][2];
    auto load_step = [&](int sidx, bf16x8_t (&a)[2], bf16x8_t (&b)[2]) {
      const int tap = sidx >> 2, kk = sidx & 3;
      const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int hy = lhy[i] + ky, hx = lhx[i] + kx;
        a[i] = *reinterpret_cast<const bf16x8_t*>(
            cur + (hy * C64_HW + hx) * 128 + (((2 * kk + half) ^ c64_swz(hy, hx)) << 4));
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
        b[tn] = *reinterpret_cast<const bf16x8_t*>(wl + tap * 8192 + tn * 4096 + w_off[kk]);
    };
    load_step(0, fa[0], fb[0]);
#pragma unroll
    for (int sidx = 0; sidx < 36; ++sidx) {
      if (sidx + 1 < 36) load_step(sidx + 1, fa[(sidx + 1) & 1], fb[(sidx + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);  // keep the next step's reads AHEAD of this step's MFMAs
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[i][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sidx & 1][i], fb[sidx & 1][tn],
                                                               acc[i][tn], 0, 0, 0);
    }
    // The next tile's halo (issued before the MFMAs) has landed; every wave is done reading `cur`,
    // which now becomes the store staging area.  Raw s_barrier + explicit counters: __syncthreads()
    // would also wait (vmcnt(0)) for the global stores of the epilogue below, a 1-2 us bubble per
    // tile; they are left in flight and only waited for after the next tile's MFMAs.
    C64_TICK(1)
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    C64_TICK(2)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    C64_TICK(3)

    const unsigned r2 = (unsigned)tile / (unsigned)p.tiles_x;
    const int tx = tile - (int)r2 * p.tiles_x;
    const int n = (int)(r2 / (unsigned)p.tiles_y), ty = (int)r2 - n * p.tiles_y;
    char* const st = cur + wave * C64_WAVE_REGION;  // wave-private: no barrier for the exchange
    if (POOL) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float v = fmaxf(fmaxf(acc[i][tn][4 * g], acc[i][tn][4 * g + 1]),
                            fmaxf(acc[i][tn][4 * g + 2], acc[i][tn][4 * g + 3]));
            if (p.relu) v = fmaxf(v, 0.f);
            const int qx = 8 * i + 2 * g + half;
            *reinterpret_cast<uint16_t*>(st + qx * 144 + (tn * 32 + l31) * 2) = f32_to_bf16_bits(v);
          }
      __builtin_amdgcn_wave_barrier();
      const int Ho = p.H >> 1, Wo = p.W >> 1;
      const int oy = ty * 4 + wave;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int idx = k * 64 + lane, qx = idx >> 3, part = idx & 7;
        const int ox = tx * 16 + qx;
        const uint4 v = *reinterpret_cast<const uint4*>(st + qx * 144 + part * 16);
        if (oy < Ho && ox < Wo)
          *reinterpret_cast<uint4*>(p.out + (((long)n * Ho + oy) * Wo + ox) * p.cout * 2 + co0 * 2 +
                                    part * 16) = v;
      }
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            float v = acc[i][tn][r];
            if (p.relu) v = fmaxf(v, 0.f);
            const int prow = i * 32 + acc_row(r, lane);
            *reinterpret_cast<uint16_t*>(st + prow * 144 + (tn * 32 + l31) * 2) = f32_to_bf16_bits(v);
          }
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int idx = k * 64 + lane, prow = idx >> 3, part = idx & 7;
        const int y = ty * 8 + 2 * wave + (prow >> 5), x = tx * 32 + (prow & 31);
        const uint4 v = *reinterpret_cast<const uint4*>(st + prow * 144 + part * 16);
        if (y < p.H && x < p.W)
          *reinterpret_cast<uint4*>(p.out + (((long)n * p.H + y) * p.W + x) * p.cout * 2 + co0 * 2 +
                                    part * 16) = v;
      }
    }
    // staging reads retired before the next iteration's LDS-DMA overwrites this buffer
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    C64_TICK(4)
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    C64_TICK(5)
  }
#undef C64_TICK
  if (prof && lane == 0) {
#pragma unroll
    for (int i = 0; i < 6; ++i) p.prof[i] = pt[i];
  }
}

static int launch_conv_c64(const void* in, int N, int H, int W, const void* w, const float* bias,
                           int cout, int relu, int pool, void* out, hipStream_t st) {
  C64Params p;
  p.in = (const char*)in;
  p.w = (const char*)w;
  p.bias = bias;
  p.out = (char*)out;
  p.zero = (const char*)zero_line_device_ptr();
  OIBL_REQUIRE(p.zero != nullptr, "conv3x3: zero line symbol not found");
  p.N = N;
  p.H = H;
  p.W = W;
  p.cout = cout;
  p.relu = relu;
  p.prof = g_prof_buf;
  p.tiles_x = (W + 31) / 32;
  p.tiles_y = (H + 7) / 8;
  const long nt = (long)N * p.tiles_x * p.tiles_y;
  OIBL_REQUIRE(nt < 0x7fffffffL, "conv3x3: too many tiles");
  p.ntiles = (int)nt;
  const int slices = cout / 64;
  int gx = 256 / slices;  // one resident workgroup per CU
  if (gx < 1) gx = 1;
  if (gx > p.ntiles) gx = p.ntiles;
  auto kern = pool ? conv3x3_c64_kernel<true> : conv3x3_c64_kernel<false>;
  if (pool)
    OIBL_SET_MAX_LDS(conv3x3_c64_kernel<true>, C64_LDS_BYTES);
  else
    OIBL_SET_MAX_LDS(conv3x3_c64_kernel<false>, C64_LDS_BYTES);
  hipLaunchKernelGGL(kern, dim3(gx, slices), dim3(256), C64_LDS_BYTES, st, p);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

// ---------------------------------------------------------------------------------------------
// VGG stem, fused (bf16): conv1_1 + ReLU + conv1_2 + ReLU + 2x2 max-pool in ONE launch.
//   x [N][3][H][W] fp32  ->  out [N][H/2][W/2][64] bf16
// Unfused, conv1_1 writes its 64-channel output (39 MB / image) to HBM only for conv1_2 to read it
// straight back: together 23 % of the step for 12.6 % of its FLOPs.  Here the conv1_1 activations
// never leave the CU.  A persistent workgroup of 8 waves is split by role (waves w and w + 4 share
// a SIMD, so every SIMD hosts one wave of each role):
//   producers (waves 4-7): for the NEXT 8 x 32 output tile, gather the 3-channel input window of
//     the (8+2) x (32+2) halo straight from global memory (coalesced dword buffer loads, an
//     out-of-image tap is an out-of-range offset -> 0), run conv1_1 on the matrix cores (K = 27
//     padded to 32, transposed GEMM as in conv1_1_mfma_kernel: 4 MFMAs per 32 halo pixels),
//     bias + ReLU + bf16, and write the halo tile into LDS in exactly the swizzled image the
//     conv1_2 main loop reads (halo pixels outside the image are written as zeros: conv1_2's
//     padding);
//   consumers (waves 0-3): the conv3x3_c64_kernel main loop (resident conv1_2 weights, 144 MFMAs
//     per wave per tile out of LDS), pool in registers, store the pooled pixels directly.
// One raw s_barrier per tile hands the halo buffers over (two buffers, producer one tile ahead).
// Numerics are those of the unfused bf16 path, operation for operation (same MFMA k order, same
// rounding points) -> bit-identical output.
// LDS: 72 KiB conv1_2 weights + 2 x 42.5 KiB halo = 157 KiB.
// ---------------------------------------------------------------------------------------------
constexpr int ST_HALO_PX = 10 * C64_HW;              // 340
constexpr int ST_HALO_BYTES = ST_HALO_PX * 128;      // 43520
constexpr int ST_LDS_BYTES = C64_W_BYTES + 2 * ST_HALO_BYTES;
constexpr int ST_LUT_ENTRIES = 3 * 257;              // uint8 input: per channel 256 values + "0.0"
constexpr int ST_LDS_BYTES_U8 = ST_LDS_BYTES + 1552;
constexpr int ST_BLOCKS = (ST_HALO_PX + 31) / 32;    // 11 blocks of 32 halo pixels
constexpr unsigned ST_OOB = 0xF0000000u;
OIBL_HOOK(int, g_stem_fused, 1);
OIBL_HOOK(int, g_stem_u8, 1);     // test hook: 0 = uint8 input of the bf16x3 / f16mx stems through the normalising pass
OIBL_HOOK(int, g_stem3_prio, 0);  // bf16x3 stem, test hook: producer issue priority | consumer priority << 2

struct StemParams {
  const void* x;     // U8 = false: [N][3][H][W] fp32 (normalised); U8 = true: [N][H][W][3] uint8
  float mean[3], stdv[3];  // U8 only: the loader's Normalize constants
  const float* w1;   // conv1_1 [64][3][3][3] fp32
  const float* b1;
  const char* w2;    // conv1_2 packed [9][64][64] bf16
  const float* b2;
  char* out;
  unsigned x_bytes;
  int N, H, W;
  int tiles_x, tiles_y, ntiles;
  unsigned long long* prof;  // optional (test hook): shader-clock totals of block 0, waves 0 and 4
  int prod_prio;             // bf16x3 stem: issue priority of the producer role outside its MFMAs
  unsigned* range_flag;      // f16mx stem: raised when a conv1_1 / conv1_2 output hits the fp16 bound; may be null
  float act_scale = 1.f;     // f16mx stem: conv1_1's weights and both biases are multiplied by this (g_mx_act_shift)
};

// U8 = true: the input is the loader's raw uint8 NHWC image; ToTensor + Normalize
// (ibl/utils/data/__init__.py:40-41: (u / 255 - mean) / std in fp32) followed by the bf16 rounding
// of the operand is a pure function of (channel, byte), so the producers look it up in a 3 x 257
// table built in LDS at kernel start with exactly that arithmetic (entry 256 = 0.0: conv1_1's
// zero padding) — bit-identical to feeding the normalised fp32 tensor, a quarter of the bytes.
template <bool U8>
__global__ __launch_bounds__(512) void vgg_stem_kernel(StemParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wl = smem;
  char* const hb = smem + C64_W_BYTES;
  const uint16_t* const lut = reinterpret_cast<const uint16_t*>(smem + ST_LDS_BYTES);
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  const int first = blockIdx.x, stride = gridDim.x;
  int niter = 0;
  if (first < p.ntiles) niter = (p.ntiles - first + stride - 1) / stride;
  const int Ho = p.H >> 1, Wo = p.W >> 1;
  if constexpr (U8) {
    uint16_t* lw = reinterpret_cast<uint16_t*>(smem + ST_LDS_BYTES);
    for (int i = threadIdx.x; i < ST_LUT_ENTRIES; i += 512) {
      const int c = i / 257, u = i - 257 * c;
      const float q = (float)u / 255.0f;
      lw[i] = u == 256 ? (uint16_t)0 : f32_to_bf16_bits((q - p.mean[c]) / p.stdv[c]);
    }
    __syncthreads();
  }

  if (wave >= 4) {
    // ================================ producers ================================================
    const int pw = wave - 4;
    // The consumer wave on this SIMD keeps the matrix pipe saturated and, being the older wave,
    // wins every arbitration: without a raised priority the producer's dozen MFMAs per tile only
    // issue once the consumer's loop has ended and the two roles serialise.  The priority is
    // raised around the MFMAs only; everything else in this role runs in the consumer's shadow.
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    // conv1_1 weights as the A operand: wf[t][s] element e <-> cout = 32 t + l31, k = 16 s + 8 half + e
    bf16x8_t wf[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 16 * s + 8 * half + e;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const float v = k < 27 ? p.w1[(32 * t + l31) * 27 + k] : 0.f;
          wf[t][s][e] = (short)f32_to_bf16_bits(v);
        }
      }
    float bb[2][16];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 16; ++r) bb[t][r] = p.b1[32 * t + acc_row(r, lane)];
    const int plane = p.H * p.W;

    // Tile-independent lane geometry.  Block b of this wave covers halo pixels r = 32 b + l31
    // (clamped to the last pixel for the 12 surplus lanes of block 10: they load valid data that is
    // never written).  g_rel = element offset of the pixel from the tile's halo origin, g_dk[j] =
    // element offset of input element k(j) from the pixel (j = 8 s + e, k = 16 s + 8 half + e;
    // slots with k >= 27 carry zero weights and simply re-read a valid tap of the same window).
    int g_row[3], g_swz[3], g_hy[3], g_hx[3], g_rel[3];
#pragma unroll
    for (int bi = 0; bi < 3; ++bi) {
      const int r = 32 * (pw + 4 * bi) + l31;
      const int rc = r < ST_HALO_PX ? r : ST_HALO_PX - 1;
      g_row[bi] = r;
      g_hy[bi] = rc / C64_HW;
      g_hx[bi] = rc - g_hy[bi] * C64_HW;
      g_swz[bi] = c64_swz(g_hy[bi], g_hx[bi]);
      g_rel[bi] = (g_hy[bi] * p.W + g_hx[bi]) * (U8 ? 3 : 1);
      asm volatile("" : "+v"(g_rel[bi]));
    }
    int g_dk[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      int k = 16 * (j >> 3) + 8 * half + (j & 7);
      if (k >= 27) k -= 8;
      const int c = k / 9, t = k - 9 * c;
      g_dk[j] = U8 ? ((t / 3 - 1) * p.W + (t % 3 - 1)) * 3 + c
                   : c * plane + (t / 3 - 1) * p.W + (t % 3 - 1);
      asm volatile("" : "+v"(g_dk[j]));  // keep it in a register: re-deriving it costs a v_mul per load
    }
    // U8: table row (channel) of slot j — two compile-time candidates selected by the lane half
    auto lut_row = [&](int j) __attribute__((always_inline)) {
      const int kA = 16 * (j >> 3) + (j & 7), kB = kA + 8 >= 27 ? kA : kA + 8;
      return (half ? kB / 9 : kA / 9) * 257;
    };
    auto tap_of = [&](int j) __attribute__((always_inline)) {  // border tiles only
      const int kA = 16 * (j >> 3) + (j & 7), kB = kA + 8 >= 27 ? kA : kA + 8;
      return half ? kB % 9 : kA % 9;
    };

    float xv[3][16];
    auto decode = [&](int tile, int& n, int& ty, int& tx) __attribute__((always_inline)) {
      const unsigned r2 = (unsigned)tile / (unsigned)p.tiles_x;
      tx = tile - (int)r2 * p.tiles_x;
      n = (int)(r2 / (unsigned)p.tiles_y);
      ty = (int)r2 - n * p.tiles_y;
    };
    // a tile is interior when every halo pixel and every tap of every halo pixel lies in the image
    auto is_interior = [&](int ty, int tx) __attribute__((always_inline)) {
      return ty >= 1 && ty * 8 + 10 <= p.H && tx >= 1 && tx * 32 + 34 <= p.W;
    };
    // issue the input gathers of one tile (48 coalesced dword loads per lane); nothing waits here
    auto issue_loads = [&](int tile) __attribute__((always_inline)) {
      int n, ty, tx;
      decode(tile, n, ty, tx);
      const int y0 = ty * 8 - 1, x0 = tx * 32 - 1;
      // element (fp32) / byte (uint8) offset of the halo origin; may be "negative" for border tiles
      const int origin = U8 ? ((n * p.H + y0) * p.W + x0) * 3 : ((n * 3) * p.H + y0) * p.W + x0;
      if (is_interior(ty, tx)) {
#pragma unroll
        for (int bi = 0; bi < 3; ++bi) {
          if (pw + 4 * bi >= ST_BLOCKS) continue;  // wave-uniform
          const int base = origin + g_rel[bi];
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            if constexpr (U8)
              xv[bi][j] = __builtin_bit_cast(
                  float, (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_x, base + g_dk[j], 0, 0));
            else
              xv[bi][j] = __builtin_bit_cast(
                  float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (base + g_dk[j]) * 4, 0, 0));
          }
        }
      } else {
#pragma unroll
        for (int bi = 0; bi < 3; ++bi) {
          if (pw + 4 * bi >= ST_BLOCKS) continue;
          const int y = y0 + g_hy[bi], x = x0 + g_hx[bi];
          unsigned mk = 0;
          if (y >= 0 && y < p.H && x >= 0 && x < p.W) {
            const bool ya = y > 0, yc = y + 1 < p.H, xa = x > 0, xc = x + 1 < p.W;
            mk = (ya && xa ? 1u : 0u) | (ya ? 2u : 0u) | (ya && xc ? 4u : 0u) | (xa ? 8u : 0u) | 16u |
                 (xc ? 32u : 0u) | (yc && xa ? 64u : 0u) | (yc ? 128u : 0u) | (yc && xc ? 256u : 0u);
          }
          const int base = origin + g_rel[bi];
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const bool ok = (mk >> tap_of(j)) & 1u;
            if constexpr (U8) {
              // an invalid tap must read as 0.0 AFTER normalisation: table entry 256
              const unsigned off = ok ? (unsigned)(base + g_dk[j]) : ST_OOB;
              const unsigned u = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_x, (int)off, 0, 0);
              xv[bi][j] = __builtin_bit_cast(float, ok ? u : 256u);
            } else {
              const unsigned off = ok ? (unsigned)(base + g_dk[j]) * 4u : ST_OOB;
              xv[bi][j] = __builtin_bit_cast(float,
                                             __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)off, 0, 0));
            }
          }
        }
      }
    };
    // conv1_1 on the loaded window, bias + ReLU + bf16, halo tile -> LDS
    auto finish = [&](int tile, char* buf) __attribute__((always_inline)) {
      int n, ty, tx;
      decode(tile, n, ty, tx);
      const int y0 = ty * 8 - 1, x0 = tx * 32 - 1;
      const bool interior = is_interior(ty, tx);
#pragma unroll
      for (int bi = 0; bi < 3; ++bi) {
        if (pw + 4 * bi >= ST_BLOCKS) continue;  // wave-uniform
        bf16x8_t xf[2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int e = 0; e < 8; ++e) {
            if constexpr (U8)
              xf[s][e] = (short)lut[lut_row(8 * s + e) + (int)__builtin_bit_cast(unsigned, xv[bi][8 * s + e])];
            else
              xf[s][e] = (short)f32_to_bf16_bits(xv[bi][8 * s + e]);
          }
        f32x16_t acc[2];
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] = bb[t][r];
#pragma unroll
          for (int s = 0; s < 2; ++s)
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[t][s], xf[s], acc[t], 0, 0, 0);
        }
        __builtin_amdgcn_s_setprio(0);
        // D[row = cout][col = pixel]: registers 4g..4g+3 = couts 32 t + 8 g + 4 half + 0..3 of the
        // lane's pixel -> 8 bytes of its 128-byte halo row, 16-B slot 4 t + g (swizzled), +8 half.
        // A halo pixel outside the image is conv1_2's zero padding, not a conv1_1 output.
        const int y = y0 + g_hy[bi], x = x0 + g_hx[bi];
        const bool pix_ok = interior || (y >= 0 && y < p.H && x >= 0 && x < p.W);
        if (g_row[bi] < ST_HALO_PX) {
          char* row = buf + g_row[bi] * 128 + 8 * half;
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              uint2 v;
              v.x = relu_bf16x2(pack_bf16x2(acc[t][4 * g], acc[t][4 * g + 1]));
              v.y = relu_bf16x2(pack_bf16x2(acc[t][4 * g + 2], acc[t][4 * g + 3]));
              if (!pix_ok) v = make_uint2(0u, 0u);
              *reinterpret_cast<uint2*>(row + (((4 * t + g) ^ g_swz[bi]) << 4)) = v;
            }
        }
      }
    };

    const bool prof = p.prof != nullptr && blockIdx.x == 0 && wave == 4;
    unsigned long long pt[2] = {0, 0};
    // producer runs one tile ahead of the consumers; the gathers of the tile after that are
    // already in flight while it waits at the hand-over barrier
    if (niter > 0) {
      issue_loads(first);
      finish(first, hb);
      if (niter > 1) issue_loads(first + stride);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int it = 0; it < niter; ++it) {
      const unsigned long long t0 = prof ? __builtin_amdgcn_s_memtime() : 0;
      if (it + 1 < niter) {
        finish(first + (it + 1) * stride, hb + ((it + 1) & 1) * ST_HALO_BYTES);
        if (it + 2 < niter) issue_loads(first + (it + 2) * stride);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long t1 = prof ? __builtin_amdgcn_s_memtime() : 0;
      __builtin_amdgcn_s_barrier();
      if (prof) {
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        pt[0] += t1 - t0;
        pt[1] += t2 - t1;
      }
    }
    if (prof && lane == 0) {
      p.prof[4] = pt[0];
      p.prof[5] = pt[1];
    }
    return;
  }

  // ================================== consumers ================================================
  {
    const int piece = ((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) * 16;
#pragma unroll
    for (int j = 0; j < 18; ++j) {
      const int q = j * 4 + wave;
      const int r = q * 8 + (lane >> 3);
      const int tap = r >> 6, c = r & 63;
      glds16(p.w2 + ((long)(tap * 64 + c) * 64) * 2 + piece, wl + q * 1024);
    }
  }
  int lhy[2], lhx[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    lhy[i] = 2 * wave + ((l31 >> 1) & 1);
    lhx[i] = 16 * i + 2 * (l31 >> 2) + (l31 & 1);
  }
  int w_off[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) w_off[kk] = l31 * 128 + (((2 * kk + half) ^ ((l31 >> 1) & 7)) << 4);
  float bvals[2];
#pragma unroll
  for (int tn = 0; tn < 2; ++tn) bvals[tn] = p.b2[tn * 32 + l31];

  const bool cprof = p.prof != nullptr && blockIdx.x == 0 && wave == 0;
  unsigned long long ct[3] = {0, 0, 0};
  // The pooled outputs of tile i are held back (8 packed registers) and stored one pixel at a time
  // between the matrix steps of tile i+1, so the store issue hides in the MFMA shadow.  A pending
  // pixel is ONE dword per lane: lanes l31 and l31 ^ 1 exchange halves (DPP + v_perm), the even lane
  // stores channels (l31, l31 + 1), the odd lane channels (32 + l31 - 1, 32 + l31) — a half-wave
  // writes the full 128-byte line of its pixel — through a buffer descriptor of ONE OUTPUT ROW
  // (rebuilt per tile from scalars; zero records for a row below the map): a pixel right of the map
  // is out of range and dropped by the hardware, its distance is the instruction's immediate.  No
  // mask, no compare, no branch in the matrix loop (as in vgg_stem_x3_kernel).
  uint32_t pend[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) pend[e] = 0;
  __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0, 0x00020000);  // nothing pending
  const unsigned lane_off = half * 128 + ((l31 & 1) ? 64 + (l31 - 1) * 2 : l31 * 2);
  const uint32_t psel = (l31 & 1) ? 0x03020706u : 0x05040100u;
  unsigned poff = lane_off;
  auto store_px = [&](int e) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_buffer_store_b32(pend[e], rs_o, (int)(poff + 256u * e), 0, 0);
  };
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), as the builtin: see the f16mx consumers
  __builtin_amdgcn_s_barrier();
  for (int it = 0; it < niter; ++it) {
    const unsigned long long c0 = cprof ? __builtin_amdgcn_s_memtime() : 0;
    const int tile = first + it * stride;
    const char* const cur = hb + (it & 1) * ST_HALO_BYTES;
    f32x16_t acc[2][2];  // start at the bias, like every convolution kernel here
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][tn][r] = bvals[tn];
    bf16x8_t fa[2][2], fb[2][2];
    auto load_step = [&](int sidx, bf16x8_t (&a)[2], bf16x8_t (&b)[2]) __attribute__((always_inline)) {
      const int tap = sidx >> 2, kk = sidx & 3;
      const int ky = tap / 3, kx = tap - 3 * ky;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int hy = lhy[i] + ky, hx = lhx[i] + kx;
        a[i] = *reinterpret_cast<const bf16x8_t*>(
            cur + (hy * C64_HW + hx) * 128 + (((2 * kk + half) ^ c64_swz(hy, hx)) << 4));
      }
#pragma unroll
      for (int tn = 0; tn < 2; ++tn)
        b[tn] = *reinterpret_cast<const bf16x8_t*>(wl + tap * 8192 + tn * 4096 + w_off[kk]);
    };
    load_step(0, fa[0], fb[0]);
#pragma unroll
    for (int sidx = 0; sidx < 36; ++sidx) {
      if (sidx + 1 < 36) load_step(sidx + 1, fa[(sidx + 1) & 1], fb[(sidx + 1) & 1]);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          acc[i][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[sidx & 1][i], fb[sidx & 1][tn],
                                                               acc[i][tn], 0, 0, 0);
      if ((sidx & 3) == 1 && (sidx >> 2) < 8) store_px(sidx >> 2);
    }
    // all fragment reads of `cur` have been consumed by the MFMAs above: hand the buffer back
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = cprof ? __builtin_amdgcn_s_memtime() : 0;
    __builtin_amdgcn_s_barrier();
    const unsigned long long c2 = cprof ? __builtin_amdgcn_s_memtime() : 0;
    // pooled epilogue from registers: lanes 0-31 / 32-63 each hold the 32 channels of one pooled
    // pixel (64 contiguous bytes); the two tn halves (low / high 16 bits) complete the 128-byte line
    const unsigned r2 = (unsigned)tile / (unsigned)p.tiles_x;
    const int tx = tile - (int)r2 * p.tiles_x;
    const int n = (int)(r2 / (unsigned)p.tiles_y), ty = (int)r2 - n * p.tiles_y;
    const int oy = ty * 4 + wave;
    // lane's first pooled pixel (i = g = 0) of this wave's pooled row; pixel e = 4 i + g is 2 e
    // pixels = 256 e bytes further: immediate offsets on one base offset into the row's descriptor
    rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out + ((long)n * Ho + oy) * Wo * 128, 0,
                                             oy < Ho ? Wo * 128 : 0, 0x00020000);
    poff = lane_off + (unsigned)tx * (16 * 128);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float v[2];
#pragma unroll
        for (int tn = 0; tn < 2; ++tn)
          v[tn] = fmaxf(fmaxf(fmaxf(acc[i][tn][4 * g], acc[i][tn][4 * g + 1]),
                              fmaxf(acc[i][tn][4 * g + 2], acc[i][tn][4 * g + 3])), 0.f);
        const uint32_t mine = pack_bf16x2(v[0], v[1]);
        const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);  // lane ^ 1
        pend[4 * i + g] = __builtin_amdgcn_perm(other, mine, psel);
      }
    if (cprof) {
      const unsigned long long c3 = __builtin_amdgcn_s_memtime();
      ct[0] += c1 - c0;
      ct[1] += c2 - c1;
      ct[2] += c3 - c2;
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) store_px(e);
  if (cprof && lane == 0) {
    p.prof[0] = ct[0];
    p.prof[1] = ct[1];
    p.prof[2] = ct[2];
  }
}

// mean3 / std3: host pointers, used only when U8
template <bool U8>
static int launch_vgg_stem(const void* x, int N, int H, int W, const float* mean3, const float* std3,
                           const float* w1, const float* b1, const void* packed_w2, const float* b2,
                           void* out, hipStream_t st) {
  StemParams p;
  p.x = x;
  for (int c = 0; c < 3; ++c) {
    p.mean[c] = U8 ? mean3[c] : 0.f;
    p.stdv[c] = U8 ? std3[c] : 1.f;
  }
  p.w1 = w1;
  p.b1 = b1;
  p.w2 = (const char*)packed_w2;
  p.b2 = b2;
  p.out = (char*)out;
  p.x_bytes = (unsigned)((size_t)N * 3 * H * W * (U8 ? 1 : 4));
  p.N = N;
  p.H = H;
  p.W = W;
  p.tiles_x = (W + 31) / 32;
  p.tiles_y = (H + 7) / 8;
  const long nt = (long)N * p.tiles_x * p.tiles_y;
  OIBL_REQUIRE(nt < 0x7fffffffL, "vgg stem: too many tiles");
  p.ntiles = (int)nt;
  p.prof = g_prof_buf;
  int gx = 256;  // one persistent workgroup per CU
  if (gx > p.ntiles) gx = p.ntiles;
  constexpr int lds = U8 ? ST_LDS_BYTES_U8 : ST_LDS_BYTES;
  auto kern = vgg_stem_kernel<U8>;
  OIBL_SET_MAX_LDS(kern, lds);
  hipLaunchKernelGGL(kern, dim3(gx), dim3(512), lds, st, p);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

// ---------------------------------------------------------------------------------------------
// VGG stem, fused, bf16x3: conv1_1 + ReLU + conv1_2 + ReLU + 2x2 max-pool in ONE launch.
//   x [N][3][H][W] fp32  ->  out [N][H/2][W/2][64] bf16x3 ((hi, lo) groups)
// Unfused, conv1_1 writes 2.5 GB of split activations per 32 images (0.86 ms, write-bound) and
// conv1_2 — Cout = 64: no ring tile shape — runs on the generic core at 45 % of the matrix pipe
// (2.5 ms): 22 % of the bf16x3 step for 13 % of its FLOPs.  The bf16 stem's plan (72 KiB of
// conv1_2 weights + two 42.5 KiB halo buffers of all 64 channels) does not survive 4-byte
// elements; this kernel keeps its producer / consumer structure and splits the work so that it does:
//   * a workgroup serves HALF of conv1_2's output channels (blockIdx.y = 0 / 1; both halves run
//     at the same time on different CUs and gather the same input window through L2): 9 taps x 2
//     channel halves x 32 cout x 128 B = 72 KiB of split weights stay resident in LDS;
//   * a tile is consumed in two PASSES, one per half of conv1_2's input channels: a halo buffer holds
//     340 pixels x [32 hi | 32 lo] of ONE 32-channel half (42.5 KiB, exactly the bf16 stem's buffer)
//     and the two buffers alternate between the passes.  Producers (waves 4-15) run conv1_1 for the
//     32 channels of the next pass (K = 27 padded to 32: lo.hi + hi.lo + hi.hi, 6 MFMAs per 32 halo
//     pixels; twelve waves, one 32-pixel block of the halo each), ReLU, split, and write the halo
//     image the consumers read; the gathered input window
//     and its (hi, lo) fragments are kept in registers for both passes of a tile.  Consumers (waves
//     0-3) accumulate both passes in registers (9 taps x 12 MFMAs per pass), then pool and store.
// Only 4 waves read LDS for the main contraction (the generic kernel has 16 competing for it).
// Numerics: operation for operation those of conv1_1_mfma_kernel<X3> followed by the generic
// bf16x3 convolution in K order (channel chunk, tap) — bit-identical to that pair (tested).
// LDS: 72 KiB + 2 x 42.5 KiB = 157 KiB.
// ---------------------------------------------------------------------------------------------
// two fp32 values -> the dword of their hi parts and the dword of their lo parts (packed bf16 pairs)
__device__ static inline void x3_split_pair(float v0, float v1, uint32_t& hi, uint32_t& lo) {
  hi = pack_bf16x2(v0, v1);
  lo = pack_bf16x2(v0 - __builtin_bit_cast(float, hi << 16), v1 - __builtin_bit_cast(float, hi & 0xffff0000u));
}

// the lane id, recomputed where it is needed (two VALU instructions, never hoisted): at 128 VGPRs the f16mx stem has
// no register to keep `lane >> 5` alive across its loops — the allocator parked it in scratch and reloaded it once
// per tile in front of a vmcnt(0) (round 6, after the b128 tails took three more registers)
__device__ static inline int fresh_lane_id() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// -DOIBL_STEM_R6_LDS (debug library; tests/gpu_stem_lds_ab.py): the conflict-free LDS access pattern built in round 6
// — MX tails as ONE ds_read_b128 (scale = register 7) and producer lanes 0-7 on a block's even pixels.  It removes
// 83 % of the kernel's SQ_LDS_BANK_CONFLICT cycles (2.01e8 -> 3.47e7 per launch; tools/lds_stem_model.py predicts
// 2612 -> 452 per tile) and a quarter of its LDS-active cycles — and the launch is NOT faster: 1.609 against 1.598 ms,
// slower in each of five alternating rounds (profiles/r06_d_stem_lds_ab.txt).  The consumers' passes do shrink
// (6.4k -> 5.2k cycles per tile) but the time moves into their waits: the tile is bound by what the four waves of a
// SIMD can ISSUE on the vector ALU (1374 VALU wave-instructions per SIMD and tile = 5.5k of its 9.8k cycles at four
// cycles each, three quarters of them the producers' conv1_1 + line packing, which BOTH workgroups of a tile run), and
// the b128 operands cost 6 % more VALU instructions (register moves).  The product keeps the rounds 3-5 pattern.
// ... for the two-workgroups-per-tile kernel (-DOIBL_STEM_SPLIT).  With ONE workgroup per tile (S3_DUAL below: the
// producers' work per consumer pass is halved) the consumers' passes are what bounds the tile again, and the
// conflict-free pattern is the default there.
#if defined(OIBL_STEM_R6_LDS) || !defined(OIBL_STEM_SPLIT)
constexpr bool S3_TAIL128 = true, S3_LANE_PERM = true;
#else
constexpr bool S3_TAIL128 = false, S3_LANE_PERM = false;
#endif
// ONE workgroup per tile for the f16mx stem (round 6): conv1_1 + the f16mx line packing of a tile's 340-pixel halo — the
// vector-ALU work that bounds the kernel — is done ONCE instead of in both workgroups of the tile, and that workgroup
// runs BOTH halves of conv1_2's output channels one after the other over the same two halo buffers:
//   stage S1  consumers: channels 0-31,  input half 0 (halo buffer 0)   producers: conv1_1 half 1 of this tile -> buffer 1
//   stage S2  consumers: channels 0-31,  input half 1 (buffer 1)         producers: convert the next tile's window, gather the one after
//             consumers: pool / pack / store channels 0-31
//   stage S3  consumers: channels 32-63, input half 0 (buffer 0)         producers: —
//   stage S4  consumers: channels 32-63, input half 1 (buffer 1)         producers: conv1_1 half 0 of the NEXT tile -> buffer 0
//             consumers: pool / pack / store channels 32-63
// (buffer 0 is last read in S3, buffer 1 in S4: the producers keep their one-stage run-ahead).  conv1_2's weights no
// longer fit as residents (2 x 72 KB): the 36 KB image of the NEXT stage (its output-channel half x input half: 9 taps x
// 32 rows x 128 B) streams from L2 into the weight buffer the PREVIOUS stage used while the current one computes —
// buffer_load ... lds issued by the consumers at the head of their pass, vmcnt(0) in front of the stage's barrier:
// 147 KB per tile through the LDS-DMA path, the same 72 KB of LDS.  -DOIBL_STEM_SPLIT restores the two workgroups per tile.
#ifdef OIBL_STEM_SPLIT
constexpr bool S3_DUAL = false;
#else
constexpr bool S3_DUAL = true;
#endif
constexpr int S3_W_BYTES = 2 * 9 * 32 * 128;
constexpr int S3_BIAS_OFF = S3_W_BYTES + 2 * ST_HALO_BYTES;   // conv1_1 bias: 64 floats; MX: + this half's 32 of conv1_2
constexpr int S3_LDS_BYTES = S3_BIAS_OFF + 256 + 256;    // conv1_1's 64 biases + conv1_2's (32 of a half; all 64: S3_DUAL)

// MX = true: the f16mx stem.  Same roles, tiles, passes and hand-overs; what changes is the arithmetic of
// conv1_2 (2 f16 + 1 scaled-fp6 MFMA per 32 K instead of 6 bf16 ones) and therefore every format:
//   * the halo buffers hold f16mx lines (common.h) of conv1_1's output.  A 32-channel group lies along the
//     ROWS of conv1_1's accumulator tile — a lane holds 16 of a pixel's 32 channels, lane ^ 32 the rest — and
//     which channel sits in which row is free: row 8g + 4hp + r carries channel 16hp + 4g + r, so that a lane
//     owns 16 CONSECUTIVE elements of the line and packs them alone (mx_pack_half: one cross-lane maximum);
//   * conv1_2 runs transposed for the same reason — A = weights (rows = output channels, in the same row
//     order), B = halo pixels — its 2x2 max-pool is two DPP quad steps over lanes, and the pooled pixel's
//     32 channels are again one lane pair: the f16mx lines of the output map are written straight from
//     registers (a workgroup's 32 output channels are exactly one group);
//   * conv1_2's weights are the packed f16mx tensor of oibl_pack_conv3x3_weights(OIBL_F16MX).
// conv1_1 itself stays split bf16 (K = 27: 6 MFMAs per 32 pixels x 32 channels, a quarter of a pass).
// U8 = true: the input is the loader's raw uint8 NHWC image (StemParams::x [N][H][W][3]).  The producers gather
// a pixel's 3 x 3 x 3 window as THREE 12-byte loads (one per window row: 9 consecutive bytes = 3 pixels x RGB,
// fetched from the enclosing aligned dwords and shifted into place), the K slots of conv1_1 then follow the
// bytes — (ky, kx, c) order, the lower lane half the first 16 of the 27, the upper half the rest — and
// ToTensor + Normalize become ONE fma per value: v = u * a_c + b_c with a_c = 1 / (255 std_c), b_c =
// -mean_c / std_c (StemParams::mean = a, ::stdv = b, rounded from double on the host).  That is NOT the loader's
// three rounded operations (u / 255 - mean) / std, but within 2^-16 of them on values up to 151 — the rounding
// the loader's own intermediate carries (u / 255 - mean is rounded at magnitude <= 1, then scaled by 255); over
// all 768 (channel, byte) pairs the bf16 hi parts conv1_1 multiplies are identical and 42 lo parts differ by one
// unit (tests/test_stem_u8_cpu.py, exhaustive) — far inside the arithmetic's own error.  (The
// exact route, a 3 x 257 table of split values as in the bf16 stem, needs 3 KB of LDS: this kernel has 1.6 KB
// left; three exactly rounded operations per value are ~110 more VALU instructions per tile and lane on the
// role that is already issue-bound.)  Out-of-image taps are zeroed AFTER the normalisation (conv1_1 pads the
// normalised tensor), from a 27-bit validity mask per lane, on border tiles only.
template <bool MX, bool U8 = false>
__global__ __launch_bounds__(1024) void vgg_stem_x3_kernel(StemParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* const wl = smem;                 // [pass h][tap][32 cout][32 hi | 32 lo of input channels 32h..]
  char* const hb = smem + S3_W_BYTES;    // halo buffer h: channels 32h..32h+31 of the current tile
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int half = lane >> 5, l31 = lane & 31;
  constexpr bool DUAL = MX && S3_DUAL;   // one workgroup per tile, both output-channel halves in turn (above)
  // The tiles of a workgroup: tile `first + it * stride`, it < niter.  Workgroups are dealt to the 8 XCDs round-robin
  // (id % 8), and every XCD has its own L2: with tile = blockIdx.x + it * gridDim.x (rounds 2-5) the four neighbours
  // of a tile — whose 10 x 34-pixel halo windows overlap its own by a third — were gathered through four OTHER L2s
  // and the kernel fetched 256 MB for a 118 MB input.  Round 6: XCD x owns the contiguous eighth [x T / 8, (x + 1) T / 8)
  // of the tile sequence (whole tile rows of whole images) and deals it to its gridDim.x / 8 workgroups.
  int first = blockIdx.x, stride = gridDim.x, tile_end = p.ntiles;
  if ((gridDim.x & 7) == 0) {
    const int x = blockIdx.x & 7;
    stride = gridDim.x >> 3;
    first = (int)((long)p.ntiles * x / 8) + (int)(blockIdx.x >> 3);
    tile_end = (int)((long)p.ntiles * (x + 1) / 8);
  }
  const int co0 = DUAL ? 0 : blockIdx.y * 32;       // this workgroup's (first) conv1_2 output channels
  int niter = 0;
  if (first < tile_end) niter = (tile_end - first + stride - 1) / stride;
  const int nstages = 2 * niter;
  const int Ho = p.H >> 1, Wo = p.W >> 1;
  // conv1_1's bias lives in LDS (the producers have no registers to spare for 2 x 16 values per lane)
  // (f16mx: conv1_1's weights and both biases carry the activation scale — every activation of the kernel, the
  //  halo tile in LDS included, is stored scaled; vgg_forward_impl)
  if (threadIdx.x < 64) reinterpret_cast<float*>(smem + S3_BIAS_OFF)[threadIdx.x] = p.b1[threadIdx.x] * (MX ? p.act_scale : 1.f);
  if (MX && threadIdx.x >= 64 && threadIdx.x < (DUAL ? 128 : 96))
    reinterpret_cast<float*>(smem + S3_BIAS_OFF)[threadIdx.x] = p.b2[co0 + threadIdx.x - 64] * (MX ? p.act_scale : 1.f);
  __syncthreads();
  // f16mx range guard (common.h): the largest group maximum this lane has packed, in a register; the flag —
  // a kernel argument and a global store — is touched once, behind the role's loops, whose lgkmcnt / vmcnt
  // waits are counted by hand (a scalar argument load inside them was seen to break the counts)
  float range_seen = 0.f;
  auto raise_if_out_of_range = [&]() __attribute__((always_inline)) {
    if constexpr (MX) {
      if (__builtin_amdgcn_ballot_w64(range_seen >= 65504.f) != 0) {
        if (range_seen >= 65504.f) mx_raise_range_flag(p.range_flag);
      }
    }
  };

  if (wave >= 4) {
    // ================================ producers (waves 4-15) ====================================
    // Twelve producer waves (three per SIMD, one 32-pixel block each) against four consumers: with one
    // producer wave per SIMD handling three blocks, the VALU-heavy role (two splits per value, ~1400
    // dependent VALU instructions per tile at ~8 cycles each) was 1.7x the consumers' time; three
    // waves per SIMD hide each other's instruction latency.  The kernel therefore runs 16 waves per
    // workgroup at <= 128 VGPRs.  prod_prio (test hook) sets the roles' issue priorities (producers:
    // outside their MFMAs, which always go out at priority 3).  Measured (tests/gpu_stem3_prof.py):
    // every setting lands within 5 % — skewed priorities save shader cycles per tile (10.6k vs
    // 12.9k), the chip answers with a lower clock and the wall time is 2.07-2.19 ms either way; equal
    // priorities (the default, 0 / 0) are the fastest.
    const int pprio = p.prod_prio & 3;
    if (pprio == 1) __builtin_amdgcn_s_setprio(1);
    else if (pprio == 2) __builtin_amdgcn_s_setprio(2);
    else if (pprio == 3) __builtin_amdgcn_s_setprio(3);
    const int pw = wave - 4;   // 0..11: producer wave pw owns block pw of the 11 (wave 15 only keeps step)
    const bool has_block = pw < ST_BLOCKS;
    const __amdgpu_buffer_rsrc_t rs_x =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    // conv1_1 weights (A operand), split: w?[h][s] element e <-> channel 32 h + l31, k = 16 s + 8 half + e
    bf16x8_t wh[2][2], wlo[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        // K slot (s, half, e) of the 32: bf16x3 — window element k = 16 s + 8 half + e (c, ky, kx order);
        // f16mx — the window travels as ROWS (below): slot 8 s + e of a lane half is element kx = idx % 3 of
        // its row idx / 3, the lower half owning rows (c, ky) = 0..4, the upper half rows 5..8
        int k = 16 * s + 8 * half + e;
        if (MX) {
          const int idx = 8 * s + e;
          k = idx < (half ? 12 : 15) ? (idx / 3 + (half ? 5 : 0)) * 3 + idx % 3 : 27;
        }
        if (U8) {   // slot 8 s + e of lane half `half` = window byte idx in (ky, kx, c) order; w1 is [c][ky][kx]
          const int idx = 16 * half + 8 * s + e;
          k = idx < 27 ? (idx % 3) * 9 + (idx / 9) * 3 + (idx % 9) / 3 : 27;
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int ch = MX ? 16 * ((l31 >> 2) & 1) + 4 * (l31 >> 3) + (l31 & 3) : l31;   // channel of row l31
          const float v = k < 27 ? p.w1[(32 * h + ch) * 27 + k] * (MX ? p.act_scale : 1.f) : 0.f;
          uint16_t hi, lo;
          x3_split(v, hi, lo);
          wh[h][s][e] = (short)hi;
          wlo[h][s][e] = (short)lo;
        }
      }
    // accumulator rows of this lane: channels 8 j + 4 half + 0..3 (j = 0..3) of the pass's 32
    // (MX: registers 4 j + r = channels 16 half + 4 j + r)
    const float* const bias_l = reinterpret_cast<const float*>(smem + S3_BIAS_OFF) + (MX ? 16 : 4) * half;
    const int plane = p.H * p.W;

    // tile-independent lane geometry (as in vgg_stem_kernel)
    // halo pixel of this lane in block bi: r = 32 (pw + 4 bi) + l31 (clamped to the last pixel for the
    // 12 surplus lanes of block 10), (hy, hx) = (r / 34, r % 34); kept per block: the swizzle of its
    // LDS row and its element offset from the tile's halo origin (the rest is recomputed where needed)
    // f16mx (round 6): lanes 0-7 of a block take its EVEN pixels 0, 2, .., 14, lanes 8-15 the odd ones (and so on
    // for pixels 16-31).  A line write is serviced in contiguous 8-lane groups over 32 banks, and the slot swizzle
    // ((hx >> 1) & 7) gives two neighbouring pixels the same slot: with lane = pixel every 16-byte write of a line
    // was a 2-way conflict (884 extra LDS cycles per tile, tools/lds_stem_model.py — with the consumers' tail reads,
    // below, the 2.0e8 SQ_LDS_BANK_CONFLICT cycles per launch of profiles/r05_z_pmc.md: 2604 per tile measured,
    // 2612 modelled).  Which halo pixel a producer lane computes is free: only this function says.
    const int lpix = (MX && S3_LANE_PERM) ? 2 * (l31 & 7) + ((l31 >> 3) & 1) + 16 * (l31 >> 4) : l31;
    auto row_of = [&](int bi) __attribute__((always_inline)) { return 32 * (pw + bi) + lpix; };
    auto hyx_of = [&](int bi, int& hy, int& hx) __attribute__((always_inline)) {
      const int r = row_of(bi), rc = r < ST_HALO_PX ? r : ST_HALO_PX - 1;
      hy = rc / C64_HW;
      hx = rc - hy * C64_HW;
    };
    int g_swz[1], g_rel[1];
#pragma unroll
    for (int bi = 0; bi < 1; ++bi) {
      int hy, hx;
      hyx_of(bi, hy, hx);
      g_swz[bi] = c64_swz(hy, hx);
      g_rel[bi] = hy * p.W + hx;
      asm volatile("" : "+v"(g_rel[bi]));
    }
    // element offset of input element k(j) from the pixel, j = 8 s + e, k = 16 s + 8 half + e (slots
    // with k >= 27 carry zero weights and re-read a valid tap): wave-uniform per lane half, so the two
    // candidates stay in scalar registers and a lane selects (no 16 VGPRs per lane)
    auto dk_of = [&](int j) __attribute__((always_inline)) {
      const int kA = 16 * (j >> 3) + (j & 7), kB = kA + 8 >= 27 ? kA : kA + 8;
      const int oA = (kA / 9) * plane + ((kA % 9) / 3 - 1) * p.W + ((kA % 9) % 3 - 1);
      const int oB = (kB / 9) * plane + ((kB % 9) / 3 - 1) * p.W + ((kB % 9) % 3 - 1);
      int hsel = half;
      asm volatile("" : "+v"(hsel));   // not loop-invariant for the compiler: no 16 hoisted VGPRs
      return hsel ? oB : oA;
    };
    auto tap_of = [&](int j) __attribute__((always_inline)) {  // border tiles only
      const int kA = 16 * (j >> 3) + (j & 7), kB = kA + 8 >= 27 ? kA : kA + 8;
      return half ? kB % 9 : kA % 9;
    };
    auto decode = [&](int tile, int& n, int& ty, int& tx) __attribute__((always_inline)) {
      const unsigned r2 = (unsigned)tile / (unsigned)p.tiles_x;
      tx = tile - (int)r2 * p.tiles_x;
      n = (int)(r2 / (unsigned)p.tiles_y);
      ty = (int)r2 - n * p.tiles_y;
    };
    auto is_interior = [&](int ty, int tx) __attribute__((always_inline)) {
      return ty >= 1 && ty * 8 + 10 <= p.H && tx >= 1 && tx * 32 + 34 <= p.W;
    };
    float xv[1][16];
    int xfix[1] = {0};   // f16mx, image edge: 1 = this pixel's rows were fetched from x (not x - 1), 2 = from x - 2
    // U8: the three window rows as fetched (aligned dwords), the byte shift of row 0, the validity mask of the
    // 27 taps (border tiles) and whether the tile in flight is a border tile
    unsigned xr[3][3] = {}, xsh = 0, xmask = 0;
    bool xborder = false;
    // per-lane Normalize constants in slot order: slot s of this lane half is channel (s + half) % 3
    float ka[3] = {0.f, 0.f, 0.f}, kb[3] = {0.f, 0.f, 0.f};
    if constexpr (U8) {
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        ka[i] = half ? p.mean[(i + 1) % 3] : p.mean[i];
        kb[i] = half ? p.stdv[(i + 1) % 3] : p.stdv[i];
      }
    }
    auto issue_loads = [&](int tile) __attribute__((always_inline)) {
      int n, ty, tx;
      decode(tile, n, ty, tx);
      const int y0 = ty * 8 - 1, x0 = tx * 32 - 1;
      const int origin = ((n * 3) * p.H + y0) * p.W + x0;
      if constexpr (U8) {
        const bool interior = is_interior(ty, tx);
        xborder = !interior;
        if (has_block) {
          constexpr int bi = 0;
          // byte offset of the window's first byte: pixel (y - 1, x - 1), channel 0 (negative at the image's
          // first pixels: the unsigned offset is then out of range and the load returns zeros)
          const int b0 = ((n * p.H + y0) * p.W + x0 + g_rel[bi] - p.W - 1) * 3;
          xsh = (unsigned)b0 & 3u;
          bool in = true, ya = true, yc = true;
          if (!interior) {
            int hy, hx;
            hyx_of(bi, hy, hx);
            const int y = y0 + hy, x = x0 + hx;
            in = y >= 0 && y < p.H && x >= 0 && x < p.W;
            ya = y > 0;
            yc = y + 1 < p.H;
            const bool xa = x > 0, xc = x + 1 < p.W;
            // bit ky * 9 + kx * 3 + c: the tap is inside the image
            const unsigned rows = (ya ? 0x1ffu : 0u) | 0x3fe00u | (yc ? 0x7fc0000u : 0u);
            const unsigned cols = (xa ? 0x0040201u * 7u : 0u) | (0x0040201u * 7u << 3) | (xc ? 0x0040201u * 7u << 6 : 0u);
            xmask = in ? rows & cols : 0u;
          }
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int offs = (b0 + ky * 3 * p.W) & ~3;
            unsigned off = (unsigned)offs;
            bool early = false;   // the tensor's very first pixel: its row starts 3 bytes before the tensor
            if (!interior) {
              const bool ok = in && (ky == 0 ? ya : ky == 2 ? yc : true);
              early = ok && offs < 0;
              off = !ok ? ST_OOB : early ? 0u : off;
            }
            const auto d = __builtin_amdgcn_raw_buffer_load_b96(rs_x, (int)off, 0, 0);
            const unsigned d0 = d[0], d1 = d[1], d2 = d[2];   // (elements through scalars: see the f16mx branch)
            xr[ky][0] = early ? 0u : d0;          // (fetched from 0: one dword late)
            xr[ky][1] = early ? d0 : d1;
            xr[ky][2] = early ? d1 : d2;
          }
        }
      } else if constexpr (MX) {
        // The 3 x 3 x 3 window as nine ROWS (c, ky) of three consecutive pixels x - 1 .. x + 1: five 12-byte
        // loads per lane (the lower lane half rows 0-4, the upper half rows 5-8 and row 8 once more under zero
        // weights) instead of sixteen 4-byte ones — the sixteen were ~3k cycles of the texture path per tile.
        // Image edges: a row above / below the image is fetched from the out-of-range offset (zeros);
        // at x = 0 the fetch starts at x, at x = W - 1 at x - 2 (no byte outside the tensor is ever touched)
        // and convert() moves the elements into place.
        const bool interior = is_interior(ty, tx);
        if (has_block) {
          constexpr int bi = 0;
          const int hsel = fresh_lane_id() >> 5;
          bool in = true, ya = true, yc = true;
          int shift = -1;
          xfix[bi] = 0;
          if (!interior) {
            int hy, hx;
            hyx_of(bi, hy, hx);
            const int y = y0 + hy, x = x0 + hx;
            in = y >= 0 && y < p.H && x >= 0 && x < p.W;
            ya = y > 0;
            yc = y + 1 < p.H;
            xfix[bi] = !in ? 0 : x == 0 ? 1 : x + 1 >= p.W ? 2 : 0;
            shift = xfix[bi] == 1 ? 0 : xfix[bi] == 2 ? -2 : -1;
          }
          const int base = origin + g_rel[bi] + shift;
#pragma unroll
          for (int i = 0; i < 5; ++i) {
            const int rA = i, rB = i + 5 < 9 ? i + 5 : 8;
            const int oA = (rA / 3) * plane + (rA % 3 - 1) * p.W, oB = (rB / 3) * plane + (rB % 3 - 1) * p.W;
            const int kyA = rA % 3, kyB = rB % 3;
            unsigned off = (unsigned)(base + (hsel ? oB : oA)) * 4u;
            if (!interior) {
              const int ky = hsel ? kyB : kyA;
              const bool ok = in && (ky == 0 ? ya : ky == 2 ? yc : true);
              off = ok ? off : ST_OOB;
            }
            // (elements through scalars: __builtin_bit_cast applied to a vector-element lvalue reads element 0)
            const auto d = __builtin_amdgcn_raw_buffer_load_b96(rs_x, (int)off, 0, 0);
            const unsigned d0 = d[0], d1 = d[1], d2 = d[2];
            xv[bi][3 * i] = __builtin_bit_cast(float, d0);
            xv[bi][3 * i + 1] = __builtin_bit_cast(float, d1);
            xv[bi][3 * i + 2] = __builtin_bit_cast(float, d2);
          }
          xv[bi][15] = 0.f;
        }
      } else if (is_interior(ty, tx)) {
#pragma unroll
        for (int bi = 0; bi < 1; ++bi) {
          if (!has_block) continue;  // wave-uniform
          const int base = origin + g_rel[bi];
#pragma unroll
          for (int j = 0; j < 16; ++j)
            xv[bi][j] = __builtin_bit_cast(
                float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (base + dk_of(j)) * 4, 0, 0));
        }
      } else {
#pragma unroll
        for (int bi = 0; bi < 1; ++bi) {
          if (!has_block) continue;
          int hy, hx;
          hyx_of(bi, hy, hx);
          const int y = y0 + hy, x = x0 + hx;
          unsigned mk = 0;
          if (y >= 0 && y < p.H && x >= 0 && x < p.W) {
            const bool ya = y > 0, yc = y + 1 < p.H, xa = x > 0, xc = x + 1 < p.W;
            mk = (ya && xa ? 1u : 0u) | (ya ? 2u : 0u) | (ya && xc ? 4u : 0u) | (xa ? 8u : 0u) | 16u |
                 (xc ? 32u : 0u) | (yc && xa ? 64u : 0u) | (yc ? 128u : 0u) | (yc && xc ? 256u : 0u);
          }
          const int base = origin + g_rel[bi];
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const bool ok = (mk >> tap_of(j)) & 1u;
            const unsigned off = ok ? (unsigned)(base + dk_of(j)) * 4u : ST_OOB;
            xv[bi][j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_x, (int)off, 0, 0));
          }
        }
      }
    };
    // the gathered window as (hi, lo) B fragments, kept for both passes of the tile
    bf16x8_t xh[1][2], xl[1][2];
    auto convert = [&]() __attribute__((always_inline)) {
      typedef __attribute__((ext_vector_type(4))) uint32_t u32x4_t;
#pragma unroll
      for (int bi = 0; bi < 1; ++bi) {
        if (!has_block) continue;
        if constexpr (U8) {
          // each row's 9 window bytes shifted into place: w[ky][0..1] = bytes 0-7, w[ky][2] byte 0 = byte 8
          const unsigned w3 = (unsigned)(3 * p.W) & 3u;
          unsigned w[3][3];
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const unsigned sh = (xsh + ky * w3) & 3u;
            w[ky][0] = __builtin_amdgcn_alignbyte(xr[ky][1], xr[ky][0], sh);
            w[ky][1] = __builtin_amdgcn_alignbyte(xr[ky][2], xr[ky][1], sh);
            w[ky][2] = xr[ky][2] >> (8u * sh);
          }
          // the 16 bytes of this lane half: lower = row 0 bytes 0-8, row 1 bytes 0-6; upper = row 1 bytes 7-8,
          // row 2 bytes 0-8, five slots under zero weights (any finite value)
          const unsigned a2 = __builtin_amdgcn_perm(w[1][0], w[0][2], 0x06050400u);
          const unsigned a3 = __builtin_amdgcn_alignbyte(w[1][1], w[1][0], 3u);
          const unsigned t = __builtin_amdgcn_perm(w[1][2], w[1][1], 0x00000403u);
          const unsigned b0 = __builtin_amdgcn_perm(w[2][0], t, 0x05040100u);
          const unsigned b1 = __builtin_amdgcn_alignbyte(w[2][1], w[2][0], 2u);
          const unsigned b2 = __builtin_amdgcn_alignbyte(w[2][2], w[2][1], 2u) & 0x00ffffffu;
          const int hsel = fresh_lane_id() >> 5;
          const unsigned dw[4] = {hsel ? b0 : w[0][0], hsel ? b1 : w[0][1], hsel ? b2 : a2, hsel ? 0u : a3};
#pragma unroll
          for (int j = 0; j < 16; ++j) {
            const float u = (float)((dw[j >> 2] >> (8 * (j & 3))) & 0xffu);   // v_cvt_f32_ubyteN
            xv[bi][j] = fmaf(u, ka[j % 3], kb[j % 3]);
          }
          if (xborder) {   // wave-uniform: zero the taps outside the image (conv1_1 pads the NORMALISED tensor)
            const unsigned mk = xmask >> (hsel ? 16 : 0);
#pragma unroll
            for (int j = 0; j < 16; ++j) xv[bi][j] = ((mk >> j) & 1u) ? xv[bi][j] : 0.f;
          }
        }
        if constexpr (MX && !U8) {
          if (__builtin_amdgcn_ballot_w64(xfix[bi] != 0) != 0) {   // an image edge inside this block (rare)
            const bool left = xfix[bi] == 1, right = xfix[bi] == 2;
#pragma unroll
            for (int i = 0; i < 5; ++i) {
              const float a = xv[bi][3 * i], b = xv[bi][3 * i + 1], c = xv[bi][3 * i + 2];
              xv[bi][3 * i] = left ? 0.f : right ? b : a;
              xv[bi][3 * i + 1] = left ? a : right ? c : b;
              xv[bi][3 * i + 2] = left ? b : right ? 0.f : c;
            }
          }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          u32x4_t hi4, lo4;
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            uint32_t hi, lo;
            x3_split_pair(xv[bi][8 * s + 2 * e2], xv[bi][8 * s + 2 * e2 + 1], hi, lo);
            hi4[e2] = hi;
            lo4[e2] = lo;
          }
          xh[bi][s] = __builtin_bit_cast(bf16x8_t, hi4);
          xl[bi][s] = __builtin_bit_cast(bf16x8_t, lo4);
        }
      }
    };
    // conv1_1 of channel half h on the converted window, bias + ReLU + split, halo tile -> LDS
    // f16mx: the packed half line of the lane's pixel, between produce() and flush()
    unsigned ph16[8] = {}, ph6[3] = {}, pl6[3] = {}, pbh = 0, pbl = 0;
    auto flush = [&](char* buf) __attribute__((always_inline)) {
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      typedef __attribute__((ext_vector_type(3))) unsigned u3;
      constexpr int bi = 0;
      if (!has_block || row_of(bi) >= ST_HALO_PX) return;
      // this lane's 16 consecutive elements of the pixel's line: fp16 parts = 16-B slots 2 half, 2 half + 1;
      // e2m3 images = dwords 3 half .. 3 half + 2 of the 6-dword strings, which the line keeps as
      // slot 4 / 5 (hi / lo: dwords 0-3) and slot 6 / 7 (dwords 4, 5, zero, scale byte)
      char* row = buf + row_of(bi) * 128;
      const int sw = g_swz[bi];
      const int half = fresh_lane_id() >> 5;     // (shadows the kernel's: see fresh_lane_id)
      *reinterpret_cast<u4*>(row + (((2 * half) ^ sw) << 4)) = (u4){ph16[0], ph16[1], ph16[2], ph16[3]};
      *reinterpret_cast<u4*>(row + (((2 * half + 1) ^ sw) << 4)) = (u4){ph16[4], ph16[5], ph16[6], ph16[7]};
      if (half == 0) {
        *reinterpret_cast<u3*>(row + ((4 ^ sw) << 4)) = (u3){ph6[0], ph6[1], ph6[2]};
        *reinterpret_cast<u3*>(row + ((5 ^ sw) << 4)) = (u3){pl6[0], pl6[1], pl6[2]};
      } else {
        *reinterpret_cast<unsigned*>(row + ((4 ^ sw) << 4) + 12) = ph6[0];
        *reinterpret_cast<unsigned*>(row + ((5 ^ sw) << 4) + 12) = pl6[0];
        *reinterpret_cast<u4*>(row + ((6 ^ sw) << 4)) = (u4){ph6[1], ph6[2], 0u, pbh};
        *reinterpret_cast<u4*>(row + ((7 ^ sw) << 4)) = (u4){pl6[1], pl6[2], 0u, pbl};
      }
    };
    auto produce = [&](int tile, auto h_c, char* buf) __attribute__((always_inline)) {
      constexpr int h = decltype(h_c)::value;
      int n, ty, tx;
      decode(tile, n, ty, tx);
      const int y0 = ty * 8 - 1, x0 = tx * 32 - 1;
      const bool interior = is_interior(ty, tx);
#pragma unroll
      for (int bi = 0; bi < 1; ++bi) {
        if (!has_block) continue;  // wave-uniform
        f32x16_t acc;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float4 b = *reinterpret_cast<const float4*>(bias_l + 32 * h + (MX ? 4 : 8) * j);
          acc[4 * j] = b.x;
          acc[4 * j + 1] = b.y;
          acc[4 * j + 2] = b.z;
          acc[4 * j + 3] = b.w;
        }
        __builtin_amdgcn_s_setprio(3);
#pragma unroll
        for (int s = 0; s < 2; ++s) {   // the order of conv1_1_mfma_kernel<X3>
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo[h][s], xh[bi][s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[h][s], xl[bi][s], acc, 0, 0, 0);
          acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[h][s], xh[bi][s], acc, 0, 0, 0);
        }
        if (pprio == 0) __builtin_amdgcn_s_setprio(0);
        else if (pprio == 1) __builtin_amdgcn_s_setprio(1);
        else if (pprio == 2) __builtin_amdgcn_s_setprio(2);
        // D[row = channel][col = pixel]: registers 4g..4g+3 = channels 8 g + 4 half + 0..3 of the
        // lane's pixel -> hi: 8 bytes of 16-B slot g, lo: of slot 4 + g (both swizzled), + 8 half.
        bool pix_ok = true;
        if (!interior) {   // border tiles only: is this halo pixel inside the image?
          int hy, hx;
          hyx_of(bi, hy, hx);
          const int y = y0 + hy, x = x0 + hx;
          pix_ok = y >= 0 && y < p.H && x >= 0 && x < p.W;
        }
        if constexpr (MX) {
          // ReLU, the fp16 bound and conv1_2's zero padding (a halo pixel outside the image) in ONE v_med3
          float c[16];
          const float lim = pix_ok ? 65504.f : 0.f;
#pragma unroll
          for (int j = 0; j < 16; ++j) c[j] = __builtin_amdgcn_fmed3f(acc[j], 0.f, lim);
          mx_pack_half<false>(c, ph16, ph6, pl6, pbh, pbl, range_seen);
          flush(buf);
        } else if (row_of(bi) < ST_HALO_PX) {
          char* row = buf + row_of(bi) * 128 + 8 * half;
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            uint2 hi, lo;
            ring_split4(fmaxf(acc[4 * g], 0.f), fmaxf(acc[4 * g + 1], 0.f), fmaxf(acc[4 * g + 2], 0.f),
                        fmaxf(acc[4 * g + 3], 0.f), hi, lo);
            if (!pix_ok) {   // outside the image: conv1_2's zero padding, not a conv1_1 output
              hi = make_uint2(0u, 0u);
              lo = make_uint2(0u, 0u);
            }
            *reinterpret_cast<uint2*>(row + ((g ^ g_swz[bi]) << 4)) = hi;
            *reinterpret_cast<uint2*>(row + (((4 + g) ^ g_swz[bi]) << 4)) = lo;
          }
        }
      }
    };

    // stage s = (tile s >> 1, channel half s & 1) goes to halo buffer s & 1; the producers run one
    // stage ahead of the consumers, the gathers one tile ahead of that
    using H0 = std::integral_constant<int, 0>;
    using H1 = std::integral_constant<int, 1>;
    const bool prof = p.prof != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && wave == 4;
    unsigned long long pt[4] = {0, 0, 0, 0};   // work, wait; of which in the stage beside the consumers' pass 0
    auto hand_over = [&](unsigned long long t0, int stage = 1) __attribute__((always_inline)) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const unsigned long long t1 = prof ? __builtin_amdgcn_s_memtime() : 0;
      __builtin_amdgcn_s_barrier();
      if (prof) {
        const unsigned long long t2 = __builtin_amdgcn_s_memtime();
        pt[0] += t1 - t0;
        pt[1] += t2 - t1;
        if (stage == 0) {
          pt[2] += t1 - t0;
          pt[3] += t2 - t1;
        }
      }
    };
    if (niter > 0) {
      issue_loads(first);
      convert();
      if (niter > 1) issue_loads(first + stride);
      produce(first, H0{}, hb);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if constexpr (DUAL) {
      for (int it = 0; it < niter; ++it) {
        // S1: the tile's second channel half into buffer 1 (last read in S4 of the previous tile)
        unsigned long long t0 = prof ? __builtin_amdgcn_s_memtime() : 0;
        produce(first + it * stride, H1{}, hb + ST_HALO_BYTES);
        hand_over(t0, 0);
        // S2: the tile's fragments are no longer needed — the window of tile it+1 (it has had a whole tile to
        // arrive) is converted and the gathers of tile it+2 go out
        t0 = prof ? __builtin_amdgcn_s_memtime() : 0;
        if (it + 1 < niter) {
          convert();
          if (it + 2 < niter) issue_loads(first + (it + 2) * stride);
        }
        hand_over(t0);
        // S3: nothing (buffer 0 is still read)
        t0 = prof ? __builtin_amdgcn_s_memtime() : 0;
        hand_over(t0);
        // S4: the first channel half of the next tile into buffer 0 (last read in S3)
        t0 = prof ? __builtin_amdgcn_s_memtime() : 0;
        if (it + 1 < niter) produce(first + (it + 1) * stride, H0{}, hb);
        hand_over(t0);
      }
    } else
    for (int it = 0; it < niter; ++it) {
      // while the consumers run pass 0 of tile it: its second channel half; then — its fragments
      // are no longer needed — the window of tile it+1 (it has had a whole tile to arrive) is
      // converted and the gathers of tile it+2 go out
      unsigned long long t0 = prof ? __builtin_amdgcn_s_memtime() : 0;
      produce(first + it * stride, H1{}, hb + ST_HALO_BYTES);
      if (it + 1 < niter) {
        convert();
        if (it + 2 < niter) issue_loads(first + (it + 2) * stride);
      }
      hand_over(t0, 0);
      // while they run pass 1: the first channel half of the next tile.  (Computing and packing it beside pass 0
      // and only writing it here was measured: 1.66 instead of 1.54 ms — that stage already carries the
      // consumers' epilogue and is bound by what the four waves of a SIMD can issue.)
      t0 = prof ? __builtin_amdgcn_s_memtime() : 0;
      if (it + 1 < niter) produce(first + (it + 1) * stride, H0{}, hb);
      hand_over(t0);
    }
    if (prof && lane == 0) {
      p.prof[4] = pt[0];
      p.prof[5] = pt[1];
      p.prof[6] = pt[2];
      p.prof[7] = pt[3];
    }
    raise_if_out_of_range();
    return;
  }

  // ================================== consumers ================================================
  if constexpr (MX) {
    typedef __attribute__((ext_vector_type(4))) unsigned u4;
    typedef __attribute__((ext_vector_type(2))) unsigned u2;
    typedef __attribute__((ext_vector_type(4))) int i4;
    {
      const int cprio = (p.prod_prio >> 2) & 3;   // test hook: the consumers' issue priority
      if (cprio == 1) __builtin_amdgcn_s_setprio(1);
      else if (cprio == 2) __builtin_amdgcn_s_setprio(2);
      else if (cprio == 3) __builtin_amdgcn_s_setprio(3);
    }
    // conv1_2's weights: LDS row (h * 9 + tap) * 32 + m = the f16mx line of output channel co0 + chan(m),
    // input group h; 16-byte slots swizzled by (m >> 1) & 7
    const int wpiece = ((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) * 16;
    const __amdgpu_buffer_rsrc_t rs_w2 =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.w2), 0, 9 * 64 * 256, 0x00020000);
    // DUAL: the 36 KB image (output-channel half c, input half h) into weight buffer `buf`: nine LDS-DMA
    // instructions per consumer wave (buffer loads: the compiler counts them in vmcnt only — a global_load_lds it has
    // not seen waited for turns every counted lgkmcnt of the pass into lgkmcnt(0))
    auto stream_w = [&](int c, int h, int buf) __attribute__((always_inline)) {
#pragma unroll
      for (int j = 0; j < 9; ++j) {
        const int q = j * 4 + wave;
        const int r = q * 8 + (lane >> 3);
        const int tap = r >> 5, m = r & 31;
        const int ch = 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
        buf_glds16(rs_w2, (unsigned)((tap * 64 + 32 * c + ch) * 256 + h * 128 + wpiece), 0u,
                   wl + buf * (9 * 32 * 128) + q * 1024);
      }
    };
    if constexpr (DUAL) {
      stream_w(0, 0, 0);
    } else {
#pragma unroll
      for (int j = 0; j < 18; ++j) {
        const int q = j * 4 + wave;
        const int r = q * 8 + (lane >> 3);
        const int h = r / 288, rem = r - 288 * h;
        const int tap = rem >> 5, m = rem & 31;
        const int ch = 16 * ((m >> 2) & 1) + 4 * (m >> 3) + (m & 3);
        glds16(p.w2 + ((long)(tap * 64 + co0 + ch) * 256 + h * 128) + wpiece, wl + q * 1024);
      }
    }
    // Pixel (B) fragments: the bf16x3 consumers' addressing (below): lane (pixel l31 of block i, k-half hc)
    // reads fp16 slot 2 s + hc for the two f16 MFMAs and — the B side of the scaled MFMA pairs q6(lo) with
    // the weights' q6(hi) in K-block 0 and the other way round in block 1 — slots 5 - hc / 7 - hc: all of
    // them (e_kx ^ const) + pc, the constants 0, 32, 80, 112 (^ 64 on odd tap rows).
    int e_kx[3], pxb;
    {
      const int ly = 2 * wave + ((l31 >> 1) & 1);
      const int lx = 2 * (l31 >> 2) + (l31 & 1);
      const int fix = half ^ ((ly & 1) << 2);
      e_kx[0] = (fix ^ ((lx >> 1) & 7)) << 4;
      e_kx[2] = (fix ^ (((lx >> 1) + 1) & 7)) << 4;
      e_kx[1] = (lx & 1) ? e_kx[2] : e_kx[0];
      pxb = (ly * C64_HW + lx) * 128;
    }
    // weight row of this lane; slot 2 kk + half of its line = w_base ^ (kk << 5) (the slot bits XOR; ONE register for
    // the four fragment addresses: the b128 tails cost three registers this 128-VGPR kernel did not have)
    int w_base = l31 * 128 + ((half ^ ((l31 >> 1) & 7)) << 4);
    auto w_off = [&](int kk) __attribute__((always_inline)) { return w_base ^ (kk << 5); };
    // (the builtin, not inline asm: the compiler models a pending global_load_lds as a FLAT access and turns
    //  every later lgkmcnt wait into lgkmcnt(0) until IT has seen vmcnt(0))
    __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0)
    __builtin_amdgcn_s_barrier();
    const bool cprof = p.prof != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0;
    unsigned long long ct[3] = {0, 0, 0}, ct0w = 0;
    f32x16_t acc[2];
    // operand registers, single-buffered: a fragment is reloaded for the next tap right behind the last
    // MFMA that reads it, and the MFMA order (w0.b0, w0.b1, w1.b0, w1.b1, wm.b0, wm.b1 — accumulators
    // alternate) leaves every reload at least four MFMAs (128 cycles) before its first use
    // (round 6) the MX operand's tail — e2m3 dwords 4, 5, a zero dword, the scale byte: one 16-byte slot of the line —
    // is read as ONE ds_read_b128, the scale taken from dword 7 of the operand.  As ds_read_b64 + ds_read_b32 (rounds
    // 3-5: the form the ring kernels keep, whose loop is not bound by LDS cycles) the 32 lanes of a b64 group reach
    // 16 of their 32 bank pairs and the 32 scale dwords sit on 8 banks: 12 LDS cycles per tail instead of 4, 1728
    // extra cycles per tile on the one role whose passes ARE the LDS port's time (4 waves x 48 cycles x 18 taps =
    // the 3456 matrix-pipe cycles of a SIMD: tools/lds_stem_model.py).
    f16x8_t w0, w1, b0[2], b1[2];
    u4 wma, bma[2];
    u4 wmt, bmt[2];
    // DUAL: (nc, nh) = the stage after this one (nc < 0: none): its weight image streams into the buffer this pass
    // does not read (the previous stage's, handed back behind that stage's barrier)
    auto run_pass = [&](auto h_c, int nc = -1, int nh = 0) __attribute__((always_inline)) {
      constexpr int h = decltype(h_c)::value;
      const unsigned long long c0 = cprof ? __builtin_amdgcn_s_memtime() : 0;
      if constexpr (DUAL) {
        if (nc >= 0) stream_w(nc, nh, h ^ 1);
      }
      const int pc = pxb + S3_W_BYTES + h * ST_HALO_BYTES;
      if (h == 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
      }
      auto paddr = [&](int tap, int part) __attribute__((always_inline)) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int c = (part == 0 ? 0 : part == 1 ? 32 : part == 2 ? 80 : 112) ^ ((ky & 1) << 6);
        int a;
        asm("v_xad_u32 %0, %1, %2, %3" : "=v"(a) : "v"(e_kx[kx]), "s"(c), "v"(pc));
        return a;
      };
      auto pbase = [&](int tap, int i) __attribute__((always_inline)) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        return smem + (ky * C64_HW + kx) * 128 + i * 2048;
      };
      auto ld_b0 = [&](int tap, int i, int a) __attribute__((always_inline)) { b0[i] = *reinterpret_cast<const f16x8_t*>(pbase(tap, i) + a); };
      auto ld_b1 = [&](int tap, int i, int a) __attribute__((always_inline)) { b1[i] = *reinterpret_cast<const f16x8_t*>(pbase(tap, i) + a); };
      auto ld_tail = [&](const char* a) __attribute__((always_inline)) -> u4 {
        if constexpr (S3_TAIL128) {
          return *reinterpret_cast<const u4*>(a);
        } else {
          const u2 d = *reinterpret_cast<const u2*>(a);
          const unsigned sc = *reinterpret_cast<const unsigned*>(a + 12);
          return (u4){d.x, d.y, 0u, sc};
        }
      };
      auto ld_bm = [&](int tap, int i, int a2, int a3) __attribute__((always_inline)) {
        bma[i] = *reinterpret_cast<const u4*>(pbase(tap, i) + a2);
        bmt[i] = ld_tail(pbase(tap, i) + a3);
      };
      auto ld_w0 = [&](int tap) __attribute__((always_inline)) { w0 = *reinterpret_cast<const f16x8_t*>(smem + tap * 4096 + w_off(0)); };
      auto ld_w1 = [&](int tap) __attribute__((always_inline)) { w1 = *reinterpret_cast<const f16x8_t*>(smem + tap * 4096 + w_off(1)); };
      auto ld_wm = [&](int tap) __attribute__((always_inline)) {
        wma = *reinterpret_cast<const u4*>(smem + tap * 4096 + w_off(2));
        wmt = ld_tail(smem + tap * 4096 + w_off(3));
      };
      auto mx = [&](int i) __attribute__((always_inline)) {
        // e2m3 x e2m3 (cbsz = blgp = 2): registers 0-5 of either operand; its scale: byte 0 of register 7 = the
        // tail slot's last dword [d4 d5 0 scale]
        const i32x8_t a8 = __builtin_shufflevector(__builtin_bit_cast(i4, wma), __builtin_bit_cast(i4, wmt),
                                                   0, 1, 2, 3, 4, 5, 6, 7);
        const i32x8_t b8 = __builtin_shufflevector(__builtin_bit_cast(i4, bma[i]), __builtin_bit_cast(i4, bmt[i]),
                                                   0, 1, 2, 3, 4, 5, 6, 7);
        acc[i] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[i], 2, 2, 0, a8[7], 0, b8[7]);
      };
      {
        const int a0 = paddr(0, 0), a1 = paddr(0, 1), a2 = paddr(0, 2), a3 = paddr(0, 3);
        ld_w0(0);
        ld_b0(0, 0, a0);
        ld_b0(0, 1, a0);
        ld_w1(0);
        ld_b1(0, 0, a1);
        ld_b1(0, 1, a1);
        ld_wm(0);
        ld_bm(0, 0, a2, a3);
        ld_bm(0, 1, a2, a3);
      }
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        // (sched_barrier: the order below IS the schedule — left alone, the scheduler sinks every reload to
        //  just in front of its use, for register pressure it does not have, and waits lgkmcnt(0) per MFMA)
        const bool more = tap + 1 < 9;
        const int n = tap + 1;
        int a0 = 0, a1 = 0, a2 = 0, a3 = 0;
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b0[0], acc[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (tap > 0) ld_wm(tap);   // (not behind the previous tap's last MFMA: 15 reads in flight there, and the
                                   //  compiler answers a full lgkmcnt counter with lgkmcnt(0))
        if (more) {
          a0 = paddr(n, 0);
          ld_b0(n, 0, a0);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w0, b0[1], acc[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
          ld_w0(n);
          ld_b0(n, 1, a0);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b1[0], acc[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
          a1 = paddr(n, 1);
          ld_b1(n, 0, a1);
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w1, b1[1], acc[1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
          ld_w1(n);
          ld_b1(n, 1, a1);
        }
        __builtin_amdgcn_sched_barrier(0);
        mx(0);
        __builtin_amdgcn_sched_barrier(0);
        if (more) {
          a2 = paddr(n, 2);
          a3 = paddr(n, 3);
          ld_bm(n, 0, a2, a3);
        }
        __builtin_amdgcn_sched_barrier(0);
        mx(1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) ld_bm(n, 1, a2, a3);
      }
      // every fragment read of this halo buffer has been consumed: hand it back (DUAL: and the next stage's weight
      // image has landed — it has had the whole pass)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if constexpr (DUAL) wait_vmcnt<0>();
      const unsigned long long c1 = cprof ? __builtin_amdgcn_s_memtime() : 0;
      __builtin_amdgcn_s_barrier();
      if (cprof) {
        const unsigned long long c2 = __builtin_amdgcn_s_memtime();
        ct[0] += c1 - c0;
        ct[1] += c2 - c1;
        if (h == 0) ct0w += c2 - c1;
      }
      w_base += h == 0 ? 9 * 32 * 128 : -(9 * 32 * 128);     // (a multiple of 4096: the slot bits stay)
    };
    using C0 = std::integral_constant<int, 0>;
    using C1 = std::integral_constant<int, 1>;
    constexpr int NCZ = DUAL ? 2 : 1;     // output-channel halves a workgroup runs per tile
#pragma unroll 1
    for (int itc = 0; itc < niter * NCZ; ++itc) {
      const int it = DUAL ? (itc >> 1) : itc, cz = DUAL ? (itc & 1) : 0;
      if constexpr (DUAL) {
        // S1 / S3 stream the image of S2 / S4 (same output half, input half 1); S2 streams (half 1, input 0); S4 the
        // next tile's (half 0, input 0)
        run_pass(C0{}, cz, 1);
        run_pass(C1{}, (cz == 0 || it + 1 < niter) ? (cz ^ 1) : -1, 0);
      } else {
        run_pass(C0{});
        run_pass(C1{});
      }
      const unsigned long long e0 = cprof ? __builtin_amdgcn_s_memtime() : 0;
      // 2x2 max-pool (the window = the lane quad: two DPP steps), bias, ReLU, pack, store.  Lane quad q of
      // block i is pooled pixel (wave, 8 i + q) of the tile's 4 x 16.
      const int lane_e = fresh_lane_id();
      const int half = lane_e >> 5, l31 = lane_e & 31;   // (shadow the kernel's: nothing lane-derived stays live across the passes)
      const float* const bias2 = reinterpret_cast<const float*>(smem + S3_BIAS_OFF) + 64 + 32 * cz + 16 * half;
      const int tile = first + it * stride;
      const unsigned r2 = (unsigned)tile / (unsigned)p.tiles_x;
      const int tx = tile - (int)r2 * p.tiles_x;
      const int n = (int)(r2 / (unsigned)p.tiles_y), ty = (int)r2 - n * p.tiles_y;
      const int oy = ty * 4 + wave;
      // ONE OUTPUT ROW of the map as the buffer: a pixel right of the map, a row below it (zero records) and
      // the three non-leader lanes of a quad (offset 2^31) are dropped by the hardware — no branch
      const __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(
          p.out + ((long)n * Ho + oy) * Wo * 256, 0, oy < Ho ? Wo * 256 : 0, 0x00020000);
      float bv[16];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 b = *reinterpret_cast<const float4*>(bias2 + 4 * j);
        bv[4 * j] = b.x;
        bv[4 * j + 1] = b.y;
        bv[4 * j + 2] = b.z;
        bv[4 * j + 3] = b.w;
      }
      {
        // pooling: the window's x pair first (lane ^ 1), then the two blocks merge — even lanes keep block 0,
        // odd lanes block 1 — and the y pair (lane ^ 2) follows on the merged values: ONE line per lane pair
        // (v_max_f32_dpp by hand: through fmaxf + mov_dpp every step is a v_mov_dpp, a v_max and two
        //  canonicalising v_max x, x — 176 instructions instead of 48.  A DPP read needs two wait states behind
        //  the VALU write of its source, which the compiler does not see inside asm: the first round reads
        //  accumulators written long ago, the second carries its own s_nop.)
        float v[16], m0[16], m1[16];
        const bool odd = l31 & 1;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          asm volatile("v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(m0[j]) : "v"(acc[0][j]));
          asm volatile("v_max_f32_dpp %0, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "=v"(m1[j]) : "v"(acc[1][j]));
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = odd ? m1[j] : m0[j];
#pragma unroll
        for (int j = 0; j < 16; ++j)   // (s_nop: the select above may be scheduled right in front of its reader)
          asm volatile("s_nop 1\n\tv_max_f32_dpp %0, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf" : "=v"(m0[j]) : "v"(v[j]));
#pragma unroll
        for (int j = 0; j < 16; ++j) v[j] = __builtin_amdgcn_fmed3f(m0[j] + bv[j], 0.f, 65504.f);   // bias, ReLU, the fp16 bound
        unsigned h16[8], h6[3], l6[3], bh, bl;
        mx_pack_half<false>(v, h16, h6, l6, bh, bl, range_seen);
        // lanes 0, 1 of a quad store: pooled pixel 8 (l31 & 1) + (l31 >> 2) of the tile row
        const unsigned off = (l31 & 2) ? 0x80000000u
                                       : (unsigned)(tx * 16 + 8 * (l31 & 1) + (l31 >> 2)) * 256u +
                                             (DUAL ? (unsigned)cz : blockIdx.y) * 128u;
        __builtin_amdgcn_raw_buffer_store_b128((u4){h16[0], h16[1], h16[2], h16[3]}, rs_o, (int)(off + 32 * half), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b128((u4){h16[4], h16[5], h16[6], h16[7]}, rs_o, (int)(off + 32 * half + 16), 0, 0);
        // e2m3 strings: this lane's dwords 3 half .. 3 half + 2; dwords 0-3 in slot 4 / 5, dwords 4, 5 in slot 6 / 7
        const unsigned oa = off + (half ? 76u : 64u), ob = off + (half ? 96u : 68u);
        __builtin_amdgcn_raw_buffer_store_b32(h6[0], rs_o, (int)oa, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64((u2){h6[1], h6[2]}, rs_o, (int)ob, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(l6[0], rs_o, (int)(oa + 16), 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64((u2){l6[1], l6[2]}, rs_o, (int)(ob + 16), 0, 0);
        const unsigned ot = half ? off + 104u : 0x80000000u;   // the tails' zero dword + scale byte
        __builtin_amdgcn_raw_buffer_store_b64((u2){0u, bh}, rs_o, (int)ot, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b64((u2){0u, bl}, rs_o, (int)(ot + 16), 0, 0);
      }
      if (cprof) ct[2] += __builtin_amdgcn_s_memtime() - e0;
    }
    if (cprof && lane == 0) {
      p.prof[0] = ct[0];
      p.prof[1] = ct[1];
      p.prof[2] = ct[2];
      p.prof[3] = ct0w;   // of the waits: behind pass 0
    }
    raise_if_out_of_range();
    return;
  }
  {
    const int cprio = (p.prod_prio >> 2) & 3;   // test hook: the consumers' issue priority
    if (cprio == 1) __builtin_amdgcn_s_setprio(1);
    else if (cprio == 2) __builtin_amdgcn_s_setprio(2);
    else if (cprio == 3) __builtin_amdgcn_s_setprio(3);
  }
  {
    const int piece = ((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) * 16;
#pragma unroll
    for (int j = 0; j < 18; ++j) {
      const int q = j * 4 + wave;
      const int r = q * 8 + (lane >> 3);          // LDS row: (h * 9 + tap) * 32 + c
      const int h = r / 288, rem = r - 288 * h;
      const int tap = rem >> 5, c = rem & 31;
      glds16(p.w2 + ((long)(tap * 64 + co0 + c) * 256 + h * 128) + piece, wl + q * 1024);
    }
  }
  // A-fragment addressing.  Block i's pixel of this lane is (ly, lx) of the 8 x 32 tile; tap (ky, kx)
  // reads halo pixel (ly + ky, lx + kx): its row is a compile-time distance (ky * 34 + kx) * 128 from
  // the lane's own, and its 16-B slot (2 pr + half) ^ swz(hy, hx) differs from the tap-(0,0) slot only
  // by   (lx + kx) >> 1 = (lx >> 1) + {0, lx & 1, 1}[kx]   and the constant bits pr << 1, (ky & 1) << 2.
  // Block 1's pixel is 16 columns right of block 0's: the same slot ((lx >> 1) & 7 repeats every 16
  // columns) 2048 bytes on.  So three per-lane slot offsets (kx = 0, 1, 2), one pixel offset and ONE
  // v_xad_u32 per pair of reads replace ~12 VALU instructions per read (hoisted, the 18 x 2 x 2
  // addresses would cost ~70 of this kernel's 128 VGPRs).
  int e_kx[3], pxb;
  {
    const int ly = 2 * wave + ((l31 >> 1) & 1);
    const int lx = 2 * (l31 >> 2) + (l31 & 1);
    const int fix = half ^ ((ly & 1) << 2);
    e_kx[0] = (fix ^ ((lx >> 1) & 7)) << 4;
    e_kx[2] = (fix ^ (((lx >> 1) + 1) & 7)) << 4;
    e_kx[1] = (lx & 1) ? e_kx[2] : e_kx[0];
    pxb = (ly * C64_HW + lx) * 128;
  }
  int w_off[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) w_off[kk] = l31 * 128 + (((2 * kk + half) ^ ((l31 >> 1) & 7)) << 4);
  const float bval = p.b2[co0 + l31];

  // The pooled outputs of tile i are held back and stored one pixel at a time between the matrix
  // steps of tile i+1.  A pending pixel is ONE dword per lane: lanes l31 and l31 ^ 1 exchange halves
  // (DPP + v_perm), the even lane stores the hi parts of channels (l31, l31 + 1), the odd lane the lo
  // parts of (l31 - 1, l31) — a half-wave writes the full 128-byte line of its pixel.  The store is a
  // buffer store into ONE OUTPUT ROW of the map (descriptor rebuilt per tile from scalars, zero
  // records for a row below the map): a pixel right of the map is out of range and dropped by the
  // hardware, the pixel's distance is the instruction's immediate — no mask, no compare, no branch,
  // so a pass stays one basic block and the scheduler can place the fragment reads freely.
  uint32_t pend[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) pend[e] = 0;
  __amdgpu_buffer_rsrc_t rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, 0, 0x00020000);  // nothing pending
  const unsigned lane_off =
      blockIdx.y * 128 + half * 256 + ((l31 & 1) ? 64 + (l31 - 1) * 2 : l31 * 2);
  unsigned poff = lane_off;
  auto store_px = [&](int e) __attribute__((always_inline)) {
    __builtin_amdgcn_raw_buffer_store_b32(pend[e], rs_o, (int)(poff + 512u * e), 0, 0);
  };
  __builtin_amdgcn_s_waitcnt(0x0F70);   // vmcnt(0), as the builtin: see the f16mx consumers
  __builtin_amdgcn_s_barrier();
  const bool cprof = p.prof != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && wave == 0;
  unsigned long long ct[3] = {0, 0, 0};
  f32x16_t acc[2];
  // one pass = 18 steps (tap, 16-wide half pr of the K-tile) over channel half h of the halo
  auto run_pass = [&](auto h_c) __attribute__((always_inline)) {
    constexpr int h = decltype(h_c)::value;
    const unsigned long long c0 = cprof ? __builtin_amdgcn_s_memtime() : 0;
    // LDS addresses beyond the 16-bit immediate of a DS instruction live in the base registers:
    // pc = this lane's pixel in halo buffer h, w_off[] = its weight row in pass h's weight image
    const int pc = pxb + S3_W_BYTES + h * ST_HALO_BYTES;
    if (h == 0) {
      float b = bval;
      asm volatile("" : "+v"(b));   // per pass: a hoisted 16-register splat would be spilled
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = b;
    }
    bf16x8_t fah[2][2], fal[2][2], fbh[2], fbl[2];
    auto load_step = [&](int sidx, bf16x8_t (&ah)[2], bf16x8_t (&al)[2], bf16x8_t& bh, bf16x8_t& bl)
        __attribute__((always_inline)) {
      const int tap = sidx >> 1, pr = sidx & 1;
      const int ky = tap / 3, kx = tap - 3 * ky;
      const int c = (pr << 5) ^ ((ky & 1) << 6);   // hi chunk; the lo chunk sits 4 slots on
      int a_hi, a_lo;
      asm("v_xad_u32 %0, %1, %2, %3" : "=v"(a_hi) : "v"(e_kx[kx]), "s"(c), "v"(pc));
      asm("v_xad_u32 %0, %1, %2, %3" : "=v"(a_lo) : "v"(e_kx[kx]), "s"(c ^ 64), "v"(pc));
      const char* px = smem + (ky * C64_HW + kx) * 128;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        ah[i] = *reinterpret_cast<const bf16x8_t*>(px + i * 2048 + a_hi);
        al[i] = *reinterpret_cast<const bf16x8_t*>(px + i * 2048 + a_lo);
      }
      bh = *reinterpret_cast<const bf16x8_t*>(smem + tap * 4096 + w_off[pr]);
      bl = *reinterpret_cast<const bf16x8_t*>(smem + tap * 4096 + w_off[pr + 2]);
    };
    load_step(0, fah[0], fal[0], fbh[0], fbl[0]);
#pragma unroll
    for (int sidx = 0; sidx < 18; ++sidx) {
      const int b = sidx & 1;
      if (sidx + 1 < 18) load_step(sidx + 1, fah[b ^ 1], fal[b ^ 1], fbh[b ^ 1], fbl[b ^ 1]);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fal[b][i], fbh[b], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[b][i], fbl[b], acc[i], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < 2; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fah[b][i], fbh[b], acc[i], 0, 0, 0);
      if (h == 0 && (sidx & 1) == 1 && (sidx >> 1) < 8) store_px(sidx >> 1);
      if (sidx + 1 < 18) {
        // the six fragment reads of step sidx+1 (and their two address instructions) go out one per
        // MFMA of this step: a full step of latency cover, no read burst in front of the matrix pipe
#pragma unroll
        for (int q = 0; q < 6; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          if (q < 2) __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          if (q == 3 && h == 0 && (sidx & 1) == 1 && (sidx >> 1) < 8)
            __builtin_amdgcn_sched_group_barrier(0x040, 1, 0);   // this step's pending-pixel store
        }
      }
    }
    // every fragment read of this halo buffer has been consumed by the MFMAs above: hand it back
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long c1 = cprof ? __builtin_amdgcn_s_memtime() : 0;
    __builtin_amdgcn_s_barrier();
    if (cprof) {
      ct[0] += c1 - c0;
      ct[1] += __builtin_amdgcn_s_memtime() - c1;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) w_off[kk] += h == 0 ? 9 * 32 * 128 : -(9 * 32 * 128);   // the other pass's image
  };
  using C0 = std::integral_constant<int, 0>;
  using C1 = std::integral_constant<int, 1>;
  const uint32_t psel = (l31 & 1) ? 0x03020706u : 0x05040100u;
  for (int it = 0; it < niter; ++it) {
    run_pass(C0{});
    run_pass(C1{});
    // pool, split, pair up: the tile's eight pixels of this lane become pending
    const int tile = first + it * stride;
    const unsigned r2 = (unsigned)tile / (unsigned)p.tiles_x;
    const int tx = tile - (int)r2 * p.tiles_x;
    const int n = (int)(r2 / (unsigned)p.tiles_y), ty = (int)r2 - n * p.tiles_y;
    const int oy = ty * 4 + wave;
    rs_o = __builtin_amdgcn_make_buffer_rsrc(p.out + ((long)n * Ho + oy) * Wo * 256, 0,
                                             oy < Ho ? Wo * 256 : 0, 0x00020000);
    poff = lane_off + (unsigned)tx * (16 * 256);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float v = fmaxf(fmaxf(fmaxf(acc[i][4 * g], acc[i][4 * g + 1]),
                                    fmaxf(acc[i][4 * g + 2], acc[i][4 * g + 3])), 0.f);
        uint16_t hi, lo;
        x3_split(v, hi, lo);
        const uint32_t mine = (uint32_t)hi | ((uint32_t)lo << 16);
        const uint32_t other = (uint32_t)__builtin_amdgcn_mov_dpp((int)mine, 0xB1, 0xF, 0xF, true);  // lane ^ 1
        pend[4 * i + g] = __builtin_amdgcn_perm(other, mine, psel);
      }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) store_px(e);
  if (cprof && lane == 0) {
    p.prof[0] = ct[0];
    p.prof[1] = ct[1];
    p.prof[2] = 0;
  }
}

// u8_mean3 / u8_std3 (host pointers, both or neither): x is the raw uint8 NHWC image and the kernel normalises
// (vgg_stem_x3_kernel, U8)
static int launch_vgg_stem_x3(const void* x, int N, int H, int W, const float* w1, const float* b1,
                              const void* packed_w2, const float* b2, void* out, hipStream_t st,
                              bool mx = false, unsigned* range_flag = nullptr, const float* u8_mean3 = nullptr,
                              const float* u8_std3 = nullptr, float act_scale = 1.f) {
  StemParams p = {};
  p.range_flag = range_flag;
  p.act_scale = act_scale;
  const bool u8 = u8_mean3 != nullptr && u8_std3 != nullptr;
  if (u8) {
    for (int c = 0; c < 3; ++c) {   // v = u * a + b (see the kernel): a in `mean`, b in `stdv`
      p.mean[c] = (float)(1.0 / (255.0 * (double)u8_std3[c]));
      p.stdv[c] = (float)(-(double)u8_mean3[c] / (double)u8_std3[c]);
    }
  }
  p.x = x;
  p.w1 = w1;
  p.b1 = b1;
  p.w2 = (const char*)packed_w2;
  p.b2 = b2;
  p.out = (char*)out;
  // (uint8: the descriptor covers the tensor rounded up to whole dwords — the 12-byte window loads are dword
  //  aligned, and the range check drops a dword that is only partly inside; the up to 3 extra bytes share the
  //  last valid byte's dword, hence its page, and are never used: the taps they belong to are masked)
  p.x_bytes = u8 ? (unsigned)align_up((size_t)N * 3 * H * W, 4) : (unsigned)((size_t)N * 3 * H * W * 4);
  p.N = N;
  p.H = H;
  p.W = W;
  p.tiles_x = (W + 31) / 32;
  p.tiles_y = (H + 7) / 8;
  const long nt = (long)N * p.tiles_x * p.tiles_y;
  OIBL_REQUIRE(nt < 0x7fffffffL, "vgg stem: too many tiles");
  p.ntiles = (int)nt;
  p.prof = g_prof_buf;
  // f16mx: producers at priority 2, consumers at 3 unless the hook says otherwise (tests/gpu_stem_mx_bench.py:
  // 1.72 ms against 1.80 ms with equal priorities; the bf16x3 roles are at their best with equal ones)
  p.prod_prio = (mx && g_stem3_prio == 0) ? 14 : g_stem3_prio;
  int gx = 128;  // two workgroups (output-channel halves) per tile range: one persistent workgroup per CU
  if (gx > p.ntiles) gx = p.ntiles;
  // f16mx, S3_DUAL: one workgroup per tile range and CU, both output-channel halves inside it
  int gmx = 256, gmy = 1;
  if (!S3_DUAL) {
    gmx = gx;
    gmy = 2;
  }
  if (gmx > p.ntiles) gmx = p.ntiles;
  if (mx && u8) {
    auto kern = vgg_stem_x3_kernel<true, true>;
    OIBL_SET_MAX_LDS(kern, S3_LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(gmx, gmy), dim3(1024), S3_LDS_BYTES, st, p);
  } else if (mx) {
    auto kern = vgg_stem_x3_kernel<true>;
    OIBL_SET_MAX_LDS(kern, S3_LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(gmx, gmy), dim3(1024), S3_LDS_BYTES, st, p);
  } else if (u8) {
    auto kern = vgg_stem_x3_kernel<false, true>;
    OIBL_SET_MAX_LDS(kern, S3_LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(gx, 2), dim3(1024), S3_LDS_BYTES, st, p);
  } else {
    auto kern = vgg_stem_x3_kernel<false>;
    OIBL_SET_MAX_LDS(kern, S3_LDS_BYTES);
    hipLaunchKernelGGL(kern, dim3(gx, 2), dim3(1024), S3_LDS_BYTES, st, p);
  }
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

__global__ void clear_word_kernel(unsigned* w) { *w = 0u; }

// uint8 NHWC -> normalised fp32 NCHW with the loader's arithmetic ((u / 255 - mean) / std, fp32,
// correctly rounded divisions): the route of the uint8 entry point whenever the fused stem is not
// used (fp32 precision, test hooks)
__global__ void u8_nhwc_to_nchw_f32_kernel(const uint8_t* __restrict__ x, float* __restrict__ out,
                                           long npix_total, long plane, float m0, float m1, float m2,
                                           float s0, float s1, float s2) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < npix_total;
       i += (long)gridDim.x * blockDim.x) {
    const long n = i / plane, pix = i - n * plane;
    const uint8_t* px = x + i * 3;
    float* o = out + n * 3 * plane + pix;
    o[0] = ((float)px[0] / 255.0f - m0) / s0;
    o[plane] = ((float)px[1] / 255.0f - m1) / s1;
    o[2 * plane] = ((float)px[2] / 255.0f - m2) / s2;
  }
}

static bool stem_eligible(int N, int H, int W) {
  return H >= 2 && W >= 2 && (size_t)N * 3 * H * W * 4 < (size_t)0xE0000000u;
}
// every activation an f16mx convolution reads stays inside a 32-bit buffer offset (ring_variant): the
// largest is conv2_2's input [N][H/2][W/2][128] x 4 bytes
static bool vgg16_f16mx_fits(int N, int H, int W) {
  return stem_eligible(N, H, W) && (size_t)N * (H / 2) * (W / 2) * 128 * 4 < (size_t)0xE0000000u;
}

// ---------------------------------------------------------------------------------------------
// small layout / pooling helpers
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void global_maxpool_kernel(const T* __restrict__ feat, float* __restrict__ out, int P,
                                      int C) {
  __shared__ float red[4][64];
  const int n = blockIdx.x, cg = blockIdx.y;
  const int ch = cg * 64 + (threadIdx.x & 63), pg = threadIdx.x >> 6;
  float m = -INFINITY;
  if (ch < C)
    for (int p = pg; p < P; p += 4) m = fmaxf(m, Elem<T>::load(feat + ((size_t)n * P + p) * C + ch));
  red[pg][threadIdx.x & 63] = m;
  __syncthreads();
  if (pg == 0 && ch < C)
    out[(size_t)n * C + ch] = fmaxf(fmaxf(red[0][threadIdx.x], red[1][threadIdx.x]),
                                    fmaxf(red[2][threadIdx.x], red[3][threadIdx.x]));
}

// [N][P][C] T -> [N][C][P] fp32 through a 32x33 LDS tile
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int P,
                                    int C) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty 0..7
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, ch = c0 + tx;
    tile[r][tx] = (p < P && ch < C) ? Elem<T>::load(in + ((size_t)n * P + p) * C + ch) : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int ch = c0 + r, p = p0 + tx;
    if (p < P && ch < C) out[((size_t)n * C + ch) * P + p] = tile[tx][r];
  }
}

// [N][C][P] fp32 -> [N][P][C] T
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int C,
                                    int P) {
  __shared__ float tile[32][33];
  const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  for (int r = ty; r < 32; r += 8) {
    const int ch = c0 + r, p = p0 + tx;
    tile[r][tx] = (p < P && ch < C) ? in[((size_t)n * C + ch) * P + p] : 0.f;
  }
  __syncthreads();
  for (int r = ty; r < 32; r += 8) {
    const int p = p0 + r, ch = c0 + tx;
    if (p < P && ch < C) Elem<T>::store(out + ((size_t)n * P + p) * C + ch, tile[tx][r]);
  }
}

struct VggLayer {
  int cin, cout, relu, pool;
};
static const VggLayer kVgg[OIBL_VGG16_NUM_CONV] = {
    {3, 64, 1, 0},    {64, 64, 1, 1},   {64, 128, 1, 0},  {128, 128, 1, 1}, {128, 256, 1, 0},
    {256, 256, 1, 0}, {256, 256, 1, 1}, {256, 512, 1, 0}, {512, 512, 1, 0}, {512, 512, 1, 1},
    {512, 512, 1, 0}, {512, 512, 1, 0}, {512, 512, 0, 0}};

static bool precision_ok(int precision) {
  return precision == OIBL_BF16 || precision == OIBL_F32 || precision == OIBL_BF16X3 || precision == OIBL_F16MX;
}

// out_f32 (bf16x3 only): write the output as plain fp32 NHWC
OIBL_HOOK(int, g_conv_splitk, 1);  // test hook: 0 = never split K

// splitk_ws (optional): scratch for the split-K partials of layers with too few tiles
// (conv_splitk_bytes); without it every layer runs one-pass
static size_t conv_splitk_bytes(long m_total, int cin, int cout, int precision) {
  if (precision == OIBL_F16MX) return 0;  // (its own plan: mx_split_bytes)
  const int steps = 9 * (cin / (precision == OIBL_BF16 ? 64 : 32));
  const int s = conv_splitk_factor(m_total, cout, steps);
  return s ? align_up((size_t)s * m_total * cout * sizeof(float), 256) : 0;
}

// scratch of one layer's split-K partials (0: the layer runs in one pass), any precision
static size_t conv_layer_scratch_bytes(int N, int h, int w, int cin, int cout, int pool, int precision) {
  if (precision == OIBL_F16MX) {   // (either K order: the order is a per-layer default a test hook can change)
    const size_t a = mx_split_bytes((long)N * h * w, cin, cout, pool, 0), b = mx_split_bytes((long)N * h * w, cin, cout, pool, 1);
    return a > b ? a : b;
  }
  const long m_total = pool ? (long)N * (h / 2) * (w / 2) * 4 : (long)N * h * w;
  return conv_splitk_bytes(m_total, cin, cout, precision);
}

static int conv3x3_impl(const void* in, int N, int H, int W, int cin, const void* packed_w,
                        const float* bias, int cout, int relu, int pool, int precision, void* out,
                        hipStream_t st, int out_f32 = 0, void* splitk_ws = nullptr,
                        unsigned* range_flag = nullptr, float bias_mul = 1.f, float out_mul = 1.f) {
  OIBL_REQUIRE(in && packed_w && bias && out, "conv3x3: null pointer");
  OIBL_REQUIRE(precision_ok(precision), "conv3x3: bad precision %d", precision);
  const int bk = precision == OIBL_BF16 ? 64 : 32;
  OIBL_REQUIRE(N > 0 && H > 0 && W > 0, "conv3x3: bad shape N=%d H=%d W=%d", N, H, W);
  OIBL_REQUIRE(cin % bk == 0 && cout % 64 == 0, "conv3x3: unsupported channels cin=%d cout=%d", cin,
               cout);
  OIBL_REQUIRE(!pool || (H >= 2 && W >= 2), "conv3x3: pooling needs H,W >= 2");
  OIBL_REQUIRE((long)N * H * W < 0x7fffffffL, "conv3x3: N*H*W must be < 2^31 (split the batch)");
  OIBL_REQUIRE((uintptr_t)in % 16 == 0 && (uintptr_t)packed_w % 16 == 0 && (uintptr_t)out % 16 == 0,
               "conv3x3: pointers must be 16-byte aligned");
  // Cin = 64: the resident-weights / LDS-halo kernel serves Cout = 64 (conv1_2); from Cout = 128 on
  // the 512 x 128 ring kernel is faster (conv2_1: 795 vs 680 TFLOP/s) unless the hook forces c64.
  if (precision == OIBL_BF16 && cin == 64 && g_conv_c64 && !g_regstage && !g_conv_ablate &&
      (g_conv_c64 == 2 || cout % 128 != 0))
    return launch_conv_c64(in, N, H, W, packed_w, bias, cout, relu, pool, out, st);
  ConvParams p;
  p.in = in;
  p.w = packed_w;
  p.bias = bias;
  p.out = out;
  p.zero = zero_line_device_ptr();
  OIBL_REQUIRE(p.zero != nullptr, "conv3x3: zero line symbol not found");
  p.N = N;
  p.H = H;
  p.W = W;
  p.cin = cin;
  p.cout = cout;
  p.relu = relu;
  p.ablate = g_conv_ablate;
  p.out_f32 = (precision == OIBL_BF16X3 || precision == OIBL_F16MX) ? out_f32 : 0;
  p.korder = conv_korder_for(precision, cin, cout);
  p.tiles_n = 0;
  if (pool) {
    p.out_rows = (long)N * (H / 2) * (W / 2);
    p.m_total = p.out_rows * 4;
  } else {
    p.out_rows = (long)N * H * W;
    p.m_total = p.out_rows;
  }
  p.ksplit = 0;
  p.partial = (float*)splitk_ws;
  p.range_flag = range_flag;
  p.bias_mul = bias_mul;
  p.out_mul = out_mul;
  if (splitk_ws && g_conv_splitk && precision != OIBL_F16MX)
    p.ksplit = conv_splitk_factor(p.m_total, cout, 9 * (cin / bk));
  if (precision == OIBL_F16MX) return launch_conv_mx(p, pool, st);
  if (precision == OIBL_BF16X3) return launch_conv<bf16x3_t>(p, pool, st);
  return precision == OIBL_BF16 ? launch_conv<bf16_t>(p, pool, st) : launch_conv<float>(p, pool, st);
}

}  // namespace oibl

using namespace oibl;

extern "C" {

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_prof_buffer(void* dev_u64x8) {
  g_prof_buf = (unsigned long long*)dev_u64x8;
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_stem_fused(int on) {
  g_stem_fused = on ? 1 : 0;
  return OIBL_OK;
}
#endif

int oibl_vgg16_stem_bf16(const float* x_nchw, int N, int H, int W, const float* w1_oihw,
                         const float* b1, const void* packed_w2, const float* b2, void* out,
                         void* stream) {
  OIBL_REQUIRE(x_nchw && w1_oihw && b1 && packed_w2 && b2 && out, "vgg16_stem: null pointer");
  OIBL_REQUIRE(N > 0 && H >= 2 && W >= 2, "vgg16_stem: bad shape N=%d H=%d W=%d", N, H, W);
  OIBL_REQUIRE(stem_eligible(N, H, W), "vgg16_stem: input of %d x 3 x %d x %d exceeds 3.5 GB", N, H, W);
  OIBL_REQUIRE((uintptr_t)packed_w2 % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)x_nchw % 4 == 0,
               "vgg16_stem: packed weights / output must be 16-byte aligned");
  return launch_vgg_stem<false>(x_nchw, N, H, W, nullptr, nullptr, w1_oihw, b1, packed_w2, b2, out,
                                (hipStream_t)stream);
}

int oibl_vgg16_stem_x3(const float* x_nchw, int N, int H, int W, const float* w1_oihw, const float* b1,
                       const void* packed_w2, const float* b2, void* out, void* stream) {
  OIBL_REQUIRE(x_nchw && w1_oihw && b1 && packed_w2 && b2 && out, "vgg16_stem_x3: null pointer");
  OIBL_REQUIRE(N > 0 && H >= 2 && W >= 2, "vgg16_stem_x3: bad shape N=%d H=%d W=%d", N, H, W);
  OIBL_REQUIRE(stem_eligible(N, H, W), "vgg16_stem_x3: input of %d x 3 x %d x %d exceeds 3.5 GB", N, H, W);
  OIBL_REQUIRE((uintptr_t)packed_w2 % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)x_nchw % 4 == 0,
               "vgg16_stem_x3: packed weights / output must be 16-byte aligned");
  return launch_vgg_stem_x3(x_nchw, N, H, W, w1_oihw, b1, packed_w2, b2, out, (hipStream_t)stream);
}

int oibl_vgg16_stem_mx(const float* x_nchw, int N, int H, int W, const float* w1_oihw, const float* b1,
                       const void* packed_w2, const float* b2, void* out, void* stream) {
  OIBL_REQUIRE(x_nchw && w1_oihw && b1 && packed_w2 && b2 && out, "vgg16_stem_mx: null pointer");
  OIBL_REQUIRE(N > 0 && H >= 2 && W >= 3, "vgg16_stem_mx: bad shape N=%d H=%d W=%d (needs H >= 2, W >= 3)", N, H, W);
  OIBL_REQUIRE(stem_eligible(N, H, W), "vgg16_stem_mx: input of %d x 3 x %d x %d exceeds 3.5 GB", N, H, W);
  OIBL_REQUIRE((uintptr_t)packed_w2 % 16 == 0 && (uintptr_t)out % 16 == 0 && (uintptr_t)x_nchw % 4 == 0,
               "vgg16_stem_mx: packed weights / output must be 16-byte aligned");
  return launch_vgg_stem_x3(x_nchw, N, H, W, w1_oihw, b1, packed_w2, b2, out, (hipStream_t)stream, true);
}

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_conv_c64(int on) {  // 0 = off, 1 = auto, 2 = every Cin = 64 layer
  g_conv_c64 = on < 0 ? 0 : (on > 2 ? 2 : on);
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_conv_ablate(int mode) {
  g_conv_ablate = mode;
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_ring_ablate(int mode) {
  g_ring_ablate = mode;
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_stem3_prio(int prio) {
  g_stem3_prio = prio < 0 ? 0 : (prio & 15);
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_conv_splitk(int on) {
  g_conv_splitk = on ? 1 : 0;
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_mx_variant(int v) {
  g_halo_var = v >= 16 ? v - 16 : 0;   // 19: halo kernel with the unsafe waits (experiment)
  if (v >= 16) v = 3;
  g_mx_variant = v;
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_conv_korder(int mode) {
  g_conv_korder = mode < 0 ? -1 : (mode ? 1 : 0);
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_stem_u8(int on) {
  g_stem_u8 = on ? 1 : 0;
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_mx_act_shift(int shift) {
  g_mx_act_shift = shift < 0 ? 0 : (shift > 12 ? 12 : shift);
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_mx_splitk(int on) {
  g_mx_splitk = on == 2 ? 2 : (on ? 1 : 0);
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_ring_stagger(int sleeps) {
  g_ring_stagger = sleeps < 0 ? 0 : sleeps;
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_ring_bar1(int on) {
  g_ring_bar1 = on ? 1 : 0;
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_ring_raster(int mode) {
  g_ring_raster = mode;
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_conv_tile(int mode) {
  g_conv_tile = mode;
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_conv11_valu(int on) {
  g_conv11_valu = on ? 1 : 0;
  return OIBL_OK;
}
#endif

size_t oibl_conv3x3_packed_bytes(int cout, int cin, int precision) {
  return (size_t)9 * cout * cin * oibl_elem_size(precision);
}

int oibl_pack_conv3x3_weights(const float* w_oihw, int cout, int cin, int precision, void* packed,
                              void* stream) {
  OIBL_REQUIRE(w_oihw && packed, "pack_conv3x3_weights: null pointer");
  OIBL_REQUIRE(cout > 0 && cin > 0, "pack_conv3x3_weights: bad shape");
  OIBL_REQUIRE(precision_ok(precision), "pack_conv3x3_weights: bad precision %d", precision);
  OIBL_REQUIRE((precision != OIBL_BF16X3 && precision != OIBL_F16MX) || cin % 32 == 0,
               "pack_conv3x3_weights: bf16x3 / f16mx need Cin %% 32 == 0");
  const size_t total = (size_t)9 * cout * cin;
  unsigned blocks = (unsigned)((total + 255) / 256);
  if (blocks > 8192) blocks = 8192;
  if (precision == OIBL_F16MX) {
    unsigned b = (unsigned)((total / 32 + 127) / 128);
    hipLaunchKernelGGL(pack_conv3x3_mx_kernel, dim3(b > 8192 ? 8192 : b), dim3(128), 0, (hipStream_t)stream, w_oihw,
                       (char*)packed, cout, cin);
  } else if (precision == OIBL_BF16)
    hipLaunchKernelGGL(pack_conv3x3_kernel<bf16_t>, dim3(blocks), dim3(256), 0,
                       (hipStream_t)stream, w_oihw, (bf16_t*)packed, cout, cin);
  else if (precision == OIBL_BF16X3)
    hipLaunchKernelGGL(pack_conv3x3_kernel<bf16x3_t>, dim3(blocks), dim3(256), 0,
                       (hipStream_t)stream, w_oihw, (bf16x3_t*)packed, cout, cin);
  else
    hipLaunchKernelGGL(pack_conv3x3_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                       w_oihw, (float*)packed, cout, cin);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_conv3x3_nhwc(const void* in, int N, int H, int W, int cin, const void* packed_w,
                      const float* bias, int cout, int relu, int pool, int precision, void* out,
                      void* stream) {
  return conv3x3_impl(in, N, H, W, cin, packed_w, bias, cout, relu, pool, precision, out,
                      (hipStream_t)stream);
}

int oibl_conv3x3_nhwc_flagged(const void* in, int N, int H, int W, int cin, const void* packed_w,
                              const float* bias, int cout, int relu, int pool, int precision, void* out,
                              uint32_t* range_flag, void* stream) {
  OIBL_REQUIRE(range_flag == nullptr || (uintptr_t)range_flag % 4 == 0, "conv3x3: range flag must be 4-byte aligned");
  return conv3x3_impl(in, N, H, W, cin, packed_w, bias, cout, relu, pool, precision, out,
                      (hipStream_t)stream, 0, nullptr, range_flag);
}

size_t oibl_conv3x3_workspace_bytes(int N, int H, int W, int cin, int cout, int pool, int precision) {
  if (N <= 0 || H <= 0 || W <= 0 || cin <= 0 || cout <= 0 || !precision_ok(precision)) return 0;
  return conv_layer_scratch_bytes(N, H, W, cin, cout, pool, precision);
}

int oibl_conv3x3_nhwc_ws(const void* in, int N, int H, int W, int cin, const void* packed_w, const float* bias,
                         int cout, int relu, int pool, int precision, void* out, void* ws, size_t ws_bytes,
                         uint32_t* range_flag, void* stream) {
  OIBL_REQUIRE(range_flag == nullptr || (uintptr_t)range_flag % 4 == 0, "conv3x3: range flag must be 4-byte aligned");
  // ws == NULL: the layer runs in one pass whatever its size (exactly oibl_conv3x3_nhwc_flagged)
  const size_t need = ws ? oibl_conv3x3_workspace_bytes(N, H, W, cin, cout, pool, precision) : 0;
  if (need && ws_bytes < need) {
    set_error("conv3x3: workspace %zu < required %zu bytes", ws_bytes, need);
    return OIBL_E_WORKSPACE;
  }
  OIBL_REQUIRE(!need || (uintptr_t)ws % 256 == 0, "conv3x3: workspace must be 256-byte aligned");
  return conv3x3_impl(in, N, H, W, cin, packed_w, bias, cout, relu, pool, precision, out,
                      (hipStream_t)stream, 0, need ? ws : nullptr, range_flag);
}

int oibl_conv1_1_nchw(const float* x_nchw, int N, int H, int W, const float* w_oihw,
                      const float* bias, int precision, void* out, void* stream) {
  OIBL_REQUIRE(x_nchw && w_oihw && bias && out, "conv1_1: null pointer");
  OIBL_REQUIRE(N > 0 && H > 0 && W > 0, "conv1_1: bad shape N=%d H=%d W=%d", N, H, W);
  OIBL_REQUIRE(precision_ok(precision), "conv1_1: bad precision %d", precision);
  const long nstrips = (long)N * H * ((W + 7) / 8);
  const long grid = (nstrips + 3) / 4;
  OIBL_REQUIRE(grid <= 0x7fffffffL, "conv1_1: grid too large");
  if (precision == OIBL_BF16X3 || (precision == OIBL_BF16 && !g_conv11_valu)) {
    const int tiles_per_row = (W + C11_TW - 1) / C11_TW;
    const long ntiles = (long)N * H * tiles_per_row;
    const unsigned blocks = (unsigned)(ntiles < 4096 ? ntiles : 4096);
    if (precision == OIBL_BF16X3)
      hipLaunchKernelGGL(conv1_1_mfma_kernel<true>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                         x_nchw, w_oihw, bias, (char*)out, N, H, W, tiles_per_row, ntiles);
    else
      hipLaunchKernelGGL(conv1_1_mfma_kernel<false>, dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                         x_nchw, w_oihw, bias, (char*)out, N, H, W, tiles_per_row, ntiles);
  } else if (precision == OIBL_BF16)
    hipLaunchKernelGGL(conv1_1_kernel<bf16_t>, dim3((unsigned)grid), dim3(256), 0,
                       (hipStream_t)stream, x_nchw, w_oihw, bias, (bf16_t*)out, N, H, W);
  else
    hipLaunchKernelGGL(conv1_1_kernel<float>, dim3((unsigned)grid), dim3(256), 0,
                       (hipStream_t)stream, x_nchw, w_oihw, bias, (float*)out, N, H, W);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_global_maxpool_nhwc(const void* feat, int N, int P, int C, int precision, float* out,
                             void* stream) {
  OIBL_REQUIRE(feat && out, "global_maxpool: null pointer");
  OIBL_REQUIRE(N > 0 && P > 0 && C > 0, "global_maxpool: bad shape");
  dim3 grid(N, (C + 63) / 64);
  if (precision == OIBL_BF16)
    hipLaunchKernelGGL(global_maxpool_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)feat, out, P, C);
  else if (precision == OIBL_F32)
    hipLaunchKernelGGL(global_maxpool_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)feat, out, P, C);
  else
    OIBL_REQUIRE(false, "global_maxpool: bad precision %d", precision);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_nhwc_to_nchw_f32(const void* feat, int N, int P, int C, int precision, float* out,
                          void* stream) {
  OIBL_REQUIRE(feat && out, "nhwc_to_nchw: null pointer");
  OIBL_REQUIRE(N > 0 && P > 0 && C > 0 && N < 65536, "nhwc_to_nchw: bad shape");
  dim3 grid((P + 31) / 32, (C + 31) / 32, N);
  if (precision == OIBL_BF16)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const bf16_t*)feat, out, P, C);
  else if (precision == OIBL_F32)
    hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream,
                       (const float*)feat, out, P, C);
  else
    OIBL_REQUIRE(false, "nhwc_to_nchw: bad precision %d", precision);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_nchw_f32_to_nhwc(const float* x, int N, int C, int P, int precision, void* out,
                          void* stream) {
  OIBL_REQUIRE(x && out, "nchw_to_nhwc: null pointer");
  OIBL_REQUIRE(N > 0 && P > 0 && C > 0 && N < 65536, "nchw_to_nhwc: bad shape");
  dim3 grid((P + 31) / 32, (C + 31) / 32, N);
  if (precision == OIBL_BF16)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, grid, dim3(256), 0, (hipStream_t)stream, x,
                       (bf16_t*)out, C, P);
  else if (precision == OIBL_F32)
    hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, x,
                       (float*)out, C, P);
  else
    OIBL_REQUIRE(false, "nchw_to_nhwc: bad precision %d", precision);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

// ping-pong activation buffers: A holds outputs of even layers, B of odd layers
static void vgg_buffer_elems(int N, int H, int W, size_t* a, size_t* b) {
  size_t ea = 0, eb = 0;
  int h = H, w = W;
  for (int l = 0; l < OIBL_VGG16_NUM_CONV - 1; ++l) {  // the last layer writes `feat`
    if (kVgg[l].pool) {
      h /= 2;
      w /= 2;
    }
    const size_t e = (size_t)N * h * w * kVgg[l].cout;
    if (l % 2 == 0)
      ea = e > ea ? e : ea;
    else
      eb = e > eb ? e : eb;
  }
  *a = ea;
  *b = eb;
}

// scratch for the split-K partials of the layers whose tiling leaves the chip idle (small batches)
static size_t vgg_splitk_bytes(int N, int H, int W, int precision) {
  size_t mx = 0;
  int h = H, w = W;
  for (int l = 1; l < OIBL_VGG16_NUM_CONV; ++l) {
    const size_t b = conv_layer_scratch_bytes(N, h, w, kVgg[l].cin, kVgg[l].cout, kVgg[l].pool, precision);
    mx = b > mx ? b : mx;
    if (kVgg[l].pool) {
      h /= 2;
      w /= 2;
    }
  }
  return mx;
}

// the workspace starts with the f16mx range flag (one 32-bit word, see the header) in a 256-byte slot
constexpr size_t VGG_WS_HEAD = 256;

size_t oibl_vgg16_workspace_bytes(int N, int H, int W, int precision) {
  if (N <= 0 || H < 16 || W < 16) return 0;
  size_t ea, eb;
  vgg_buffer_elems(N, H, W, &ea, &eb);
  const size_t es = oibl_elem_size(precision);
  return VGG_WS_HEAD + align_up(ea * es, 256) + align_up(eb * es, 256) + vgg_splitk_bytes(N, H, W, precision);
}

int oibl_vgg16_conv5_forward(const float* x_nchw, int N, int H, int W,
                             const void* const* packed_w_host, const float* const* bias_host,
                             int precision, void* feat, void* ws, size_t ws_bytes, void* stream) {
  return oibl_vgg16_conv5_forward_ev(x_nchw, N, H, W, packed_w_host, bias_host, precision, feat, ws,
                                     ws_bytes, stream, nullptr, nullptr);
}

// x: fp32 NCHW (u8 = 0) or uint8 NHWC + Normalize constants (u8 = 1, host pointers mean3 / std3)
static int vgg_forward_impl(const void* x, int u8, const float* mean3, const float* std3, int N, int H,
                            int W, const void* const* packed_w_host, const float* const* bias_host,
                            int precision, void* feat, void* ws, size_t ws_bytes, void* stream,
                            void* ev_igemm_begin, void* ev_igemm_end) {
  OIBL_REQUIRE(x && packed_w_host && bias_host && feat && ws, "vgg16: null pointer");
  OIBL_REQUIRE(!u8 || (mean3 && std3), "vgg16: uint8 input needs the mean / std constants");
  OIBL_REQUIRE(precision_ok(precision), "vgg16: bad precision %d", precision);
  OIBL_REQUIRE(N > 0 && H >= 16 && W >= 16, "vgg16: bad shape N=%d H=%d W=%d", N, H, W);
  OIBL_REQUIRE((uintptr_t)ws % 256 == 0, "vgg16: workspace must be 256-byte aligned");
  const size_t base_need = oibl_vgg16_workspace_bytes(N, H, W, precision);
  const size_t need = u8 ? oibl_vgg16_u8_workspace_bytes(N, H, W, precision) : base_need;
  if (ws_bytes < need) {
    set_error("vgg16: workspace %zu < required %zu bytes", ws_bytes, need);
    return OIBL_E_WORKSPACE;
  }
  size_t ea, eb;
  vgg_buffer_elems(N, H, W, &ea, &eb);
  const size_t es = oibl_elem_size(precision);
  char* bufA = (char*)ws + VGG_WS_HEAD;
  char* bufB = bufA + align_up(ea * es, 256);
  char* splitk = vgg_splitk_bytes(N, H, W, precision) ? bufB + align_up(eb * es, 256) : nullptr;
  hipStream_t st = (hipStream_t)stream;
  // f16mx: the pass starts with a clear range flag; every packer of the pass may raise it (common.h)
  // (cleared by a one-thread KERNEL: a hipMemsetAsync of 4 bytes goes to a DMA engine and, replayed inside a
  //  graph, waited there behind the next batch's 118 MB input copy — extract_features from fp32 host batches
  //  lost 8 % to it)
  unsigned* const range_flag = precision == OIBL_F16MX ? (unsigned*)ws : nullptr;
  if (range_flag) {
    hipLaunchKernelGGL(clear_word_kernel, dim3(1), dim3(1), 0, st, range_flag);
    OIBL_LAUNCH_CHECK();
  }

  int rc;
  int h = H, w = W, l0 = 1;
  const void* cur = bufA;
  const bool fused = precision == OIBL_BF16 && g_stem_fused && stem_eligible(N, H, W) &&
                     !g_regstage && !g_conv_ablate;
  const float* x_f32 = (const float*)x;
  const bool mx = precision == OIBL_F16MX;
  const bool fused3 = (precision == OIBL_BF16X3 || mx) && g_stem_fused && stem_eligible(N, H, W) && !g_regstage &&
                      !g_conv_ablate && g_conv_tile == 0;
  // uint8 input: every fused stem normalises inside the kernel (g_stem_u8 = 0, test hook: the 4-byte stems
  // take the normalising pass below instead)
  // (the 4-byte stems fetch the image with dword-aligned 12-byte buffer loads over a descriptor rounded up to whole
  //  dwords: an input that does not start on a 4-byte boundary — an odd-offset slice of a caller's buffer — takes
  //  the normalising pass instead, which reads bytes)
  const bool u8_fused3 = u8 && fused3 && g_stem_u8 && (uintptr_t)x % 4 == 0;
  if (u8 && !fused && !u8_fused3) {  // normalise into the fp32 staging area behind the activation buffers
    float* stage = (float*)((char*)ws + base_need);
    const long npix = (long)N * H * W;
    hipLaunchKernelGGL(u8_nhwc_to_nchw_f32_kernel, dim3(2048), dim3(256), 0, st, (const uint8_t*)x,
                       stage, npix, (long)H * W, mean3[0], mean3[1], mean3[2], std3[0], std3[1],
                       std3[2]);
    OIBL_LAUNCH_CHECK();
    x_f32 = stage;
  }
  // f16mx: every activation the backbone STORES is multiplied by 2^-g_mx_act_shift (1/8).  ReLU, max-pool and the
  // convolutions are positively homogeneous and a power of two commutes with every rounding of the format (fp16
  // hi, e2m3 lo under a power-of-two block scale, fp32 sums): the stored values are the unscaled ones' exact
  // images, and the fp16 bound moves from 65 504 to 5.2e5 in activation units — 11.7x above the peak of the
  // calibrated-activation test (4.5e4, tests/test_gpu_range.py) instead of 1.46x.  The price is at the other
  // end of fp16: a scaled value below 6.1e-5 (|x| < 4.9e-4 before scaling) is a subnormal hi.  At the reference's
  // input scale (0-255-range pixels, activations in the tens to thousands) that is nothing: the fp32 map is
  // BIT-IDENTICAL to the unscaled one (tests/gpu_scale_probe2.py, test_activation_scale_is_an_exact_image); on
  // unit-range images through the synthetic state (conv5_3 mean 0.01 .. 0.05) a shift of 2 moves the map by 5.9e-6,
  // 3 by 7.9e-6, 4 by 1.45e-5, 5 by 2.1e-5 — the size of the format's own error — which is why the shift is 3.  Costs nothing:
  // the stem multiplies conv1_1's weights and the two biases as it stages them (its halo tile in LDS and
  // conv1_2's sums are then scaled too), the layers start their accumulators at bias * scale, and the last
  // layer multiplies its accumulators by 1 / scale once behind the loop (the head gets the fp32 map in
  // activation units).
  const float act_scale = mx ? ldexpf(1.f, -g_mx_act_shift) : 1.f;
  // f16mx has no unfused front (Cout = 64 fits no f16mx tile): conv1_2's weights are packed for the stem
  OIBL_REQUIRE(!mx || fused3, "vgg16: the f16mx backbone needs the fused stem (input below 3.5 GB, no stem / tile hooks)");
  // ... and only 32-bit-offset kernels behind it: the largest activation they read is conv2_2's input
  OIBL_REQUIRE(!mx || vgg16_f16mx_fits(N, H, W),
               "vgg16 (f16mx): a batch of %d x %d x %d exceeds the 3.5 GB per-activation limit of the f16mx kernels "
               "(conv2_2 reads N (H/2) (W/2) 128 4-byte elements): split the batch or use OIBL_BF16X3", N, H, W);
  if (fused3) {
    // bf16x3 / f16mx: conv1_1 + conv1_2 + pool in one launch (the uint8 entry has normalised into x_f32)
    if (ev_igemm_begin) OIBL_HIP_CHECK(hipEventRecord((hipEvent_t)ev_igemm_begin, st));
    rc = u8_fused3 ? launch_vgg_stem_x3(x, N, H, W, (const float*)packed_w_host[0], bias_host[0], packed_w_host[1],
                                        bias_host[1], bufB, st, mx, range_flag, mean3, std3, act_scale)
                   : launch_vgg_stem_x3(x_f32, N, H, W, (const float*)packed_w_host[0], bias_host[0], packed_w_host[1],
                                        bias_host[1], bufB, st, mx, range_flag, nullptr, nullptr, act_scale);
    if (rc) return rc;
    h /= 2;
    w /= 2;
    cur = bufB;
    l0 = 2;
  } else if (fused) {
    // conv1_1 + conv1_2 + pool in one launch (the matrix-core span then starts with it)
    if (ev_igemm_begin) OIBL_HIP_CHECK(hipEventRecord((hipEvent_t)ev_igemm_begin, st));
    rc = u8 ? launch_vgg_stem<true>(x, N, H, W, mean3, std3, (const float*)packed_w_host[0],
                                    bias_host[0], packed_w_host[1], bias_host[1], bufB, st)
            : launch_vgg_stem<false>(x, N, H, W, nullptr, nullptr, (const float*)packed_w_host[0],
                                     bias_host[0], packed_w_host[1], bias_host[1], bufB, st);
    if (rc) return rc;
    h /= 2;
    w /= 2;
    cur = bufB;
    l0 = 2;
  } else {
    rc = oibl_conv1_1_nchw(x_f32, N, H, W, (const float*)packed_w_host[0], bias_host[0], precision, bufA, stream);
    if (rc) return rc;
    if (ev_igemm_begin) OIBL_HIP_CHECK(hipEventRecord((hipEvent_t)ev_igemm_begin, st));
  }
  for (int l = l0; l < OIBL_VGG16_NUM_CONV; ++l) {
    void* dst = (l == OIBL_VGG16_NUM_CONV - 1) ? feat : (l % 2 == 0 ? (void*)bufA : (void*)bufB);
    // bf16x3 / f16mx: the last layer hands the head a plain fp32 map
    const bool last = l == OIBL_VGG16_NUM_CONV - 1;
    rc = conv3x3_impl(cur, N, h, w, kVgg[l].cin, packed_w_host[l], bias_host[l], kVgg[l].cout,
                      kVgg[l].relu, kVgg[l].pool, precision, dst, st, last, splitk, range_flag, act_scale,
                      last ? 1.f / act_scale : 1.f);
    if (rc) return rc;
    if (kVgg[l].pool) {
      h /= 2;
      w /= 2;
    }
    cur = dst;
  }
  if (ev_igemm_end) OIBL_HIP_CHECK(hipEventRecord((hipEvent_t)ev_igemm_end, st));
  return OIBL_OK;
}

int oibl_vgg16_conv5_forward_ev(const float* x_nchw, int N, int H, int W,
                                const void* const* packed_w_host, const float* const* bias_host,
                                int precision, void* feat, void* ws, size_t ws_bytes, void* stream,
                                void* ev_igemm_begin, void* ev_igemm_end) {
  return vgg_forward_impl(x_nchw, 0, nullptr, nullptr, N, H, W, packed_w_host, bias_host, precision,
                          feat, ws, ws_bytes, stream, ev_igemm_begin, ev_igemm_end);
}

size_t oibl_vgg16_u8_workspace_bytes(int N, int H, int W, int precision) {
  const size_t b = oibl_vgg16_workspace_bytes(N, H, W, precision);
  return b ? b + align_up((size_t)N * 3 * H * W * sizeof(float), 256) : 0;
}

int oibl_vgg16_conv5_forward_u8(const uint8_t* x_nhwc, int N, int H, int W, const float* mean3_host,
                                const float* std3_host, const void* const* packed_w_host,
                                const float* const* bias_host, int precision, void* feat, void* ws,
                                size_t ws_bytes, void* stream, void* ev_igemm_begin,
                                void* ev_igemm_end) {
  return vgg_forward_impl(x_nhwc, 1, mean3_host, std3_host, N, H, W, packed_w_host, bias_host,
                          precision, feat, ws, ws_bytes, stream, ev_igemm_begin, ev_igemm_end);
}

int oibl_x3_split_rows(const float* src, void* dst, size_t rows, int C, void* stream) {
  OIBL_REQUIRE(src && dst, "x3_split_rows: null pointer");
  OIBL_REQUIRE(C > 0 && C % 32 == 0, "x3_split_rows: C=%d must be a positive multiple of 32", C);
  if (rows == 0) return OIBL_OK;
  const size_t n = rows * (size_t)C;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(x3_split_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src,
                     (char*)dst, n, C);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_mx_split_rows_flagged(const float* src, void* dst, size_t rows, int C, uint32_t* range_flag,
                               void* stream) {
  OIBL_REQUIRE(src && dst, "mx_split_rows: null pointer");
  OIBL_REQUIRE(C > 0 && C % 32 == 0, "mx_split_rows: C %% 32 != 0");
  OIBL_REQUIRE(range_flag == nullptr || (uintptr_t)range_flag % 4 == 0, "mx_split_rows: range flag must be 4-byte aligned");
  if (rows == 0) return OIBL_OK;
  const size_t lines = rows * (size_t)(C / 32);
  unsigned b = (unsigned)((lines + 255) / 256);
  hipLaunchKernelGGL(mx_pack_rows_kernel<0>, dim3(b > 16384 ? 16384 : b), dim3(256), 0, (hipStream_t)stream,
                     (const char*)src, (char*)dst, lines, (unsigned*)range_flag);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_mx_split_rows(const float* src, void* dst, size_t rows, int C, void* stream) {
  return oibl_mx_split_rows_flagged(src, dst, rows, C, nullptr, stream);
}

int oibl_mx_join_rows(const void* src, float* dst, size_t rows, int C, int which, void* stream) {
  OIBL_REQUIRE(src && dst, "mx_join_rows: null pointer");
  OIBL_REQUIRE(C > 0 && C % 32 == 0 && which >= 0 && which <= 3, "mx_join_rows: bad arguments");
  if (rows == 0) return OIBL_OK;
  const size_t n = rows * (size_t)C;
  unsigned b = (unsigned)((n + 255) / 256);
  hipLaunchKernelGGL(mx_join_rows_kernel, dim3(b > 16384 ? 16384 : b), dim3(256), 0, (hipStream_t)stream,
                     (const char*)src, dst, n, which);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_x3_join_rows(const void* src, float* dst, size_t rows, int C, void* stream) {
  OIBL_REQUIRE(src && dst, "x3_join_rows: null pointer");
  OIBL_REQUIRE(C > 0 && C % 32 == 0, "x3_join_rows: C=%d must be a positive multiple of 32", C);
  if (rows == 0) return OIBL_OK;
  const size_t n = rows * (size_t)C;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(x3_join_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream,
                     (const char*)src, dst, n, C);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

}  // extern "C"
