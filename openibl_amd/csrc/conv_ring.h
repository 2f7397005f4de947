// Implicit-GEMM 3x3 convolution, bf16, on the ring schedule (ring_core.h): 256 x 256 tile
// (Cout % 256 == 0) or 512 x 128 tile (Cout % 128 == 0).
//
// Same contraction and the same summation order as conv3x3_igemm_kernel (conv.hip): M = output
// pixels, N = Cout, K = 9 * Cin ordered (tap, cin), one K-tile = 64 bf16 = one 128-byte line per
// row, so the results are bit-identical to the generic kernel's.
//
// Addressing: the per-lane part of an operand address is a 32-bit buffer offset that is constant
// for the whole kernel (weights) or for one tap (pixels); the per-K-tile part is a scalar offset.
// A tap that leaves the image gets an offset beyond num_records: the buffer unit returns zeros —
// zero padding without a padded copy, a zero line or any per-tile select.
//
// X3 = bf16x3 activations and weights (common.h): a pixel's channels are groups of [32 hi | 32 lo],
// a K-tile is one such 128-byte group (32 real channels), so the loaders only see 4-byte elements;
// the main loop runs 12 MFMAs per phase and the epilogue splits every output into its (hi, lo) pair
// again — or, for the layer that feeds the fp32 head, stores plain fp32 (RingParams::out_f32).
//
// P = RING_MX: f16mx activations and weights (common.h) — the same 128-byte groups of 32 channels, so
// the loaders are those of bf16x3; 6 MFMAs per phase; the epilogue stages the tile as fp32 and turns
// every (pixel, 32-channel group) into its f16mx line (mx_pack_line) before the copy-out.
#pragma once

#include "ring_core.h"

namespace oibl {

struct RingParams {
  const void* in;
  const void* w;
  const float* bias;
  void* out;
  unsigned in_bytes, w_bytes;
  int N, H, W, cin, cout;
  int m_total;   // GEMM rows (pixels enumerated, 4 per pooled output when POOL)
  int out_rows;  // rows of the output tensor
  int tiles_n;
  int relu;
  unsigned long long* prof;  // optional (test hook): shader-clock stamps of block 0, wave 0
  // floor(m / d) for m < 2^31 as (m * mul) >> sh (ring_magic_u31): d = pixels (pooled: quads) per
  // image, and per image row.  An emulated 32-bit divide costs ~25 instructions, and every lane
  // decomposes 8-16 GEMM rows before the first load can be issued.
  unsigned hw_mul, hw_sh, w_mul, w_sh;
  int ablate;  // timing experiments only (WRONG results): 1 = pixel loads of taps != 0 all hit one
               // line, 2 = weight loads after the first K-tile all hit one line, 3 = both
  int out_f32;  // X3, !POOL only: write plain fp32 NHWC instead of the (hi, lo) groups
  int tiles_m;
  int raster;   // xcd_tile() mode
  int korder;   // 0 = (tap, channel chunk), 1 = (channel chunk, tap): see ConvRingALoader::begin_tile
  unsigned* range_flag;  // f16mx output: raised when an output is beyond fp16 (common.h, mx_raise_range_flag); may be null
  // Activation scale of the f16mx backbone (conv.hip, g_mx_act_shift): the accumulators start at bias * bias_mul, and
  // out_mul != 1 (the layer that hands the fp32 map to the head) multiplies them once behind the loop.  Both 1
  // for a stand-alone layer.
  float bias_mul = 1.f, out_mul = 1.f;
  // Row / K sub-ranges (all zero: the whole problem in one pass).  A launch covers the GEMM rows from m_base
  // on (tiles_m tiles of them); with nsteps_part != 0 it is a SPLIT-K launch: workgroup (tile, blockIdx.y)
  // contracts nsteps_part K-tiles starting at OUTER K index blockIdx.y * k_outer_step (outer = tap in K order
  // 0, channel chunk in order 1: a split starts where the inner index is 0), from ZERO accumulators, and its
  // epilogue (host: out_f32 = 1, relu = 0) writes them as fp32 rows to out + blockIdx.y * part_stride bytes,
  // row (m - m_base); conv_mx_splitk_reduce_kernel adds bias and partials in a fixed order.
  int m_base, nsteps_part, k_outer_step;
  size_t part_stride;
  // Experiment (test hook, 0 = off): the workgroups of the FIRST round start (blockIdx & 3) * stagger sleeps of
  // 8128 cycles late.  All workgroups of a layer run for the same time, so the rounds stay in phase across the
  // chip: every CU loads, then every CU stores its tile (conv2_1: 64 MB of lines per round inside ~9 us); four
  // phase groups spread those bursts over the round.
  int stagger;
};

// mul, sh with floor(m / d) == (m * mul) >> sh for every m < 2^31 (d >= 1):  sh = 31 + ceil(log2 d),
// mul = ceil(2^sh / d) < 2^32; the rounding error (mul d - 2^sh) m / (d 2^sh) < m / 2^31 / d < 1 / d.
static inline void ring_magic_u31(unsigned d, unsigned* mul, unsigned* sh) {
  unsigned s = 0;
  while (((unsigned long long)1 << s) < d) ++s;
  *sh = 31 + s;
  *mul = (unsigned)((((unsigned long long)1 << (31 + s)) + d - 1) / d);
}
__device__ static inline unsigned ring_div_u31(unsigned m, unsigned mul, unsigned sh) {
  return (unsigned)(((unsigned long long)m * mul) >> sh);
}

// epilogue staging: the output tile (X3 without pooling: one half of its rows at a time)
template <int WM, bool POOL, int P = RING_BF16, bool OUTMX = (P >= RING_MX)>
constexpr int ring_lds_bytes() {
  using G = RingGeo<WM>;
  constexpr bool E4 = P != RING_BF16;
  constexpr int rows = POOL ? G::BM / 4 : (E4 ? G::BM / 2 : G::BM);
  constexpr int epi = rows * (G::BN * (E4 ? 4 : 2) + (OUTMX ? 0 : 16));
  return epi > G::MAIN_LDS ? epi : G::MAIN_LDS;
}

// A operand: im2col rows of the NHWC input.  Per lane and LDS-DMA instruction: the byte offset of
// the pixel (centre tap) and a 9-bit tap-validity mask; per tap: the offsets actually used
// (RG_OOB when the tap leaves the image); per K-tile: a scalar channel-chunk offset.
template <int NA, bool POOL, bool X3 = false>
struct ConvRingALoader {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned base[2 * NA], mask[2 * NA], cur[2 * NA];
  unsigned soff, abl_piece;
  int in, out, per, W, pix_bytes, ablate, korder;   // K cursor: inner / outer counter, inner period
  __device__ inline void init(const RingParams& p, int m0, const int (&tile_row)[2 * NA], int piece,
                              int outer0 = 0) {
    korder = p.korder;
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
    pix_bytes = p.cin * (X3 ? 4 : 2);
    per = korder ? 9 : (p.cin >> (X3 ? 5 : 6));
    W = p.W;
    in = -1;
    out = outer0 - 1;   // the first begin_tile() lands on (outer0, inner 0)
    soff = 0;
    ablate = p.ablate & 1;
    abl_piece = (unsigned)piece;
    const int Hq = POOL ? (p.H >> 1) : p.H, Wq = POOL ? (p.W >> 1) : p.W;
    const unsigned hw = (unsigned)Hq * (unsigned)Wq;
#pragma unroll
    for (int j = 0; j < 2 * NA; ++j) {
      // 32-bit index math (the host checks m_total < 2^31)
      const unsigned m = (unsigned)m0 + (unsigned)tile_row[j];
      unsigned mk = 0, off = 0;
      if (m < (unsigned)p.m_total) {
        const unsigned q = POOL ? (m >> 2) : m;
        const unsigned sub = POOL ? (m & 3u) : 0u;
        const unsigned n = ring_div_u31(q, p.hw_mul, p.hw_sh);
        const unsigned rem = q - n * hw;
        const unsigned yq = ring_div_u31(rem, p.w_mul, p.w_sh);
        int y = (int)yq, x = (int)(rem - yq * (unsigned)Wq);
        if (POOL) {
          y = 2 * y + (int)(sub >> 1);
          x = 2 * x + (int)(sub & 1);
        }
        const bool y0 = y > 0, y2 = y + 1 < p.H, x0 = x > 0, x2 = x + 1 < p.W;
        mk = (y0 && x0 ? 1u : 0u) | (y0 ? 2u : 0u) | (y0 && x2 ? 4u : 0u) | (x0 ? 8u : 0u) | 16u |
             (x2 ? 32u : 0u) | (y2 && x0 ? 64u : 0u) | (y2 ? 128u : 0u) | (y2 && x2 ? 256u : 0u);
        off = ((n * (unsigned)p.H + (unsigned)y) * (unsigned)p.W + (unsigned)x) * (unsigned)pix_bytes;
      }
      mask[j] = mk;
      base[j] = off + piece;
      cur[j] = RG_OOB;
    }
  }
  // K order.  korder 0 (default): tap outer, channel chunk inner — per K-tile only the scalar chunk
  // offset changes, the per-lane offsets change once per tap.  korder 1: chunk outer, tap inner — the
  // nine taps of one 128-byte chunk are consecutive K-tiles, so horizontally adjacent taps re-read
  // their lines from the XCD's L2.  Measured (profiles/r02_*): korder 1 cuts the fetched bytes 3-8x
  // (every tap of korder 0 misses L2 once 32 workgroups x 9 taps of footprint exceed 4 MiB) and is
  // nevertheless 3-12 % SLOWER in bf16 and 2-5 % slower in bf16x3 on all layers but one: the kernels are
  // not bound by fetch volume (Infinity-Cache hits are cheap), while korder 0 streams each pixel's
  // channel run as consecutive lines.
  // The cursor is ONE inner counter with a period and an outer counter, advanced without branches
  // (a handful of scalar instructions: this runs between the MFMAs of a COMPUTE segment, ring_core.h);
  // only the per-lane tap offsets — 5 vector instructions per row, once per tap — sit behind a branch.
  __device__ inline void begin_tile() {
    const bool wrap = in == per - 1;
    in = wrap ? 0 : in + 1;
    out += (in == 0) ? 1 : 0;
    const int tap = korder ? in : out, cc = korder ? out : in;
    if (korder != 0 || in == 0) {
      const int ky = (tap * 11) >> 5, kx = tap - 3 * ky;
      const int toff = ((ky - 1) * W + (kx - 1)) * pix_bytes;
#pragma unroll
      for (int j = 0; j < 2 * NA; ++j)
        cur[j] = ((mask[j] >> tap) & 1u) ? base[j] + (unsigned)toff : RG_OOB;
      if (ablate && tap != 0) {
#pragma unroll
        for (int j = 0; j < 2 * NA; ++j) cur[j] = abl_piece;
      }
    }
    soff = (unsigned)cc * 128u;
  }
  __device__ inline void stage(int h, char* dst, int i0 = 0, int i1 = NA) const {
#pragma unroll
    for (int i = 0; i < NA; ++i)
      if (i >= i0 && i < i1) buf_glds16(rsrc, cur[NA * h + i], soff, dst + i * 8192);
  }
};

// B operand: packed weights [tap][Cout][Cin]; row = output channel.
template <int NB, bool X3 = false>
struct ConvRingBLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned off[2 * NB];
  unsigned soff, step_in, step_wrap;   // K cursor as a running byte offset: + step_in, or + step_wrap when
  int in, per;                         // the inner counter (period `per`) wraps — no branches
  __device__ inline void init(const RingParams& p, int n0, const int (&tile_row)[2 * NB], int piece,
                              int outer0 = 0) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
    const unsigned pix_bytes = (unsigned)p.cin * (X3 ? 4u : 2u);
    const unsigned cchunks = (unsigned)(p.cin >> (X3 ? 5 : 6));
    const unsigned tap_stride = (unsigned)p.cout * pix_bytes;
    // offset of K-tile (tap, cc) = tap * tap_stride + cc * 128, same K order as the A loader
    if (p.korder == 0) {   // (tap, chunk): chunk inner
      per = (int)cchunks;
      step_in = 128u;
      step_wrap = tap_stride - (cchunks - 1u) * 128u;
    } else {               // (chunk, tap): tap inner
      per = 9;
      step_in = tap_stride;
      step_wrap = 128u - 8u * tap_stride;
    }
    in = -1;
    soff = (unsigned)outer0 * (p.korder == 0 ? tap_stride : 128u) - step_in;   // first begin_tile(): K-tile (outer0, 0)
#pragma unroll
    for (int j = 0; j < 2 * NB; ++j) off[j] = (unsigned)(n0 + tile_row[j]) * pix_bytes + piece;
    if (p.ablate & 2) {    // timing experiment: every K-tile reads ONE line (wrong results)
      step_in = step_wrap = 0u;
      soff = 0u;
#pragma unroll
      for (int j = 0; j < 2 * NB; ++j) off[j] = (unsigned)piece;
    }
  }
  __device__ inline void begin_tile() {
    const bool wrap = in == per - 1;
    in = wrap ? 0 : in + 1;
    soff += wrap ? step_wrap : step_in;
  }
  __device__ inline void stage(int h, char* dst, int i0 = 0, int i1 = NB) const {
#pragma unroll
    for (int i = 0; i < NB; ++i)
      if (i >= i0 && i < i1) buf_glds16(rsrc, off[NB * h + i], soff, dst + i * 8192);
  }
};

// (hi, lo) pairs of four fp32 values as two dwords each: hi = bf16(v), lo = bf16(v - hi)
__device__ static inline void ring_split4(float a0, float a1, float a2, float a3, uint2& hi, uint2& lo) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
  hi.x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){a0, a1}, bf2));
  hi.y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){a2, a3}, bf2));
  const float r0 = a0 - __builtin_bit_cast(float, hi.x << 16);
  const float r1 = a1 - __builtin_bit_cast(float, hi.x & 0xffff0000u);
  const float r2 = a2 - __builtin_bit_cast(float, hi.y << 16);
  const float r3 = a3 - __builtin_bit_cast(float, hi.y & 0xffff0000u);
  lo.x = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){r0, r1}, bf2));
  lo.y = __builtin_bit_cast(uint32_t, __builtin_convertvector((f2){r2, r3}, bf2));
}

// OUTMX: the output is written as f16mx lines (always with f16mx operands; the parameter exists because the
// epilogue only depends on it: bf16x3 operands with f16mx output compile too — round 3 ran conv2_1 that way
// until the stem itself became f16mx).
// The kernel's body.  GROUP: -1, or (BAR1) the stagger group of the calling wave as a compile-time constant —
// the kernel then holds one copy of the body per group (ring_core.h, GROUP: two copies of the LOOP that merged
// again in front of a shared epilogue made the register allocator spill; a lambda around the body put the
// kernel arguments on the stack).
template <int WM, bool POOL, bool ODD, int P, bool OUTMX, bool BAR1, int GROUP>
__device__ __forceinline__ void conv3x3_ring_body(const RingParams& p, char* smem, const int lane, const int wave) {
  using G = RingGeo<WM>;
  constexpr int NA = G::NA, NB = G::NB;
  constexpr bool X3 = P != RING_BF16;   // 4-byte elements, 32 channels per K-tile (bf16x3 and f16mx)
  constexpr bool MX = P >= RING_MX;     // f16mx operands
  static_assert(!OUTMX || X3, "f16mx output needs 4-byte elements");
  const int wm = wave / G::WN, wn = wave % G::WN;
  const bool prof = p.prof != nullptr && blockIdx.x == 0 && wave == 0;
  const unsigned long long t_start = prof ? __builtin_amdgcn_s_memtime() : 0;
  // (test hook, tests/gpu_wg_turnover.py: every workgroup's start / end stamp and the CU it ran on — what a CU
  //  does between two tiles is not visible from inside one workgroup)
  // Debug library only: three correlated branches on one flag made the compiler clone everything between them
  // — the whole body, twice the code in a kernel that already holds one body per stagger group — and the product
  // step lost 4.5 % (instruction cache).
#ifdef OIBL_DEBUG_HOOKS
  const bool wgprof = OUTMX && p.prof != nullptr && wave == 0 && lane == 0 && blockIdx.y == 0;
#else
  constexpr bool wgprof = false;
#endif
  if (wgprof) {
    unsigned hw, xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    p.prof[64 + 4 * (size_t)blockIdx.x] = __builtin_amdgcn_s_memtime();
    p.prof[64 + 4 * (size_t)blockIdx.x + 1] = (unsigned long long)hw | ((unsigned long long)xcc << 32);
  }
  int tm, tn;
  xcd_tile(blockIdx.x, (unsigned)p.tiles_m, (unsigned)p.tiles_n, p.raster, tm, tn);
  const int m0 = p.m_base + tm * G::BM, n0 = tn * G::BN;
  const bool splitk = p.nsteps_part != 0;
  const int nsteps = splitk ? p.nsteps_part : 9 * (p.cin >> (X3 ? 5 : 6));
  const int outer0 = (int)blockIdx.y * p.k_outer_step;

  const int piece = ring_piece(wave, lane);
  int rows_a[2 * NA], rows_b[2 * NB];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int i = 0; i < NA; ++i) rows_a[NA * h + i] = ring_a_row<WM>(wave, lane, h, i);
#pragma unroll
    for (int i = 0; i < NB; ++i) rows_b[NB * h + i] = ring_b_row<WM>(wave, lane, h, i);
  }
  ConvRingALoader<NA, POOL, X3> la;
  ConvRingBLoader<NB, X3> lb;
  la.init(p, m0, rows_a, piece, outer0);
  lb.init(p, n0, rows_b, MX ? ring_piece_mxb(wave, lane) : piece, outer0);   // (operand format, not output format)

  // The accumulators start at the bias (the fma chain of every output begins with it), laid out
  // like the results: natural layout = one channel per lane and column tile, transposed layout
  // (!POOL) = 4 consecutive channels per register quad.  Nothing is added in the epilogue, and the
  // loads are long done when the loop ends.
  f32x16_t acc[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if constexpr (POOL) {
      float b = p.bias[n0 + wn * 64 + j * 32 + (lane & 31)] * p.bias_mul;
      b = splitk ? 0.f : b;   // (split-K partials start from zero: the reduction adds the bias)
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          asm volatile("" : "+v"(b));  // 64 distinct registers, not 64 aliases of one value
          acc[i][j][r] = b;
        }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 b =
            *reinterpret_cast<const float4*>(p.bias + n0 + wn * 64 + j * 32 + 8 * g + 4 * (lane >> 5));
        b = make_float4(b.x * p.bias_mul, b.y * p.bias_mul, b.z * p.bias_mul, b.w * p.bias_mul);
        if (splitk) b = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          acc[i][j][4 * g] = b.x;
          acc[i][j][4 * g + 1] = b.y;
          acc[i][j][4 * g + 2] = b.z;
          acc[i][j][4 * g + 3] = b.w;
        }
      }
    }
  }

  const unsigned long long t_loop = prof ? __builtin_amdgcn_s_memtime() : 0;
  ring_mainloop<WM, ODD, !POOL, P, BAR1, GROUP>(acc, smem, wave, lane, la, lb, nsteps,
                                                (P == RING_MX_PROF && blockIdx.x == 0 && p.prof) ? p.prof + 8 : nullptr);
  // (the main loop ends on a workgroup barrier: the staging LDS is free for the epilogue)
  if (p.out_mul != 1.f) {   // (uniform; one layer of the f16mx backbone)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= p.out_mul;
  }
  const unsigned long long t_epi = prof ? __builtin_amdgcn_s_memtime() : 0;
  if (wgprof) p.prof[64 + 4 * (size_t)blockIdx.x + 3] = __builtin_amdgcn_s_memtime();

  // ---- epilogue: bias (+ReLU) (+2x2 max-pool over register quads), transpose through LDS,
  //      full-line NHWC stores.
  //      POOL : accumulators in the natural layout (lane = channel, registers = pixels): a pooling
  //             window is one register quad -> three v_max, 2-byte LDS writes of the pooled values.
  //      !POOL: accumulators transposed (lane = pixel, register quad = 4 consecutive channels):
  //             one packed 8-byte LDS write per quad, a quarter of the write instructions.
  //      X3   : an output element is 4 bytes — the (hi, lo) pair at x3_off(channel) / + 64, or the
  //             fp32 value when out_f32 is set; without pooling the tile is staged in two passes of
  //             BM / 2 rows (accumulator row tiles {0, 1}, then {2, 3} of every wave).
  if constexpr (OUTMX) {
    // ---- f16mx epilogue.  The tile (without pooling: one half of its rows at a time) is staged as
    // fp32, rows of BN floats = CPR 16-byte chunks, chunk q of row r at physical chunk q ^ (r % CPR):
    // the 8 chunks of a (row, 32-channel group) item stay inside one aligned 128-byte block, 16
    // consecutive rows of one logical chunk fall on 16 distinct bank slots (the accumulator writes
    // and the item reads walk rows), and a row's chunks read in order are a permutation of the row
    // (the copy-out walks chunks).  Then one thread per item reads its 32 values, packs the f16mx
    // line (mx_pack_line) and writes it back in place; the copy-out moves full lines.
    // out_f32 (the layer feeding the fp32 head): no packing, the fp32 rows go out as they are.
    constexpr int CPR = G::BN / 4;
    constexpr int ROWB = G::BN * 4;
    constexpr int PASSES = POOL ? 1 : 2;
    constexpr int ROWS = (POOL ? G::BM / 4 : G::BM) / PASSES;
    constexpr int ITEMS = ROWS * (G::BN / 32);
    constexpr int ITERS = ROWS * CPR / 512, BATCH = 8;
    static_assert(ITEMS % 512 == 0 && ITERS % BATCH == 0, "f16mx epilogue shape");
    // (split-K: the partial tensor of this K range, rows counted from m_base)
    const long row0 = (POOL ? (m0 >> 2) : m0) - (splitk ? p.m_base : 0);
    char* obase = reinterpret_cast<char*>(p.out) + (long)n0 * 4 + (splitk ? (size_t)blockIdx.y * p.part_stride : 0);
    const long orow_bytes = (long)p.cout * 4;
    const float floor_v = p.relu ? 0.f : -INFINITY;
    unsigned long long t_copy = 0;
    unsigned long long ep_[6] = {};
#pragma unroll
    for (int pass = 0; pass < PASSES; ++pass) {
      if constexpr (POOL) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int col = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
          for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const float v = fmaxf(fmaxf(fmaxf(acc[i][j][4 * g], acc[i][j][4 * g + 1]),
                                          fmaxf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])), floor_v);
              const int row = (wm * 4 + i) * 8 + 2 * g + (lane >> 5);
              *reinterpret_cast<float*>(smem + row * ROWB + (((col >> 2) ^ (row & (CPR - 1))) << 4) + (col & 3) * 4) = v;
            }
        }
      } else {
        const int half = lane >> 5;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const int q = (wn * 64 + j * 32 + 8 * g + 4 * half) >> 2;  // chunk of this quad's 4 channels
#pragma unroll
            for (int i2 = 0; i2 < 2; ++i2) {
              const int i = 2 * pass + i2;
              const int row = wm * 64 + i2 * 32 + (lane & 31);
              *reinterpret_cast<float4*>(smem + row * ROWB + ((q ^ (row & (CPR - 1))) << 4)) =
                  make_float4(fmaxf(acc[i][j][4 * g], floor_v), fmaxf(acc[i][j][4 * g + 1], floor_v),
                              fmaxf(acc[i][j][4 * g + 2], floor_v), fmaxf(acc[i][j][4 * g + 3], floor_v));
            }
          }
      }
      __syncthreads();
      if (pass == 0 && prof) t_copy = __builtin_amdgcn_s_memtime();
      if (prof) ep_[3 * pass] = __builtin_amdgcn_s_memtime();       // staged
      if (!p.out_f32) {
#pragma unroll 1
        for (int it = 0; it < ITEMS / 512; ++it) {
          const int item = it * 512 + (int)threadIdx.x;
          const int row = item % ROWS, grp = item / ROWS;
          char* const rowp = smem + row * ROWB;
          const int sw = row & (CPR - 1);
          float v[32];
#pragma unroll
          for (int k = 0; k < 8; ++k) {
            const float4 t = *reinterpret_cast<const float4*>(rowp + (((grp * 8 + k) ^ sw) << 4));
            v[4 * k] = t.x;
            v[4 * k + 1] = t.y;
            v[4 * k + 2] = t.z;
            v[4 * k + 3] = t.w;
          }
          uint4 line[8];
          mx_pack_line(v, line, p.range_flag);
#pragma unroll
          for (int k = 0; k < 8; ++k) *reinterpret_cast<uint4*>(rowp + (((grp * 8 + k) ^ sw) << 4)) = line[k];
        }
        __syncthreads();
      }
      if (prof) ep_[3 * pass + 1] = __builtin_amdgcn_s_memtime();   // packed
#pragma unroll
      for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
        uint4 v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const int idx = (it0 + u) * 512 + (int)threadIdx.x;
          const int lr = idx / CPR, q = idx % CPR;
          v[u] = *reinterpret_cast<const uint4*>(smem + lr * ROWB + ((q ^ (lr & (CPR - 1))) << 4));
        }
#pragma unroll
        for (int u = 0; u < BATCH; ++u) {
          const int idx = (it0 + u) * 512 + (int)threadIdx.x;
          const int lr = idx / CPR;
          const long grow = row0 + (PASSES == 1 ? lr : (lr >> 6) * 128 + pass * 64 + (lr & 63));
          if (grow < p.out_rows)
            *reinterpret_cast<uint4*>(obase + grow * orow_bytes + (idx % CPR) * 16) = v[u];
        }
      }
      if (prof) ep_[3 * pass + 2] = __builtin_amdgcn_s_memtime();   // stores issued
      if (pass + 1 < PASSES) __syncthreads();
    }
    if (prof) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const unsigned long long t_end = __builtin_amdgcn_s_memtime();
      if (lane == 0) {
        p.prof[0] = t_loop - t_start;
        p.prof[1] = t_epi - t_loop;
        p.prof[2] = t_copy - t_epi;
        p.prof[3] = t_end - t_copy;
        p.prof[4] = ep_[1] - ep_[0];   // pass 0: pack
        p.prof[5] = ep_[2] - ep_[1];   //         copy-out (issue)
        p.prof[6] = PASSES > 1 ? ep_[4] - ep_[3] : 0;
        p.prof[7] = PASSES > 1 ? ep_[5] - ep_[4] : 0;
      }
    }
    if (wgprof) p.prof[64 + 4 * (size_t)blockIdx.x + 2] = __builtin_amdgcn_s_memtime();   // (stores issued, not drained)
    return;
  }
  constexpr int EB = X3 ? 4 : 2;
  constexpr int PITCH = G::BN * EB + 16;
  constexpr int PASSES = (X3 && !POOL) ? 2 : 1;
  constexpr int OUT_ROWS = (POOL ? G::BM / 4 : G::BM) / PASSES;  // rows staged per pass
  constexpr int CPR = G::BN * EB / 16;                           // 16-byte chunks per output row
  constexpr int ITERS = OUT_ROWS * CPR / 512, BATCH = ITERS < 8 ? ITERS : 8;
  static_assert(OUT_ROWS * CPR % 512 == 0 && ITERS % BATCH == 0, "copy-out shape");
  const long row0 = POOL ? (m0 >> 2) : m0;
  char* obase = reinterpret_cast<char*>(p.out) + (long)n0 * EB;
  const long orow_bytes = (long)p.cout * EB;
  unsigned long long t_copy = 0;
#pragma unroll
  for (int pass = 0; pass < PASSES; ++pass) {
    if constexpr (POOL) {
      const float floor_v = p.relu ? 0.f : -INFINITY;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int col = wn * 64 + j * 32 + (lane & 31);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const float v = fmaxf(fmaxf(fmaxf(acc[i][j][4 * g], acc[i][j][4 * g + 1]),
                                        fmaxf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])), floor_v);
            const int row = (wm * 4 + i) * 8 + 2 * g + (lane >> 5);
            if constexpr (X3)
              x3_store(smem + row * PITCH, col, v);
            else
              *reinterpret_cast<uint16_t*>(smem + row * PITCH + col * 2) = f32_to_bf16_bits(v);
          }
      }
    } else if constexpr (X3) {
      const float floor_v = p.relu ? 0.f : -INFINITY;
      const int half = lane >> 5;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = wn * 64 + j * 32 + 8 * g + 4 * half;  // first of this quad's 4 channels
#pragma unroll
          for (int i2 = 0; i2 < 2; ++i2) {
            const int i = 2 * pass + i2;
            const float a0 = fmaxf(acc[i][j][4 * g], floor_v), a1 = fmaxf(acc[i][j][4 * g + 1], floor_v);
            const float a2 = fmaxf(acc[i][j][4 * g + 2], floor_v), a3 = fmaxf(acc[i][j][4 * g + 3], floor_v);
            char* rowp = smem + (wm * 64 + i2 * 32 + (lane & 31)) * PITCH;
            if (p.out_f32) {
              *reinterpret_cast<float4*>(rowp + c0 * 4) = make_float4(a0, a1, a2, a3);
            } else {
              uint2 hi, lo;
              ring_split4(a0, a1, a2, a3, hi, lo);
              char* q = rowp + (c0 >> 5) * 128 + (c0 & 31) * 2;
              *reinterpret_cast<uint2*>(q) = hi;
              *reinterpret_cast<uint2*>(q + 64) = lo;
            }
          }
        }
    } else {
      typedef __attribute__((ext_vector_type(2))) float f2;
      typedef __attribute__((ext_vector_type(2))) __bf16 bf2;
      typedef __attribute__((ext_vector_type(2))) short s2;
      // ReLU on the packed pair: as signed 16-bit integers every negative bf16 is < 0 (see conv.hip)
      const short fl = p.relu ? (short)0 : (short)-32768;
      const s2 floor2 = {fl, fl};
      const int half = lane >> 5;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int c0 = wn * 64 + j * 32 + 8 * g + 4 * half;  // first of this quad's 4 channels
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const bf2 lo = __builtin_convertvector((f2){acc[i][j][4 * g], acc[i][j][4 * g + 1]}, bf2);
            const bf2 hi = __builtin_convertvector((f2){acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]}, bf2);
            uint2 pk;
            pk.x = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2, lo), floor2));
            pk.y = __builtin_bit_cast(uint32_t, __builtin_elementwise_max(__builtin_bit_cast(s2, hi), floor2));
            const int row = wm * 128 + i * 32 + (lane & 31);
            *reinterpret_cast<uint2*>(smem + row * PITCH + c0 * 2) = pk;
          }
        }
    }
    __syncthreads();
    if (pass == 0 && prof) t_copy = __builtin_amdgcn_s_memtime();
    // LDS reads of a batch first, then its stores: the loads' latency is paid once per batch
#pragma unroll
    for (int it0 = 0; it0 < ITERS; it0 += BATCH) {
      uint4 v[BATCH];
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = (it0 + u) * 512 + (int)threadIdx.x;
        v[u] = *reinterpret_cast<const uint4*>(smem + (idx / CPR) * PITCH + (idx % CPR) * 16);
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = (it0 + u) * 512 + (int)threadIdx.x;
        const int lr = idx / CPR;
        // staged row -> tile row: all rows in order, or (two passes) 64 of every wave row's 128
        const long grow = row0 + (PASSES == 1 ? lr : (lr >> 6) * 128 + pass * 64 + (lr & 63));
        if (grow < p.out_rows)
          *reinterpret_cast<uint4*>(obase + grow * orow_bytes + (idx % CPR) * 16) = v[u];
      }
    }
    if (pass + 1 < PASSES) __syncthreads();  // the staging rows are rewritten by the next pass
  }
  if (prof) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = __builtin_amdgcn_s_memtime();
    if (lane == 0) {
      p.prof[0] = t_loop - t_start;
      p.prof[1] = t_epi - t_loop;
      p.prof[2] = t_copy - t_epi;
      p.prof[3] = t_end - t_copy;
    }
  }
}

template <int WM, bool POOL, bool ODD, int P = RING_BF16, bool OUTMX = (P >= RING_MX), bool BAR1 = false>
__global__ __launch_bounds__(512) void conv3x3_ring_kernel(RingParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
#ifdef OIBL_DEBUG_HOOKS
  if (p.stagger > 0 && blockIdx.x < 256 && blockIdx.y == 0) {
    const int n = (int)(blockIdx.x & 3) * p.stagger;
    for (int i = 0; i < n; ++i) __builtin_amdgcn_s_sleep(127);
  }
#endif
  if constexpr (BAR1) {
    if ((wave >> 2) == 0) conv3x3_ring_body<WM, POOL, ODD, P, OUTMX, true, 0>(p, smem, lane, wave);
    else conv3x3_ring_body<WM, POOL, ODD, P, OUTMX, true, 1>(p, smem, lane, wave);
  } else {
    conv3x3_ring_body<WM, POOL, ODD, P, OUTMX, false, -1>(p, smem, lane, wave);
  }
}

}  // namespace oibl
