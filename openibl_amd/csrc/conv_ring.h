// Implicit-GEMM 3x3 convolution, bf16, 256 x 256 or 512 x 128 tile — "ring" schedule for gfx950.
//
// Same contraction and the same summation order as conv3x3_igemm_kernel (conv.hip): M = output
// pixels, N = Cout, K = 9 * Cin ordered (tap, cin), one K-tile = 64 bf16 = one 128-byte line per
// row.  What changes is the pipeline.  The generic core drains its LDS-DMA queue (vmcnt(0)) and
// crosses one workgroup barrier per K-tile, so the whole workgroup waits for the slowest line of
// every tile.  Here
//
//   * a K-tile is staged as four 16-KiB UNITS (A0, A1: the two 64-row halves of every wave row's
//     128 pixels; B0, B1: the two 32-column halves of every wave column's 64 output channels), and
//     LDS holds two K-tiles (128 KiB).  A unit is re-filled two phases after its last fragment
//     read, i.e. with the data of K-tile t + 2, and is read six phases after it was issued: at any
//     time five units (80 KiB per CU) are in flight, waited for with a COUNTED s_waitcnt vmcnt(10)
//     — the queue is never drained inside the loop;
//   * a K-tile is four PHASES, one per 64 x 32 quadrant of the wave's 128 x 64 accumulator
//     (8 MFMAs 32x32x16 each).  Every phase is a LOAD segment (fragment ds_reads of the operand
//     half that changes: 8, 4, 8, 4 reads; two LDS-DMA instructions; the counted wait) and a
//     COMPUTE segment (lgkmcnt(0); 8 MFMAs), separated by raw s_barriers;
//   * the two wave rows (waves 0-3 / 4-7: one wave of each on every SIMD) run ONE BARRIER APART:
//     while one wave of a SIMD is in its COMPUTE segment the other is in its LOAD segment, so the
//     matrix pipe of the SIMD always has a wave with operands in registers (s_setprio 1 around the
//     MFMAs lets it win issue arbitration against the loading partner).
//
// Hazard rules (cdna_hip_programming.md, "256^2 8-phase template"), with phases numbered globally:
//   RAW  a unit is read in phase >= w + 1 where w is the phase whose LOAD segment holds the
//        vmcnt that retires it (own loads) and whose closing barriers make the other waves' loads
//        visible;  here w = read - 1 and vmcnt(10) after the phase's own 2 issues leaves exactly
//        the 5 youngest units outstanding.
//   WAR  a unit is re-staged in phase >= r + 2 where r is the last phase that reads it (the
//        lagging wave row retires those reads after the barrier that ends phase r).
// Unit schedule for K-tile t (phases 4t .. 4t+3), reads / (re)stages:
//   P0: read A0(t)            stage A1(t+1)          P1: read B1(t)       stage B0(t+2)
//   P2: read A1(t)            stage A0(t+2)          P3: read B0(t+1)     stage B1(t+2)
// B0 lives in one of two fragment register sets (X/Y) that swap roles every K-tile, so that the
// next tile's B0 can be fetched during P3 while the current B0 is still being multiplied.
//
// Addressing: both operands are fetched with buffer_load_dwordx4 ... lds (16 B per lane, straight
// into LDS).  The per-lane part of the address is a 32-bit offset that is constant for the whole
// kernel (B) or for one tap (A); the per-K-tile part is a scalar offset.  A tap that leaves the
// image gets an offset beyond num_records: the buffer unit returns zeros — zero padding without a
// padded copy, a zero line or any per-tile select.
#pragma once

#include <type_traits>

#include "gemm_core.h"

namespace oibl {

constexpr unsigned RG_OOB = 0xF0000000u;  // voffset of an out-of-image tap (>= num_records)

// Geometry of one instantiation.  WM = wave rows (2 or 4); the 8 waves form a WM x (8 / WM) grid,
// every wave owns 128 x 64 outputs, so the tile is 256 x 256 (WM = 2, Cout % 256 == 0) or
// 512 x 128 (WM = 4, Cout % 128 == 0).  Stagger group of a wave = wave >> 2 (waves w and w + 4
// share a SIMD).
template <int WM_>
struct RingGeo {
  static constexpr int WM = WM_, WN = 8 / WM_;
  static constexpr int BM = WM * 128, BN = WN * 64;
  static constexpr int NA = WM;       // LDS-DMA instructions per wave per A unit (WM * 64 rows)
  static constexpr int NB = WN / 2;   // ... per B unit (WN * 32 rows)
  static constexpr int A_UNIT = WM * 64 * 128, B_UNIT = WN * 32 * 128;
  static constexpr int TILE = 2 * A_UNIT + 2 * B_UNIT;  // one K-tile: A0 A1 B0 B1
  static constexpr int MAIN_LDS = 2 * TILE;
  static_assert(WM == 2 || WM == 4, "wave grid");
};

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
};

template <int WM, bool POOL>
constexpr int ring_lds_bytes() {
  using G = RingGeo<WM>;
  constexpr int rows = POOL ? G::BM / 4 : G::BM;
  constexpr int epi = rows * (G::BN * 2 + 16);
  return epi > G::MAIN_LDS ? epi : G::MAIN_LDS;
}

__device__ static inline void buf_glds16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff,
                                         char* lds_wave_base) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(
      rsrc, (__attribute__((address_space(3))) void*)lds_wave_base, 16, (int)voff, (int)soff, 0, 0);
}

template <int N>
__device__ static inline void wait_vmcnt() {
  static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int WM, bool POOL, bool ODD>
__global__ __launch_bounds__(512) void conv3x3_ring_kernel(RingParams p) {
  using G = RingGeo<WM>;
  constexpr int NA = G::NA, NB = G::NB;
  constexpr int OFF_A0 = 0, OFF_A1 = G::A_UNIT, OFF_B0 = 2 * G::A_UNIT, OFF_B1 = 2 * G::A_UNIT + G::B_UNIT;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / G::WN, wn = wave % G::WN;
  const int group = wave >> 2;  // stagger group: one wave of each group on every SIMD
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
  const int m0 = tm * G::BM, n0 = tn * G::BN;
  const int pix_bytes = p.cin * 2;
  const int cchunks = p.cin >> 6;
  const int nsteps = 9 * cchunks;

  const __amdgpu_buffer_rsrc_t rs_a =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rs_b =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);

  // ---- staging geometry: LDS-DMA instruction i of this wave fills unit rows
  //      u = 8 * (wave + 8 i) + (lane >> 3); lane's 16-B piece is XOR-swizzled on the SOURCE side
  //      (physical slot = logical ^ ((u >> 1) & 7) = logical ^ (4 (wave & 1) + (lane >> 4))).
  const int piece = ((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) * 16;
  unsigned a_base[2 * NA];  // [NA h + i]: byte offset of the pixel (centre tap) + piece
  unsigned a_mask[2 * NA];  // 9-bit tap validity
  unsigned b_off[2 * NB];   // [NB h + i]: byte offset of the weight row (tap 0, chunk 0) + piece
  {
    const int Hq = POOL ? (p.H >> 1) : p.H, Wq = POOL ? (p.W >> 1) : p.W;
    const unsigned hw = (unsigned)Hq * (unsigned)Wq;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < NA; ++i) {
        const int u = 8 * (wave + 8 * i) + (lane >> 3);
        const unsigned m = (unsigned)m0 + (unsigned)((u >> 6) * 128 + h * 64 + (u & 63));
        unsigned mk = 0, off = 0;
        if (m < (unsigned)p.m_total) {
          const unsigned q = POOL ? (m >> 2) : m;
          const unsigned sub = POOL ? (m & 3u) : 0u;
          const unsigned n = q / hw;
          const unsigned rem = q - n * hw;
          const unsigned yq = rem / (unsigned)Wq;
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
        a_mask[NA * h + i] = mk;
        a_base[NA * h + i] = off + piece;
      }
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int u = 8 * (wave + 8 * i) + (lane >> 3);
        const int col = (u >> 5) * 64 + h * 32 + (u & 31);
        b_off[NB * h + i] = (unsigned)(n0 + col) * (unsigned)pix_bytes + piece;
      }
  }
  const unsigned tap_stride = (unsigned)p.cout * (unsigned)pix_bytes;

  // staging cursor: describes the K-tile whose units are currently being issued
  int s_tap = 0, s_cc = -1;
  unsigned a_cur[2 * NA];  // per-tap A offsets (RG_OOB when the tap leaves the image)
  unsigned a_soff = 0, b_soff = 0;
  auto begin_tile = [&]() __attribute__((always_inline)) {
    ++s_cc;
    if (s_cc == cchunks) {
      s_cc = 0;
      ++s_tap;
    }
    if (s_cc == 0) {
      const int ky = (s_tap * 11) >> 5, kx = s_tap - 3 * ky;
      const int toff = ((ky - 1) * p.W + (kx - 1)) * pix_bytes;
#pragma unroll
      for (int j = 0; j < 2 * NA; ++j)
        a_cur[j] = ((a_mask[j] >> s_tap) & 1u) ? a_base[j] + (unsigned)toff : RG_OOB;
    }
    a_soff = (unsigned)s_cc * 128u;
    b_soff = (unsigned)s_tap * tap_stride + (unsigned)s_cc * 128u;
  };
  char* const st_base = smem + wave * 1024;
  auto stage_a = [&](int buf, int h) __attribute__((always_inline)) {
    char* d = st_base + buf * G::TILE + (h ? OFF_A1 : OFF_A0);
#pragma unroll
    for (int i = 0; i < NA; ++i) buf_glds16(rs_a, a_cur[NA * h + i], a_soff, d + i * 8192);
  };
  auto stage_b = [&](int buf, int h) __attribute__((always_inline)) {
    char* d = st_base + buf * G::TILE + (h ? OFF_B1 : OFF_B0);
#pragma unroll
    for (int i = 0; i < NB; ++i) buf_glds16(rs_b, b_off[NB * h + i], b_soff, d + i * 8192);
  };

  // ---- fragment read geometry
  int frag_off[4];
  {
    const int row = lane & 31, half = lane >> 5, swz = (lane >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = row * 128 + (((2 * kk + half) ^ swz) * 16);
  }
  const char* const rd_a = smem + wm * 8192;  // + buf * TILE + OFF_A{h} + i2 * 4096
  const char* const rd_b = smem + wn * 4096;  // + buf * TILE + OFF_B{h}

  bf16x8_t fa[2][4], fbx[4], fby[4];
  auto read_a = [&](int buf, int h) __attribute__((always_inline)) {
    const char* s = rd_a + buf * G::TILE + (h ? OFF_A1 : OFF_A0);
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk)
        fa[i2][kk] = *reinterpret_cast<const bf16x8_t*>(s + i2 * 4096 + frag_off[kk]);
  };
  auto read_b = [&](int buf, int h, bf16x8_t (&f)[4]) __attribute__((always_inline)) {
    const char* s = rd_b + buf * G::TILE + (h ? OFF_B1 : OFF_B0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f[kk] = *reinterpret_cast<const bf16x8_t*>(s + frag_off[kk]);
  };

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  auto compute = [&](auto h_c, auto j_c, const bf16x8_t (&fb)[4]) __attribute__((always_inline)) {
    constexpr int h = decltype(h_c)::value, j = decltype(j_c)::value;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2)
        acc[2 * h + i2][j] =
            __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i2][kk], fb[kk], acc[2 * h + i2][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  // ---- prologue: B0 A0 B1 A1 of K-tile 0, B0 A0 B1 of K-tile 1 (the steady-state issue order)
  begin_tile();
  stage_b(0, 0);
  stage_a(0, 0);
  stage_b(0, 1);
  stage_a(0, 1);
  begin_tile();
  stage_b(1, 0);
  stage_a(1, 0);
  stage_b(1, 1);
  wait_vmcnt<2 * NA + 3 * NB>();  // B0(0), A0(0) of this wave have landed
  bar();
  read_b(0, 0, fbx);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if (group == 1) bar();  // group 1 runs one barrier behind group 0

  // One K-tile = 4 phases.  PAR = tile parity (LDS buffer; which register set holds B0).
  // TAIL: 0 = steady state, 1 = tile nsteps-2, 2 = tile nsteps-1 (nothing left to stage).
  // The counted waits leave exactly the five youngest units in flight (steady state); in the tail
  // the units that are no longer issued are subtracted.
  auto ktile = [&](auto par_c, auto tail_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr int TAIL = decltype(tail_c)::value;
    bf16x8_t(&b0)[4] = PAR ? fby : fbx;  // B0 of this tile
    bf16x8_t(&b1)[4] = PAR ? fbx : fby;  // B1 of this tile; from P3 on: B0 of the next tile
    // P0: A0 x B0
    read_a(PAR, 0);
    if constexpr (TAIL <= 1) {
      stage_a(PAR ^ 1, 1);  // A1(t+1)
      wait_vmcnt<3 * NA + 2 * NB>();
    } else wait_vmcnt<NA>();
    bar();
    compute(I0{}, I0{}, b0);
    bar();
    // P1: A0 x B1
    read_b(PAR, 1, b1);
    if constexpr (TAIL == 0) {
      begin_tile();
      stage_b(PAR, 0);  // B0(t+2)
      wait_vmcnt<2 * NA + 3 * NB>();
    } else if constexpr (TAIL == 1) wait_vmcnt<2 * NA + 2 * NB>();
    else wait_vmcnt<0>();
    bar();
    compute(I0{}, I1{}, b1);
    bar();
    // P2: A1 x B1
    read_a(PAR, 1);
    if constexpr (TAIL == 0) {
      stage_a(PAR, 0);  // A0(t+2)
      wait_vmcnt<3 * NA + 2 * NB>();
    } else if constexpr (TAIL == 1) wait_vmcnt<2 * NA + NB>();
    bar();
    compute(I1{}, I1{}, b1);
    bar();
    // P3: A1 x B0   (B0 of the next tile goes into the register set B1 just vacated)
    if constexpr (TAIL <= 1) read_b(PAR ^ 1, 0, b1);
    if constexpr (TAIL == 0) {
      stage_b(PAR, 1);  // B1(t+2)
      wait_vmcnt<2 * NA + 3 * NB>();
    } else if constexpr (TAIL == 1) wait_vmcnt<NA + NB>();
    bar();
    compute(I1{}, I0{}, b0);
    bar();
  };
  for (int t = 0; t + 3 < nsteps; t += 2) {  // pairs of steady-state tiles
    ktile(I0{}, I0{});
    ktile(I1{}, I0{});
  }
  if constexpr (ODD) {  // Cin = 64: nine K-tiles
    ktile(I0{}, I0{});
    ktile(I1{}, I1{});
    ktile(I0{}, I2{});
  } else {
    ktile(I0{}, I1{});
    ktile(I1{}, I2{});
  }
  if (group == 0) bar();
  __syncthreads();  // staging LDS is free for the epilogue

  // ---- epilogue: bias (+ReLU) (+2x2 max-pool over register quads), transpose through LDS,
  //      full-line NHWC stores
  constexpr int PITCH = G::BN * 2 + 16;
  constexpr int OUT_ROWS = POOL ? G::BM / 4 : G::BM;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int col = wn * 64 + j * 32 + (lane & 31);
    const float b = p.bias[n0 + col];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if constexpr (POOL) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float v = fmaxf(fmaxf(acc[i][j][4 * g], acc[i][j][4 * g + 1]),
                          fmaxf(acc[i][j][4 * g + 2], acc[i][j][4 * g + 3])) + b;
          if (p.relu) v = fmaxf(v, 0.f);
          const int row = (wm * 4 + i) * 8 + 2 * g + (lane >> 5);
          *reinterpret_cast<uint16_t*>(smem + row * PITCH + col * 2) = f32_to_bf16_bits(v);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float v = acc[i][j][r] + b;
          if (p.relu) v = fmaxf(v, 0.f);
          const int row = (wm * 4 + i) * 32 + acc_row(r, lane);
          *reinterpret_cast<uint16_t*>(smem + row * PITCH + col * 2) = f32_to_bf16_bits(v);
        }
      }
    }
  }
  __syncthreads();
  constexpr int CPR = G::BN * 2 / 16;  // 16-byte chunks per output row
  const long row0 = POOL ? (m0 >> 2) : m0;
  char* obase = reinterpret_cast<char*>(p.out) + (long)n0 * 2;
  const long orow_bytes = (long)p.cout * 2;
  for (int idx = threadIdx.x; idx < OUT_ROWS * CPR; idx += 512) {
    const int row = idx / CPR, ch = idx - row * CPR;
    const long grow = row0 + row;
    if (grow < p.out_rows)
      *reinterpret_cast<uint4*>(obase + grow * orow_bytes + ch * 16) =
          *reinterpret_cast<const uint4*>(smem + row * PITCH + ch * 16);
  }
}

}  // namespace oibl
