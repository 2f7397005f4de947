// Implicit-GEMM 3x3 convolution with a RESIDENT INPUT HALO, f16mx / bf16x3 (4-byte elements), 256-pixel
// x 256-channel tile.
//
// Why.  The ring kernels (conv_ring.h) stage every K-tile's A unit — one 128-byte channel chunk of 256
// pixels for ONE tap — from L2: a pixel's line crosses the L2 -> LDS path nine times per channel chunk
// (once per tap) and the tile moves 64 KB per K-tile.  In bf16x3 the 12 MFMAs of a phase cover that;
// with the 6 MFMAs of f16mx the kernel is bound by operand delivery instead (profiles/r03_*: the same
// ~1.5 us per K-tile as bf16 with a third less matrix work; every load redirected to one L1-resident line:
// -22 %; no LDS-DMA at all: -41 %; 32 CUs of an XCD pull 2 MB per K-tile through one L2).  Here
//   * the tile's pixels are a PH x PW PATCH of one image (PH * PW <= 256, e.g. 8 x 32, 12 x 20, 6 x 40) and
//     the (PH + 2) x (PW + 2) halo of one channel chunk — <= 344 lines of 128 bytes — is staged ONCE and
//     multiplied by all nine taps: the A fragments of tap (dy, dx) are read from the halo at a shifted
//     position, so K runs (chunk outer, tap inner) and the next chunk's halo streams in behind the nine
//     taps of the current one (two halo buffers);
//   * only the weights travel per K-tile: 32 KB + 4.8 KB of halo instead of 64 KB.
// LDS: 2 x 43 KiB halo + 2 K-tiles x 32 KiB of weights (B0 | B1 as in ring_core.h) + 8 KiB sink = 158 KiB.
//
// Schedule: the ring's — four phases per K-tile (A0 x B0, A0 x B1, A1 x B1, A1 x B0), two stagger groups
// one barrier apart, B0 in two register sets — with the A units replaced by the halo:
//   P0: read A0 (halo, tap)   issue 1 LDS-DMA: slot `tap` of the next chunk's halo   vmcnt(2 NB)
//   P1: read B1(t)            issue B0(t+2)
//   P2: read A1 (halo, tap)                                                         vmcnt(2 NB)
//   P3: read B0(t+1)          issue B1(t+2)
// Every K-tile issues exactly 1 + NB + NB = 5 LDS-DMA instructions per wave (a halo slot that does not
// exist — taps 6..8, waves whose sixth slot is beyond the halo, the chunk after the last — and the
// weight units beyond the last K-tile load some in-range line into a per-wave sink), so the waits are the
// same constants everywhere; they count the weight instructions only (see the note at the waits: fully
// out-of-range LDS-DMA instructions do not retire in order).  A halo slot is at least 19 instructions
// old when its chunk starts.  Hazards as in ring_core.h (read >= 1 phase after the
// retiring wait; re-stage >= 2 phases after the last read: the halo buffer of chunk c+1 was last read in
// P2 of tap 8 of chunk c-1, its first slot is issued in P0 of tap 0 of chunk c).
//
// Halo image: position pos = hy * (PW + 2) + hx, one 128-byte line each, 16-byte slots XOR-swizzled by
// f(hx, hy) = ((hx >> 1) & 7) ^ ((hy & 1) << 2) on the LDS-DMA source side and on the fragment read: 16
// consecutive positions of a row, and the 2 x 8 positions of four pooling quads, fall on 16 distinct bank
// slots.  A lane keeps, per 32-row fragment and per dx, the byte address of its pixel's top-left tap with
// the swizzle folded in (12 registers); a tap adds a scalar line offset and XORs the constant
// (kk << 5) ^ ((dy & 1) << 6).
#pragma once

#include "conv_ring.h"

namespace oibl {

constexpr int HALO_MAX_POS = 344;  // 43 LDS-DMA instructions of 8 positions
constexpr int HALO_SLOTS = 6;      // per wave: instruction wave + 8 j, j < 6
constexpr int HALO_BYTES = HALO_MAX_POS * 128;
constexpr int HALO_OFF_B = 2 * HALO_BYTES;
constexpr int HALO_B_TILE = 2 * RingGeo<2>::B_UNIT;
constexpr int HALO_OFF_SINK = HALO_OFF_B + 2 * HALO_B_TILE;
constexpr int HALO_LDS = HALO_OFF_SINK + 8 * 1024;
static_assert(HALO_LDS <= 160 * 1024, "LDS budget");

struct HaloParams {
  const void* in;
  const void* w;
  const float* bias;
  void* out;
  unsigned in_bytes, w_bytes;
  int N, H, W, cin, cout;
  int PH, PW;              // patch (both even, PH * PW <= 256, (PH + 2) * (PW + 2) <= 344)
  int tiles_y, tiles_x;    // patches per image
  int tiles_m, tiles_n, raster;
  unsigned img_mul, img_sh;  // tile -> image: divide by tiles_y * tiles_x
  unsigned tx_mul, tx_sh;    // divide by tiles_x
  unsigned pw_mul, pw_sh;    // divide by PW (pooled: PW / 2): tile row -> pixel / quad
  unsigned hp_mul, hp_sh;    // divide by PW + 2: halo position -> (hy, hx)
  int relu, out_f32;
  unsigned* range_flag;      // raised when an output is beyond fp16 (common.h, mx_raise_range_flag); may be null
  unsigned long long* prof;  // test hook (debug library, conv_halo4.h): per-workgroup stamps, tests/gpu_halo4_phase.py
  float bias_mul = 1.f, out_mul = 1.f;   // the f16mx backbone's activation scale (conv_ring.h, RingParams)
};

// one phase's matrix work on two 32x32 accumulator tiles (ring_core.h, compute)
template <int P, bool SWAP, typename Prep>
__device__ static inline void halo_phase_mma(f32x16_t& acc0, f32x16_t& acc1, const bf16x8_t (&fa)[2][4],
                                             const bf16x8_t (&fb)[4], Prep&& prep, bool prio2 = false) {
#ifdef OIBL_RING_LGKM0   // (ring_core.h, read_a: the compiler's counted waits instead)
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
  __builtin_amdgcn_sched_barrier(0);
  if (prio2) __builtin_amdgcn_s_setprio(2);   // (BAR1, group 0, a compile-time constant at every call: ring_core.h)
  else __builtin_amdgcn_s_setprio(1);
  if constexpr (P >= RING_MX) {
    typedef __attribute__((ext_vector_type(4))) int i4;
    auto f16 = [&](f32x16_t& acc, int i2, int k) __attribute__((always_inline)) {
      const f16x8_t a = __builtin_bit_cast(f16x8_t, fa[i2][k]), b = __builtin_bit_cast(f16x8_t, fb[k]);
      acc = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0)
                 : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    };
    const i32x8_t b8 = __builtin_shufflevector(__builtin_bit_cast(i4, fb[2]), __builtin_bit_cast(i4, fb[3]), 0, 1, 2,
                                               3, 4, 5, 6, 7);
    auto mx = [&](f32x16_t& acc, int i2) __attribute__((always_inline)) {
      const i32x8_t a8 = __builtin_shufflevector(__builtin_bit_cast(i4, fa[i2][2]), __builtin_bit_cast(i4, fa[i2][3]),
                                                 0, 1, 2, 3, 4, 5, 6, 7);
#ifdef OIBL_MX_TAIL_B128
      constexpr int SC = 7;   // (ring_core.h)
#else
      constexpr int SC = 6;
#endif
      acc = SWAP ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, acc, 2, 2, 0, b8[SC], 0, a8[SC])
                 : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc, 2, 2, 0, a8[SC], 0, b8[SC]);
    };
    f16(acc0, 0, 0);
    f16(acc1, 1, 0);
    prep();   // scalar work of the next LOAD segment, in the shadow of the matrix pipe (ring_core.h)
    f16(acc0, 0, 1);
    f16(acc1, 1, 1);
    mx(acc0, 0);
    mx(acc1, 1);
    asm volatile("" : "+v"(acc0), "+v"(acc1));  // pin the results inside the segment (ring_core.h)
  } else {
    static_assert(P == RING_X3, "halo kernel: 4-byte element types only");
    auto mma = [&](f32x16_t& acc, int i2, int ka, int kb) __attribute__((always_inline)) {
      acc = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb], fa[i2][ka], acc, 0, 0, 0)
                 : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i2][ka], fb[kb], acc, 0, 0, 0);
    };
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) {
      if (pr == 1) prep();
      mma(acc0, 0, pr + 2, pr);
      mma(acc1, 1, pr + 2, pr);
      mma(acc0, 0, pr, pr + 2);
      mma(acc1, 1, pr, pr + 2);
      mma(acc0, 0, pr, pr);
      mma(acc1, 1, pr, pr);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (P3's pre-read of the next B0: ring_core.h, compute)
  __builtin_amdgcn_s_setprio(0);
  __builtin_amdgcn_sched_barrier(0);
}

// VAR (experiments kept for tests/gpu_halo_determinism.py): 0 = production; 3 = the waits that count the
// halo instructions as outstanding (vmcnt(6) / vmcnt(5)) with out-of-range dummies: rarely WRONG, see below.
// BAR1: one barrier per phase and wave (ring_core.h) — the same hazard distances hold here; GROUP as in
// conv3x3_ring_body (conv_ring.h).
template <bool POOL, int P, int VAR, bool BAR1, int GROUP>
__device__ __forceinline__ void conv3x3_halo_body(const HaloParams& p, char* smem, const int lane, const int wave) {
  using G = RingGeo<2>;
  constexpr bool MX = P >= RING_MX;
  constexpr bool SWAP = !POOL;
  constexpr int NB = G::NB;
  const int wm = wave / G::WN, wn = wave % G::WN;
  const int group = wave >> 2;
  int tm, tn;
  xcd_tile(blockIdx.x, (unsigned)p.tiles_m, (unsigned)p.tiles_n, p.raster, tm, tn);
  const unsigned img = ring_div_u31((unsigned)tm, p.img_mul, p.img_sh);
  const unsigned trem = (unsigned)tm - img * (unsigned)(p.tiles_y * p.tiles_x);
  const unsigned tyi = ring_div_u31(trem, p.tx_mul, p.tx_sh);
  const int y0 = (int)tyi * p.PH, x0 = (int)(trem - tyi * (unsigned)p.tiles_x) * p.PW;
  const int n0 = tn * G::BN;
  const int HP = p.PW + 2;
  const int npos = (p.PH + 2) * HP;
  const int chunks = p.cin >> 5;
  const int nk = 9 * chunks;
  const unsigned pix_bytes = (unsigned)p.cin * 4u;

  // ---- halo loader: slot j of this wave = instruction wave + 8 j = halo positions 8 (wave + 8 j) .. + 7
  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  unsigned hvoff[HALO_SLOTS];
#pragma unroll
  for (int j = 0; j < HALO_SLOTS; ++j) {
    const int pos = (wave + 8 * j) * 8 + (lane >> 3);
    const int hy = (int)ring_div_u31((unsigned)pos, p.hp_mul, p.hp_sh), hx = pos - hy * HP;
    const int y = y0 - 1 + hy, x = x0 - 1 + hx;
    const bool ok = pos < npos && y >= 0 && y < p.H && x >= 0 && x < p.W;
    const int f = ((hx >> 1) & 7) ^ ((hy & 1) << 2);
    hvoff[j] = ok ? ((img * (unsigned)p.H + (unsigned)y) * (unsigned)p.W + (unsigned)x) * pix_bytes +
                        (unsigned)(((lane & 7) ^ f) << 4)
                  : RG_OOB;
  }
  char* const sink = smem + HALO_OFF_SINK + wave * 1024;
  // slot j (compile-time) of chunk cc into halo buffer hb; anything that does not exist goes to the sink
  auto stage_halo = [&](auto j_c, int hb, int cc) __attribute__((always_inline)) {
    constexpr int j = decltype(j_c)::value;
    if constexpr (j < HALO_SLOTS) {
      const int ii = wave + 8 * j;
      const bool real = ii * 8 < npos && cc < chunks;
      char* dst = real ? smem + hb * HALO_BYTES + ii * 1024 : sink;
      buf_glds16(rs_in, hvoff[j], (unsigned)(real ? cc : 0) * 128u, dst);
    } else {
      buf_glds16(rs_in, VAR == 3 ? RG_OOB : (unsigned)(lane * 16), 0u, sink);   // in range: see the waits
    }
  };

  // ---- weights: the ring's B loader in K order (chunk, tap)
  RingParams rp = {};
  rp.w = p.w;
  rp.w_bytes = p.w_bytes;
  rp.cin = p.cin;
  rp.cout = p.cout;
  rp.korder = 1;
  int rows_b[2 * NB];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < NB; ++i) rows_b[NB * h + i] = ring_b_row<2>(wave, lane, h, i);
  ConvRingBLoader<NB, true> lb;
  lb.init(rp, n0, rows_b, MX ? ring_piece_mxb(wave, lane) : ring_piece(wave, lane));
  char* const st_b = smem + HALO_OFF_B + wave * 1024;
  auto stage_b = [&](int buf, int h, bool real) __attribute__((always_inline)) {
    if (real) {
      lb.stage(h, st_b + buf * HALO_B_TILE + h * G::B_UNIT);
    } else {
#pragma unroll
      for (int i = 0; i < NB; ++i) buf_glds16(lb.rsrc, VAR == 3 ? RG_OOB : (unsigned)(lane * 16), 0u, sink);
    }
  };

  // ---- fragment addresses
  int frag_off[4];
  {
    const int row = lane & 31, half = lane >> 5, swz = (lane >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = row * 128 + (((2 * kk + half) ^ swz) * 16);
  }
  const char* const rd_b = smem + HALO_OFF_B + wn * 4096;
  // pre[h][i2][dx]: byte address (inside a halo buffer) of the top-left tap of this lane's pixel of
  // fragment (h, i2), slot bits = swizzle of the column the tap dx reads ^ lane half ^ row parity
  int pre[2][2][3];
  {
    const int half = lane >> 5;
    const int npix = p.PH * p.PW;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        int r = wm * 128 + h * 64 + i2 * 32 + (lane & 31);
        if (r >= npix) r = 0;  // dead rows of a patch smaller than 256 pixels: any valid position
        int py, px;
        if constexpr (POOL) {
          const int q = r >> 2, sub = r & 3, hw = p.PW >> 1;
          const int qy = (int)ring_div_u31((unsigned)q, p.pw_mul, p.pw_sh), qx = q - qy * hw;
          py = 2 * qy + (sub >> 1);
          px = 2 * qx + (sub & 1);
        } else {
          py = (int)ring_div_u31((unsigned)r, p.pw_mul, p.pw_sh);
          px = r - py * p.PW;
        }
        const int lb0 = (py * HP + px) * 128;
#pragma unroll
        for (int dxi = 0; dxi < 3; ++dxi) {
          const int hx = px + dxi;  // = (px + 1) + (dxi - 1)
          const int t = ((hx >> 1) & 7) ^ half ^ (((py + 1) & 1) << 2);
          pre[h][i2][dxi] = lb0 + (t << 4);
        }
      }
  }
  const int row_pitch = HP * 128;

  bf16x8_t fa[2][4], fbx[4], fby[4];
  auto ld_frag = [&](const char* a, int kk) __attribute__((always_inline)) -> bf16x8_t {
#ifndef OIBL_MX_TAIL_B128
    if constexpr (MX) {
      if (kk == 3) {
        typedef __attribute__((ext_vector_type(2))) unsigned u2;   // (not uint2: see ring_core.h, read_frag)
        const u2 d = *reinterpret_cast<const u2*>(a);
        const unsigned sc = *reinterpret_cast<const unsigned*>(a + 12);
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        return __builtin_bit_cast(bf16x8_t, (u4){d.x, d.y, sc, 0u});
      }
    }
#endif
    return *reinterpret_cast<const bf16x8_t*>(a);
  };
  auto read_a = [&](int hb, auto h_c, auto tap_c) __attribute__((always_inline)) {
    constexpr int h = decltype(h_c)::value, tap = decltype(tap_c)::value;
    constexpr int dyi = tap / 3, dxi = tap % 3;
    constexpr int cdy = (dyi != 1) ? 64 : 0;   // tap row parity differs from the centre row's
    // (opaque: every fragment address is loop invariant — 2 buffers x 9 taps x 16 fragments — and the
    //  compiler would otherwise hoist all of them out of the K loop into scratch)
    int rp_ = row_pitch;
    asm volatile("" : "+s"(rp_));
    const int tapoff = hb * HALO_BYTES + dyi * rp_ + dxi * 128;
    int a0[2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
      a0[i2] = pre[h][i2][dxi];
      asm volatile("" : "+v"(a0[i2]));   // (see above: keeps the XORed fragment addresses out of scratch)
      a0[i2] += tapoff;
    }
    // in the order the MFMAs consume them (k-chunk outer: ring_core.h, read_a)
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) fa[i2][kk] = ld_frag(smem + (a0[i2] ^ ((kk << 5) ^ cdy)), kk);
  };
  auto read_b = [&](int buf, int h, bf16x8_t (&f)[4]) __attribute__((always_inline)) {
    const char* s = rd_b + buf * HALO_B_TILE + h * G::B_UNIT;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f[kk] = ld_frag(s + frag_off[kk], kk);
  };
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // BAR1: the stagger group is a compile-time constant of one of two copies of the K loop (ring_core.h, GROUP)
  auto bar_g = [&](auto grp_c, int g) __attribute__((always_inline)) {
    if constexpr (BAR1) {
      __builtin_amdgcn_sched_barrier(0);
      if (decltype(grp_c)::value == g) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    } else {
      bar();
    }
  };

  // accumulators start at the bias (conv_ring.h)
  f32x16_t acc[4][2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    if constexpr (POOL) {
      float b = p.bias[n0 + wn * 64 + j * 32 + (lane & 31)] * p.bias_mul;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          asm volatile("" : "+v"(b));
          acc[i][j][r] = b;
        }
    } else {
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        float4 b = *reinterpret_cast<const float4*>(p.bias + n0 + wn * 64 + j * 32 + 8 * g + 4 * (lane >> 5));
        b = make_float4(b.x * p.bias_mul, b.y * p.bias_mul, b.z * p.bias_mul, b.w * p.bias_mul);
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

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  // ---- prologue: the halo of chunk 0, the weights of K-tiles 0 and 1; everything landed
#define HALO_IC(x) std::integral_constant<int, (x)> {}
  stage_halo(HALO_IC(0), 0, 0);
  stage_halo(HALO_IC(1), 0, 0);
  stage_halo(HALO_IC(2), 0, 0);
  stage_halo(HALO_IC(3), 0, 0);
  stage_halo(HALO_IC(4), 0, 0);
  stage_halo(HALO_IC(5), 0, 0);
  lb.begin_tile();
  stage_b(0, 0, true);
  stage_b(0, 1, true);
  lb.begin_tile();
  stage_b(1, 0, true);
  stage_b(1, 1, true);
  wait_vmcnt<0>();
  bar();
  read_b(0, 0, fbx);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if constexpr (!BAR1) {
    if (group == 1) bar();  // group 1 runs one barrier behind group 0
  }

  // one K-tile = tap `TAP` of channel chunk cc (halo buffer hb); PAR = parity of the K-tile (weight buffer,
  // register set of B0); kt = its index
  auto ktile = [&](auto grp_c, auto par_c, auto tap_c, int hb, int cc, int kt) __attribute__((always_inline)) {
    constexpr bool prio2 = BAR1 && decltype(grp_c)::value == 0;
    constexpr int PAR = decltype(par_c)::value;
    constexpr int TAP = decltype(tap_c)::value;
    bf16x8_t(&b0)[4] = PAR ? fby : fbx;
    bf16x8_t(&b1)[4] = PAR ? fbx : fby;
    const bool more = kt + 2 < nk;
    // P0: A0 x B0
    read_a(hb, I0{}, tap_c);
    stage_halo(tap_c, hb ^ 1, cc + 1);
    // Counted waits.  In issue order the unit read next (B1(t), issued in P3(t-2)) has 1 + NB + NB + 1
    // younger instructions here, B0(t+1) in P2 has NB + 1 + NB.  But an LDS-DMA instruction whose 64 lanes are
    // ALL out of range — the halo slots of a border tile's outside rows — was seen to retire ahead of older
    // in-range loads: with vmcnt(6) / vmcnt(5) one tile in ~100 launches came out wrong, always a top- or
    // bottom-row tile, more often with another stream loading the memory system (tests/gpu_halo_determinism.py:
    // 8 of 270 launches; 0 of 270 with the waits below).  So the halo instructions are not counted — the
    // waits allow only the 2 x NB younger WEIGHT instructions in flight — and every dummy is an in-range load.
    wait_vmcnt<(VAR == 3 ? 1 + 2 * NB + 1 : 2 * NB)>();
    bar_g(grp_c, 0);
    halo_phase_mma<P, SWAP>(acc[0][0], acc[1][0], fa, b0, [&] { lb.begin_tile(); }, prio2);  // (cursor -> K-tile
    bar_g(grp_c, 1);                          // t+2; unconditional: advanced under a branch it ends up in a VGPR)
    // P1: A0 x B1
    read_b(PAR, 1, b1);
    stage_b(PAR, 0, more);  // B0(t+2)
    bar_g(grp_c, 0);
    halo_phase_mma<P, SWAP>(acc[0][1], acc[1][1], fa, b1, [] {}, prio2);
    bar_g(grp_c, 1);
    // P2: A1 x B1
    read_a(hb, I1{}, tap_c);
    wait_vmcnt<(VAR == 3 ? 2 * NB + 1 : 2 * NB)>();
    bar_g(grp_c, 0);
    halo_phase_mma<P, SWAP>(acc[2][1], acc[3][1], fa, b1, [] {}, prio2);
    bar_g(grp_c, 1);
    // P3: A1 x B0   (B0 of the next K-tile goes into the register set B1 just vacated)
    read_b(PAR ^ 1, 0, b1);
    stage_b(PAR, 1, more);  // B1(t+2)
    bar_g(grp_c, 0);
    halo_phase_mma<P, SWAP>(acc[2][0], acc[3][0], fa, b0, [] {}, prio2);
    bar_g(grp_c, 1);
    (void)TAP;
  };
  // two chunks = 18 K-tiles per trip (nine taps each: the K-tile parity flips from chunk to chunk)
  auto run = [&](auto grp_c) __attribute__((always_inline)) {
    for (int cc = 0; cc < chunks; cc += 2) {
      const int kt = 9 * cc;
      ktile(grp_c, I0{}, HALO_IC(0), 0, cc, kt);
      ktile(grp_c, I1{}, HALO_IC(1), 0, cc, kt + 1);
      ktile(grp_c, I0{}, HALO_IC(2), 0, cc, kt + 2);
      ktile(grp_c, I1{}, HALO_IC(3), 0, cc, kt + 3);
      ktile(grp_c, I0{}, HALO_IC(4), 0, cc, kt + 4);
      ktile(grp_c, I1{}, HALO_IC(5), 0, cc, kt + 5);
      ktile(grp_c, I0{}, HALO_IC(6), 0, cc, kt + 6);
      ktile(grp_c, I1{}, HALO_IC(7), 0, cc, kt + 7);
      ktile(grp_c, I0{}, HALO_IC(8), 0, cc, kt + 8);
      ktile(grp_c, I1{}, HALO_IC(0), 1, cc + 1, kt + 9);
      ktile(grp_c, I0{}, HALO_IC(1), 1, cc + 1, kt + 10);
      ktile(grp_c, I1{}, HALO_IC(2), 1, cc + 1, kt + 11);
      ktile(grp_c, I0{}, HALO_IC(3), 1, cc + 1, kt + 12);
      ktile(grp_c, I1{}, HALO_IC(4), 1, cc + 1, kt + 13);
      ktile(grp_c, I0{}, HALO_IC(5), 1, cc + 1, kt + 14);
      ktile(grp_c, I1{}, HALO_IC(6), 1, cc + 1, kt + 15);
      ktile(grp_c, I0{}, HALO_IC(7), 1, cc + 1, kt + 16);
      ktile(grp_c, I1{}, HALO_IC(8), 1, cc + 1, kt + 17);
    }
  };
  run(std::integral_constant<int, (GROUP == 1 ? 1 : 0)>{});
#undef HALO_IC
  if constexpr (!BAR1) {
    if (group == 0) bar();
  }
  wait_vmcnt<0>();  // (sink writes of the last dummies)
  __syncthreads();
  if (p.out_mul != 1.f) {   // (uniform; the layer handing the fp32 map to the head: conv_ring.h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= p.out_mul;
  }

  // ---- epilogue: as the f16mx ring kernel's (conv_ring.h) — fp32 staging with the chunk swizzle, one
  //      thread per (row, 32-channel group) packs its f16mx line in place, full lines out — except that
  //      a tile row is a pixel (pooled: a quad) of the patch, not a run of the flattened tensor.
  static_assert(MX, "halo kernel epilogue: f16mx");
  constexpr int CPR = G::BN / 4;
  constexpr int ROWB = G::BN * 4;
  constexpr int PASSES = POOL ? 1 : 2;
  constexpr int ROWS = (POOL ? G::BM / 4 : G::BM) / PASSES;
  constexpr int ITEMS = ROWS * (G::BN / 32);
  constexpr int ITERS = ROWS * CPR / 512, BATCH = 8;
  static_assert(ITEMS % 512 == 0 && ITERS % BATCH == 0, "epilogue shape");
  char* obase = reinterpret_cast<char*>(p.out) + (long)n0 * 4;
  const long orow_bytes = (long)p.cout * 4;
  const float floor_v = p.relu ? 0.f : -INFINITY;
  const int Ho = POOL ? (p.H >> 1) : p.H, Wo = POOL ? (p.W >> 1) : p.W;
  const int oy0 = POOL ? (y0 >> 1) : y0, ox0 = POOL ? (x0 >> 1) : x0;
  const int opw = POOL ? (p.PW >> 1) : p.PW;
  const int orows = POOL ? (p.PH * p.PW) >> 2 : p.PH * p.PW;
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
          const int q = (wn * 64 + j * 32 + 8 * g + 4 * half) >> 2;
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
        const int r = PASSES == 1 ? lr : (lr >> 6) * 128 + pass * 64 + (lr & 63);  // tile row = pixel / quad
        const int ry = (int)ring_div_u31((unsigned)r, p.pw_mul, p.pw_sh), rx = r - ry * opw;
        const int oy = oy0 + ry, ox = ox0 + rx;
        if (r < orows && oy < Ho && ox < Wo) {
          const long orow = ((long)img * Ho + oy) * Wo + ox;
          *reinterpret_cast<uint4*>(obase + orow * orow_bytes + (idx % CPR) * 16) = v[u];
        }
      }
    }
    if (pass + 1 < PASSES) __syncthreads();
  }
}

template <bool POOL, int P, int VAR = 0, bool BAR1 = false>
__global__ __launch_bounds__(512) void conv3x3_halo_kernel(HaloParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if constexpr (BAR1) {
    if ((wave >> 2) == 0) conv3x3_halo_body<POOL, P, VAR, true, 0>(p, smem, lane, wave);
    else conv3x3_halo_body<POOL, P, VAR, true, 1>(p, smem, lane, wave);
  } else {
    conv3x3_halo_body<POOL, P, VAR, false, -1>(p, smem, lane, wave);
  }
}

}  // namespace oibl
