// Implicit-GEMM 3x3 convolution with a resident input halo for the 128-OUTPUT-CHANNEL layers (conv2_1, conv2_2;
// ibl/models/vgg.py:40-42, modules 5 and 7), f16mx: 256-pixel x 128-channel tile, FOUR waves, TWO workgroups per CU.
//
// Why.  Cout = 128 leaves the ring kernels no 256-wide tile: their
// 512 x 128 tile stages 80 KB per K-tile through the L2 -> LDS path (64 KB of it pixel rows, re-fetched once per
// tap) for the same 1536 matrix-pipe cycles as a 256 x 256 tile, and runs 2310-2430 cycles per K-tile against
// 1740 — on the fill rate of ~36 B per cycle and CU.  Here
//   * a tile is a PH x PW patch (<= 256 pixels) of one image x 128 output channels, and the (PH + 2) x (PW + 2)
//     halo of ONE 32-channel chunk (<= 344 lines of 128 bytes, 43 KB) is staged once and read by all nine taps
//     (conv_halo.h has the halo image, its swizzle and the tap addressing — shared through halo_patch /
//     HaloParams); per K-tile only 16 KB of weights travel: 20.9 KB per K-tile and workgroup instead of 80;
//   * a workgroup is 4 waves (2 x 2, 128 pixels x 64 channels each — the wave tile of every ring kernel: 4
//     fragment reads of A and 2 of B per 8 accumulator tiles), ONE per SIMD, 79 KB of LDS: TWO workgroups share
//     a CU, each wave with the 256 registers it has in the 8-wave kernels.  The second workgroup is what the
//     second stagger group is in ring_core.h — a wave with operands in registers while the other one loads —
//     without any coupling: no barrier between the two, and the single-buffered halo reload of one
//     workgroup (every 9 K-tiles; it needs a drained queue: conv_halo.h, "fully out-of-range LDS-DMA
//     instructions do not retire in order") runs under the other one's matrix work.  Measured with per-workgroup
//     stamps (tests/gpu_halo4_phase.py, profiles/r05_*): the two workgroups of a CU de-phase by themselves within
//     the first tile — 95-100 % of every epilogue lies inside the other workgroup's K loop — so no start-up
//     stagger is needed (three variants of one were tried: no effect).  What the design does NOT reach is the
//     8-wave kernels' matrix-pipe occupancy inside the K loop: 2260-2660 cycles per K-tile and workgroup with two
//     resident (57-68 % of 2 x 768) against 1660-1740 per 256 x 256 K-tile (88-92 %) — two UNCOORDINATED waves per
//     SIMD do not alternate load and compute segments the way the barrier-staggered groups of ring_core.h do.  Net:
//     conv2_1 0.76 -> 0.71 ms, conv2_2 1.10 -> 1.03 ms at batch 32 (the halved L2 -> LDS traffic pays for it).
//   * with one wave per SIMD and no stagger discipline the weights need TWO barriers per K-tile, not four / eight.
//
// Schedule of K-tile t = tap `tap` of chunk cc (weights: two K-tile buffers of B0 | B1, 64 rows x 128 B each):
//   P0  read A0(tap) from the halo                                             mma  A0 x B0(t)     [B0(t) in registers]
//   P1  vmcnt(2 NB); barrier; read B1(t); issue B0(t+2) over B0(t)             mma  A0 x B1(t)
//   P2  read A1(tap)                                                           mma  A1 x B1(t)
//   P3  vmcnt(2 NB); barrier; read B0(t+1) into the registers B1 vacates;
//       issue B1(t+2) over B1(t); tap 8: issue the next chunk's halo           mma  A1 x B0(t)
//   chunk start (tap 0, not the first): vmcnt(0); barrier   — the halo has landed
// Hazards.  RAW: a weight unit is read behind a barrier that every wave crosses after its own counted wait (the
// unit's NB instructions are the oldest outstanding: 2 NB younger ones in flight).  WAR: B0(t) is last read in
// P3(t-1) and overwritten in P1(t), B1(t) read in P1(t) and overwritten in P3(t) — a barrier in between, crossed
// after the lgkmcnt(0) that closes every mma segment.  The halo is last read in P2 of tap 8 and overwritten behind
// the barrier of P3.  Units beyond the last K-tile load an in-range line into a per-wave sink, so the counts
// are constants.
// -DOIBL_HALO4_CONT (debug library, round 6; VERDICT r05 item 3 "give the two workgroups of a CU a stagger discipline"):
// a CONTINUOUS-ISSUE K loop instead of the phases below.
// Rounds 5's phases were LOAD (ten fragment reads, wait) then MMA (six matrix instructions): with ONE wave per SIMD and
// workgroup nothing covers a wave's own LOAD, and two uncoordinated workgroups per CU reached 57-68 % of the matrix
// pipe inside the K loop (2260-2660 cycles per K-tile against 2 x 768).  Now a wave never stops issuing matrix
// instructions to load: every A fragment is re-read for its NEXT use right behind the last instruction that reads it
// (the order of halo4_mma6 leaves each reload five instructions = 160 pipe cycles before its first reader — the
// discipline of the f16mx stem's consumers), and the B sets are read a whole phase ahead:
//   P0  read B1(t) into the set B0(t-1) vacated                              mma  A0 x B0(t)
//   P1  lgkmcnt(0); barrier SA; issue B0(t+2), B1(t+2) over K-tile t         mma  A0 x B1(t), A0 <- A1(tap) behind its readers
//   P2                                                                        mma  A1 x B1(t)
//   P3  vmcnt(2 NB); barrier SB; read B0(t+1) into the set B1(t) vacated     mma  A1 x B0(t), A1 <- A0(tap + 1) behind its readers
//   chunk start (tap 0, not the first): vmcnt(0); barrier; read A0(tap 0)    — the halo has landed (tap 8 pre-reads
//       nothing: the next chunk's halo is issued behind SB of tap 8 and overwrites the buffer the reads would hit)
// Hazards.  RAW: K-tile t+1's units were issued behind SA(t-1); at SB(t) a wave allows only the 2 NB instructions of
// K-tile t+2 (issued behind SA(t)) in flight, every wave crosses SB behind its own wait: B0(t+1) (read in P3(t)) and
// B1(t+1) (read in P0(t+1)) have landed.  WAR: K-tile t's buffer is last read in P3(t-1) (B0) and P0(t) (B1); both
// reads are retired by the lgkmcnt(0) in front of SA(t) — which P1's first instruction needs anyway — so the units
// of K-tile t+2 issued behind SA(t) overwrite nothing a wave still reads.  Nothing is outstanding at SB (P1's A
// reloads were consumed by P2), so neither barrier exposes an LDS latency.  The halo is last read in P1 of tap 8 (A1)
// and overwritten behind SB of tap 8.  Units beyond the last K-tile load an in-range line into a per-wave sink, so
// the counts are constants.  The MX operand's tail is read as ONE ds_read_b128 ([d4 d5 0 scale], scale = register 7):
// with both workgroups issuing continuously the LDS port is the next limit, and the b64 + b32 tails' bank conflicts
// (12 cycles instead of 4, tools/lds_stem_model.py) were a third of its read cycles.
// Measured (one box, gpurun r06_e; profiles/r06_e_halo4_cont.txt): all 136 tests of test_gpu_mx / splitk / range green,
// K loop + prologue per workgroup 45.5k -> 42.5k shader cycles on conv2_1 and 81.6k -> 79.0k on conv2_2 — and the
// layers' WALL time unchanged (0.721 -> 0.721 ms, 1.010 -> 1.034 ms against an unchanged ring kernel on the same
// boxes): these launches sit at the board's power cap (1.75-1.87 GHz of 2.4), where cycles saved on stalls come back as
// a lower clock — what moves them is fewer joules per tile, not a tighter interleave.  256 VGPRs against
// 239; the pooled variant parks ten values in scratch.  Not adopted: the product keeps the phases.
#pragma once

#include "conv_halo.h"

namespace oibl {

constexpr int H4_WAVES = 4, H4_THREADS = 256;
constexpr int H4_BN = 128, H4_BM = 256;
constexpr int H4_NB = 2;                         // LDS-DMA instructions per wave and weight unit (64 rows)
constexpr int H4_B_UNIT = 64 * 128;              // 8 KB
constexpr int H4_B_TILE = 2 * H4_B_UNIT;         // one K-tile: B0 | B1
constexpr int H4_HALO_SLOTS = (HALO_MAX_POS / 8 + H4_WAVES - 1) / H4_WAVES;   // 11 per wave
constexpr int H4_OFF_B = HALO_BYTES;             // one halo buffer
constexpr int H4_OFF_SINK = H4_OFF_B + 2 * H4_B_TILE;
constexpr int H4_LDS = H4_OFF_SINK + H4_WAVES * 1024;
static_assert(2 * H4_LDS <= 160 * 1024, "two workgroups per CU");

template <bool POOL>
__global__ __launch_bounds__(H4_THREADS, 2) void conv3x3_halo4_kernel(HaloParams p) {
  constexpr int P = RING_MX_EARLY;
  constexpr bool SWAP = !POOL;
  constexpr int NB = H4_NB;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
#ifdef OIBL_DEBUG_HOOKS   // (debug library only: conv_ring.h, wgprof)
  const bool wgprof = p.prof != nullptr && wave == 0 && lane == 0;
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
  xcd_tile(blockIdx.x, (unsigned)p.tiles_m, (unsigned)p.tiles_n, p.raster & 255, tm, tn);
  const unsigned img = ring_div_u31((unsigned)tm, p.img_mul, p.img_sh);
  const unsigned trem = (unsigned)tm - img * (unsigned)(p.tiles_y * p.tiles_x);
  const unsigned tyi = ring_div_u31(trem, p.tx_mul, p.tx_sh);
  const int y0 = (int)tyi * p.PH, x0 = (int)(trem - tyi * (unsigned)p.tiles_x) * p.PW;
  const int n0 = tn * H4_BN;
  const int HP = p.PW + 2;
  const int npos = (p.PH + 2) * HP;
  const int chunks = p.cin >> 5;
  const int nk = 9 * chunks;
  const unsigned pix_bytes = (unsigned)p.cin * 4u;

  const __amdgpu_buffer_rsrc_t rs_in =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.in), 0, (int)p.in_bytes, 0x00020000);
  char* const sink = smem + H4_OFF_SINK + wave * 1024;
  // the whole halo of chunk cc: instruction ii = wave + 4 j fills positions 8 ii .. 8 ii + 7 (per-lane offsets are
  // recomputed here — once per nine K-tiles — instead of living in eleven registers)
  auto stage_halo = [&](int cc) __attribute__((always_inline)) {
    // (a LOOP, not eleven copies: unrolled, the compiler hoists all eleven loop-invariant offsets out of the K loop
    //  into registers it does not have — two of them went to scratch)
#pragma unroll 1
    for (int j = 0; j < H4_HALO_SLOTS; ++j) {
      const int ii = wave + H4_WAVES * j;
      const int pos = ii * 8 + (lane >> 3);
      const int hy = (int)ring_div_u31((unsigned)pos, p.hp_mul, p.hp_sh), hx = pos - hy * HP;
      const int y = y0 - 1 + hy, x = x0 - 1 + hx;
      const bool ok = pos < npos && y >= 0 && y < p.H && x >= 0 && x < p.W;
      const int f = ((hx >> 1) & 7) ^ ((hy & 1) << 2);
      const unsigned voff = ok ? ((img * (unsigned)p.H + (unsigned)y) * (unsigned)p.W + (unsigned)x) * pix_bytes +
                                     (unsigned)(((lane & 7) ^ f) << 4)
                               : RG_OOB;
      if (ii * 8 < npos) buf_glds16(rs_in, voff, (unsigned)cc * 128u, smem + ii * 1024);   // (wave-uniform)
    }
  };

  // ---- weights, K order (chunk, tap): K-tile (cc, tap) at tap * tap_stride + cc * 128
  const __amdgpu_buffer_rsrc_t rs_w =
      __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  const unsigned tap_stride = (unsigned)p.cout * pix_bytes;
  unsigned woff[2 * NB];
  {
    const int piece = ring_piece_mxb(wave, lane);   // (row >> 1) & 7 of unit row 8 (wave + 4 i) + (lane >> 3)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i = 0; i < NB; ++i) {
        const int u = 8 * (wave + H4_WAVES * i) + (lane >> 3);          // row of the 64-row unit
        const int row = (u >> 5) * 64 + h * 32 + (u & 31);              // output channel inside the tile
        woff[NB * h + i] = (unsigned)(n0 + row) * pix_bytes + (unsigned)piece;
      }
  }
  unsigned wsoff = 0u - tap_stride;   // cursor of the next K-tile to stage; begin_tile() moves it
  int win = -1;
  auto begin_tile = [&]() __attribute__((always_inline)) {
    const bool wrap = win == 8;
    win = wrap ? 0 : win + 1;
    wsoff += wrap ? (128u - 8u * tap_stride) : tap_stride;
  };
  char* const st_b = smem + H4_OFF_B + wave * 1024;
  // (branch-free: a unit beyond the last K-tile becomes an in-range load into the sink by SELECTING offsets and
  //  destination — with `if (real)` every LDS-DMA instruction of the loop sat behind its own scalar branch)
  const unsigned sink_voff = (unsigned)(lane * 16);
  auto stage_b = [&](int buf, int h, bool real) __attribute__((always_inline)) {
    const unsigned so = real ? wsoff : 0u;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      char* const dst = st_b + buf * H4_B_TILE + h * H4_B_UNIT + i * (H4_WAVES * 1024);
      buf_glds16(rs_w, real ? woff[NB * h + i] : sink_voff, so, real ? dst : sink);   // in range: conv_halo.h, the waits
    }
  };

  // ---- fragment addresses (conv_halo.h)
  int frag_off[4];
  {
    const int row = lane & 31, half = lane >> 5, swz = (lane >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = row * 128 + (((2 * kk + half) ^ swz) * 16);
  }
  const char* const rd_b = smem + H4_OFF_B + wn * 4096;
  int pre[2][2][3];
  {
    const int half = lane >> 5;
    const int npix = p.PH * p.PW;
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) {
        int r = wm * 128 + h * 64 + i2 * 32 + (lane & 31);
        if (r >= npix) r = 0;
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
          const int hx = px + dxi;
          const int t = ((hx >> 1) & 7) ^ half ^ (((py + 1) & 1) << 2);
          pre[h][i2][dxi] = lb0 + (t << 4);
        }
      }
  }
  const int row_pitch = HP * 128;

#ifdef OIBL_HALO4_CONT
  bf16x8_t fa[2][4], fbx[4], fby[4];
  // (every piece of a fragment — fp16 k-halves, e2m3 dwords 0-3, the tail slot [d4 d5 0 scale] — is one 16-byte slot)
  auto ld_frag = [&](const char* a) __attribute__((always_inline)) -> bf16x8_t {
    return *reinterpret_cast<const bf16x8_t*>(a);
  };
  // the two row blocks' base addresses of A(h) at tap: fragment kk of block i2 is at a0[i2] ^ ((kk << 5) ^ cdy)
  int a_cur[2];
  auto a_base = [&](auto h_c, auto tap_c) __attribute__((always_inline)) {
    constexpr int h = decltype(h_c)::value, tap = decltype(tap_c)::value;
    constexpr int dyi = tap / 3, dxi = tap % 3;
    int rp_ = row_pitch;
    asm volatile("" : "+s"(rp_));     // (opaque: keeps 9 taps x 16 loop-invariant addresses out of scratch)
    const int tapoff = dyi * rp_ + dxi * 128;
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
      a_cur[i2] = pre[h][i2][dxi];
      asm volatile("" : "+v"(a_cur[i2]));
      a_cur[i2] += tapoff;
    }
  };
  auto ld_a = [&](auto tap_c, int i2, int kk) __attribute__((always_inline)) {
    constexpr int tap = decltype(tap_c)::value;
    constexpr int cdy = (tap / 3 != 1) ? 64 : 0;
    fa[i2][kk] = ld_frag(smem + (a_cur[i2] ^ ((kk << 5) ^ cdy)));
  };
  auto read_a = [&](auto h_c, auto tap_c) __attribute__((always_inline)) {   // all of A(h) at once (prologue, chunk starts)
    a_base(h_c, tap_c);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) ld_a(tap_c, i2, kk);
  };
  auto read_b = [&](int buf, int h, bf16x8_t (&f)[4]) __attribute__((always_inline)) {
    const char* s = rd_b + buf * H4_B_TILE + h * H4_B_UNIT;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f[kk] = ld_frag(s + frag_off[kk]);
  };
  // one phase: six matrix instructions on two accumulator tiles; after(q) runs right behind instruction q (the
  // reloads of the fragments it was the last reader of).  sched_barrier pins the written order: left alone, the
  // scheduler sinks every reload to just in front of its use (conv.hip, the f16mx stem's consumers).
  auto mma6 = [&](f32x16_t& acc0, f32x16_t& acc1, const bf16x8_t (&fb)[4], auto&& after) __attribute__((always_inline)) {
    typedef __attribute__((ext_vector_type(4))) int i4;
    auto f16 = [&](f32x16_t& acc, int i2, int k) __attribute__((always_inline)) {
      const f16x8_t a = __builtin_bit_cast(f16x8_t, fa[i2][k]), b = __builtin_bit_cast(f16x8_t, fb[k]);
      acc = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc, 0, 0, 0)
                 : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
    };
    auto mx = [&](f32x16_t& acc, int i2) __attribute__((always_inline)) {
      // (the tail slot's zero dword is never read: without a use the allocator hands that register to a temporary
      //  while the ds_read_b128 that writes it is still in flight — and protects the temporary with an lgkmcnt(0) in
      //  front of the phase's first matrix instruction, i.e. the exposed LOAD this schedule is there to remove)
      asm volatile("" ::"v"(fa[i2][3]), "v"(fb[3]));
      const i32x8_t a8 = __builtin_shufflevector(__builtin_bit_cast(i4, fa[i2][2]), __builtin_bit_cast(i4, fa[i2][3]),
                                                 0, 1, 2, 3, 4, 5, 6, 7);
      const i32x8_t b8 = __builtin_shufflevector(__builtin_bit_cast(i4, fb[2]), __builtin_bit_cast(i4, fb[3]), 0, 1, 2,
                                                 3, 4, 5, 6, 7);
      // e2m3 x e2m3 (cbsz = blgp = 2); scales: byte 0 of register 7 of either operand (the tail slot's last dword)
      acc = SWAP ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, acc, 2, 2, 0, b8[7], 0, a8[7])
                 : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc, 2, 2, 0, a8[7], 0, b8[7]);
    };
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
    f16(acc0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    after(0);
    __builtin_amdgcn_sched_barrier(0);
    f16(acc1, 1, 0);
    __builtin_amdgcn_sched_barrier(0);
    after(1);
    __builtin_amdgcn_sched_barrier(0);
    f16(acc0, 0, 1);
    __builtin_amdgcn_sched_barrier(0);
    after(2);
    __builtin_amdgcn_sched_barrier(0);
    f16(acc1, 1, 1);
    __builtin_amdgcn_sched_barrier(0);
    after(3);
    __builtin_amdgcn_sched_barrier(0);
    mx(acc0, 0);
    __builtin_amdgcn_sched_barrier(0);
    after(4);
    __builtin_amdgcn_sched_barrier(0);
    mx(acc1, 1);
    __builtin_amdgcn_sched_barrier(0);
    after(5);
    asm volatile("" : "+v"(acc0), "+v"(acc1));  // pin the results inside the segment (ring_core.h)
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  // the reload of A behind its readers: instruction q was the last reader of fa[q & 1][q >> 1] (q < 4), of
  // fa[0][2..3] (q = 4), of fa[1][2..3] (q = 5)
  auto reload_a = [&](auto tap_c, int q) __attribute__((always_inline)) {
    if (q < 4) {
      ld_a(tap_c, q & 1, q >> 1);
    } else {
      ld_a(tap_c, q - 4, 2);
      ld_a(tap_c, q - 4, 3);
    }
  };
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // accumulators start at the bias (conv_halo.h)
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
  // ---- prologue: the halo of chunk 0, the weights of K-tiles 0 and 1; everything landed; B0(0) and A0(tap 0) read
  stage_halo(0);
  begin_tile();
  stage_b(0, 0, true);
  stage_b(0, 1, true);
  begin_tile();
  stage_b(1, 0, true);
  stage_b(1, 1, true);
  wait_vmcnt<0>();
  bar();
  read_b(0, 0, fbx);
  read_a(I0{}, std::integral_constant<int, 0>{});

#define H4_IC(x) std::integral_constant<int, (x)> {}
  auto ktile = [&](auto par_c, auto tap_c, int cc, int kt) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr int TAP = decltype(tap_c)::value;
    constexpr int NXT = TAP == 8 ? 0 : TAP + 1;
    bf16x8_t(&b0)[4] = PAR ? fby : fbx;
    bf16x8_t(&b1)[4] = PAR ? fbx : fby;
    const bool more = kt + 2 < nk;
    if constexpr (TAP == 0) {
      if (kt > 0) {          // a new chunk: its halo (issued behind SB of the previous tap 8) has landed
        wait_vmcnt<0>();
        bar();
        read_a(I0{}, tap_c);
      }
    }
    // P0: A0 x B0; B1(t) arrives under it
    read_b(PAR, 1, b1);
    mma6(acc[0][0], acc[1][0], b0, [&](int q) __attribute__((always_inline)) {
      if (q == 1) begin_tile();   // (scalar cursor of the K-tile staged in P1, in the shadow of the matrix pipe)
    });
    // P1: A0 x B1; A1(tap) behind A0's readers.  SA: every wave's reads of K-tile t's buffer are retired
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    bar();
    stage_b(PAR, 0, more);    // K-tile t + 2 over K-tile t
    stage_b(PAR, 1, more);
    a_base(I1{}, tap_c);
    mma6(acc[0][1], acc[1][1], b1, [&](int q) __attribute__((always_inline)) { reload_a(tap_c, q); });
    // P2: A1 x B1
    mma6(acc[2][1], acc[3][1], b1, [](int) __attribute__((always_inline)) {});
    // P3: A1 x B0; B0(t+1) into the set B1 vacated; A0(tap + 1) behind A1's readers.  SB: K-tile t+1 has landed
    wait_vmcnt<2 * NB>();
    bar();
    read_b(PAR ^ 1, 0, b1);
    if constexpr (TAP == 8) {
      if (cc + 1 < chunks) stage_halo(cc + 1);     // (over the halo A1(tap 8) was the last to read: no pre-read of A0)
      mma6(acc[2][0], acc[3][0], b0, [](int) __attribute__((always_inline)) {});
    } else {
      a_base(I0{}, H4_IC(NXT));
      mma6(acc[2][0], acc[3][0], b0, [&](int q) __attribute__((always_inline)) { reload_a(H4_IC(NXT), q); });
    }
  };
  for (int cc = 0; cc < chunks; cc += 2) {
    const int kt = 9 * cc;
    ktile(I0{}, H4_IC(0), cc, kt);
    ktile(I1{}, H4_IC(1), cc, kt + 1);
    ktile(I0{}, H4_IC(2), cc, kt + 2);
    ktile(I1{}, H4_IC(3), cc, kt + 3);
    ktile(I0{}, H4_IC(4), cc, kt + 4);
    ktile(I1{}, H4_IC(5), cc, kt + 5);
    ktile(I0{}, H4_IC(6), cc, kt + 6);
    ktile(I1{}, H4_IC(7), cc, kt + 7);
    ktile(I0{}, H4_IC(8), cc, kt + 8);
    ktile(I1{}, H4_IC(0), cc + 1, kt + 9);
    ktile(I0{}, H4_IC(1), cc + 1, kt + 10);
    ktile(I1{}, H4_IC(2), cc + 1, kt + 11);
    ktile(I0{}, H4_IC(3), cc + 1, kt + 12);
    ktile(I1{}, H4_IC(4), cc + 1, kt + 13);
    ktile(I0{}, H4_IC(5), cc + 1, kt + 14);
    ktile(I1{}, H4_IC(6), cc + 1, kt + 15);
    ktile(I0{}, H4_IC(7), cc + 1, kt + 16);
    ktile(I1{}, H4_IC(8), cc + 1, kt + 17);
  }
#undef H4_IC
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // (the last P3's pre-read of a K-tile that does not exist)
#else
  bf16x8_t fa[2][4], fbx[4], fby[4];
  auto ld_frag = [&](const char* a, int kk) __attribute__((always_inline)) -> bf16x8_t {
    if (kk == 3) {
      typedef __attribute__((ext_vector_type(2))) unsigned u2;   // (not uint2: ring_core.h, read_frag)
      const u2 d = *reinterpret_cast<const u2*>(a);
      const unsigned sc = *reinterpret_cast<const unsigned*>(a + 12);
      typedef __attribute__((ext_vector_type(4))) unsigned u4;
      return __builtin_bit_cast(bf16x8_t, (u4){d.x, d.y, sc, 0u});
    }
    return *reinterpret_cast<const bf16x8_t*>(a);
  };
  auto read_a = [&](auto h_c, auto tap_c) __attribute__((always_inline)) {
    constexpr int h = decltype(h_c)::value, tap = decltype(tap_c)::value;
    constexpr int dyi = tap / 3, dxi = tap % 3;
    constexpr int cdy = (dyi != 1) ? 64 : 0;
    int rp_ = row_pitch;
    asm volatile("" : "+s"(rp_));     // (opaque: keeps 9 taps x 16 loop-invariant addresses out of scratch)
    const int tapoff = dyi * rp_ + dxi * 128;
    int a0[2];
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2) {
      a0[i2] = pre[h][i2][dxi];
      asm volatile("" : "+v"(a0[i2]));
      a0[i2] += tapoff;
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) fa[i2][kk] = ld_frag(smem + (a0[i2] ^ ((kk << 5) ^ cdy)), kk);
  };
  auto read_b = [&](int buf, int h, bf16x8_t (&f)[4]) __attribute__((always_inline)) {
    const char* s = rd_b + buf * H4_B_TILE + h * H4_B_UNIT;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f[kk] = ld_frag(s + frag_off[kk], kk);
  };
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };

  // accumulators start at the bias (conv_halo.h)
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
  stage_halo(0);
  begin_tile();
  stage_b(0, 0, true);
  stage_b(0, 1, true);
  begin_tile();
  stage_b(1, 0, true);
  stage_b(1, 1, true);
  wait_vmcnt<0>();
  bar();
  read_b(0, 0, fbx);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

#define H4_IC(x) std::integral_constant<int, (x)> {}
  auto ktile = [&](auto par_c, auto tap_c, int cc, int kt) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr int TAP = decltype(tap_c)::value;
    bf16x8_t(&b0)[4] = PAR ? fby : fbx;
    bf16x8_t(&b1)[4] = PAR ? fbx : fby;
    const bool more = kt + 2 < nk;
    if constexpr (TAP == 0) {
      if (kt > 0) {          // a new chunk: its halo (issued in P3 of the previous tap 8) has landed
        wait_vmcnt<0>();
        bar();
      }
    }
    // P0: A0 x B0
    read_a(I0{}, tap_c);
    halo_phase_mma<P, SWAP>(acc[0][0], acc[1][0], fa, b0, [&] { begin_tile(); });
    // P1: A0 x B1
    wait_vmcnt<2 * NB>();
    bar();
    read_b(PAR, 1, b1);
    stage_b(PAR, 0, more);    // B0(t+2)
    halo_phase_mma<P, SWAP>(acc[0][1], acc[1][1], fa, b1, [] {});
    // P2: A1 x B1
    read_a(I1{}, tap_c);
    halo_phase_mma<P, SWAP>(acc[2][1], acc[3][1], fa, b1, [] {});
    // P3: A1 x B0   (B0 of the next K-tile goes into the register set B1 just vacated)
    wait_vmcnt<2 * NB>();
    bar();
    read_b(PAR ^ 1, 0, b1);
    stage_b(PAR, 1, more);    // B1(t+2)
    if constexpr (TAP == 8) {
      if (cc + 1 < chunks) stage_halo(cc + 1);
    }
    halo_phase_mma<P, SWAP>(acc[2][0], acc[3][0], fa, b0, [] {});
  };
  for (int cc = 0; cc < chunks; cc += 2) {
    const int kt = 9 * cc;
    ktile(I0{}, H4_IC(0), cc, kt);
    ktile(I1{}, H4_IC(1), cc, kt + 1);
    ktile(I0{}, H4_IC(2), cc, kt + 2);
    ktile(I1{}, H4_IC(3), cc, kt + 3);
    ktile(I0{}, H4_IC(4), cc, kt + 4);
    ktile(I1{}, H4_IC(5), cc, kt + 5);
    ktile(I0{}, H4_IC(6), cc, kt + 6);
    ktile(I1{}, H4_IC(7), cc, kt + 7);
    ktile(I0{}, H4_IC(8), cc, kt + 8);
    ktile(I1{}, H4_IC(0), cc + 1, kt + 9);
    ktile(I0{}, H4_IC(1), cc + 1, kt + 10);
    ktile(I1{}, H4_IC(2), cc + 1, kt + 11);
    ktile(I0{}, H4_IC(3), cc + 1, kt + 12);
    ktile(I1{}, H4_IC(4), cc + 1, kt + 13);
    ktile(I0{}, H4_IC(5), cc + 1, kt + 14);
    ktile(I1{}, H4_IC(6), cc + 1, kt + 15);
    ktile(I0{}, H4_IC(7), cc + 1, kt + 16);
    ktile(I1{}, H4_IC(8), cc + 1, kt + 17);
  }
#undef H4_IC
#endif
  wait_vmcnt<0>();  // (sink writes of the last dummies)
  __syncthreads();
  if (wgprof) p.prof[64 + 4 * (size_t)blockIdx.x + 3] = __builtin_amdgcn_s_memtime();
  if (p.out_mul != 1.f) {   // (uniform; the layer handing the fp32 map to the head — only when the "every layer on
                            //  this kernel" experiment, hook 13, sends conv5_3 here: conv_halo.h)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] *= p.out_mul;
  }

  // ---- epilogue: conv_halo.h's — fp32 staging with the chunk swizzle, one thread per (row, 32-channel group) packs
  //      its f16mx line in place, full lines out — for 128 channels and 256 threads
  constexpr int CPR = H4_BN / 4;
  constexpr int ROWB = H4_BN * 4;
  constexpr int PASSES = POOL ? 1 : 2;
  constexpr int ROWS = (POOL ? H4_BM / 4 : H4_BM) / PASSES;
  constexpr int ITEMS = ROWS * (H4_BN / 32);
  constexpr int ITERS = ROWS * CPR / H4_THREADS, BATCH = 8;
  static_assert(ITEMS % H4_THREADS == 0 && ITERS % BATCH == 0 && ROWS * ROWB <= H4_LDS, "epilogue shape");
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
      for (int it = 0; it < ITEMS / H4_THREADS; ++it) {
        const int item = it * H4_THREADS + (int)threadIdx.x;
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
        const int idx = (it0 + u) * H4_THREADS + (int)threadIdx.x;
        const int lr = idx / CPR, q = idx % CPR;
        v[u] = *reinterpret_cast<const uint4*>(smem + lr * ROWB + ((q ^ (lr & (CPR - 1))) << 4));
      }
#pragma unroll
      for (int u = 0; u < BATCH; ++u) {
        const int idx = (it0 + u) * H4_THREADS + (int)threadIdx.x;
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
  if (wgprof) p.prof[64 + 4 * (size_t)blockIdx.x + 2] = __builtin_amdgcn_s_memtime();
}

}  // namespace oibl
