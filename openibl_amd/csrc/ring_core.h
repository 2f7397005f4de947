// "Ring" schedule for bf16 NT contractions on gfx950:  C[m][n] += sum_k A[m][k] * B[n][k], one
// K-tile = 64 bf16 = one 128-byte line per row, 8 waves, 128 x 64 outputs per wave.
// Users: the implicit-GEMM convolutions (conv_ring.h) and the query x gallery distance kernels
// (match.hip); they differ in the Loader of each operand and in the epilogue.
//
// The generic core (gemm_core.h) drains its LDS-DMA queue (vmcnt(0)) and crosses one workgroup
// barrier per K-tile, so the whole workgroup waits for the slowest line of every tile.  Here
//
//   * a K-tile is staged as four UNITS (A0, A1: the two 64-row halves of every wave row's 128
//     rows; B0, B1: the two 32-column halves of every wave column's 64 columns), and LDS holds two
//     K-tiles.  A unit is re-filled two phases after its last fragment read, i.e. with the data of
//     K-tile t + 2, and is read six phases after it was issued: at any time five units are in
//     flight, waited for with a COUNTED s_waitcnt vmcnt(N) — the queue is never drained inside
//     the loop;
//   * a K-tile is four PHASES, one per 64 x 32 quadrant of the wave's 128 x 64 accumulator
//     (8 MFMAs 32x32x16 each).  Every phase is a LOAD segment (fragment ds_reads of the operand
//     half that changes: 8, 4, 8, 4 reads; the LDS-DMA instructions of one unit; the counted wait)
//     and a COMPUTE segment (8 MFMAs behind counted lgkmcnt waits), separated by raw s_barriers;
//   * the two stagger groups (waves 0-3 / 4-7: one wave of each on every SIMD) run ONE BARRIER
//     APART: while one wave of a SIMD is in its COMPUTE segment the other is in its LOAD segment,
//     so the matrix pipe of the SIMD always has a wave with operands in registers (s_setprio 1
//     around the MFMAs lets it win issue arbitration against the loading partner).
//   * BAR1 (round 4): ONE barrier per phase and wave instead of two.  Every wave runs the same sequence
//     L(0) C(0) L(1) C(1) ...; group 0 crosses its barrier between L(p) and C(p), group 1 between C(p) and
//     L(p+1).  Barrier interval n then holds  group 0: C(n-1) L(n)   |   group 1: L(n) C(n)  — the two
//     COMPUTE segments of a SIMD run back to back on its matrix pipe (group 0 at the higher priority: its
//     LOAD is still to come), each LOAD segment in the shadow of the partner's COMPUTE, and the barrier
//     latency + arrival skew is paid once per 2 x COMPUTE instead of once per COMPUTE (with 6 MFMAs =
//     192 pipe cycles per f16mx phase the two-barrier interval measured 245-277 cycles).  Hazards, with
//     intervals counted like phases: the fragment reads of phase r are retired inside C(r) (counted
//     waits in front of the MFMAs that use them, lgkmcnt(0) behind the last one): in interval r (group 1) or at the head of interval r+1 (group 0) — a unit re-staged in
//     L(q), q >= r+2, is issued in interval >= r+2 by either group: WAR as before.  A wait in L(w) is
//     followed by barrier w in both groups (directly in group 0, behind C(w) in group 1); the reads of
//     L(w+1) come after barrier w in both: RAW as before (read >= 1 phase after the retiring wait).
//
// Hazard rules (cdna_hip_programming.md, "256^2 8-phase template"), with phases numbered globally:
//   RAW  a unit is read in phase >= w + 1 where w is the phase whose LOAD segment holds the
//        vmcnt that retires it (own loads) and whose closing barriers make the other waves' loads
//        visible;  here w = read - 1 and the wait after the phase's own issues leaves exactly the
//        5 youngest units outstanding.
//   WAR  a unit is re-staged in phase >= r + 2 where r is the last phase that reads it (the
//        lagging group retires those reads after the barrier that ends phase r).
// Unit schedule for K-tile t (phases 4t .. 4t+3), reads / (re)stages:
//   P0: read A0(t)            stage A1(t+1)          P1: read B1(t)       stage B0(t+2)
//   P2: read A1(t)            stage A0(t+2)          P3: read B0(t+1)     stage B1(t+2)
// B0 lives in one of two fragment register sets (X/Y) that swap roles every K-tile, so that the
// next tile's B0 can be fetched during P3 while the current B0 is still being multiplied.
//
// LDS image of a unit: rows of 128 B, 16-B slots XOR-swizzled by ((row >> 1) & 7) — applied on the
// SOURCE side of the LDS-DMA (the destination is wave-linear) and on the fragment read, so every
// ds_read_b128 lane group touches 16 distinct bank slots (same image as gemm_core.h).
//
// Loader concept (one object per operand, each with its own K cursor):
//   void begin_tile();                 advance the cursor to the next K-tile; called once per
//                                      K-tile and operand, before the first stage() of that tile
//   void stage(int h, char* dst, int i0, int i1);   issue LDS-DMA instructions i0 <= i < i1 (of N)
//                                      of half h of the cursor's K-tile; instruction i fills
//                                      dst + i * 8192 (wave-relative)
// Both operands are fetched with buffer_load_dwordx4 ... lds (16 B per lane); an offset beyond the
// descriptor's num_records returns zeros, which is how the convolution pads.
#pragma once

#include <type_traits>

#include "gemm_core.h"

namespace oibl {

constexpr unsigned RG_OOB = 0xF0000000u;  // voffset that is out of range for every descriptor here

// operand arithmetic of an instantiation (template argument P; false / true of the older bool
// parameter convert to RING_BF16 / RING_X3)
constexpr int RING_F16 = -1;  // fp16 rows: the bf16 stream with v_mfma_f32_32x32x16_f16 (match.hip: the filter pass
                              // of the fp16-filter + exact-rescore top-k; every `P >= ...` test below reads it as bf16)
constexpr int RING_BF16 = 0;  // bf16 rows, 64 K per 128-byte K-tile, 8 MFMAs per phase
constexpr int RING_X3 = 1;    // bf16x3 rows ([32 hi | 32 lo]), 32 K per K-tile, 12 MFMAs per phase
constexpr int RING_MX = 2;    // f16mx rows (common.h), 32 K per K-tile, 4 f16 + 2 MX-fp6 MFMAs per phase;
                              // the LDS-DMA instructions of a phase are issued inside its COMPUTE segment
constexpr int RING_MX_EARLY = 3;  // the same with the LDS-DMA issue in the LOAD segment (as bf16 / bf16x3)
// timing experiments on the RING_MX_EARLY stream (WRONG results): one ingredient of the loop removed
constexpr int RING_MX_NOMFMA = 4, RING_MX_NODMA = 5, RING_MX_NOREAD = 6, RING_MX_NOBAR = 7;
// RING_MX_EARLY with shader-clock stamps of phases P0 / P1 of the last steady-state K-tile (diagnostic)
constexpr int RING_MX_PROF = 8;

// Geometry of one instantiation.  WM = wave rows (2 or 4); the 8 waves form a WM x (8 / WM) grid,
// every wave owns 128 x 64 outputs, so the tile is 256 x 256 (WM = 2) or 512 x 128 (WM = 4).
// Stagger group of a wave = wave >> 2 (waves w and w + 4 share a SIMD).
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

// Tile row of the unit row this lane fetches with LDS-DMA instruction i of half h.
//   unit row u = 8 * (wave + 8 i) + (lane >> 3);  A: wave row u >> 6, B: wave column u >> 5.
template <int WM>
__device__ static inline int ring_a_row(int wave, int lane, int h, int i) {
  const int u = 8 * (wave + 8 * i) + (lane >> 3);
  return (u >> 6) * 128 + h * 64 + (u & 63);
}
template <int WM>
__device__ static inline int ring_b_row(int wave, int lane, int h, int i) {
  const int u = 8 * (wave + 8 * i) + (lane >> 3);
  return (u >> 5) * 64 + h * 32 + (u & 31);
}
// Byte offset of this lane's (swizzled) 16-B piece inside its row's 128-byte K segment.
__device__ static inline int ring_piece(int wave, int lane) {
  return ((lane & 7) ^ (4 * (wave & 1) + (lane >> 4))) * 16;
}

// f16mx, B operand: the same LDS position receives the line's slot s ^ 1 for s >= 4 (q6(hi) <-> q6(lo))
__device__ static inline int ring_piece_mxb(int wave, int lane) {
  const int s = (lane & 7) ^ (4 * (wave & 1) + (lane >> 4));
  return (s >= 4 ? s ^ 1 : s) * 16;
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

// Plain row-major operand: row r of the tile is `base + min(row0 + r, nrows - 1) * ld_bytes`
// (clamped so that partial tiles never read out of bounds; their results are discarded).
template <int N_INSTR>
struct RingRowLoader {
  __amdgpu_buffer_rsrc_t rsrc;
  unsigned voff[2 * N_INSTR];
  unsigned soff;
  __device__ inline void init(const void* base, unsigned bytes, long row0, long nrows, long ld_bytes,
                              const int (&tile_row)[2 * N_INSTR], int piece) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
#pragma unroll
    for (int j = 0; j < 2 * N_INSTR; ++j) {
      long r = row0 + tile_row[j];
      if (r > nrows - 1) r = nrows - 1;
      voff[j] = (unsigned)(r * ld_bytes) + piece;
    }
    soff = 0u - 128u;
  }
  __device__ inline void begin_tile() { soff += 128u; }
  __device__ inline void stage(int h, char* dst, int i0 = 0, int i1 = N_INSTR) const {
#pragma unroll
    for (int i = 0; i < N_INSTR; ++i)
      if (i >= i0 && i < i1) buf_glds16(rsrc, voff[N_INSTR * h + i], soff, dst + i * 8192);
  }
};

// The main loop.  acc[i][j]: 32x32 tile (row tile i of 4, column tile j of 2) of this wave's
// 128 x 64 block, in the MFMA C/D layout.  nsteps >= 3; ODD = nsteps is odd.
// SWAP = true feeds the B fragment as the MFMA's first operand: every accumulator tile is then
// held TRANSPOSED (register index <-> column of the block, lane <-> row), i.e. a lane owns 4
// consecutive columns of one row per register quad — what an epilogue that writes row-major
// 16-bit outputs wants (8-byte stores instead of 2-byte ones).  Each output element is the same
// k-ordered fma chain either way: identical bits.
// P = RING_X3: bf16x3 operands (common.h): a K-tile row is [32 hi | 32 lo], so the four fragments of a
// row are hi[0:16], hi[16:32], lo[0:16], lo[16:32] and a phase multiplies lo.hi + hi.lo + hi.hi per
// 16-wide half: 12 MFMAs per phase instead of 8 on the same LDS traffic.
// P = RING_MX: f16mx operands (common.h): fragments 0, 1 are hi[0:16], hi[16:32] in fp16, fragments
// 2, 3 together are one MX operand (6 dwords of e2m3 + the scale byte in dword 6) — q6(hi) in the
// lower lane half and q6(lo) in the upper one on the A side, the other way round on the B side: the B
// LOADER stages slots 4..7 of every line pairwise exchanged (ring_piece_mxb: the exchange sits in the
// per-lane source offset of the LDS-DMA, fixed for the kernel) — so that ONE K = 64 MX instruction adds
// both cross terms: per 32x32 tile and K-tile 2 f16 MFMAs + 1 MX MFMA, 6 per phase, on the same LDS
// traffic and with the same fragment addresses for both operands.
// On return every wave has passed a workgroup barrier: the staging LDS is free.
// GROUP (BAR1 only): the stagger group of the calling wave as a compile-time constant — the caller branches
// ONCE on wave >> 2 into one of two copies of the loop, so that which barrier a wave crosses and at which
// priority it computes cost no instructions inside the loop (with run-time tests — four branches and ~10 scalar
// instructions per phase — the one-barrier schedule measured 3-8 % SLOWER than the two-barrier one).
template <int WM, bool ODD, bool SWAP, int P = RING_BF16, bool BAR1 = false, int GROUP = -1, typename LA, typename LB>
__device__ static inline void ring_mainloop(f32x16_t (&acc)[4][2], char* smem, int wave, int lane,
                                            LA& la, LB& lb, int nsteps, unsigned long long* stamps = nullptr) {
  using G = RingGeo<WM>;
  constexpr int NA = G::NA, NB = G::NB;
  constexpr int OFF_A0 = 0, OFF_A1 = G::A_UNIT, OFF_B0 = 2 * G::A_UNIT,
                OFF_B1 = 2 * G::A_UNIT + G::B_UNIT;
  const int wm = wave / G::WN, wn = wave % G::WN;
  const int group = wave >> 2;
  static_assert(!BAR1 || GROUP == 0 || GROUP == 1, "BAR1: the caller fixes the stagger group");

  char* const st_base = smem + wave * 1024;
  // SPLIT (512 x 128 tile: an A unit is 4 instructions, a B unit 1): the A unit is issued in two
  // halves, the second one phase later together with the B unit, so that every phase carries 2-3
  // LDS-DMA instructions instead of 4, 1, 4, 1 — the LOAD segments of the A phases were longer
  // than a COMPUTE segment.
  constexpr bool SPLIT = NA >= 4 * NB;
  auto stage_a = [&](int buf, int h, int part) __attribute__((always_inline)) {
    if constexpr (P == RING_MX_NODMA) return;
    char* d = st_base + buf * G::TILE + (h ? OFF_A1 : OFF_A0);
    if constexpr (SPLIT) la.stage(h, d, part * (NA / 2), (part + 1) * (NA / 2));
    else if (part == 0) la.stage(h, d, 0, NA);
  };
  auto stage_b = [&](int buf, int h) __attribute__((always_inline)) {
    if constexpr (P == RING_MX_NODMA) return;
    lb.stage(h, st_base + buf * G::TILE + (h ? OFF_B1 : OFF_B0));
  };

  constexpr bool X3 = P == RING_X3, MX = P >= RING_MX;   // (every code >= RING_MX is an f16mx stream)
  // LATE: with 6 MFMAs (192 cycles) per phase the LOAD segment (fragment reads + 2-4 LDS-DMA issues at
  // 100-185 cycles each next to the reads) is longer than the COMPUTE segment it is paired with, so the
  // DMA issue moves into COMPUTE, between the MFMAs (~60 cycles each there); every counted wait then
  // sits BEFORE its phase's issues and allows that many fewer instructions in flight.
  constexpr bool LATE = P == RING_MX;
  constexpr bool PROF = P == RING_MX_PROF;
  unsigned long long st_[14] = {};
#define RING_STAMP(i) do { if constexpr (PROF && TAIL == 0) st_[i] = __builtin_amdgcn_s_memtime(); } while (0)
  int frag_off[4];
  {
    const int row = lane & 31, half = lane >> 5, swz = (lane >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = row * 128 + (((2 * kk + half) ^ swz) * 16);
  }
  const char* const rd_a = smem + wm * 8192;  // + buf * TILE + OFF_A{h} + i2 * 4096
  const char* const rd_b = smem + wn * 4096;  // + buf * TILE + OFF_B{h}

  bf16x8_t fa[2][4], fbx[4], fby[4];
  // One fragment.  MX, fragment 3 (the tail of the MX operand): 8 data bytes + the scale byte are
  // fetched as ds_read_b64 + ds_read_b32 instead of one ds_read_b128 — the same LDS cycles, and the
  // 6-register operand of the MX instruction is then the register sequence {b128, b64} of two loads
  // (coalesced by the allocator); with a b128 tail the operand had to be re-assembled by copies into
  // fresh registers, ~20 VGPRs the distance kernels do not have.
  auto read_frag = [&](const char* s, int kk) __attribute__((always_inline)) -> bf16x8_t {
#ifndef OIBL_MX_TAIL_B128
    if constexpr (MX) {
      if (kk == 3) {
        // (an ext_vector load, not HIP's uint2 struct: a struct load carries no TBAA, and the compiler
        //  drains the LDS-DMA queue — s_waitcnt vmcnt(0) — in front of every LDS read without it)
        typedef __attribute__((ext_vector_type(2))) unsigned u2;
        const u2 d = *reinterpret_cast<const u2*>(s + frag_off[3]);
        const unsigned sc = *reinterpret_cast<const unsigned*>(s + frag_off[3] + 12);
        typedef __attribute__((ext_vector_type(4))) unsigned u4;
        return __builtin_bit_cast(bf16x8_t, (u4){d.x, d.y, sc, 0u});
      }
    }
#endif
    return *reinterpret_cast<const bf16x8_t*>(s + frag_off[kk]);
  };
  auto read_a = [&](int buf, int h) __attribute__((always_inline)) {
    if constexpr (P == RING_MX_NOREAD) return;
    const char* s = rd_a + buf * G::TILE + (h ? OFF_A1 : OFF_A0);
    // Fragments in the order the MFMAs consume them (k-chunk outer), and NO lgkmcnt(0) in front of COMPUTE:
    // the compiler's own counted waits (lgkmcnt(9), (8), (7), (6), (2), (0) in an f16mx A phase) let the first
    // MFMAs start while the tails are still in flight.  +1.5 % on the f16mx layers, nothing in bf16, same bits
    // (profiles/r04_h_lgkm_ab.txt; OIBL_RING_LGKM0 restores the row-outer order and the full wait).
#ifndef OIBL_RING_LGKM0
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i2 = 0; i2 < 2; ++i2) fa[i2][kk] = read_frag(s + i2 * 4096, kk);
#else
#pragma unroll
    for (int i2 = 0; i2 < 2; ++i2)
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) fa[i2][kk] = read_frag(s + i2 * 4096, kk);
#endif
  };
  auto read_b = [&](int buf, int h, bf16x8_t (&f)[4]) __attribute__((always_inline)) {
    if constexpr (P == RING_MX_NOREAD) return;
    const char* s = rd_b + buf * G::TILE + (h ? OFF_B1 : OFF_B0);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) f[kk] = read_frag(s, kk);
  };

  auto compute = [&](auto h_c, auto j_c, const bf16x8_t (&fb)[4], auto&& issue) __attribute__((always_inline)) {
    constexpr int h = decltype(h_c)::value, j = decltype(j_c)::value;
#ifdef OIBL_RING_LGKM0
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#endif
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (BAR1 && GROUP == 0) {   // both groups compute inside one barrier interval: the one that still
      __builtin_amdgcn_s_setprio(2);      // has to LOAD goes first
    } else {
      __builtin_amdgcn_s_setprio(1);
    }
    auto mma = [&](int i2, int ka, int kb) __attribute__((always_inline)) {
      if constexpr (P == RING_F16) {
        const f16x8_t a = __builtin_bit_cast(f16x8_t, fa[i2][ka]), b = __builtin_bit_cast(f16x8_t, fb[kb]);
        acc[2 * h + i2][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[2 * h + i2][j], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[2 * h + i2][j], 0, 0, 0);
      } else {
        acc[2 * h + i2][j] =
            SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_bf16(fb[kb], fa[i2][ka], acc[2 * h + i2][j], 0, 0, 0)
                 : __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[i2][ka], fb[kb], acc[2 * h + i2][j], 0, 0, 0);
      }
    };
    if constexpr (P == RING_MX_NOMFMA) {
      asm volatile("" : "+v"(acc[2 * h][j]), "+v"(acc[2 * h + 1][j]));
    } else if constexpr (MX) {
      typedef __attribute__((ext_vector_type(4))) int i4;
      auto f16 = [&](int i2, int k) __attribute__((always_inline)) {
        const f16x8_t a = __builtin_bit_cast(f16x8_t, fa[i2][k]), b = __builtin_bit_cast(f16x8_t, fb[k]);
        acc[2 * h + i2][j] = SWAP ? __builtin_amdgcn_mfma_f32_32x32x16_f16(b, a, acc[2 * h + i2][j], 0, 0, 0)
                                  : __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[2 * h + i2][j], 0, 0, 0);
      };
      const i32x8_t b8 = __builtin_shufflevector(__builtin_bit_cast(i4, fb[2]), __builtin_bit_cast(i4, fb[3]),
                                                 0, 1, 2, 3, 4, 5, 6, 7);
      auto mx = [&](int i2) __attribute__((always_inline)) {
        const i32x8_t a8 = __builtin_shufflevector(__builtin_bit_cast(i4, fa[i2][2]),
                                                   __builtin_bit_cast(i4, fa[i2][3]), 0, 1, 2, 3, 4, 5, 6, 7);
        // e2m3 x e2m3 (cbsz = blgp = 2); scales: byte 0 of dword 6 of either operand
#ifdef OIBL_MX_TAIL_B128
        constexpr int SC = 7;   // the tail slot read as one ds_read_b128: [d4 d5 0 scale]
#else
        constexpr int SC = 6;
#endif
        acc[2 * h + i2][j] =
            SWAP ? __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(b8, a8, acc[2 * h + i2][j], 2, 2, 0, b8[SC], 0, a8[SC])
                 : __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[2 * h + i2][j], 2, 2, 0, a8[SC], 0, b8[SC]);
      };
      f16(0, 0);
      f16(1, 0);
      issue();
      f16(0, 1);
      f16(1, 1);
      mx(0);
      mx(1);
      // (the MX instruction's 6-register operands are assembled by copies, and copy + MFMA were seen
      //  to be sunk out of the segment, past the barrier, into the next LOAD segment: pin the results)
      asm volatile("" : "+v"(acc[2 * h][j]), "+v"(acc[2 * h + 1][j]));
    } else if constexpr (X3) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        if (pr == 1) issue();
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) mma(i2, pr + 2, pr);  // lo . hi
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) mma(i2, pr, pr + 2);  // hi . lo
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) mma(i2, pr, pr);      // hi . hi
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        if (kk == 1) issue();
#pragma unroll
        for (int i2 = 0; i2 < 2; ++i2) mma(i2, kk, kk);
      }
    }
    // every fragment read of this phase is retired INSIDE it (the WAR rule above counts on that): the ones this
    // phase multiplies were waited for by its MFMAs, P3's pre-read of the next tile's B0 — issued a whole
    // COMPUTE segment ago — is waited for here, behind the last MFMA's issue
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
  };
  auto bar = [&]() __attribute__((always_inline)) {
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (P != RING_MX_NOBAR) __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  };
  // the barrier in front of (g = 0) / behind (g = 1) a COMPUTE segment: with BAR1 only group g crosses it
  auto bar_g = [&](int g) __attribute__((always_inline)) {
    if constexpr (BAR1) {
      __builtin_amdgcn_sched_barrier(0);
      if (GROUP == g) __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
    } else {
      bar();
    }
  };

  using I0 = std::integral_constant<int, 0>;
  using I1 = std::integral_constant<int, 1>;
  using I2 = std::integral_constant<int, 2>;

  // ---- prologue: B0 A0 B1 A1 of K-tile 0, B0 A0 B1 of K-tile 1 (the steady-state issue order)
  la.begin_tile();
  lb.begin_tile();
  stage_b(0, 0);
  stage_a(0, 0, 0);
  stage_a(0, 0, 1);
  stage_b(0, 1);
  stage_a(0, 1, 0);
  stage_a(0, 1, 1);
  lb.begin_tile();
  stage_b(1, 0);
  la.begin_tile();
  stage_a(1, 0, 0);
  stage_a(1, 0, 1);
  stage_b(1, 1);
  wait_vmcnt<2 * NA + 3 * NB>();  // B0(0), A0(0) of this wave have landed
  bar();
  read_b(0, 0, fbx);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  if constexpr (!BAR1) {
    if (group == 1) bar();  // group 1 runs one barrier behind group 0
  }

  // One K-tile = 4 phases.  PAR = tile parity (LDS buffer; which register set holds B0).
  // TAIL: 0 = steady state, 1 = tile nsteps-2, 2 = tile nsteps-1 (nothing left to stage).
  // The counted waits leave exactly the five youngest units in flight (steady state; with SPLIT the
  // not-yet-issued second half of the phase's A unit is subtracted); in the tail the units that
  // are no longer issued are subtracted.
  // Cursor discipline: A1(t+1) is staged (P0) before la moves on to t+2 (P2); B0(t+2) is the first
  // unit of tile t+2 (P1), so lb moves there.
  auto ktile = [&](auto par_c, auto tail_c) __attribute__((always_inline)) {
    constexpr int PAR = decltype(par_c)::value;
    constexpr int TAIL = decltype(tail_c)::value;
    bf16x8_t(&b0)[4] = PAR ? fby : fbx;  // B0 of this tile
    bf16x8_t(&b1)[4] = PAR ? fbx : fby;  // B1 of this tile; from P3 on: B0 of the next tile
    // A phase = reads; [early: issues;] counted wait; barrier; COMPUTE [late: issues inside]; barrier.
    // `cnt` = what the wait allows in flight when it stands AFTER the phase's `n_issue` instructions.
    // `prep` (EARLY schedules): the K-cursor advance of the NEXT phase's issue — scalar code with branches,
    // ~100 cycles at the head of a LOAD segment (RING_MX_PROF: LDS-DMA issue 172 cycles in P1 against 64
    // in P0) — runs between the MFMAs of this phase instead, where the wave waits for the matrix pipe anyway.
    auto phase = [&](auto cnt_c, auto n_issue_c, auto h_c, auto j_c, const bf16x8_t (&fb)[4], auto&& issue,
                     auto sb_c, auto&& prep) __attribute__((always_inline)) {
      constexpr int CNT = decltype(cnt_c)::value, NI = decltype(n_issue_c)::value;
      constexpr int SB = decltype(sb_c)::value;   // first stamp slot of this phase, -1 = none
      if constexpr (!LATE) {
        if constexpr (SB >= 0) RING_STAMP(SB + 1);       // reads issued
        issue();
        if constexpr (SB >= 0) RING_STAMP(SB + 2);       // LDS-DMA issued
        if constexpr (CNT >= 0) wait_vmcnt<CNT>();
        if constexpr (SB >= 0) RING_STAMP(SB + 3);       // counted wait passed
        bar_g(0);
        if constexpr (SB >= 0) RING_STAMP(SB + 4);       // barrier passed: COMPUTE starts
        compute(h_c, j_c, fb, prep);
        if constexpr (SB >= 0) RING_STAMP(SB + 5);       // MFMAs issued
      } else {
        if constexpr (CNT >= 0) wait_vmcnt<(CNT - NI)>();
        bar_g(0);
        compute(h_c, j_c, fb, issue);
      }
      bar_g(1);
      if constexpr (SB >= 0) RING_STAMP(SB + 6);         // closing barrier passed
    };
#define RING_IC(x) std::integral_constant<int, (x)> {}
    constexpr int HA = SPLIT ? NA / 2 : NA;   // instructions of the A issue in P0 / P2
    constexpr int HB = SPLIT ? NA / 2 : 0;    // A instructions issued next to the B unit in P1 / P3
    // P0: A0 x B0
    RING_STAMP(0);
    read_a(PAR, 0);
    if constexpr (TAIL <= 1)
      phase(RING_IC(3 * NA + 2 * NB - (SPLIT ? NA / 2 : 0)), RING_IC(HA), I0{}, I0{}, b0,
            [&] { stage_a(PAR ^ 1, 1, 0); }, RING_IC(0),  // A1(t+1) (SPLIT: its first half)
            [&] { if constexpr (TAIL == 0 && !LATE) lb.begin_tile(); });
    else
      phase(RING_IC(NA), RING_IC(0), I0{}, I0{}, b0, [] {}, RING_IC(-1), [] {});
    // P1: A0 x B1
    RING_STAMP(7);
    read_b(PAR, 1, b1);
    if constexpr (TAIL == 0)
      phase(RING_IC(2 * NA + 3 * NB), RING_IC(HB + NB), I0{}, I1{}, b1, [&] {
        stage_a(PAR ^ 1, 1, 1);  // SPLIT: second half of A1(t+1)
        if constexpr (LATE) lb.begin_tile();   // (EARLY: advanced in P0's COMPUTE)
        stage_b(PAR, 0);  // B0(t+2)
      }, RING_IC(7), [&] { if constexpr (!LATE) la.begin_tile(); });
    else if constexpr (TAIL == 1)
      phase(RING_IC(2 * NA + 2 * NB), RING_IC(HB), I0{}, I1{}, b1, [&] { stage_a(PAR ^ 1, 1, 1); }, RING_IC(-1), [] {});
    else
      phase(RING_IC(0), RING_IC(0), I0{}, I1{}, b1, [] {}, RING_IC(-1), [] {});
    // P2: A1 x B1
    read_a(PAR, 1);
    if constexpr (TAIL == 0)
      phase(RING_IC(3 * NA + 2 * NB - (SPLIT ? NA / 2 : 0)), RING_IC(HA), I1{}, I1{}, b1, [&] {
        if constexpr (LATE) la.begin_tile();   // (EARLY: advanced in P1's COMPUTE)
        stage_a(PAR, 0, 0);  // A0(t+2) (SPLIT: its first half)
      }, RING_IC(-1), [] {});
    else if constexpr (TAIL == 1)
      phase(RING_IC(2 * NA + NB), RING_IC(0), I1{}, I1{}, b1, [] {}, RING_IC(-1), [] {});
    else
      phase(RING_IC(-1), RING_IC(0), I1{}, I1{}, b1, [] {}, RING_IC(-1), [] {});
    // P3: A1 x B0   (B0 of the next tile goes into the register set B1 just vacated)
    if constexpr (TAIL <= 1) read_b(PAR ^ 1, 0, b1);
    if constexpr (TAIL == 0)
      phase(RING_IC(2 * NA + 3 * NB), RING_IC(HB + NB), I1{}, I0{}, b0, [&] {
        stage_a(PAR, 0, 1);  // SPLIT: second half of A0(t+2)
        stage_b(PAR, 1);     // B1(t+2)
      }, RING_IC(-1), [] {});
    else if constexpr (TAIL == 1)
      phase(RING_IC(NA + NB), RING_IC(0), I1{}, I0{}, b0, [] {}, RING_IC(-1), [] {});
    else
      phase(RING_IC(-1), RING_IC(0), I1{}, I0{}, b0, [] {}, RING_IC(-1), [] {});
#undef RING_IC
  };
  for (int t = 0; t + 3 < nsteps; t += 2) {  // pairs of steady-state tiles
    ktile(I0{}, I0{});
    ktile(I1{}, I0{});
  }
  if constexpr (ODD) {
    ktile(I0{}, I0{});
    ktile(I1{}, I1{});
    ktile(I0{}, I2{});
  } else {
    ktile(I0{}, I1{});
    ktile(I1{}, I2{});
  }
  if constexpr (!BAR1) {
    if (group == 0) bar();
  }
  __syncthreads();
  if constexpr (PROF) {
    if (stamps != nullptr && lane == 0 && (wave & 3) == 0) {
#pragma unroll
      for (int i = 0; i < 14; ++i) stamps[(wave >> 2) * 14 + i] = st_[i];
    }
  }
#undef RING_STAMP
}

}  // namespace oibl
