// NT GEMM core for gfx950:  C[m][n] += sum_k A[m][k] * B[n][k]   (both operands K-contiguous).
//
// One core serves every contraction on the hot path: the 3x3 convolutions (A rows are im2col
// gathers of NHWC pixels), NetVLAD soft-assignment, the PCA projection and the query x gallery
// distance matrix.  What differs per user is (a) the Loader that yields, for every K-step, the
// global address of each 128-byte row segment, and (b) the epilogue applied to the accumulators.
//
// Geometry
//   * a K-step is 128 bytes of K per row (64 bf16 / 32 fp32) — one full cache line per row;
//   * a workgroup of WAVES_M x WAVES_N wave64s owns a BM x BN tile, each wave TM x TN tiles of
//     32x32 accumulated with v_mfma_f32_32x32x16_bf16 (bf16) or v_mfma_f32_32x32x2_f32 (fp32);
//   * LDS stage = [BM + BN rows][128 B]; two stages, tile t+1 is in flight (global_load_lds,
//     16 B per lane straight into LDS) while tile t is multiplied.
//
// LDS image.  global_load_lds writes wave-linear (base + lane*16), so one wave-instruction fills
// 8 consecutive rows x 8 slots of 16 B.  A fragment read is "lane -> row (lane & 31), slot
// 2*kk + (lane >> 5)" with ds_read_b128, which would put a 16-lane group on two 16-B bank slots
// (8-way conflict).  The image is therefore XOR-swizzled: physical slot = logical slot ^
// ((row >> 1) & 7).  The permutation is applied on the SOURCE address of the load (lane l of a
// load reads logical slot (l & 7) ^ swz(row) of its row — still the same 128-B line, so global
// coalescing is unchanged) and on the READ address; the LDS destination stays linear.  With it
// every ds_read_b128 lane group touches 16 distinct slots.
//
// fp32 K order.  v_mfma_f32_32x32x2_f32 takes one float per lane (k = lane >> 5).  A lane reads
// 16 B = 4 consecutive floats of its row and feeds them to 4 successive MFMAs; A and B use the
// same (lane half, element) -> k assignment, so the products pair up correctly and only the
// summation order over k differs from ascending (irrelevant to the result beyond fp32 rounding).
//
// bf16x3 (common.h): a K-step row is [32 hi | 32 lo]; the four k-chunks a lane reads are hi[0:16],
// hi[16:32], lo[0:16], lo[16:32] and the step evaluates lo.hi + hi.lo + hi.hi per 16-wide half
// (6 MFMAs instead of 4 per tile pair, 32 real K elements instead of 64).
#pragma once

#include <type_traits>

#include "common.h"

namespace oibl {

template <typename T_, int WAVES_M_, int WAVES_N_, int TM_, int TN_>
struct GemmCfg {
  using T = T_;
  static constexpr int WAVES_M = WAVES_M_, WAVES_N = WAVES_N_, TM = TM_, TN = TN_;
  static constexpr int NWAVES = WAVES_M * WAVES_N;
  static constexpr int NTHREADS = 64 * NWAVES;
  static constexpr int BM = WAVES_M * TM * 32;
  static constexpr int BN = WAVES_N * TN * 32;
  static constexpr int BK = 128 / (int)sizeof(T);  // elements of K per step
  static constexpr int A_BYTES = BM * 128;
  static constexpr int B_BYTES = BN * 128;
  static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
  static constexpr int MAIN_LDS_BYTES = 2 * STAGE_BYTES;
  static constexpr int A_LOADS = BM / (8 * NWAVES);  // wave-instructions per wave per step
  static constexpr int B_LOADS = BN / (8 * NWAVES);
  static_assert(BM % (8 * NWAVES) == 0 && BN % (8 * NWAVES) == 0, "tile/wave mismatch");
  static_assert((NWAVES & 1) == 0, "swizzle term assumes an even wave count");
};

template <typename T>
struct Mma;
template <>
struct Mma<bf16_t> {
  using Frag = bf16x8_t;
  __device__ static inline void mma(f32x16_t& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
};
template <>
struct Mma<bf16x3_t> : Mma<bf16_t> {};
template <>
struct Mma<float> {
  using Frag = f32x4_t;
  __device__ static inline void mma(f32x16_t& acc, const Frag& a, const Frag& b) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b[1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b[2], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b[3], acc, 0, 0, 0);
  }
};

// Position of this lane inside the workgroup tile.
struct WaveCoord {
  int lane, wave, wm, wn;
};
template <typename Cfg>
__device__ static inline WaveCoord wave_coord() {
  WaveCoord c;
  c.lane = threadIdx.x & 63;
  c.wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  c.wm = c.wave / Cfg::WAVES_N;
  c.wn = c.wave % Cfg::WAVES_N;
  return c;
}

// For load-instruction j of this wave: which tile row this lane fetches, and the byte offset of
// its (swizzled) 16-byte piece inside that row's 128-byte K segment.
template <typename Cfg>
__device__ static inline int load_row(const WaveCoord& c, int j) {
  return (j * Cfg::NWAVES + c.wave) * 8 + (c.lane >> 3);
}
template <typename Cfg>
__device__ static inline int load_piece_bytes(const WaveCoord& c) {
  // swz(row) = (row >> 1) & 7 with row = 8*(j*NWAVES + wave) + (lane >> 3); NWAVES even.
  const int swz = 4 * (c.wave & 1) + (c.lane >> 4);
  return ((c.lane & 7) ^ swz) * 16;
}

__device__ static inline void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void*)gsrc,
      (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// Loader concept:
//   const char* src(int j) const;  // this lane's 16-B source for load j at the current K-step
//   bool active() const;            // false only in ablation builds (skip this operand's loads)
//   void next();                    // advance to the next K-step
//
// GLDS = true : global_load_lds straight into the next stage (the fast path).
// GLDS = false: same LDS image staged through registers (global_load -> ds_write_b128); kept as
//               the cross-check of the LDS-DMA path.
template <typename Cfg, bool GLDS, typename ALoader, typename BLoader>
__device__ static inline void gemm_nt_mainloop(f32x16_t (&acc)[Cfg::TM][Cfg::TN], char* lds,
                                               const WaveCoord& c, ALoader& la, BLoader& lb,
                                               int nsteps) {
  using T = typename Cfg::T;
  using Frag = typename Mma<T>::Frag;
  constexpr int TM = Cfg::TM, TN = Cfg::TN;

  // per-lane fragment read offsets (same for A and B): row (lane & 31), logical slot
  // 2*kk + (lane >> 5), physical slot = logical ^ ((row >> 1) & 7).
  int frag_off[4];
  {
    const int row = c.lane & 31, half = c.lane >> 5, swz = (c.lane >> 1) & 7;
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) frag_off[kk] = row * 128 + (((2 * kk + half) ^ swz) * 16);
  }
  const int a_wave_off = c.wm * TM * 4096;
  const int b_wave_off = Cfg::A_BYTES + c.wn * TN * 4096;
  const int wave_ld_off = c.wave * 1024;

  uint4 ra[Cfg::A_LOADS], rb[Cfg::B_LOADS];

  auto fetch = [&](int stage) {
    char* sbase = lds + stage * Cfg::STAGE_BYTES + wave_ld_off;
    if constexpr (GLDS) {
      if (la.active()) {
#pragma unroll
        for (int j = 0; j < Cfg::A_LOADS; ++j) glds16(la.src(j), sbase + j * Cfg::NWAVES * 1024);
      }
      if (lb.active()) {
#pragma unroll
        for (int j = 0; j < Cfg::B_LOADS; ++j)
          glds16(lb.src(j), sbase + Cfg::A_BYTES + j * Cfg::NWAVES * 1024);
      }
    } else {
#pragma unroll
      for (int j = 0; j < Cfg::A_LOADS; ++j) ra[j] = *reinterpret_cast<const uint4*>(la.src(j));
#pragma unroll
      for (int j = 0; j < Cfg::B_LOADS; ++j) rb[j] = *reinterpret_cast<const uint4*>(lb.src(j));
    }
    la.next();
    lb.next();
  };
  auto commit = [&](int stage) {  // register-staged path only
    if constexpr (!GLDS) {
      char* sbase = lds + stage * Cfg::STAGE_BYTES + wave_ld_off + c.lane * 16;
#pragma unroll
      for (int j = 0; j < Cfg::A_LOADS; ++j)
        *reinterpret_cast<uint4*>(sbase + j * Cfg::NWAVES * 1024) = ra[j];
#pragma unroll
      for (int j = 0; j < Cfg::B_LOADS; ++j)
        *reinterpret_cast<uint4*>(sbase + Cfg::A_BYTES + j * Cfg::NWAVES * 1024) = rb[j];
    }
  };
  auto sync_stage = [&]() {
    if constexpr (GLDS) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  };

  fetch(0);
  commit(0);
  sync_stage();

  // One wave-instruction of LDS-DMA costs the issuing wave ~100-200 cycles of issue time
  // (MI355X_MICROARCH.md "LDS-DMA piece ... issue cost"): issued back to back at the top of a
  // K-step they stall the wave for longer than the step's MFMAs take.  The step's loads are
  // therefore dealt out over its four k-sub-steps, each group placed in front of that
  // sub-step's MFMAs, so that their issue overlaps matrix work already in the pipe.
  constexpr int NLD = Cfg::A_LOADS + Cfg::B_LOADS;
  auto fetch_part = [&](int stage, int kk) {
    char* sbase = lds + stage * Cfg::STAGE_BYTES + wave_ld_off;
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
      if ((i * 4) / NLD != kk) continue;
      if (i < Cfg::A_LOADS) {
        if (la.active()) glds16(la.src(i), sbase + i * Cfg::NWAVES * 1024);
      } else {
        const int j = i - Cfg::A_LOADS;
        if (lb.active()) glds16(lb.src(j), sbase + Cfg::A_BYTES + j * Cfg::NWAVES * 1024);
      }
    }
  };

  for (int t = 0; t < nsteps; ++t) {
    const int cur = t & 1;
    const bool more = (t + 1) < nsteps;
    if constexpr (!GLDS) {
      if (more) fetch(cur ^ 1);
    }

    const char* abase = lds + cur * Cfg::STAGE_BYTES + a_wave_off;
    const char* bbase = lds + cur * Cfg::STAGE_BYTES + b_wave_off;
    if constexpr (std::is_same<T, bf16x3_t>::value) {
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {  // 16-wide half of the step's 32 K elements
        Frag ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          ah[i] = *reinterpret_cast<const Frag*>(abase + i * 4096 + frag_off[pr]);
          al[i] = *reinterpret_cast<const Frag*>(abase + i * 4096 + frag_off[pr + 2]);
        }
#pragma unroll
        for (int i = 0; i < TN; ++i) {
          bh[i] = *reinterpret_cast<const Frag*>(bbase + i * 4096 + frag_off[pr]);
          bl[i] = *reinterpret_cast<const Frag*>(bbase + i * 4096 + frag_off[pr + 2]);
        }
        if constexpr (GLDS) {
          if (more) {
            fetch_part(cur ^ 1, 2 * pr);
            fetch_part(cur ^ 1, 2 * pr + 1);
          }
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) Mma<T>::mma(acc[i][jn], al[i], bh[jn]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) Mma<T>::mma(acc[i][jn], ah[i], bl[jn]);
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) Mma<T>::mma(acc[i][jn], ah[i], bh[jn]);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) {
        Frag a[TM], b[TN];
#pragma unroll
        for (int i = 0; i < TM; ++i)
          a[i] = *reinterpret_cast<const Frag*>(abase + i * 4096 + frag_off[kk]);
#pragma unroll
        for (int i = 0; i < TN; ++i)
          b[i] = *reinterpret_cast<const Frag*>(bbase + i * 4096 + frag_off[kk]);
        if constexpr (GLDS) {
          if (more) fetch_part(cur ^ 1, kk);
        }
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int jn = 0; jn < TN; ++jn) Mma<T>::mma(acc[i][jn], a[i], b[jn]);
      }
    }
    if constexpr (GLDS) {
      if (more) {
        la.next();
        lb.next();
      }
    }

    if (more) commit(cur ^ 1);
    sync_stage();
  }
}

// C/D fragment geometry of the 32x32 MFMA: register r of lane l holds
//   row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5),  col = l & 31.
__device__ static inline int acc_row(int r, int lane) {
  return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
}

// Plain row-major loader: row i of the tile is `base + (row0 + i) * ld_bytes`, clamped to the
// last valid row so that partial tiles never read out of bounds (their results are discarded).
template <typename Cfg, int NLOADS>
struct RowLoader {
  const char* p[NLOADS];
  __device__ inline void init(const WaveCoord& c, const void* base, long row0, long nrows,
                              long ld_bytes) {
    const int piece = load_piece_bytes<Cfg>(c);
#pragma unroll
    for (int j = 0; j < NLOADS; ++j) {
      long r = row0 + load_row<Cfg>(c, j);
      if (r > nrows - 1) r = nrows - 1;
      p[j] = reinterpret_cast<const char*>(base) + r * ld_bytes + piece;
    }
  }
  __device__ inline const char* src(int j) const { return p[j]; }
  __device__ inline bool active() const { return true; }
  __device__ inline void next() {
#pragma unroll
    for (int j = 0; j < NLOADS; ++j) p[j] += 128;
  }
};

}  // namespace oibl
