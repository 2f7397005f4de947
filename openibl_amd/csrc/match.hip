// Query x gallery squared-L2 distance matrix and per-row top-k on gfx950.
// Reference behaviour: pairwise_distance (ibl/evaluators.py:105-130) and the argsort consumed by
// evaluate_all (ibl/evaluators.py:142-159).
#undef OIBL_MX_TAIL_B128   // the distance kernels have no registers for the 16-byte tail (ring_core.h)
#include "gemm_core.h"
#include "ring_core.h"

namespace oibl {

// squared L2 norm per row (torch.pow(x, 2).sum(dim=1), evaluators.py:127-128); one wave per row
__global__ void row_sqnorm_kernel(const float* __restrict__ x, float* __restrict__ out, int rows,
                                  int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const float* xr = x + (size_t)row * d;
  float s = 0.f;
  for (int i = lane * 4; i < d; i += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xr + i);
    s = fmaf(v.x, v.x, s);
    s = fmaf(v.y, v.y, s);
    s = fmaf(v.z, v.z, s);
    s = fmaf(v.w, v.w, s);
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

// Descriptor rows in their storage type (OIBL_ST_*), widened to fp32 four at a time.
template <int ST>
__device__ static inline float4 load4_widen(const void* row, int i) {
  if constexpr (ST == OIBL_ST_F32) {
    return *reinterpret_cast<const float4*>(static_cast<const float*>(row) + i);
  } else {
    const uint2 r = *reinterpret_cast<const uint2*>(static_cast<const uint16_t*>(row) + i);
    float4 v;
    if constexpr (ST == OIBL_ST_F16) {
      v.x = f16_bits_to_f32((uint16_t)r.x);
      v.y = f16_bits_to_f32((uint16_t)(r.x >> 16));
      v.z = f16_bits_to_f32((uint16_t)r.y);
      v.w = f16_bits_to_f32((uint16_t)(r.y >> 16));
    } else {
      v.x = __builtin_bit_cast(float, r.x << 16);
      v.y = __builtin_bit_cast(float, r.x & 0xffff0000u);
      v.z = __builtin_bit_cast(float, r.y << 16);
      v.w = __builtin_bit_cast(float, r.y & 0xffff0000u);
    }
    return v;
  }
}

// The same reduction over the widened row, with the operand copy the contraction reads written on
// the way: OUT = 1 the bf16 rounding (bf16 mode; nothing to write for bf16 storage, xo == nullptr),
// OUT = 2 the fp32 widening of a 16-bit row (fp32 mode), OUT = 3 the (hi, lo) split of bf16x3 mode
// (rows of d / 32 groups [32 hi | 32 lo]).  The descriptors are read once.
template <int ST, int OUT>
__global__ void row_sqnorm_cast_kernel(const void* __restrict__ x, float* __restrict__ out,
                                       void* __restrict__ xo, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t es = ST == OIBL_ST_F32 ? 4 : 2;
  const char* xr = static_cast<const char*>(x) + (size_t)row * d * es;
  float s = 0.f;
  for (int i = lane * 4; i < d; i += 256) {
    const float4 v = load4_widen<ST>(xr, i);
    s = fmaf(v.x, v.x, s);
    s = fmaf(v.y, v.y, s);
    s = fmaf(v.z, v.z, s);
    s = fmaf(v.w, v.w, s);
    if constexpr (OUT == 1) {
      if (ST != OIBL_ST_BF16) {
        uint2 b;
        b.x = (uint32_t)f32_to_bf16_bits(v.x) | ((uint32_t)f32_to_bf16_bits(v.y) << 16);
        b.y = (uint32_t)f32_to_bf16_bits(v.z) | ((uint32_t)f32_to_bf16_bits(v.w) << 16);
        *reinterpret_cast<uint2*>(static_cast<uint16_t*>(xo) + (size_t)row * d + i) = b;
      }
    } else if constexpr (OUT == 2) {
      *reinterpret_cast<float4*>(static_cast<float*>(xo) + (size_t)row * d + i) = v;
    } else if constexpr (OUT == 3) {
      uint2 hi, lo;
      hi.x = (uint32_t)f32_to_bf16_bits(v.x) | ((uint32_t)f32_to_bf16_bits(v.y) << 16);
      hi.y = (uint32_t)f32_to_bf16_bits(v.z) | ((uint32_t)f32_to_bf16_bits(v.w) << 16);
      const float r0 = v.x - __builtin_bit_cast(float, hi.x << 16);
      const float r1 = v.y - __builtin_bit_cast(float, hi.x & 0xffff0000u);
      const float r2 = v.z - __builtin_bit_cast(float, hi.y << 16);
      const float r3 = v.w - __builtin_bit_cast(float, hi.y & 0xffff0000u);
      lo.x = (uint32_t)f32_to_bf16_bits(r0) | ((uint32_t)f32_to_bf16_bits(r1) << 16);
      lo.y = (uint32_t)f32_to_bf16_bits(r2) | ((uint32_t)f32_to_bf16_bits(r3) << 16);
      char* q = static_cast<char*>(xo) + (size_t)row * d * 4 + x3_off((size_t)i);
      *reinterpret_cast<uint2*>(q) = hi;
      *reinterpret_cast<uint2*>(q + 64) = lo;
    }
  }
  s = wave_sum(s);
  if (lane == 0) out[row] = s;
}

// f16mx operand rows (common.h): one wave per row, a lane packs whole 32-element groups (their 128-byte
// lines); the norm is that of the widened row, as in the other modes.  A row with an element beyond fp16
// (|v| > 65504: its line is not a 1e-4 image of it, common.h) gets the norm +inf — every distance to it is
// +inf, never a finite wrong number.  (Descriptors are unit vectors: ibl/evaluators.py:33.)
template <int ST>
__global__ void row_sqnorm_mx_kernel(const void* __restrict__ x, float* __restrict__ out,
                                     void* __restrict__ xo, int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t es = ST == OIBL_ST_F32 ? 4 : 2;
  const char* xr = static_cast<const char*>(x) + (size_t)row * d * es;
  uint4* orow = reinterpret_cast<uint4*>(static_cast<char*>(xo) + (size_t)row * d * 4);
  float s = 0.f;
  bool over = false;
  for (int g = lane; g < (d >> 5); g += 64) {
    float v[32];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const float4 t = load4_widen<ST>(xr, g * 32 + 4 * k);
      v[4 * k] = t.x;
      v[4 * k + 1] = t.y;
      v[4 * k + 2] = t.z;
      v[4 * k + 3] = t.w;
    }
#pragma unroll
    for (int e = 0; e < 32; ++e) s = fmaf(v[e], v[e], s);
#pragma unroll
    for (int e = 0; e < 32; ++e) over |= fabsf(v[e]) > 65504.f;   // (a NaN row keeps its NaN norm)
    uint4 line[8];
    mx_pack_line(v, line);
#pragma unroll
    for (int k = 0; k < 8; ++k) orow[g * 8 + k] = line[k];
  }
  s = wave_sum(s);
  if (__builtin_amdgcn_ballot_w64(over) != 0) s = INFINITY;
  if (lane == 0) out[row] = s;
}

struct PairParams {
  const void* x;  // [m][d] T
  const void* y;  // [n][d] T
  const float* xn;
  const float* yn;
  float* dist;
  size_t ldd;
  int m, n, d, tiles_n;
};

// dist[i][j] = (xn[i] + yn[j]) - 2 * x_i . y_j
template <typename Cfg, bool GLDS>
__global__ __launch_bounds__(Cfg::NTHREADS) void pairwise_kernel(PairParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = typename Cfg::T;
  const WaveCoord c = wave_coord<Cfg>();
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = tile % p.tiles_n, tm = tile / p.tiles_n;
  const long m0 = (long)tm * Cfg::BM, n0 = (long)tn * Cfg::BN;

  RowLoader<Cfg, Cfg::A_LOADS> la;
  RowLoader<Cfg, Cfg::B_LOADS> lb;
  la.init(c, p.x, m0, p.m, (long)p.d * sizeof(T));
  lb.init(c, p.y, n0, p.n, (long)p.d * sizeof(T));

  f32x16_t acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gemm_nt_mainloop<Cfg, GLDS>(acc, smem, c, la, lb, p.d / Cfg::BK);

#pragma unroll
  for (int j = 0; j < Cfg::TN; ++j) {
    const long n = n0 + (c.wn * Cfg::TN + j) * 32 + (c.lane & 31);
    const bool nok = n < p.n;
    const float yn = nok ? p.yn[n] : 0.f;
#pragma unroll
    for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + (c.wm * Cfg::TM + i) * 32 + acc_row(r, c.lane);
        if (nok && m < p.m) p.dist[m * p.ldd + n] = fmaf(-2.0f, acc[i][j][r], p.xn[m] + yn);
      }
  }
}

// ---------------------------------------------------------------------------------------------
// bf16 distance kernel on the ring schedule (ring_core.h): 256 x 256 tile, A = queries, B = gallery,
// K = d.  Same K order and the same epilogue expression as pairwise_kernel -> identical matrices.
// Tile order: the hardware deals block b to XCD b % 8; every XCD gets a contiguous range of tile
// ids, and ids walk groups of 8 query tiles x all gallery tiles with the query tile fastest, so the
// 32 workgroups resident on an XCD form an 8 x 4 block of tiles that shares 8 query panels and 4
// gallery panels through that XCD's L2 (12 panel streams for 32 tiles instead of 33-64).
//   FILTER = false: write dist[m][ldd].
//   FILTER = true : fused top-k front end.  thr[row] is an upper bound of the row's k-th smallest
//     distance (k-th smallest over a strided sample of the gallery, computed with this same kernel,
//     so the sample's own distances reappear bit for bit); every distance <= thr[row] is appended
//     to the row's candidate list (value, global index).  The matrix is never written: for
//     8192 x 81920 the candidates are ~1 % of its 2.7 GB.  Lists that outgrow `cap` are counted,
//     not stored (the caller sees cnt > cap and falls back to the exact path).
// ---------------------------------------------------------------------------------------------
struct PairRingParams {
  const void* x;  // [m][d] bf16
  const void* y;  // [n rows at y_row_bytes][d] bf16
  const float* xn;
  const float* yn;  // yn[col * yn_stride]
  float* dist;
  size_t ldd;
  unsigned x_bytes, y_bytes;
  long y_row_bytes;
  int yn_stride;
  int m, n, d, tiles_m, tiles_n;
  const float* thr;  // thr[row * thr_stride]
  int thr_stride;
  float* cand_val;
  int32_t* cand_idx;
  int* cand_cnt;
  int cap, index_base, index_stride;  // global index of column c = index_base + c * index_stride
  int group_m;                        // query tiles per ordering group (see above)
  // 2-way split-K (threshold sample only, gridDim.y = 2): block (., h) contracts K-half h and writes
  // dist + h * part_stride; the norms enter half 0 only, so the two halves ADD UP to the distances
  // (in another summation order than the one-pass kernel: see thr_slack).
  size_t part_stride;
  // !FILTER: optional, max over the launch's columns of yn (bit pattern of a non-negative float,
  // atomicMax).  FILTER: optional, read back — the threshold of row i is widened by
  // thr_slack * (xn[i] + *yn_max) / 2, a bound on what the other summation order can move a distance.
  unsigned* yn_max;
  float thr_slack;
};

template <bool FILTER, int P = RING_BF16, bool BAR1 = false>
__global__ __launch_bounds__(512) void pairwise_ring_kernel(PairRingParams p) {
  using G = RingGeo<2>;
  constexpr bool X3 = P != RING_BF16;  // 4-byte operand elements, 32 K per K-tile (bf16x3 and f16mx)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave / G::WN, wn = wave % G::WN;
  const unsigned id = xcd_remap(blockIdx.x, gridDim.x);
  const unsigned gm = (unsigned)p.group_m;
  const unsigned width = gm * (unsigned)p.tiles_n;
  const unsigned grp = id / width, in_grp = id - grp * width;
  const unsigned first_m = grp * gm;
  const unsigned gsz = (unsigned)p.tiles_m - first_m < gm ? (unsigned)p.tiles_m - first_m : gm;
  const int tm = (int)(first_m + in_grp % gsz), tn = (int)(in_grp / gsz);
  const int m0 = tm * G::BM, n0 = tn * G::BN;

  const int piece = ring_piece(wave, lane);
  int rows_a[4], rows_b[4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      rows_a[2 * h + i] = ring_a_row<2>(wave, lane, h, i);
      rows_b[2 * h + i] = ring_b_row<2>(wave, lane, h, i);
    }
  // split-K: this block's K-half starts kh * d/2 elements into every row
  const int ksplit = (int)gridDim.y, kh = (int)blockIdx.y;
  const int d_part = p.d / ksplit;
  const unsigned koff = (unsigned)kh * (unsigned)d_part * (X3 ? 4u : 2u);
  RingRowLoader<2> la, lb;
  la.init(static_cast<const char*>(p.x) + koff, p.x_bytes - koff, m0, p.m, (long)p.d * (X3 ? 4 : 2), rows_a,
          piece);
  lb.init(static_cast<const char*>(p.y) + koff, p.y_bytes - koff, n0, p.n, p.y_row_bytes, rows_b,
          P >= RING_MX ? ring_piece_mxb(wave, lane) : piece);

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if constexpr (BAR1) {   // one copy of the loop per stagger group (ring_core.h)
    if ((wave >> 2) == 0) ring_mainloop<2, false, false, P, true, 0>(acc, smem, wave, lane, la, lb, d_part >> (X3 ? 5 : 6));
    else ring_mainloop<2, false, false, P, true, 1>(acc, smem, wave, lane, la, lb, d_part >> (X3 ? 5 : 6));
  } else {
    ring_mainloop<2, false, false, P>(acc, smem, wave, lane, la, lb, d_part >> (X3 ? 5 : 6));
  }

  // norms (and thresholds) of the tile's rows / columns -> LDS; out-of-range rows/columns are
  // clamped here and masked at the store
  float* const xn_s = reinterpret_cast<float*>(smem);
  float* const yn_s = xn_s + 256;
  float* const th_s = xn_s + 512;
  if (threadIdx.x < 256) {
    int r = m0 + (int)threadIdx.x;
    if (r > p.m - 1) r = p.m - 1;
    const float xn = p.xn[r];
    xn_s[threadIdx.x] = kh == 0 ? xn : 0.f;
    if (FILTER) {
      float th = p.thr[(long)r * p.thr_stride];
      if (p.yn_max) th += p.thr_slack * 0.5f * (xn + __uint_as_float(*p.yn_max));
      th_s[threadIdx.x] = th;
    }
  } else {
    int c = n0 + (int)threadIdx.x - 256;
    if (c > p.n - 1) c = p.n - 1;
    const float yn = p.yn[(long)c * p.yn_stride];
    yn_s[threadIdx.x - 256] = kh == 0 ? yn : 0.f;
    if (!FILTER && p.yn_max && kh == 0 && tm == 0) {  // one column of tiles covers every column once
      const float mx = wave_max(yn);                  // (norms are >= 0: their bit patterns order as integers)
      if (lane == 0) atomicMax(p.yn_max, __float_as_uint(mx));
    }
  }
  __syncthreads();
  // lane geometry: column col0 + 32 j, rows row0 + 32 i + (r & 3) + 8 (r >> 2)
  const int col0 = wn * 64 + (lane & 31), row0 = wm * 128 + 4 * (lane >> 5);
  if constexpr (!FILTER) {
    // stores go through buffer descriptors based at the first element of each 32-row accumulator tile of
    // the wave: one 32-bit lane offset, everything else is scalar (offsets inside a tile stay below
    // 31 * ldd * 4 bytes: the host guarantees ldd <= 2^24)
    const float* const dbase = p.dist + (size_t)kh * p.part_stride + ((size_t)m0 + wm * 128) * p.ldd + n0;
    const unsigned ldd4 = (unsigned)p.ldd * 4u;
    const unsigned voff = (unsigned)(4 * (lane >> 5)) * ldd4 + (unsigned)col0 * 4u;
    const int rows_left = p.m - m0 - row0;  // row r of this lane is valid iff its offset < rows_left
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(dbase + (size_t)(32 * i) * p.ldd), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool nok = n0 + col0 + 32 * j < p.n;
        const float yn = yn_s[col0 + 32 * j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rt = (r & 3) + 8 * (r >> 2), ro = 32 * i + rt;
          const float dv = fmaf(-2.0f, acc[i][j][r], xn_s[row0 + ro] + yn);
          if (nok && ro < rows_left)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dv), rs_d, (int)voff,
                                                  (int)((unsigned)rt * ldd4 + 128u * j), 0);
          // keep the scalar row offsets from being computed (and spilled) for all 128 stores at once
          if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
    // Survivors are rare (about 1 % of the distances, ~1.3 per lane and tile) but a wave sees one
    // in every second accumulator register, so nothing with a memory round trip may sit in this
    // loop: each lane first collects its own survivors in a private LDS list (count in a
    // register), then all lanes claim their slots with back-to-back atomics — one round trip per
    // tile instead of one per register.  List layout [entry][thread]: conflict-free.
    constexpr int LCAP = 8;
    float* const l_val = reinterpret_cast<float*>(smem + 4096);
    int* const l_idx = reinterpret_cast<int*>(smem + 4096 + LCAP * 512 * 4);
    int* const l_row = reinterpret_cast<int*>(smem + 4096 + 2 * LCAP * 512 * 4);
    const int rows_left = p.m - m0 - row0;
    int nl = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + col0 + 32 * j;
      const bool nok = n < p.n;
      const float yn = yn_s[col0 + 32 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = 32 * i + (r & 3) + 8 * (r >> 2);
          const float dv = fmaf(-2.0f, acc[i][j][r], xn_s[row0 + ro] + yn);
          if (nok && ro < rows_left && dv <= th_s[row0 + ro]) {
            const int m = m0 + row0 + ro;
            if (nl < LCAP) {
              l_val[nl * 512 + threadIdx.x] = dv;
              l_idx[nl * 512 + threadIdx.x] = p.index_base + n * p.index_stride;
              l_row[nl * 512 + threadIdx.x] = m;
              ++nl;
            } else {  // list full (practically never): claim the slot right away
              const int pos = atomicAdd(p.cand_cnt + m, 1);
              if (pos < p.cap) {
                p.cand_val[(size_t)m * p.cap + pos] = dv;
                p.cand_idx[(size_t)m * p.cap + pos] = p.index_base + n * p.index_stride;
              }
            }
          }
        }
    }
    int e_row[LCAP], e_pos[LCAP];
#pragma unroll
    for (int e = 0; e < LCAP; ++e) {
      e_row[e] = e < nl ? l_row[e * 512 + threadIdx.x] : 0;
      e_pos[e] = p.cap;
    }
#pragma unroll
    for (int e = 0; e < LCAP; ++e)
      if (e < nl) e_pos[e] = atomicAdd(p.cand_cnt + e_row[e], 1);
#pragma unroll
    for (int e = 0; e < LCAP; ++e)
      if (e < nl && e_pos[e] < p.cap) {
        p.cand_val[(size_t)e_row[e] * p.cap + e_pos[e]] = l_val[e * 512 + threadIdx.x];
        p.cand_idx[(size_t)e_row[e] * p.cap + e_pos[e]] = l_idx[e * 512 + threadIdx.x];
      }
  }
}

// ---------------------------------------------------------------------------------------------
// per-row top-k (k smallest, ascending, lowest index first on ties)
// One workgroup per row.  Every element becomes a 64-bit key (order-preserving bits of the value
// << 32 | index); a running threshold (the k-th best key so far) filters the stream, survivors
// are appended to an LDS candidate buffer that is bitonic-sorted only when it could overflow.
// After the first chunks the threshold is tight and nearly nothing survives, so the pass is a
// pure streaming read of the row.
// ---------------------------------------------------------------------------------------------
__device__ static inline uint32_t ordered_bits(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ static inline float from_ordered_bits(uint32_t o) {
  const uint32_t u = (o & 0x80000000u) ? (o & 0x7fffffffu) : ~o;
  return __uint_as_float(u);
}

constexpr int TOPK_CAP = 2048;    // candidate buffer entries (power of two)
constexpr int TOPK_CHUNK = 1024;  // elements examined between overflow checks (256 thr x 4)
constexpr unsigned long long TOPK_INF = 0xffffffffffffffffull;

__device__ static inline void bitonic_sort_lds(unsigned long long* buf, int n_pow2, int tid,
                                               int nthreads) {
  for (int k = 2; k <= n_pow2; k <<= 1) {
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n_pow2; i += nthreads) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = buf[i], b = buf[ixj];
          const bool up = ((i & k) == 0);
          if ((a > b) == up) {
            buf[i] = b;
            buf[ixj] = a;
          }
        }
      }
      __syncthreads();
    }
  }
}

// Short rows and small k (the two calls of the fused distance + top-k: 1024 sample distances, ~1000
// candidates; k = 10): no LDS candidate buffer and no 55-stage bitonic sort — every lane keeps up
// to 8 keys in registers, each wave extracts its k smallest by k rounds of (lane minimum, wave
// minimum, the owning lane retires that key), and wave 0 repeats the same over the 4 k survivors.
// Keys are unique except for (+inf, -1) paddings, so exactly one copy is retired per round and the
// result is the sorted prefix the bitonic path produces.
constexpr int SEL_MAX_K = 32, SEL_MAX_N = 2048;

__device__ static inline unsigned long long wave_min_u64(unsigned long long v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)v, o, 64), hi = __shfl_xor((unsigned)(v >> 32), o, 64);
    const unsigned long long w = ((unsigned long long)hi << 32) | lo;
    v = w < v ? w : v;
  }
  return v;
}

// one selection round over NQ register keys per lane: returns the wave's smallest key and retires it
template <int NQ>
__device__ static inline unsigned long long wave_extract_min(unsigned long long (&mine)[NQ], int lane) {
  unsigned long long m = mine[0];
#pragma unroll
  for (int q = 1; q < NQ; ++q) m = mine[q] < m ? mine[q] : m;
  const unsigned long long w = wave_min_u64(m);
  const unsigned long long owners = __ballot(m == w);
  if (lane == __ffsll((long long)owners) - 1) {
    bool done = false;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (!done && mine[q] == w) {
        mine[q] = TOPK_INF;
        done = true;
      }
  }
  return w;
}

// row_n (optional): per-row element count (clamped to n); a count above n raises *overflow.
__global__ __launch_bounds__(256) void row_topk_kernel(const float* __restrict__ vals,
                                                       const int32_t* __restrict__ idx_in, int n,
                                                       size_t ld, int k, int index_base,
                                                       float* __restrict__ out_val,
                                                       int32_t* __restrict__ out_idx,
                                                       const int* __restrict__ row_n,
                                                       int* __restrict__ overflow, int skip_le) {
  __shared__ unsigned long long cand[TOPK_CAP];
  __shared__ int cnt;
  __shared__ unsigned long long thr;
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* vr = vals + (size_t)row * ld;
  const int32_t* ir = idx_in ? idx_in + (size_t)row * ld : nullptr;
  if (row_n) {
    const int have_n = row_n[row];
    if (have_n > n && tid == 0 && overflow) atomicOr(overflow, 1);
    n = have_n < n ? have_n : n;
  }
  if (n <= skip_le) return;  // row_select_wave_kernel owns the short rows of this launch pair
  if (k <= SEL_MAX_K && n <= SEL_MAX_N) {  // workgroup-uniform
    unsigned long long* const wsel = cand;  // [4][SEL_MAX_K]
    const int lane = tid & 63, wave = tid >> 6;
    unsigned long long mine[SEL_MAX_N / 256];
#pragma unroll
    for (int q = 0; q < SEL_MAX_N / 256; ++q) {
      const int j = tid + 256 * q;
      mine[q] = TOPK_INF;
      if (j < n) {
        const uint32_t id = ir ? (uint32_t)ir[j] : (uint32_t)(index_base + j);
        mine[q] = ((unsigned long long)ordered_bits(vr[j]) << 32) | id;
      }
    }
    for (int r = 0; r < k; ++r) {
      const unsigned long long w = wave_extract_min(mine, lane);
      if (lane == 0) wsel[wave * SEL_MAX_K + r] = w;
    }
    __syncthreads();
    if (wave == 0) {
      unsigned long long two[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = lane + 64 * h;  // survivor e = wave (e / k), rank (e % k)
        two[h] = e < 4 * k ? wsel[(e / k) * SEL_MAX_K + (e % k)] : TOPK_INF;
      }
      unsigned long long res = TOPK_INF;
      for (int r = 0; r < k; ++r) {
        const unsigned long long w = wave_extract_min(two, lane);
        if (lane == r) res = w;
      }
      if (lane < k) {
        const bool valid = lane < n;
        out_val[(size_t)row * k + lane] = valid ? from_ordered_bits((uint32_t)(res >> 32)) : INFINITY;
        out_idx[(size_t)row * k + lane] = valid ? (int32_t)(uint32_t)(res & 0xffffffffu) : -1;
      }
    }
    return;
  }
  if (tid == 0) {
    cnt = 0;
    thr = TOPK_INF;
  }
  __syncthreads();

  for (int base = 0; base < n; base += TOPK_CHUNK) {
    const unsigned long long t = thr;
#pragma unroll
    for (int q = 0; q < TOPK_CHUNK / 256; ++q) {
      const int j = base + q * 256 + tid;
      if (j < n) {
        const uint32_t id = ir ? (uint32_t)ir[j] : (uint32_t)(index_base + j);
        const unsigned long long key = ((unsigned long long)ordered_bits(vr[j]) << 32) | id;
        if (key < t) {
          const int pos = atomicAdd(&cnt, 1);
          cand[pos] = key;
        }
      }
    }
    __syncthreads();
    const int have = cnt;
    __syncthreads();  // every thread has read cnt before anyone appends again
    // the next chunk can add at most TOPK_CHUNK entries: compact when that could overflow
    if (have > TOPK_CAP - TOPK_CHUNK) {
      for (int i = have + tid; i < TOPK_CAP; i += 256) cand[i] = TOPK_INF;
      __syncthreads();
      bitonic_sort_lds(cand, TOPK_CAP, tid, 256);
      if (tid == 0) {
        cnt = have < k ? have : k;
        if (have >= k) thr = cand[k - 1];
      }
      __syncthreads();
    }
  }
  const int have = cnt;
  int span = 2;  // smallest power of two holding the survivors and the k outputs
  while (span < have || span < k) span <<= 1;
  for (int i = have + tid; i < span; i += 256) cand[i] = TOPK_INF;
  __syncthreads();
  bitonic_sort_lds(cand, span, tid, 256);
  for (int i = tid; i < k; i += 256) {
    const unsigned long long key = cand[i];
    const bool valid = i < have;
    out_val[(size_t)row * k + i] = valid ? from_ordered_bits((uint32_t)(key >> 32)) : INFINITY;
    out_idx[(size_t)row * k + i] = valid ? (int32_t)(uint32_t)(key & 0xffffffffu) : -1;
  }
}

// Short rows (<= 1024 elements, k <= 32: the distance sample, the merge of the per-shard lists, and
// nearly every candidate list): one WAVE per row, four rows per workgroup — the selection rounds
// above without the second stage, the LDS hand-over or a barrier.  With row_n, rows longer than
// 64 NQ are left to row_topk_kernel (launched right after with skip_le = 64 NQ).
template <int NQ>
__global__ __launch_bounds__(256) void row_select_wave_kernel(const float* __restrict__ vals,
                                                              const int32_t* __restrict__ idx_in,
                                                              int m, int n, size_t ld, int k,
                                                              int index_base,
                                                              float* __restrict__ out_val,
                                                              int32_t* __restrict__ out_idx,
                                                              const int* __restrict__ row_n,
                                                              const float* __restrict__ vals2) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= m) return;  // wave-uniform
  if (row_n) {
    const int have_n = row_n[row];
    n = have_n < n ? have_n : n;
    if (n > 64 * NQ) return;
  }
  const float* vr = vals + (size_t)row * ld;
  const float* vr2 = vals2 ? vals2 + (size_t)row * ld : nullptr;  // second addend (split-K halves)
  const int32_t* ir = idx_in ? idx_in + (size_t)row * ld : nullptr;
  unsigned long long mine[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int j = lane + 64 * q;
    mine[q] = TOPK_INF;
    if (j < n) {
      const uint32_t id = ir ? (uint32_t)ir[j] : (uint32_t)(index_base + j);
      const float v = vr2 ? vr[j] + vr2[j] : vr[j];
      mine[q] = ((unsigned long long)ordered_bits(v) << 32) | id;
    }
  }
  unsigned long long res = TOPK_INF;
  for (int r = 0; r < k; ++r) {
    const unsigned long long w = wave_extract_min(mine, lane);
    if (lane == r) res = w;
  }
  if (lane < k) {
    const bool valid = lane < n;
    out_val[(size_t)row * k + lane] = valid ? from_ordered_bits((uint32_t)(res >> 32)) : INFINITY;
    out_idx[(size_t)row * k + lane] = valid ? (int32_t)(uint32_t)(res & 0xffffffffu) : -1;
  }
}

// The k-th smallest VALUE of every row, k beyond the register rounds (the threshold sample of a k = 120 call: only
// sval[k - 1] is ever read): one 256-thread workgroup per row, the row (<= 256 NQ values; vals2: a second addend,
// the split-K halves) in registers as order-preserving 32-bit images, bisection over that image — at most 32
// counting rounds of NQ compares, a wave sum and one barrier — instead of the chunked bitonic sorts of
// row_topk_kernel.  Writes out_val[row * k + k - 1] (the slot a top-k launch would fill); +inf when n < k.
template <int NQ>
__global__ __launch_bounds__(256) void row_kth_bisect_kernel(const float* __restrict__ vals,
                                                             const float* __restrict__ vals2, int n, size_t ld, int k,
                                                             float* __restrict__ out_val) {
  __shared__ int s_part[2][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row = blockIdx.x;
  const float* vr = vals + row * ld;
  const float* vr2 = vals2 ? vals2 + row * ld : nullptr;
  uint32_t key[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int j = tid + 256 * q;
    key[q] = 0xffffffffu;                       // (paddings: never counted, mid < hi <= 0xffffffff)
    if (j < n) key[q] = ordered_bits(vr2 ? vr[j] + vr2[j] : vr[j]);
  }
  uint32_t lo = 0u, hi = 0xffffffffu;           // smallest T with #{key <= T} >= k; n < k: T stays the padding image
  if (n >= k) {
    for (int it = 0; lo < hi; ++it) {           // (workgroup-uniform)
      const uint32_t mid = lo + ((hi - lo) >> 1);
      int c = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) c += key[q] <= mid ? 1 : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
      if (lane == 0) s_part[it & 1][wave] = c;  // two buffers by parity: f16r_select_bisect_kernel has the argument
      __syncthreads();
      const int total = s_part[it & 1][0] + s_part[it & 1][1] + s_part[it & 1][2] + s_part[it & 1][3];
      if (total >= k) hi = mid;
      else lo = mid + 1u;
    }
  } else {
    lo = 0xffffffffu;
  }
  if (tid == 0) out_val[row * k + (k - 1)] = lo == 0xffffffffu ? INFINITY : from_ordered_bits(lo);
}

// dispatch: wave-per-row selection for short rows, one workgroup per row for the rest
// vals2 (optional, only on the wave-per-row paths: k <= 32, n <= 1024, no per-row lengths): a second
// matrix added element-wise before selecting
static void launch_row_topk(const float* vals, const int32_t* idx_in, int m, int n, size_t ld, int k,
                            int index_base, float* out_val, int32_t* out_idx, const int* row_n,
                            int* overflow, hipStream_t st, const float* vals2 = nullptr) {
  const dim3 wgrid((m + 3) / 4), block(256);
  int skip_le = -1;
  if (k <= SEL_MAX_K && !row_n && n <= 512) {
    hipLaunchKernelGGL(row_select_wave_kernel<8>, wgrid, block, 0, st, vals, idx_in, m, n, ld, k,
                       index_base, out_val, out_idx, row_n, vals2);
    return;
  }
  if (k <= SEL_MAX_K && (row_n || n <= 1024)) {
    hipLaunchKernelGGL(row_select_wave_kernel<16>, wgrid, block, 0, st, vals, idx_in, m, n, ld, k,
                       index_base, out_val, out_idx, row_n, vals2);
    if (!row_n) return;
    skip_le = 1024;  // per-row lengths: the workgroup kernel takes the rows above 1024 (and raises
                     // the overflow flag for rows beyond the capacity)
  }
  hipLaunchKernelGGL(row_topk_kernel, dim3(m), block, 0, st, vals, idx_in, n, ld, k, index_base,
                     out_val, out_idx, row_n, overflow, skip_le);
}

// thresholds of the fused paths: sval[row * k + k - 1] = the k-th smallest of a row of S sample distances (vals2: the
// second split-K half).  Up to k = 32 the sorted prefix comes from the register rounds anyway; beyond, only the k-th
// value is formed.
constexpr int KTH_MAX_N = 8192;
static void launch_sample_kth(const float* sample, const float* sample2, int m, int S, int k, float* sval,
                              int32_t* sidx, hipStream_t st) {
  if (k > SEL_MAX_K && S <= KTH_MAX_N) {
    hipLaunchKernelGGL(row_kth_bisect_kernel<KTH_MAX_N / 256>, dim3((unsigned)m), dim3(256), 0, st, sample, sample2, S,
                       (size_t)S, k, sval);
    return;
  }
  launch_row_topk(sample, nullptr, m, S, (size_t)S, k, 0, sval, sidx, nullptr, nullptr, st, sample2);
}

// ---------------------------------------------------------------------------------------------
// Full-row ranking (torch.argsort(distmat, dim=1) of the hard-negative mining samplers,
// ibl/utils/data/sampler.py:46-54, 126-135, and the unbounded prefix evaluate_all may ask for):
// stable ascending argsort of every row.  One workgroup (16 waves) per row, least-significant-digit
// radix sort over the order-preserving 32-bit image of the value, 8 bits per pass, index payload;
// stability + initial index order = ties broken by lowest index, like the top-k kernels.
// Per pass: digit histogram of the row -> bin bases; then the row is scattered chunk by chunk
// (1024 elements) in order: inside a wave the lanes holding the same digit find each other with
// eight ballots (rank = popcount of the peers below, one leader per digit publishes the count),
// a per-digit prefix over the 16 waves orders the waves, the bin base orders the chunks.
// Rows are a few hundred KB: the ping-pong buffers stay in L2.
// ---------------------------------------------------------------------------------------------
constexpr int SORT_THREADS = 1024, SORT_WAVES = SORT_THREADS / 64;

struct RowSortParams {
  const float* vals;
  size_t ld;
  int n;
  uint32_t* key[2];  // [rows][n] ping-pong
  int32_t* idx[2];
  int32_t* out_idx;  // [rows][n]
  float* out_val;    // [rows][n] or nullptr
};

__global__ __launch_bounds__(SORT_THREADS) void row_radix_sort_kernel(RowSortParams p) {
  __shared__ unsigned wave_cnt[SORT_WAVES][256];
  __shared__ unsigned bin_base[256], chunk_base[256];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const size_t row = blockIdx.x;
  const int n = p.n;
  const float* vr = p.vals + row * p.ld;
  for (int pass = 0; pass < 4; ++pass) {
    const int shift = 8 * pass;
    const uint32_t* sk = pass == 0 ? nullptr : p.key[(pass - 1) & 1] + row * n;
    const int32_t* si = pass == 0 ? nullptr : p.idx[(pass - 1) & 1] + row * n;
    uint32_t* dk = p.key[pass & 1] + row * n;
    int32_t* di = pass == 3 ? p.out_idx + row * n : p.idx[pass & 1] + row * n;
    // ---- histogram -> exclusive bin bases
    if (tid < 256) bin_base[tid] = 0;
    __syncthreads();
    for (int j = tid; j < n; j += SORT_THREADS) {
      const uint32_t k = pass == 0 ? ordered_bits(vr[j]) : sk[j];
      atomicAdd(&bin_base[(k >> shift) & 255u], 1u);
    }
    __syncthreads();
    if (wave == 0) {  // 256 bins, 4 per lane: wave-level exclusive scan
      unsigned c[4], s = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        c[q] = bin_base[4 * lane + q];
        s += c[q];
      }
      unsigned incl = s;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
      }
      unsigned run = incl - s;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bin_base[4 * lane + q] = run;
        run += c[q];
      }
    }
    __syncthreads();
    // ---- ordered scatter, 1024 elements at a time
    for (int j0 = 0; j0 < n; j0 += SORT_THREADS) {
      for (int i = tid; i < SORT_WAVES * 256; i += SORT_THREADS) (&wave_cnt[0][0])[i] = 0;
      __syncthreads();
      const int j = j0 + tid;
      const bool act = j < n;
      uint32_t k = 0;
      int32_t id = 0;
      if (act) {
        k = pass == 0 ? ordered_bits(vr[j]) : sk[j];
        id = pass == 0 ? j : si[j];
      }
      const unsigned dg = (k >> shift) & 255u;
      unsigned long long peers = __ballot(act);
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        const unsigned long long m = __ballot(act && ((dg >> b) & 1u));
        peers &= ((dg >> b) & 1u) ? m : ~m;
      }
      const unsigned long long below = peers & ((1ull << lane) - 1ull);
      const unsigned rank = (unsigned)__popcll(below);
      if (act && below == 0) wave_cnt[wave][dg] = (unsigned)__popcll(peers);
      __syncthreads();
      if (tid < 256) {
        unsigned run = 0;
#pragma unroll
        for (int w = 0; w < SORT_WAVES; ++w) {
          const unsigned c = wave_cnt[w][tid];
          wave_cnt[w][tid] = run;
          run += c;
        }
        chunk_base[tid] = bin_base[tid];
        bin_base[tid] += run;
      }
      __syncthreads();
      if (act) {
        const unsigned pos = chunk_base[dg] + wave_cnt[wave][dg] + rank;
        di[pos] = id;
        if (pass < 3)
          dk[pos] = k;
        else if (p.out_val)
          p.out_val[row * n + pos] = from_ordered_bits(k);
      }
      __syncthreads();
    }
    __threadfence_block();
  }
}

// ---------------------------------------------------------------------------------------------
// Recall counting on the device (evaluate_all / spatial_nms, ibl/evaluators.py:132-160).
// Per query: the rank, inside the prediction list the reference would build, of the first
// prediction that is a ground-truth neighbour (-1: none).  Recall@N for every N follows on the
// host from these m integers (query q counts for N iff 0 <= rank < N).
//   topk [m][k]     ranked gallery positions (ascending distance), -1 = padding
//   gt_off [m+1], gt_val [nnz]   ground truth as CSR (gallery positions per query)
//   pids != NULL    spatial NMS: only the first nms_window predictions are considered and a
//                   prediction whose pid already occurred earlier in the list is dropped
//                   (evaluators.py:132-140 keeps the first occurrence); the rank is the position
//                   among the kept ones.
// One wave per query, 64 predictions per pass.
// ---------------------------------------------------------------------------------------------
constexpr int RANK_MAXK = 1024;
__global__ __launch_bounds__(256) void first_hit_rank_kernel(const int32_t* __restrict__ topk, int m,
                                                             int k, const int32_t* __restrict__ gt_off,
                                                             const int32_t* __restrict__ gt_val,
                                                             const int32_t* __restrict__ pids,
                                                             int nms_window,
                                                             int32_t* __restrict__ out_rank) {
  __shared__ int32_t s_pid[4][RANK_MAXK];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int q = blockIdx.x * 4 + wv;
  if (q >= m) return;
  const int32_t* pred = topk + (size_t)q * k;
  const int window = pids ? (k < nms_window ? k : nms_window) : k;
  const int g0 = gt_off[q], g1 = gt_off[q + 1];
  if (pids) {
    for (int j = lane; j < window; j += 64) {
      const int32_t id = pred[j];
      s_pid[wv][j] = id >= 0 ? pids[id] : -1;
    }
    __builtin_amdgcn_wave_barrier();
  }
  int kept_before = 0, result = -1;
  for (int j0 = 0; j0 < window && result < 0; j0 += 64) {
    const int j = j0 + lane;
    const int32_t id = j < window ? pred[j] : -1;
    bool keep = id >= 0;
    if (pids && keep) {
      const int32_t mine = s_pid[wv][j];
      for (int i = 0; i < j; ++i)
        if (s_pid[wv][i] == mine && pred[i] >= 0) {
          keep = false;
          break;
        }
    }
    bool hit = false;
    if (keep)
      for (int t = g0; t < g1; ++t)
        if (gt_val[t] == id) {
          hit = true;
          break;
        }
    const unsigned long long keep_m = __ballot(keep), hit_m = __ballot(hit);
    if (hit_m) {
      const int first = __ffsll((long long)hit_m) - 1;
      result = kept_before + __popcll(keep_m & ((1ull << first) - 1ull));
    }
    kept_before += __popcll(keep_m);
  }
  if (lane == 0) out_rank[q] = result;
}

}  // namespace oibl

#include "match_f16r.h"   // fp16 filter + exact rescoring (uses ordered_bits and the ring kernel's tile order)

using namespace oibl;

extern "C" {

OIBL_HOOK(int, g_match_ring, 1);  // test hook: 0 = never, 1 = auto, 2 = whenever legal
OIBL_HOOK(int, g_match_group, 4);  // test hook: query tiles per ordering group of the ring kernel
OIBL_HOOK(int, g_match_splitk, 1);  // test hook: 0 = never split the threshold sample's contraction
// f16mx distances: 1 = LDS-DMA issue in the LOAD segments (as the convolutions; the default since the one-barrier
// schedule: 7.72-7.76 ms against 7.85-7.93 for 8192 x 81920 x 4096 + top-10, same bits), 0 = inside COMPUTE
OIBL_HOOK(int, g_match_mx_early, 1);
OIBL_HOOK(int, g_match_bar1, 1);    // test hook: 0 = two barriers per phase in the ring kernel (ring_core.h, BAR1: 1 = 0-2.5 % faster)

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_match_splitk(int on) {
  g_match_splitk = on ? 1 : 0;
  return OIBL_OK;
}
int oibl_debug_set_match_bar1(int on) {
  g_match_bar1 = on ? 1 : 0;
  return OIBL_OK;
}
int oibl_debug_set_match_mx_early(int on) {
  g_match_mx_early = on ? 1 : 0;
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_match_group(int g) {
  g_match_group = g < 1 ? 1 : g;
  return OIBL_OK;
}
#endif

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_match_ring(int mode) {
  g_match_ring = mode;
  return OIBL_OK;
}
#endif

static size_t pw_off_yn(int m) { return align_up((size_t)m * sizeof(float), 256); }
static size_t pw_off_xt(int m, int n) { return pw_off_yn(m) + align_up((size_t)n * sizeof(float), 256); }

// f16mx needs K-tiles of 32 elements in even number >= 4 (d % 64 == 0, d >= 128); the one legal width
// below that (d = 64) is served in bf16x3 — same element size, more accurate, never on the hot path
static int eff_precision(int precision, int d) {
  return precision == OIBL_F16MX && d < 128 ? OIBL_BF16X3 : precision;
}
static bool st_ok(int st) { return st == OIBL_ST_F32 || st == OIBL_ST_F16 || st == OIBL_ST_BF16; }
// bytes of the operand copy the contraction reads instead of the stored rows (0: reads them as is)
static size_t pw_copy_bytes(int rows, int d, int precision, int st) {
  if (precision == OIBL_BF16) return st == OIBL_ST_BF16 ? 0 : align_up((size_t)rows * d * 2, 256);
  if (precision == OIBL_BF16X3 || precision == OIBL_F16MX) return align_up((size_t)rows * d * 4, 256);  // split rows
  return st == OIBL_ST_F32 ? 0 : align_up((size_t)rows * d * 4, 256);
}

size_t oibl_pairwise_st_workspace_bytes(int m, int n, int d, int precision, int x_st, int y_st) {
  precision = eff_precision(precision, d);
  if (m <= 0 || n <= 0 || d <= 0 || !st_ok(x_st) || !st_ok(y_st)) return 0;
  return pw_off_xt(m, n) + pw_copy_bytes(m, d, precision, x_st) + pw_copy_bytes(n, d, precision, y_st);
}
size_t oibl_pairwise_workspace_bytes(int m, int n, int d, int precision) {
  return oibl_pairwise_st_workspace_bytes(m, n, d, precision, OIBL_ST_F32, OIBL_ST_F32);
}

// ring kernel legality: bf16, an even number (>= 4) of 64-element K-tiles, 32-bit buffer offsets
// (es = bytes per operand element: 2 bf16, 4 bf16x3 with 32-element K-tiles)
static bool pair_ring_legal(int m, int n, int d, int es = 2) {
  const int ksteps = d * es / 128;
  return d % 64 == 0 && ksteps >= 4 && (ksteps & 1) == 0 && (size_t)m * d * es < (size_t)0xE0000000u &&
         (size_t)n * d * es < (size_t)0xE0000000u && n <= (1 << 20) && !g_regstage;
}
static bool mfma16(int precision) {
  return precision == OIBL_BF16 || precision == OIBL_BF16X3 || precision == OIBL_F16MX;
}

static int opnd_es(int precision) { return precision == OIBL_BF16 ? 2 : 4; }
static bool pair_ring_wanted(int m, int n, int d, int es = 2) {
  if (!g_match_ring || !pair_ring_legal(m, n, d, es)) return false;
  // below ~64 tiles of 256 x 256 the 128 x 128 kernel fills the chip better
  return g_match_ring == 2 || (long)((m + 255) / 256) * ((n + 255) / 256) >= 64;
}

extern "C++" {
template <bool FILTER, int P = RING_BF16>
static int launch_pairwise_ring(PairRingParams& p, hipStream_t st, int ksplit = 1) {
  p.tiles_m = (p.m + 255) / 256;
  p.tiles_n = (p.n + 255) / 256;
  p.group_m = g_match_group;
  const long grid = (long)p.tiles_m * p.tiles_n;
  OIBL_REQUIRE(grid > 0 && grid <= 0x7fffffffL, "pairwise: grid out of range");
  constexpr int lds = RingGeo<2>::MAIN_LDS;
  if (g_match_bar1) {   // one barrier per phase (ring_core.h, BAR1)
    auto kern = pairwise_ring_kernel<FILTER, P, true>;
    OIBL_SET_MAX_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)ksplit), dim3(512), lds, st, p);
  } else {
    auto kern = pairwise_ring_kernel<FILTER, P>;
    OIBL_SET_MAX_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)ksplit), dim3(512), lds, st, p);
  }
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}
}  // extern "C++"

// norms + operand copies of one descriptor matrix into the workspace
extern "C++" {
template <int ST>
static int prepare_rows(const void* x, int rows, int d, int precision, float* norms, void* copy,
                        const void** opnd, hipStream_t st) {
  const dim3 grid((rows + 3) / 4), block(256);
  if (precision == OIBL_BF16) {
    hipLaunchKernelGGL((row_sqnorm_cast_kernel<ST, 1>), grid, block, 0, st, x, norms, copy, rows, d);
    *opnd = ST == OIBL_ST_BF16 ? x : copy;
  } else if (precision == OIBL_BF16X3) {
    hipLaunchKernelGGL((row_sqnorm_cast_kernel<ST, 3>), grid, block, 0, st, x, norms, copy, rows, d);
    *opnd = copy;
  } else if (precision == OIBL_F16MX) {
    hipLaunchKernelGGL((row_sqnorm_mx_kernel<ST>), grid, block, 0, st, x, norms, copy, rows, d);
    *opnd = copy;
  } else if (ST == OIBL_ST_F32) {
    hipLaunchKernelGGL(row_sqnorm_kernel, grid, block, 0, st, (const float*)x, norms, rows, d);
    *opnd = x;
  } else {
    hipLaunchKernelGGL((row_sqnorm_cast_kernel<ST, 2>), grid, block, 0, st, x, norms, copy, rows, d);
    *opnd = copy;
  }
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}
}  // extern "C++"
static int prepare_rows_st(const void* x, int x_st, int rows, int d, int precision, float* norms,
                           void* copy, const void** opnd, hipStream_t st) {
  switch (x_st) {
    case OIBL_ST_F32: return prepare_rows<OIBL_ST_F32>(x, rows, d, precision, norms, copy, opnd, st);
    case OIBL_ST_F16: return prepare_rows<OIBL_ST_F16>(x, rows, d, precision, norms, copy, opnd, st);
    default: return prepare_rows<OIBL_ST_BF16>(x, rows, d, precision, norms, copy, opnd, st);
  }
}
// fills xo / yo (what the contraction reads) and xn / yn (fp32 squared norms)
static int pairwise_prepare(const void* x, int x_st, int m, const void* y, int y_st, int n, int d,
                            int precision, char* wsb, const void** xo, const void** yo, float** xn,
                            float** yn, void* stream) {
  hipStream_t st = (hipStream_t)stream;
  *xn = (float*)wsb;
  *yn = (float*)(wsb + pw_off_yn(m));
  char* xc = wsb + pw_off_xt(m, n);
  char* yc = xc + pw_copy_bytes(m, d, precision, x_st);
  int rc = prepare_rows_st(x, x_st, m, d, precision, *xn, xc, xo, st);
  if (rc) return rc;
  return prepare_rows_st(y, y_st, n, d, precision, *yn, yc, yo, st);
}

// dist[m rows starting at row0][ldd] from prepared operands
static int pairwise_launch(const void* xo, const void* yo, const float* xn, const float* yn, int row0,
                           int rows, int m_all, int n, int d, int precision, float* dist, size_t ldd,
                           hipStream_t st) {
  const size_t es = oibl_elem_size(precision);
  if (precision == OIBL_F16MX) {
    // the ring kernel is the only f16mx implementation: panels of rows / columns keep every launch inside
    // the 32-bit buffer offsets of its operand loaders
    OIBL_REQUIRE(ldd <= ((size_t)1 << 24), "pairwise (f16mx): row stride %zu above 2^24", ldd);
    long panel = (long)(((size_t)0xE0000000u / ((size_t)d * 4) - 1) / 256 * 256);
    if (panel > (1 << 20)) panel = 1 << 20;
    for (long r0 = 0; r0 < rows; r0 += panel)
      for (long c0 = 0; c0 < n; c0 += panel) {
        const int pr = (int)(rows - r0 < panel ? rows - r0 : panel), pc = (int)(n - c0 < panel ? n - c0 : panel);
        PairRingParams q = {};
        q.x = (const char*)xo + (size_t)(row0 + r0) * d * 4;
        q.y = (const char*)yo + (size_t)c0 * d * 4;
        q.xn = xn + row0 + r0;
        q.yn = yn + c0;
        q.dist = dist + (size_t)r0 * ldd + c0;
        q.ldd = ldd;
        q.x_bytes = (unsigned)((size_t)pr * d * 4);
        q.y_bytes = (unsigned)((size_t)pc * d * 4);
        q.y_row_bytes = (long)d * 4;
        q.yn_stride = 1;
        q.m = pr;
        q.n = pc;
        q.d = d;
        const int rc = g_match_mx_early ? launch_pairwise_ring<false, RING_MX_EARLY>(q, st)
                                        : launch_pairwise_ring<false, RING_MX>(q, st);
        if (rc) return rc;
      }
    return OIBL_OK;
  }
  if (mfma16(precision) && pair_ring_wanted(rows, n, d, (int)es) && ldd <= ((size_t)1 << 24)) {
    PairRingParams q = {};
    q.x = (const char*)xo + (size_t)row0 * d * es;
    q.y = yo;
    q.xn = xn + row0;
    q.yn = yn;
    q.dist = dist;
    q.ldd = ldd;
    q.x_bytes = (unsigned)((size_t)rows * d * es);
    q.y_bytes = (unsigned)((size_t)n * d * es);
    q.y_row_bytes = (long)d * es;
    q.yn_stride = 1;
    q.m = rows;
    q.n = n;
    q.d = d;
    return precision == OIBL_BF16X3 ? launch_pairwise_ring<false, RING_X3>(q, st)
                                    : launch_pairwise_ring<false>(q, st);
  }
  PairParams p;
  p.x = (const char*)xo + (size_t)row0 * d * es;
  p.y = yo;
  p.xn = xn + row0;
  p.yn = yn;
  p.dist = dist;
  p.ldd = ldd;
  p.m = rows;
  p.n = n;
  p.d = d;
  p.tiles_n = (n + 127) / 128;
  (void)m_all;
  const long grid = (long)((rows + 127) / 128) * p.tiles_n;
  OIBL_REQUIRE(grid <= 0x7fffffffL, "pairwise: grid too large");
  if (precision == OIBL_BF16) {
    using Cfg = GemmCfg<bf16_t, 2, 2, 2, 2>;
    if (g_regstage)
      hipLaunchKernelGGL((pairwise_kernel<Cfg, false>), dim3((unsigned)grid), dim3(Cfg::NTHREADS),
                         Cfg::MAIN_LDS_BYTES, st, p);
    else
      hipLaunchKernelGGL((pairwise_kernel<Cfg, true>), dim3((unsigned)grid), dim3(Cfg::NTHREADS),
                         Cfg::MAIN_LDS_BYTES, st, p);
  } else if (precision == OIBL_BF16X3) {
    using Cfg = GemmCfg<bf16x3_t, 2, 2, 2, 2>;
    if (g_regstage)
      hipLaunchKernelGGL((pairwise_kernel<Cfg, false>), dim3((unsigned)grid), dim3(Cfg::NTHREADS),
                         Cfg::MAIN_LDS_BYTES, st, p);
    else
      hipLaunchKernelGGL((pairwise_kernel<Cfg, true>), dim3((unsigned)grid), dim3(Cfg::NTHREADS),
                         Cfg::MAIN_LDS_BYTES, st, p);
  } else {
    using Cfg = GemmCfg<float, 2, 2, 2, 2>;
    if (g_regstage)
      hipLaunchKernelGGL((pairwise_kernel<Cfg, false>), dim3((unsigned)grid), dim3(Cfg::NTHREADS),
                         Cfg::MAIN_LDS_BYTES, st, p);
    else
      hipLaunchKernelGGL((pairwise_kernel<Cfg, true>), dim3((unsigned)grid), dim3(Cfg::NTHREADS),
                         Cfg::MAIN_LDS_BYTES, st, p);
  }
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_pairwise_sqdist_st(const void* x, int x_st, int m, const void* y, int y_st, int n, int d,
                            int precision, float* dist, size_t ldd, void* ws, size_t ws_bytes,
                            void* stream) {
  precision = eff_precision(precision, d);
  OIBL_REQUIRE(x && y && dist && ws, "pairwise: null pointer");
  OIBL_REQUIRE(mfma16(precision) || precision == OIBL_F32, "pairwise: bad precision %d", precision);
  OIBL_REQUIRE(st_ok(x_st) && st_ok(y_st), "pairwise: bad storage type %d / %d", x_st, y_st);
  OIBL_REQUIRE(m > 0 && n > 0 && d > 0 && d % 64 == 0 && ldd >= (size_t)n,
               "pairwise: unsupported shape m=%d n=%d d=%d ldd=%zu", m, n, d, ldd);
  OIBL_REQUIRE((uintptr_t)ws % 256 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0,
               "pairwise: workspace must be 256-byte, x and y 16-byte aligned");
  const size_t need = oibl_pairwise_st_workspace_bytes(m, n, d, precision, x_st, y_st);
  if (ws_bytes < need) {
    set_error("pairwise: workspace %zu < required %zu bytes", ws_bytes, need);
    return OIBL_E_WORKSPACE;
  }
  const void *xo, *yo;
  float *xn, *yn;
  int rc = pairwise_prepare(x, x_st, m, y, y_st, n, d, precision, (char*)ws, &xo, &yo, &xn, &yn, stream);
  if (rc) return rc;
  return pairwise_launch(xo, yo, xn, yn, 0, m, m, n, d, precision, dist, ldd, (hipStream_t)stream);
}
int oibl_pairwise_sqdist(const float* x, int m, const float* y, int n, int d, int precision,
                         float* dist, size_t ldd, void* ws, size_t ws_bytes, void* stream) {
  return oibl_pairwise_sqdist_st(x, OIBL_ST_F32, m, y, OIBL_ST_F32, n, d, precision, dist, ldd, ws,
                                 ws_bytes, stream);
}

// ---- fused distance + top-k -------------------------------------------------------------------
// Plan of one call (all sizes derived from m, n, d, k, precision only, so that the workspace query
// and the call agree):
//   fused (bf16, ring kernel legal, gallery large enough to sample):
//     S = sample size, cap = candidate capacity per query
//   exact: distance tiles of `chunk` query rows are materialised in the workspace and reduced
//     with row_topk (the only path in fp32 mode; the fallback when a candidate list overflows).
struct TopkPlan {
  bool fused;
  int ksplit;  // 2: the threshold sample is contracted in two K-halves on twice the workgroups
  int S, stride, cap, chunk;
  size_t off_prep, off_sample, off_sval, off_sidx, off_cnt, off_cval, off_cidx, off_chunk, total;
};
static TopkPlan topk_plan(int m, int n, int d, int k, int precision, size_t prep_bytes) {
  TopkPlan t = {};
  // Sample size: the k-th smallest of S sampled distances lets ~ k n / S gallery rows through the
  // filter.  1024 rows keep the survivor density near 1 % of a tile (the filter epilogue's per-lane
  // lists are sized for that; a 256-row sample on a 10k-row shard — 4 % — slowed the contraction
  // from 1.2 to 0.93 PFLOP/s and saved nothing: the sample pass costs the latency of one 256 x 256
  // tile over K = 4096 whatever S is); galleries beyond 82k rows sample 2048 so that the candidate
  // lists stay on the register selection path of row_topk.
  // k > SEL_MAX_K (the 120 ranks of spatial NMS): neither the sample's top-k nor the candidate lists are on the
  // register paths anyway, and the survivor density is what costs — the sample may grow to 8192 rows (k = 120 on a
  // 76k-row gallery: 1.5 % of a tile survives instead of 5.8 %, for a sample pass of a tenth of the filter pass) as
  // long as the gallery stays 16 samples deep.
  int S = 1024;
  const int s_max = k > SEL_MAX_K ? 8192 : SEL_MAX_N;
  while ((S < 4 * k || (long)S * 800 < (long)k * n) && S < s_max && (k <= SEL_MAX_K || (long)16 * S <= n)) S *= 2;
  t.fused = mfma16(precision) && g_match_ring && pair_ring_legal(m, n, d, opnd_es(precision)) && n >= 8 * S &&
            S >= 4 * k && (long)((m + 255) / 256) * ((n + 255) / 256) >= 64;
  t.S = S;
  // The sample pass is one 256 x 256 tile over the whole K per workgroup: with m/256 * S/256 <= 128
  // tiles half of the CUs idle for its ~93 us.  Two K-halves on twice the workgroups halve that
  // latency; the halves are added in the selection (wave-per-row paths only) and the thresholds get
  // the slack that covers the changed summation order.
  const int es_ = opnd_es(precision);
  const int kt_half = d * es_ / 256;      // K-tiles per half
  // (k <= 32: the halves are added by the wave-per-row selection, rows up to 1024; beyond: by row_kth_bisect_kernel)
  t.ksplit = (g_match_splitk && t.fused && (k <= SEL_MAX_K ? S <= 1024 : S <= KTH_MAX_N) && d % 128 == 0 &&
              kt_half >= 4 && (kt_half & 1) == 0 && (long)((m + 255) / 256) * (S / 256) <= 128) ? 2 : 1;
  t.stride = n / S;                       // sample = gallery rows 0, stride, 2 stride, ...
  const long expect = (long)k * t.stride + k;  // ~ n k / S survivors per query
  long cap = 4096;
  while (cap < 6 * expect) cap *= 2;
  t.cap = (int)cap;
  long chunk = ((long)1 << 28) / n;       // <= 1 GiB of distances at a time
  if (chunk < 1) chunk = 1;
  if (chunk > m) chunk = m;
  t.chunk = (int)chunk;
  size_t o = 0;
  t.off_prep = o;
  o += align_up(prep_bytes, 256);
  t.off_sample = o;
  o += align_up((size_t)m * S * sizeof(float), 256) * 2;   // two halves when the sample is split
  t.off_sval = o;
  o += align_up((size_t)m * k * sizeof(float), 256);
  t.off_sidx = o;
  o += align_up((size_t)m * k * sizeof(int32_t), 256);
  t.off_cnt = o;
  o += align_up((size_t)(m + 1) * sizeof(int), 256);
  t.off_cval = o;
  o += align_up((size_t)m * t.cap * sizeof(float), 256);
  t.off_cidx = o;
  o += align_up((size_t)m * t.cap * sizeof(int32_t), 256);
  t.off_chunk = o;
  o += align_up((size_t)t.chunk * n * sizeof(float), 256);
  t.total = o;
  return t;
}

size_t oibl_sqdist_topk_st_workspace_bytes(int m, int n, int d, int k, int precision, int x_st,
                                           int y_st) {
  precision = eff_precision(precision, d);
  if (m <= 0 || n <= 0 || d <= 0 || k <= 0 || !st_ok(x_st) || !st_ok(y_st)) return 0;
  return topk_plan(m, n, d, k, precision, oibl_pairwise_st_workspace_bytes(m, n, d, precision, x_st, y_st)).total;
}
size_t oibl_sqdist_topk_workspace_bytes(int m, int n, int d, int k, int precision) {
  return oibl_sqdist_topk_st_workspace_bytes(m, n, d, k, precision, OIBL_ST_F32, OIBL_ST_F32);
}

int oibl_sqdist_topk(const float* x, int m, const float* y, int n, int d, int k, int index_base,
                     int precision, int exact, float* out_val, int32_t* out_idx, int32_t* overflow,
                     void* ws, size_t ws_bytes, void* stream) {
  return oibl_sqdist_topk_st(x, OIBL_ST_F32, m, y, OIBL_ST_F32, n, d, k, index_base, precision, exact,
                             out_val, out_idx, overflow, ws, ws_bytes, stream);
}

// everything behind the norm / operand preparation: xo, yo are the rows the contraction reads
// (bf16, (hi, lo) groups or fp32 by precision), xn, yn their fp32 squared norms
static int sqdist_topk_core(const void* xo, const float* xn, int m, const void* yo, const float* yn,
                            int n, int d, int k, int index_base, int precision, int exact,
                            float* out_val, int32_t* out_idx, int32_t* overflow, char* wsb,
                            const TopkPlan& t, hipStream_t st) {
  int rc;
  if (overflow) OIBL_HIP_CHECK(hipMemsetAsync(overflow, 0, sizeof(int32_t), st));
  if (t.fused && !exact) {
    // 1. thresholds: k-th smallest distance to a strided sample of S gallery rows
    float* sample = (float*)(wsb + t.off_sample);
    float* sval = (float*)(wsb + t.off_sval);
    int32_t* sidx = (int32_t*)(wsb + t.off_sidx);
    PairRingParams q = {};
    q.x = xo;
    q.y = yo;
    q.xn = xn;
    q.yn = yn;
    q.dist = sample;
    q.ldd = (size_t)t.S;
    const bool x3 = precision == OIBL_BF16X3, mx = precision == OIBL_F16MX;
    const size_t es = (size_t)opnd_es(precision);
    q.x_bytes = (unsigned)((size_t)m * d * es);
    q.y_bytes = (unsigned)((size_t)n * d * es);
    q.y_row_bytes = (long)d * es * t.stride;
    q.yn_stride = t.stride;
    q.m = m;
    q.n = t.S;
    q.d = d;
    int* cnt = (int*)(wsb + t.off_cnt);
    OIBL_HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)(m + 1) * sizeof(int), st));  // counts + the yn_max slot
    const size_t half = align_up((size_t)m * t.S * sizeof(float), 256) / sizeof(float);
    if (t.ksplit == 2) {
      q.part_stride = half;
      q.yn_max = (unsigned*)(cnt + m);
    }
    rc = mx   ? (g_match_mx_early ? launch_pairwise_ring<false, RING_MX_EARLY>(q, st, t.ksplit)
                                  : launch_pairwise_ring<false, RING_MX>(q, st, t.ksplit))
         : x3 ? launch_pairwise_ring<false, RING_X3>(q, st, t.ksplit)
              : launch_pairwise_ring<false>(q, st, t.ksplit);
    if (rc) return rc;
    launch_sample_kth(sample, t.ksplit == 2 ? sample + half : nullptr, m, t.S, k, sval, sidx, st);
    OIBL_LAUNCH_CHECK();
    // 2. the full contraction, keeping only distances <= threshold (+ the slack of a split sample:
    //    a bound on what fp32 accumulation in another order can move a distance)
    q.part_stride = 0;
    // (each of the two orders is within K' 2^-24 |x||y| of the exact dot product, K' = d products —
    //  3 d in bf16x3 — so they differ by at most twice that, and a distance by twice that again)
    q.thr_slack = t.ksplit == 2 ? 4.0f * (float)d * ((x3 || mx) ? 3.0f : 1.0f) * 5.9604645e-8f : 0.f;
    q.dist = nullptr;
    q.y_row_bytes = (long)d * es;
    q.yn_stride = 1;
    q.n = n;
    q.thr = sval + (k - 1);
    q.thr_stride = k;
    q.cand_val = (float*)(wsb + t.off_cval);
    q.cand_idx = (int32_t*)(wsb + t.off_cidx);
    q.cand_cnt = cnt;
    q.cap = t.cap;
    q.index_base = index_base;
    q.index_stride = 1;
    rc = mx   ? (g_match_mx_early ? launch_pairwise_ring<true, RING_MX_EARLY>(q, st)
                                  : launch_pairwise_ring<true, RING_MX>(q, st))
         : x3 ? launch_pairwise_ring<true, RING_X3>(q, st)
              : launch_pairwise_ring<true>(q, st);
    if (rc) return rc;
    // 3. exact top-k of every candidate list ((value, index) keys: independent of append order)
    launch_row_topk(q.cand_val, q.cand_idx, m, t.cap, (size_t)t.cap, k, 0, out_val, out_idx, cnt,
                    (int*)overflow, st);
    OIBL_LAUNCH_CHECK();
    return OIBL_OK;
  }
  float* tile = (float*)(wsb + t.off_chunk);
  for (int r0 = 0; r0 < m; r0 += t.chunk) {
    const int rows = m - r0 < t.chunk ? m - r0 : t.chunk;
    rc = pairwise_launch(xo, yo, xn, yn, r0, rows, m, n, d, precision, tile, (size_t)n, st);
    if (rc) return rc;
    launch_row_topk(tile, nullptr, rows, n, (size_t)n, k, index_base, out_val + (size_t)r0 * k,
                    out_idx + (size_t)r0 * k, nullptr, nullptr, st);
    OIBL_LAUNCH_CHECK();
  }
  return OIBL_OK;
}

static int topk_args_ok(int m, int n, int d, int k, int index_base, int precision) {
  OIBL_REQUIRE(mfma16(precision) || precision == OIBL_F32, "sqdist_topk: bad precision %d", precision);
  OIBL_REQUIRE(m > 0 && n > 0 && d > 0 && d % 64 == 0, "sqdist_topk: unsupported shape m=%d n=%d d=%d",
               m, n, d);
  OIBL_REQUIRE(k >= 1 && k <= 1024, "sqdist_topk: k=%d outside [1, 1024]", k);
  OIBL_REQUIRE((long)index_base + n <= 0x7fffffffL, "sqdist_topk: index_base + n overflows int32");
  return OIBL_OK;
}

int oibl_sqdist_topk_st(const void* x, int x_st, int m, const void* y, int y_st, int n, int d, int k,
                        int index_base, int precision, int exact, float* out_val, int32_t* out_idx,
                        int32_t* overflow, void* ws, size_t ws_bytes, void* stream) {
  precision = eff_precision(precision, d);
  OIBL_REQUIRE(x && y && out_val && out_idx && ws, "sqdist_topk: null pointer");
  OIBL_REQUIRE(st_ok(x_st) && st_ok(y_st), "sqdist_topk: bad storage type %d / %d", x_st, y_st);
  int rc = topk_args_ok(m, n, d, k, index_base, precision);
  if (rc) return rc;
  OIBL_REQUIRE((uintptr_t)ws % 256 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0,
               "sqdist_topk: workspace must be 256-byte, x and y 16-byte aligned");
  const TopkPlan t =
      topk_plan(m, n, d, k, precision, oibl_pairwise_st_workspace_bytes(m, n, d, precision, x_st, y_st));
  if (ws_bytes < t.total) {
    set_error("sqdist_topk: workspace %zu < required %zu bytes", ws_bytes, t.total);
    return OIBL_E_WORKSPACE;
  }
  char* wsb = (char*)ws;
  const void *xo, *yo;
  float *xn, *yn;
  rc = pairwise_prepare(x, x_st, m, y, y_st, n, d, precision, wsb + t.off_prep, &xo, &yo, &xn, &yn,
                        stream);
  if (rc) return rc;
  return sqdist_topk_core(xo, xn, m, yo, yn, n, d, k, index_base, precision, exact, out_val, out_idx,
                          overflow, wsb, t, (hipStream_t)stream);
}

// ---- prepared operands: a gallery (or query set) that is matched many times ----------------------
size_t oibl_match_operand_bytes(int rows, int d, int precision, int st) {
  precision = eff_precision(precision, d);
  if (rows <= 0 || d <= 0 || !st_ok(st)) return 0;
  return pw_copy_bytes(rows, d, precision, st);
}

int oibl_match_prepare(const void* x, int x_st, int rows, int d, int precision, float* norms,
                       void* operand, void* stream) {
  precision = eff_precision(precision, d);
  OIBL_REQUIRE(x && norms, "match_prepare: null pointer");
  OIBL_REQUIRE(mfma16(precision) || precision == OIBL_F32, "match_prepare: bad precision %d", precision);
  OIBL_REQUIRE(st_ok(x_st), "match_prepare: bad storage type %d", x_st);
  OIBL_REQUIRE(rows > 0 && d > 0 && d % 64 == 0, "match_prepare: unsupported shape rows=%d d=%d", rows, d);
  OIBL_REQUIRE(operand || pw_copy_bytes(rows, d, precision, x_st) == 0,
               "match_prepare: this precision / storage pair needs an operand buffer");
  OIBL_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)operand % 16 == 0, "match_prepare: unaligned pointer");
  const void* opnd;
  return prepare_rows_st(x, x_st, rows, d, precision, norms, operand, &opnd, (hipStream_t)stream);
}

size_t oibl_sqdist_topk_prepared_workspace_bytes(int m, int n, int d, int k, int precision) {
  precision = eff_precision(precision, d);
  if (m <= 0 || n <= 0 || d <= 0 || k <= 0) return 0;
  return topk_plan(m, n, d, k, precision, 0).total;
}

int oibl_sqdist_topk_prepared(const void* xo, const float* xn, int m, const void* yo, const float* yn,
                              int n, int d, int k, int index_base, int precision, int exact,
                              float* out_val, int32_t* out_idx, int32_t* overflow, void* ws,
                              size_t ws_bytes, void* stream) {
  precision = eff_precision(precision, d);
  OIBL_REQUIRE(xo && xn && yo && yn && out_val && out_idx && ws, "sqdist_topk_prepared: null pointer");
  int rc = topk_args_ok(m, n, d, k, index_base, precision);
  if (rc) return rc;
  OIBL_REQUIRE((uintptr_t)ws % 256 == 0 && (uintptr_t)xo % 16 == 0 && (uintptr_t)yo % 16 == 0,
               "sqdist_topk_prepared: workspace must be 256-byte, operands 16-byte aligned");
  const TopkPlan t = topk_plan(m, n, d, k, precision, 0);
  if (ws_bytes < t.total) {
    set_error("sqdist_topk_prepared: workspace %zu < required %zu bytes", ws_bytes, t.total);
    return OIBL_E_WORKSPACE;
  }
  return sqdist_topk_core(xo, xn, m, yo, yn, n, d, k, index_base, precision, exact, out_val, out_idx,
                          overflow, (char*)ws, t, (hipStream_t)stream);
}

// ---- f16r: fp16 filter pass + exact rescoring (match_f16r.h) --------------------------------------
int oibl_match_prepare_f16r_st(const void* x, int x_st, int rows, int d, float* norms, float* aux, void* rows_f16,
                               void* stream) {
  OIBL_REQUIRE(x && norms && aux && rows_f16, "match_prepare_f16r: null pointer");
  OIBL_REQUIRE(st_ok(x_st), "match_prepare_f16r: bad storage type %d", x_st);
  OIBL_REQUIRE(rows > 0 && d > 0 && d % 64 == 0, "match_prepare_f16r: unsupported shape rows=%d d=%d", rows, d);
  OIBL_REQUIRE((uintptr_t)x % 16 == 0 && (uintptr_t)rows_f16 % 16 == 0 && (uintptr_t)aux % 16 == 0,
               "match_prepare_f16r: x, rows_f16 and aux must be 16-byte aligned");
  const dim3 grid((rows + 3) / 4), block(256);
  hipStream_t st = (hipStream_t)stream;
  switch (x_st) {
    case OIBL_ST_F32:
      hipLaunchKernelGGL(f16r_prepare_kernel<OIBL_ST_F32>, grid, block, 0, st, x, norms, (float4*)aux,
                         (uint16_t*)rows_f16, rows, d);
      break;
    case OIBL_ST_F16:
      hipLaunchKernelGGL(f16r_prepare_kernel<OIBL_ST_F16>, grid, block, 0, st, x, norms, (float4*)aux,
                         (uint16_t*)rows_f16, rows, d);
      break;
    default:
      hipLaunchKernelGGL(f16r_prepare_kernel<OIBL_ST_BF16>, grid, block, 0, st, x, norms, (float4*)aux,
                         (uint16_t*)rows_f16, rows, d);
  }
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}
int oibl_match_prepare_f16r(const float* x, int rows, int d, float* norms, float* aux, void* rows_f16,
                            void* stream) {
  return oibl_match_prepare_f16r_st(x, OIBL_ST_F32, rows, d, norms, aux, rows_f16, stream);
}

// member slots per query for the rescoring: 32 up to k = 16, 2k + 32 beyond (1024 at most: the rescoring's LDS window;
// the fused path needs the whole 2k + 32, the exact path works with what it gets)
static int f16r_k2(int k) { return k <= 16 ? 32 : (2 * k + 32 <= F16R_MAX_K2 ? 2 * k + 32 : F16R_MAX_K2); }

struct F16rPlan {
  TopkPlan t;
  bool fused;
  int K2;
  size_t off_lval, off_lidx, off_ymax, total;
};
static F16rPlan f16r_plan(int m, int n, int d, int k) {
  F16rPlan f = {};
  f.t = topk_plan(m, n, d, k, OIBL_BF16, 0);   // 2-byte operand rows: the bf16 plan's sample / capacity / legality
  f.K2 = f16r_k2(k);
  f.fused = f.t.fused && 2 * k + 32 <= F16R_MAX_K2 && f.K2 <= f.t.cap;   // (k <= 496: a member window of k + k + 32)
  size_t o = f.t.total;
  f.off_lval = o;
  o += align_up((size_t)m * f.K2 * sizeof(float), 256);
  f.off_lidx = o;
  o += align_up((size_t)m * f.K2 * sizeof(int32_t), 256);
  f.off_ymax = o;
  o += 256;
  f.total = o;
  return f;
}

size_t oibl_sqdist_topk_f16r_st_workspace_bytes(int m, int n, int d, int k, int x_st, int y_st) {
  if (m <= 0 || n <= 0 || d <= 0 || k <= 0 || !st_ok(x_st) || !st_ok(y_st)) return 0;
  // (behind the plan: what the exact path's fp32 tiles need of 16-bit stored rows — their widened copies + norms)
  const bool widen = x_st != OIBL_ST_F32 || y_st != OIBL_ST_F32;
  return f16r_plan(m, n, d, k).total + (widen ? align_up(oibl_pairwise_st_workspace_bytes(m, n, d, OIBL_F32, x_st, y_st), 256) : 0);
}
size_t oibl_sqdist_topk_f16r_workspace_bytes(int m, int n, int d, int k) {
  return oibl_sqdist_topk_f16r_st_workspace_bytes(m, n, d, k, OIBL_ST_F32, OIBL_ST_F32);
}

extern "C++" {
template <bool FILTER>
static int launch_pairwise_f16r(F16rParams& p, hipStream_t st, int ksplit = 1) {
  p.tiles_m = (p.m + 255) / 256;
  p.tiles_n = (p.n + 255) / 256;
  p.group_m = g_match_group;
  const long grid = (long)p.tiles_m * p.tiles_n;
  OIBL_REQUIRE(grid > 0 && grid <= 0x7fffffffL, "sqdist_topk_f16r: grid out of range");
  constexpr int lds = RingGeo<2>::MAIN_LDS;
  if (g_match_bar1) {
    auto kern = pairwise_f16r_kernel<FILTER, true>;
    OIBL_SET_MAX_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)ksplit), dim3(512), lds, st, p);
  } else {
    auto kern = pairwise_f16r_kernel<FILTER, false>;
    OIBL_SET_MAX_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)ksplit), dim3(512), lds, st, p);
  }
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}
}  // extern "C++"

// K2 of a k (the member slots per query) and whether (m, n, d, k) takes the fused path — what a caller that runs the
// two stages itself (sharded matching: the rescoring behind the exchange of the filter lists) has to know
int oibl_f16r_members(int k) { return k >= 1 ? f16r_k2(k) : 0; }
int oibl_f16r_fused(int m, int n, int d, int k) {
  if (m <= 0 || n <= 0 || d <= 0 || d % 64 != 0 || k <= 0) return 0;
  return f16r_plan(m, n, d, k).fused ? 1 : 0;
}

// stage 1: thresholds, filter pass, selection -> lval / lidx [m][K2]: the candidates that can belong to the top-k of
// THIS gallery (filter distances + global indices, any order, (+inf, -1) paddings); ymax_out [2] (device, may be
// NULL) = the largest |y| and |y - yh 2^-e| of the gallery (what the pair bound of a wider selection needs)
int oibl_f16r_filter_select(const void* xh, const float* xaux, const float* xn, int m, const void* yh,
                            const float* yaux, const float* yn, int n, int d, int k, int index_base, float* lval,
                            int32_t* lidx, float* ymax_out, int32_t* overflow, void* ws, size_t ws_bytes,
                            void* stream) {
  OIBL_REQUIRE(xh && xaux && xn && yh && yaux && yn && lval && lidx && ws, "f16r_filter_select: null pointer");
  int rc = topk_args_ok(m, n, d, k, index_base, OIBL_F32);
  if (rc) return rc;
  OIBL_REQUIRE((uintptr_t)ws % 256 == 0 && (uintptr_t)xh % 16 == 0 && (uintptr_t)yh % 16 == 0 &&
                   (uintptr_t)xaux % 16 == 0 && (uintptr_t)yaux % 16 == 0,
               "f16r_filter_select: workspace must be 256-byte, rows and aux 16-byte aligned");
  const F16rPlan f = f16r_plan(m, n, d, k);
  OIBL_REQUIRE(f.fused, "f16r_filter_select: (m=%d, n=%d, d=%d, k=%d) does not take the fused path (oibl_f16r_fused)", m, n,
               d, k);
  if (ws_bytes < f.total) {
    set_error("f16r_filter_select: workspace %zu < required %zu bytes", ws_bytes, f.total);
    return OIBL_E_WORKSPACE;
  }
  char* wsb = (char*)ws;
  hipStream_t st = (hipStream_t)stream;
  const TopkPlan& t = f.t;
  if (overflow) OIBL_HIP_CHECK(hipMemsetAsync(overflow, 0, sizeof(int32_t), st));
  float* sample = (float*)(wsb + t.off_sample);
  float* sval = (float*)(wsb + t.off_sval);
  int32_t* sidx = (int32_t*)(wsb + t.off_sidx);
  int* cnt = (int*)(wsb + t.off_cnt);
  unsigned* ymax = (unsigned*)(wsb + f.off_ymax);
  OIBL_HIP_CHECK(hipMemsetAsync(cnt, 0, (size_t)(m + 1) * sizeof(int), st));
  OIBL_HIP_CHECK(hipMemsetAsync(ymax, 0, 16, st));
  // 1. thresholds: k-th smallest filter distance to a strided sample of S gallery rows
  F16rParams q = {};
  q.x = xh;
  q.y = yh;
  q.xn = xn;
  q.yn = yn;
  q.xaux = (const float4*)xaux;
  q.yaux = (const float4*)yaux;
  q.dist = sample;
  q.ldd = (size_t)t.S;
  q.x_bytes = (unsigned)((size_t)m * d * 2);
  q.y_bytes = (unsigned)((size_t)n * d * 2);
  q.y_row_bytes = (long)d * 2 * t.stride;
  q.y_stride = t.stride;
  q.m = m;
  q.n = t.S;
  q.d = d;
  q.ymax = ymax;
  q.gamma = (float)d * 5.9604645e-8f;
  const size_t half = align_up((size_t)m * t.S * sizeof(float), 256) / sizeof(float);
  if (t.ksplit == 2) q.part_stride = half;
  rc = launch_pairwise_f16r<false>(q, st, t.ksplit);
  if (rc) return rc;
  launch_sample_kth(sample, t.ksplit == 2 ? sample + half : nullptr, m, t.S, k, sval, sidx, st);
  OIBL_LAUNCH_CHECK();
  // 2. the full contraction, keeping the pairs whose filter distance could belong to a member of the true top-k
  q.part_stride = 0;
  q.dist = nullptr;
  q.y_row_bytes = (long)d * 2;
  q.y_stride = 1;
  q.n = n;
  q.thr = sval + (k - 1);
  q.thr_stride = k;
  q.cand_val = (float*)(wsb + t.off_cval);
  q.cand_idx = (int32_t*)(wsb + t.off_cidx);
  q.cand_cnt = cnt;
  q.cap = t.cap;
  q.index_base = index_base;
  rc = launch_pairwise_f16r<true>(q, st);
  if (rc) return rc;
  // 3. the members of every query's rescore set (a list beyond the window / capacity, or more than K2 members,
  //    raises *overflow)
  if (k <= SEL_MAX_K) {
    hipLaunchKernelGGL((f16r_select_kernel<32, false>), dim3((unsigned)((m + 3) / 4)), dim3(256), 0, st, q.cand_val,
                       q.cand_idx, cnt, m, t.cap, k, f.K2, 2048, xn, (const float4*)xaux, ymax, q.gamma, lval, lidx,
                       (int*)overflow);
    OIBL_LAUNCH_CHECK();
    // (the rare lists beyond 2048 entries: one workgroup per such query; every other workgroup returns at once)
    hipLaunchKernelGGL((f16r_select_kernel<32, true>), dim3((unsigned)m), dim3(256), 0, st, q.cand_val, q.cand_idx,
                       cnt, m, t.cap, k, f.K2, 2048, xn, (const float4*)xaux, ymax, q.gamma, lval, lidx,
                       (int*)overflow);
  } else {
    // k beyond the register rounds: the k-th smallest by bisection, one workgroup per query (lists up to 8192 entries)
    hipLaunchKernelGGL((f16r_select_bisect_kernel<32>), dim3((unsigned)m), dim3(256), 0, st, q.cand_val, q.cand_idx,
                       cnt, m, t.cap, k, f.K2, xn, (const float4*)xaux, ymax, q.gamma, lval, lidx, (int*)overflow);
  }
  OIBL_LAUNCH_CHECK();
  if (ymax_out) OIBL_HIP_CHECK(hipMemcpyAsync(ymax_out, ymax + 2, 8, hipMemcpyDeviceToDevice, st));
  return OIBL_OK;
}

// stage 2: exact distances (fp64-accumulated, from the fp32 rows) of the listed members lidx [m][K2] (global indices
// of THIS gallery, -1 = no member), the k smallest (distance, index) per query
extern "C++" {
template <int XST>
static void launch_f16r_rescore(int y_st, unsigned m, unsigned threads, hipStream_t st, const F16rRescoreParams& r) {
  switch (y_st) {
    case OIBL_ST_F32: hipLaunchKernelGGL((f16r_rescore_kernel<XST, OIBL_ST_F32>), dim3(m), dim3(threads), 0, st, r); break;
    case OIBL_ST_F16: hipLaunchKernelGGL((f16r_rescore_kernel<XST, OIBL_ST_F16>), dim3(m), dim3(threads), 0, st, r); break;
    default: hipLaunchKernelGGL((f16r_rescore_kernel<XST, OIBL_ST_BF16>), dim3(m), dim3(threads), 0, st, r);
  }
}
}  // extern "C++"

int oibl_f16r_rescore_st(const void* xsrc, int x_st, const float* xn, int m, const void* ysrc, int y_st,
                         const float* yn, int d, int k, int members, int index_base, const int32_t* lidx,
                         float* out_val, int32_t* out_idx, void* stream) {
  OIBL_REQUIRE(xsrc && xn && ysrc && yn && lidx && out_val && out_idx, "f16r_rescore: null pointer");
  OIBL_REQUIRE(st_ok(x_st) && st_ok(y_st), "f16r_rescore: bad storage type %d / %d", x_st, y_st);
  OIBL_REQUIRE(m > 0 && d > 0 && d % 4 == 0 && k >= 1 && members >= 1 && members <= F16R_MAX_K2,
               "f16r_rescore: bad shape m=%d d=%d k=%d members=%d", m, d, k, members);
  OIBL_REQUIRE((uintptr_t)xsrc % 16 == 0 && (uintptr_t)ysrc % 16 == 0, "f16r_rescore: rows must be 16-byte aligned");
  F16rRescoreParams r = {};
  r.xsrc = xsrc;
  r.ysrc = ysrc;
  r.xn = xn;
  r.yn = yn;
  r.lidx = lidx;
  r.m = m;
  r.d = d;
  r.k = k;
  r.K2 = members;
  r.index_base = index_base;
  r.out_val = out_val;
  r.out_idx = out_idx;
  // 4 waves walk a 32-slot window; the wide windows of k > 16 (2k + 32 slots, k + a few of them members) get 16
  const unsigned threads = members <= 64 ? 256u : 1024u;
  hipStream_t st = (hipStream_t)stream;
  switch (x_st) {
    case OIBL_ST_F32: launch_f16r_rescore<OIBL_ST_F32>(y_st, (unsigned)m, threads, st, r); break;
    case OIBL_ST_F16: launch_f16r_rescore<OIBL_ST_F16>(y_st, (unsigned)m, threads, st, r); break;
    default: launch_f16r_rescore<OIBL_ST_BF16>(y_st, (unsigned)m, threads, st, r);
  }
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}
int oibl_f16r_rescore(const float* xsrc, const float* xn, int m, const float* ysrc, const float* yn, int d, int k,
                      int index_base, const int32_t* lidx, float* out_val, int32_t* out_idx, void* stream) {
  return oibl_f16r_rescore_st(xsrc, OIBL_ST_F32, xn, m, ysrc, OIBL_ST_F32, yn, d, k, f16r_k2(k), index_base, lidx,
                              out_val, out_idx, stream);
}

int oibl_f16r_keep_members(const float* lval, int32_t* lidx, int m, int k, const float* thr, const float* xn,
                           const float* xaux, const float* ymax_all, int shards, int d, void* stream) {
  OIBL_REQUIRE(lval && lidx && thr && xn && xaux && ymax_all, "f16r_keep_members: null pointer");
  OIBL_REQUIRE(m > 0 && k >= 1 && shards >= 1 && d > 0, "f16r_keep_members: bad arguments");
  const int K2 = f16r_k2(k);
  const long items = (long)m * K2;
  hipLaunchKernelGGL(f16r_keep_members_kernel, dim3((unsigned)((items + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     lval, lidx, m, K2, thr, xn, (const float4*)xaux, ymax_all, shards, (float)d * 5.9604645e-8f);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_sqdist_topk_f16r_st(const void* xh, const float* xaux, const float* xn, const void* xsrc, int x_st, int m,
                             const void* yh, const float* yaux, const float* yn, const void* ysrc, int y_st, int n,
                             int d, int k, int index_base, int exact, float* out_val, int32_t* out_idx,
                             int32_t* overflow, void* ws, size_t ws_bytes, void* stream) {
  OIBL_REQUIRE(xh && xaux && xn && xsrc && yh && yaux && yn && ysrc && out_val && out_idx && ws,
               "sqdist_topk_f16r: null pointer");
  OIBL_REQUIRE(st_ok(x_st) && st_ok(y_st), "sqdist_topk_f16r: bad storage type %d / %d", x_st, y_st);
  int rc = topk_args_ok(m, n, d, k, index_base, OIBL_F32);
  if (rc) return rc;
  OIBL_REQUIRE((uintptr_t)ws % 256 == 0 && (uintptr_t)xsrc % 16 == 0 && (uintptr_t)ysrc % 16 == 0,
               "sqdist_topk_f16r: workspace must be 256-byte, rows 16-byte aligned");
  const F16rPlan f = f16r_plan(m, n, d, k);
  const size_t need = oibl_sqdist_topk_f16r_st_workspace_bytes(m, n, d, k, x_st, y_st);
  if (ws_bytes < need) {
    set_error("sqdist_topk_f16r: workspace %zu < required %zu bytes", ws_bytes, need);
    return OIBL_E_WORKSPACE;
  }
  char* wsb = (char*)ws;
  float* lval = (float*)(wsb + f.off_lval);
  int32_t* lidx = (int32_t*)(wsb + f.off_lidx);
  if (!f.fused || exact) {
    // the exact path: the K2 nearest by fp32 distance tiles of the widened rows + row_topk (what OIBL_F32 runs), then
    // the SAME rescoring — values and tie order do not depend on which path produced the member set
    // (K2 >= k for every k <= 1024: 32 up to 16, min(2k + 32, 1024) beyond)
    const int K2 = f.K2;
    const void *xo = xsrc, *yo = ysrc;
    const float *xn2 = xn, *yn2 = yn;
    if (x_st != OIBL_ST_F32 || y_st != OIBL_ST_F32) {   // 16-bit storage: fp32 copies of the rows behind the plan
      float *xw, *yw;
      rc = pairwise_prepare(xsrc, x_st, m, ysrc, y_st, n, d, OIBL_F32, wsb + f.total, &xo, &yo, &xw, &yw, stream);
      if (rc) return rc;
      xn2 = xw;
      yn2 = yw;
    }
    rc = sqdist_topk_core(xo, xn2, m, yo, yn2, n, d, K2, index_base, OIBL_F32, 1, lval, lidx, overflow, wsb, f.t,
                          (hipStream_t)stream);
    if (rc) return rc;
    return oibl_f16r_rescore_st(xsrc, x_st, xn, m, ysrc, y_st, yn, d, k, K2, index_base, lidx, out_val, out_idx, stream);
  }
  rc = oibl_f16r_filter_select(xh, xaux, xn, m, yh, yaux, yn, n, d, k, index_base, lval, lidx, nullptr, overflow, ws,
                               ws_bytes, stream);
  if (rc) return rc;
  return oibl_f16r_rescore_st(xsrc, x_st, xn, m, ysrc, y_st, yn, d, k, f.K2, index_base, lidx, out_val, out_idx, stream);
}
int oibl_sqdist_topk_f16r(const void* xh, const float* xaux, const float* xn, const float* xsrc, int m,
                          const void* yh, const float* yaux, const float* yn, const float* ysrc, int n, int d,
                          int k, int index_base, int exact, float* out_val, int32_t* out_idx, int32_t* overflow,
                          void* ws, size_t ws_bytes, void* stream) {
  return oibl_sqdist_topk_f16r_st(xh, xaux, xn, xsrc, OIBL_ST_F32, m, yh, yaux, yn, ysrc, OIBL_ST_F32, n, d, k,
                                  index_base, exact, out_val, out_idx, overflow, ws, ws_bytes, stream);
}

int oibl_first_hit_rank(const int32_t* topk_idx, int m, int k, const int32_t* gt_offsets,
                        const int32_t* gt_values, const int32_t* gallery_pids, int nms_window,
                        int32_t* out_rank, void* stream) {
  OIBL_REQUIRE(topk_idx && gt_offsets && gt_values && out_rank, "first_hit_rank: null pointer");
  OIBL_REQUIRE(m > 0 && k >= 1 && k <= RANK_MAXK, "first_hit_rank: bad shape m=%d k=%d", m, k);
  OIBL_REQUIRE(!gallery_pids || nms_window >= 1, "first_hit_rank: nms_window must be >= 1");
  hipLaunchKernelGGL(first_hit_rank_kernel, dim3((m + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     topk_idx, m, k, gt_offsets, gt_values, gallery_pids, nms_window, out_rank);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

size_t oibl_row_argsort_workspace_bytes(int m, int n) {
  if (m <= 0 || n <= 0) return 0;
  return 4 * align_up((size_t)m * n * 4, 256);
}

int oibl_row_argsort(const float* vals, int m, int n, size_t ld, int32_t* out_idx, float* out_val,
                     void* ws, size_t ws_bytes, void* stream) {
  OIBL_REQUIRE(vals && out_idx && ws, "row_argsort: null pointer");
  OIBL_REQUIRE(m > 0 && n > 0 && ld >= (size_t)n, "row_argsort: bad shape m=%d n=%d ld=%zu", m, n, ld);
  OIBL_REQUIRE((uintptr_t)ws % 256 == 0, "row_argsort: workspace must be 256-byte aligned");
  const size_t need = oibl_row_argsort_workspace_bytes(m, n);
  if (ws_bytes < need) {
    set_error("row_argsort: workspace %zu < required %zu bytes", ws_bytes, need);
    return OIBL_E_WORKSPACE;
  }
  const size_t part = align_up((size_t)m * n * 4, 256);
  char* w = (char*)ws;
  RowSortParams p;
  p.vals = vals;
  p.ld = ld;
  p.n = n;
  p.key[0] = (uint32_t*)w;
  p.key[1] = (uint32_t*)(w + part);
  p.idx[0] = (int32_t*)(w + 2 * part);
  p.idx[1] = (int32_t*)(w + 3 * part);
  p.out_idx = out_idx;
  p.out_val = out_val;
  hipLaunchKernelGGL(row_radix_sort_kernel, dim3((unsigned)m), dim3(SORT_THREADS), 0, (hipStream_t)stream, p);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_row_topk(const float* vals, const int32_t* idx_in, int m, int n, size_t ld, int k,
                  int index_base, float* out_val, int32_t* out_idx, void* stream) {
  OIBL_REQUIRE(vals && out_val && out_idx, "row_topk: null pointer");
  OIBL_REQUIRE(m > 0 && n > 0 && ld >= (size_t)n, "row_topk: bad shape m=%d n=%d ld=%zu", m, n, ld);
  OIBL_REQUIRE(k >= 1 && k <= 1024, "row_topk: k=%d outside [1, 1024]", k);
  launch_row_topk(vals, idx_in, m, n, ld, k, index_base, out_val, out_idx, nullptr, nullptr,
                  (hipStream_t)stream);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

}  // extern "C"
