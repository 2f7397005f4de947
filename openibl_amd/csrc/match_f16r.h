// "f16r": fp16 FILTER pass + EXACT rescoring — fused distance + top-k with fp32-exact lists at the speed of a
// 2-byte operand stream.  Included by match.hip behind the ring distance kernel.
// Reference behaviour: pairwise_distance + the argsort prefix evaluate_all reads (ibl/evaluators.py:105-130,
// 142-159), i.e. the k nearest gallery rows per query by fp32 squared-L2.
//
// Idea.  The parity modes pay 1.5 (f16mx) or 3 (bf16x3) matrix-instruction times per product for EVERY pair,
// although only ~k of a query's n distances are ever read.  Here every pair costs ONE fp16 MFMA product, and
// the few pairs that can reach the top-k are recomputed exactly:
//
//   prepare   a row x becomes xh = fp16(x * 2^e) (e: the row's largest element lands in [2^14, 2^15), results
//             below the fp16 normals are flushed to zero EXPLICITLY), plus four fp32 scalars: 2^-e, |x|, and
//             |r| = |x - xh 2^-e| — the residual is formed exactly (fp32 subtraction of a value and its own
//             fp16 rounding) and its norm rounded up.  The fp32 rows stay resident for the rescoring.
//   bound     for any pair, by Cauchy-Schwarz,
//               |x.y - xh.yh| <= |rx||y| + |x||ry| + |rx||ry|,   and the fp32 accumulation of the 2-byte stream
//               moves the computed dot product by at most g (|x| + |rx|)(|y| + |ry|), g = d 2^-24 (the figure
//               the split-K sample slack of this file has always used), so the filter distance D_h and the
//               distance D the rescoring returns differ by at most
//               eps(i, j) = A_i |y_j| + B_i |ry_j| + tiny,  A_i = 2 (|rx_i| + g (|x_i| + |rx_i|)),
//                                                              B_i = 2 (|x_i| + |rx_i|) (1 + g).
//             Nothing here is statistical: a pair outside a threshold widened by eps is outside for D as well.
//   filter    thr_i = k-th smallest D_h over a strided sample of S gallery rows (same kernel).  The k-th
//             smallest TRUE distance of the whole gallery is <= thr_i + max_s eps(i, s), so every member of the
//             true top-k has D_h <= thr_i + max_s eps(i, s) + eps(i, j): that is the filter's test (two fma per
//             pair in the epilogue).  Survivors go to per-query candidate lists, as in the other modes.
//   select    one wave per query holds its candidate list in registers: k rounds of minimum extraction give T =
//             the k-th smallest D_h, and a member of the true top-k has D_h <= T + max_{e < k} eps_e + eps(i, j)
//             <= T + 2 eps_any (eps_any: the bound with the gallery-wide maxima of |y|, |ry|): the RESCORE SET,
//             typically k + 2..5 entries, compacted by ballot.  k > 32 (round 6: the 120 ranks of spatial NMS, up
//             to 496): T by BISECTION over the values' order-preserving image, one workgroup per query
//             (f16r_select_bisect_kernel).  More than K2 (32 for k <= 16, 2k + 32 beyond) members, or a list beyond
//             the register window: *overflow is raised and the caller repeats on the exact path.
//   storage   (round 6) the rows may be STORED as fp32, IEEE half or bf16 (OIBL_ST_*): prepare widens them exactly,
//             the rescoring reads the stored rows; an fp16-stored row is its own fp16 image (residual 0).  The exact
//             path (small problems, the repeat after *overflow) takes its member set from fp32 distance tiles and
//             ends in the SAME rescoring: values and tie order do not depend on the path.
//   rescore   for the set only: x.y from the resident fp32 rows, accumulated in fp64 (one wave per pair, a 16 KB
//             contiguous gather per gallery row), D = fl32((|x|^2 + |y|^2) - 2 x.y) with the fp32 norms every
//             mode uses; the k smallest (D, index) leave, lowest index first on ties.
// The lists are therefore those of an fp32 matrix whose dot products are correctly rounded — closer to the fp64
// oracle than the fp32 MFMA mode itself — and nothing about them depends on the fp16 pass except their cost.
#pragma once

#include "ring_core.h"

namespace oibl {

struct F16rParams {
  const void* x;          // [m][d] fp16, scaled rows
  const void* y;          // gallery rows at y_row_bytes
  const float* xn;        // fp32 squared norms (of the fp32 rows)
  const float* yn;        // yn[col * y_stride]
  const float4* xaux;     // {2^-e, |x| (rounded up), |residual| (rounded up), 0}
  const float4* yaux;     // yaux[col * y_stride]
  float* dist;            // !FILTER: [m][ldd]
  size_t ldd;
  unsigned x_bytes, y_bytes;
  long y_row_bytes;
  int y_stride;
  int m, n, d, tiles_m, tiles_n, group_m;
  const float* thr;       // FILTER: thr[row * thr_stride]
  int thr_stride;
  float* cand_val;
  int32_t* cand_idx;
  int* cand_cnt;
  int cap, index_base;
  size_t part_stride;     // 2-way split-K of the sample pass (gridDim.y = 2): half h writes dist + h * part_stride
  unsigned* ymax;         // [4] bit patterns of max |y|, max |ry|: [0], [1] over the SAMPLE columns — written
                          // (atomicMax) by the !FILTER launch, read by the FILTER launch; [2], [3] over ALL gallery
                          // columns — written by the FILTER launch, read by the rescoring
  float gamma;            // d * 2^-24
};

// stored rows (fp32, IEEE half or bf16: OIBL_ST_*) -> scaled fp16 rows + norms + the four scalars; one wave per row.
// The squared norm is formed exactly as row_sqnorm_kernel / row_sqnorm_cast_kernel form it (same loop over the
// widened row, same reduction): bit-identical to the other modes'.  A 16-bit stored row is widened exactly; an
// fp16-stored row is its own fp16 image up to the power-of-two scale (residual 0 unless elements fall below the
// scaled row's fp16 normals), so the filter bound of such a gallery is the accumulation slack alone.
template <int ST>
__global__ __launch_bounds__(256) void f16r_prepare_kernel(const void* __restrict__ x, float* __restrict__ norms,
                                                           float4* __restrict__ aux, uint16_t* __restrict__ xh,
                                                           int rows, int d) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  const size_t es = ST == OIBL_ST_F32 ? 4 : 2;
  const char* xr = static_cast<const char*>(x) + (size_t)row * d * es;
  float s = 0.f, mx = 0.f;
  for (int i = lane * 4; i < d; i += 256) {
    const float4 v = load4_widen<ST>(xr, i);
    s = fmaf(v.x, v.x, s);
    s = fmaf(v.y, v.y, s);
    s = fmaf(v.z, v.z, s);
    s = fmaf(v.w, v.w, s);
    mx = fmaxf(fmaxf(mx, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
  }
  s = wave_sum(s);
  mx = wave_max(mx);
  int e = 0;
  if (mx > 0.f && mx < INFINITY) {
    int ex;
    (void)frexpf(mx, &ex);   // mx = f 2^ex, f in [0.5, 1)
    e = 15 - ex;             // mx 2^e in [2^14, 2^15)
    e = e > 100 ? 100 : (e < -100 ? -100 : e);
  }
  const float sc = ldexpf(1.0f, e), isc = ldexpf(1.0f, -e);
  uint16_t* hr = xh + (size_t)row * d;
  float rs = 0.f;
  for (int i = lane * 4; i < d; i += 256) {
    const float4 v = load4_widen<ST>(xr, i);
    const float t[4] = {v.x * sc, v.y * sc, v.z * sc, v.w * sc};   // exact (power of two; no under/overflow: |e| <= 100)
    uint16_t h[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      // below the fp16 normals the element is dropped here, whatever the matrix pipe would do with a denormal;
      // it stays in the residual, so the bound holds either way
      h[c] = fabsf(t[c]) < 6.103515625e-5f ? (uint16_t)0 : f32_to_f16_bits(t[c]);
      const float r = t[c] - f16_bits_to_f32(h[c]);   // exact
      rs = fmaf(r, r, rs);
    }
    uint2 o;
    o.x = (uint32_t)h[0] | ((uint32_t)h[1] << 16);
    o.y = (uint32_t)h[2] | ((uint32_t)h[3] << 16);
    *reinterpret_cast<uint2*>(hr + i) = o;
  }
  rs = wave_sum(rs);
  if (lane == 0) {
    norms[row] = s;
    // rounded UP: the fp32 sums above carry ~d 2^-24 relative error at most
    const float up = 1.0f + 1.0f / 1024.0f;
    aux[row] = make_float4(isc, sqrtf(s) * up, sqrtf(rs) * isc * up, 0.f);
  }
}

// The ring distance kernel on fp16 rows (RING_F16: the bf16 stream with v_mfma_f32_32x32x16_f16), tile order and
// main loop exactly pairwise_ring_kernel's; the epilogue undoes the row scales and, with FILTER, widens every
// row's threshold by the pair's error bound.
template <bool FILTER, bool BAR1>
__global__ __launch_bounds__(512) void pairwise_f16r_kernel(F16rParams p) {
  using G = RingGeo<2>;
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
  const int ksplit = (int)gridDim.y, kh = (int)blockIdx.y;
  const int d_part = p.d / ksplit;
  const unsigned koff = (unsigned)kh * (unsigned)d_part * 2u;
  RingRowLoader<2> la, lb;
  la.init(static_cast<const char*>(p.x) + koff, p.x_bytes - koff, m0, p.m, (long)p.d * 2, rows_a, piece);
  lb.init(static_cast<const char*>(p.y) + koff, p.y_bytes - koff, n0, p.n, p.y_row_bytes, rows_b, piece);

  f32x16_t acc[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  if constexpr (BAR1) {
    if ((wave >> 2) == 0) ring_mainloop<2, false, false, RING_F16, true, 0>(acc, smem, wave, lane, la, lb, d_part >> 6);
    else ring_mainloop<2, false, false, RING_F16, true, 1>(acc, smem, wave, lane, la, lb, d_part >> 6);
  } else {
    ring_mainloop<2, false, false, RING_F16>(acc, smem, wave, lane, la, lb, d_part >> 6);
  }

  float* const xn_s = reinterpret_cast<float*>(smem);   // 256 floats each
  float* const yn_s = xn_s + 256;
  float* const th_s = xn_s + 512;
  float* const xs_s = xn_s + 768;     // -2 * 2^-ex
  float* const xa_s = xn_s + 1024;    // A_i
  float* const xb_s = xn_s + 1280;    // B_i
  float* const ys_s = xn_s + 1536;    // 2^-ey
  float* const yy_s = xn_s + 1792;    // |y_j|
  float* const yr_s = xn_s + 2048;    // |ry_j|
  if (threadIdx.x < 256) {
    int r = m0 + (int)threadIdx.x;
    if (r > p.m - 1) r = p.m - 1;
    const float xn = p.xn[r];
    const float4 a = p.xaux[r];
    xn_s[threadIdx.x] = kh == 0 ? xn : 0.f;
    xs_s[threadIdx.x] = -2.0f * a.x;
    if (FILTER) {
      const float nx = a.y, rx = a.z;
      const float A = 2.0f * (rx + p.gamma * (nx + rx)), B = 2.0f * (nx + rx) * (1.0f + p.gamma);
      const float ymx = __uint_as_float(p.ymax[0]), rmx = __uint_as_float(p.ymax[1]);
      xa_s[threadIdx.x] = A;
      xb_s[threadIdx.x] = B;
      // sample k-th + the largest bound over the sample + the fp32 roundings of the two final distances
      th_s[threadIdx.x] = p.thr[(long)r * p.thr_stride] + fmaf(A, ymx, B * rmx) + 1e-6f * (xn + ymx * ymx);
    }
  } else {
    int c = n0 + (int)threadIdx.x - 256;
    if (c > p.n - 1) c = p.n - 1;
    const float yn = p.yn[(long)c * p.y_stride];
    const float4 a = p.yaux[(long)c * p.y_stride];
    yn_s[threadIdx.x - 256] = kh == 0 ? yn : 0.f;
    ys_s[threadIdx.x - 256] = a.x;
    yy_s[threadIdx.x - 256] = a.y;
    yr_s[threadIdx.x - 256] = a.z;
    if (p.ymax && kh == 0 && tm == 0) {  // one column of tiles covers every column of the launch once
      const float my = wave_max(a.y), mr = wave_max(a.z);   // (non-negative: bit patterns order as integers)
      if (lane == 0) {
        atomicMax(p.ymax + (FILTER ? 2 : 0), __float_as_uint(my));
        atomicMax(p.ymax + (FILTER ? 3 : 1), __float_as_uint(mr));
      }
    }
  }
  __syncthreads();
  const int col0 = wn * 64 + (lane & 31), row0 = wm * 128 + 4 * (lane >> 5);
  if constexpr (!FILTER) {
    // (stores through one buffer descriptor per 32-row accumulator tile, scalar row offsets: pairwise_ring_kernel)
    const float* const dbase = p.dist + (size_t)kh * p.part_stride + ((size_t)m0 + wm * 128) * p.ldd + n0;
    const unsigned ldd4 = (unsigned)p.ldd * 4u;
    const unsigned voff = (unsigned)(4 * (lane >> 5)) * ldd4 + (unsigned)col0 * 4u;
    const int rows_left = p.m - m0 - row0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const __amdgpu_buffer_rsrc_t rs_d = __builtin_amdgcn_make_buffer_rsrc(
          const_cast<float*>(dbase + (size_t)(32 * i) * p.ldd), 0, 0x7fffffff, 0x00020000);
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const bool nok = n0 + col0 + 32 * j < p.n;
        const float yn = yn_s[col0 + 32 * j], ys = ys_s[col0 + 32 * j];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int rt = (r & 3) + 8 * (r >> 2), ro = 32 * i + rt;
          const float dv = fmaf(acc[i][j][r] * ys, xs_s[row0 + ro], xn_s[row0 + ro] + yn);
          if (nok && ro < rows_left)
            __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, dv), rs_d, (int)voff,
                                                  (int)((unsigned)rt * ldd4 + 128u * j), 0);
          if ((r & 7) == 7) __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  } else {
    // per-lane survivor lists, then one round of atomics per tile (pairwise_ring_kernel, FILTER)
    constexpr int LCAP = 8;
    float* const l_val = reinterpret_cast<float*>(smem + 12288);
    int* const l_idx = reinterpret_cast<int*>(smem + 12288 + LCAP * 512 * 4);
    int* const l_row = reinterpret_cast<int*>(smem + 12288 + 2 * LCAP * 512 * 4);
    const int rows_left = p.m - m0 - row0;
    int nl = 0;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int n = n0 + col0 + 32 * j;
      const bool nok = n < p.n;
      const float yn = yn_s[col0 + 32 * j], ys = ys_s[col0 + 32 * j];
      const float yy = yy_s[col0 + 32 * j], yr = yr_s[col0 + 32 * j];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int ro = 32 * i + (r & 3) + 8 * (r >> 2);
          const float dv = fmaf(acc[i][j][r] * ys, xs_s[row0 + ro], xn_s[row0 + ro] + yn);
          const float bound = th_s[row0 + ro] + fmaf(xa_s[row0 + ro], yy, xb_s[row0 + ro] * yr);
          // !(dv > bound): a NaN on either side KEEPS the pair (the list then overflows into the exact path)
          if (nok && ro < rows_left && !(dv > bound)) {
            const int m = m0 + row0 + ro;
            if (nl < LCAP) {
              l_val[nl * 512 + threadIdx.x] = dv;
              l_idx[nl * 512 + threadIdx.x] = p.index_base + n;
              l_row[nl * 512 + threadIdx.x] = m;
              ++nl;
            } else {
              const int pos = atomicAdd(p.cand_cnt + m, 1);
              if (pos < p.cap) {
                p.cand_val[(size_t)m * p.cap + pos] = dv;
                p.cand_idx[(size_t)m * p.cap + pos] = p.index_base + n;
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

// ---- selection: which candidates can belong to the true top-k ------------------------------------------------
// One wave per query, its candidate list (<= 64 NQ entries) in registers as (ordered value bits << 32 | index) keys:
// k rounds of wave-minimum extraction give T = the k-th smallest filter distance; a candidate is a MEMBER of the
// rescore set iff D_h <= T + 2 eps_any, eps_any = the pair bound with the gallery-wide maxima of |y| and |ry| (>=
// the bound of every pair of this query: both the k-th TRUE distance and the candidate's own move by at most that).
// Members are compacted (ballot prefix, any order) into lval / lidx [m][K2], padded with (+inf, -1).  A list longer
// than the register window or the candidate capacity, or more than K2 members: *overflow (exact path).
constexpr int F16R_MAX_K2 = 1024;

// WG = false: one wave per query (four per workgroup), lists up to 64 NQ entries — longer ones are left to the
// WG = true launch, one 256-thread workgroup per query, lists up to 256 NQ entries (the candidate capacity).  With
// ~800 expected candidates and the k-th order statistic's gamma tail, a list beyond 2048 entries happens about once
// per 8192 queries: the second launch finds its handful of rows by their counts and returns at once everywhere else.
template <int NQ, bool WG>
__global__ __launch_bounds__(256) void f16r_select_kernel(const float* __restrict__ cand_val,
                                                          const int32_t* __restrict__ cand_idx,
                                                          const int* __restrict__ cnt, int m, int cap, int k, int K2,
                                                          int short_rows,
                                                          const float* __restrict__ xn, const float4* __restrict__ xaux,
                                                          const unsigned* __restrict__ ymax, float gamma,
                                                          float* __restrict__ lval, int32_t* __restrict__ lidx,
                                                          int* __restrict__ overflow) {
  __shared__ unsigned long long wsel[4 * SEL_MAX_K];   // WG: the k smallest keys of every wave
  __shared__ int s_count;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tid = WG ? (int)threadIdx.x : lane, nthr = WG ? 256 : 64;
  const int row = WG ? (int)blockIdx.x : (int)blockIdx.x * 4 + wave;
  if (row >= m) return;   // (WG: workgroup-uniform; else wave-uniform)
  int n = cnt[row];
  if (WG ? n <= short_rows : n > short_rows) return;     // the other launch's row
  const int window = cap < nthr * NQ ? cap : nthr * NQ;
  if (n > window) {
    if (tid == 0 && overflow) atomicOr(overflow, 1);
    n = window;
  }
  const float* vr = cand_val + (size_t)row * cap;
  const int32_t* ir = cand_idx + (size_t)row * cap;
  unsigned long long mine[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int j = tid + nthr * q;
    mine[q] = TOPK_INF;
    if (j < n) mine[q] = ((unsigned long long)ordered_bits(vr[j]) << 32) | (unsigned)ir[j];
  }
  // the wave's k smallest keys leave `mine` one by one; lane r keeps the r-th (k <= SEL_MAX_K <= 64)
  unsigned long long res = TOPK_INF, w = TOPK_INF;
  for (int r = 0; r < k; ++r) {
    w = wave_extract_min(mine, lane);
    if (lane == r) res = w;
  }
  if constexpr (WG) {
    if (tid == 0) s_count = 0;
    if (lane < k) wsel[wave * SEL_MAX_K + lane] = res;
    __syncthreads();
    if (wave == 0) {       // the k-th smallest of the 4 k survivors (<= 128 keys: two per lane)
      unsigned long long two[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const int e = lane + 64 * h;
        two[h] = e < 4 * k ? wsel[(e / k) * SEL_MAX_K + (e % k)] : TOPK_INF;
      }
      unsigned long long t_ = TOPK_INF;
      for (int r = 0; r < k; ++r) t_ = wave_extract_min(two, lane);
      if (lane == 0) wsel[0] = t_;   // (wave 0's own entries were read into `two` / `res` above)
    }
    __syncthreads();
    w = wsel[0];
  }
  const float T = w == TOPK_INF ? INFINITY : from_ordered_bits((uint32_t)(w >> 32));
  const float4 xa = xaux[row];
  const float nx = xa.y, rx = xa.z, xnr = xn[row];
  const float A = 2.0f * (rx + gamma * (nx + rx)), B = 2.0f * (nx + rx) * (1.0f + gamma);
  const float ymx = __uint_as_float(ymax[2]), rmx = __uint_as_float(ymax[3]);
  const float eps_any = fmaf(A, ymx, B * rmx) + 1e-6f * (xnr + ymx * ymx);
  const float bound = T + 2.0f * eps_any;
  float* lv = lval + (size_t)row * K2;
  int32_t* li = lidx + (size_t)row * K2;
  // members: the extracted keys (lane r's `res`) and what is left in `mine`, if not beyond the bound
  int count = 0;
  auto emit = [&](unsigned long long key) __attribute__((always_inline)) {
    const float v = from_ordered_bits((uint32_t)(key >> 32));
    const bool in = key != TOPK_INF && !(v > bound);              // (NaN on either side: keep)
    const unsigned long long mask = __ballot(in);
    int pos;
    if constexpr (WG) {
      int base = 0;
      if (lane == 0 && mask) base = atomicAdd(&s_count, __popcll(mask));
      base = __shfl(base, 0, 64);
      pos = base + __popcll(mask & ((1ull << lane) - 1ull));
    } else {
      pos = count + __popcll(mask & ((1ull << lane) - 1ull));
      count += __popcll(mask);
    }
    if (in && pos < K2) {
      lv[pos] = v;
      li[pos] = (int32_t)(uint32_t)(key & 0xffffffffu);
    }
  };
  emit(res);
#pragma unroll
  for (int q = 0; q < NQ; ++q) emit(mine[q]);
  if constexpr (WG) {
    __syncthreads();
    count = s_count;
  }
  if (count > K2) {
    if (tid == 0 && overflow) atomicOr(overflow, 1);
    count = K2;
  }
  for (int e = count + tid; e < K2; e += nthr) {
    lv[e] = INFINITY;
    li[e] = -1;
  }
}

// k beyond the register rounds above (k > SEL_MAX_K: the 12 x 10 ranks spatial NMS reads, evaluators.py:152-153 —
// examples/test.py:130 runs Tokyo 24/7 that way): one 256-thread workgroup per query holds its list (<= 256 NQ
// entries) in registers as (ordered value bits, index) pairs and finds T = the k-th smallest filter distance by
// BISECTION over the order-preserving 32-bit image of the values — at most 32 counting rounds of one compare per
// register, a wave sum and one barrier, whatever k is — instead of k extraction rounds.  T is a value, not a key:
// entries equal to T are all members, and so is everything within 2 eps_any of it (the same rule as above).
template <int NQ>
__global__ __launch_bounds__(256) void f16r_select_bisect_kernel(const float* __restrict__ cand_val,
                                                                 const int32_t* __restrict__ cand_idx,
                                                                 const int* __restrict__ cnt, int m, int cap, int k,
                                                                 int K2, const float* __restrict__ xn,
                                                                 const float4* __restrict__ xaux,
                                                                 const unsigned* __restrict__ ymax, float gamma,
                                                                 float* __restrict__ lval, int32_t* __restrict__ lidx,
                                                                 int* __restrict__ overflow) {
  __shared__ int s_part[2][4];
  __shared__ int s_count;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int row = blockIdx.x;
  int n = cnt[row];
  const int window = cap < 256 * NQ ? cap : 256 * NQ;
  if (n > window) {
    if (tid == 0 && overflow) atomicOr(overflow, 1);
    n = window;
  }
  const float* vr = cand_val + (size_t)row * cap;
  const int32_t* ir = cand_idx + (size_t)row * cap;
  uint32_t key[NQ];
  int32_t id[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int j = tid + 256 * q;
    key[q] = 0xffffffffu;
    id[q] = -1;
    if (j < n) {
      key[q] = ordered_bits(vr[j]);
      id[q] = ir[j];
    }
  }
  if (tid == 0) s_count = 0;
  // smallest T with #{key <= T} >= k (fewer than k entries: every entry is a member, T = the largest image)
  uint32_t lo = 0u, hi = 0xffffffffu;
  if (n > k) {
    for (int it = 0; lo < hi; ++it) {      // (lo, hi are workgroup-uniform: every thread sees the same counts)
      const uint32_t mid = lo + ((hi - lo) >> 1);
      int c = 0;
#pragma unroll
      for (int q = 0; q < NQ; ++q) c += (id[q] >= 0 && key[q] <= mid) ? 1 : 0;
#pragma unroll
      for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
      // two buffers by parity, one barrier per round: round it + 2 rewrites buffer (it & 1) behind barrier it + 1,
      // which every thread reaches only after it has read round it's counts
      if (lane == 0) s_part[it & 1][wave] = c;
      __syncthreads();
      const int total = s_part[it & 1][0] + s_part[it & 1][1] + s_part[it & 1][2] + s_part[it & 1][3];
      if (total >= k) hi = mid;
      else lo = mid + 1u;
    }
  } else {
    lo = 0xffffffffu;
  }
  __syncthreads();      // (s_count = 0 is visible; the bisection's barriers are conditional on n > k)
  const float T = lo == 0xffffffffu ? INFINITY : from_ordered_bits(lo);
  const float4 xa = xaux[row];
  const float nx = xa.y, rx = xa.z, xnr = xn[row];
  const float A = 2.0f * (rx + gamma * (nx + rx)), B = 2.0f * (nx + rx) * (1.0f + gamma);
  const float ymx = __uint_as_float(ymax[2]), rmx = __uint_as_float(ymax[3]);
  const float eps_any = fmaf(A, ymx, B * rmx) + 1e-6f * (xnr + ymx * ymx);
  const float bound = T + 2.0f * eps_any;
  float* lv = lval + (size_t)row * K2;
  int32_t* li = lidx + (size_t)row * K2;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const float v = from_ordered_bits(key[q]);
    const bool in = id[q] >= 0 && !(v > bound);                   // (NaN on either side: keep)
    const unsigned long long mask = __ballot(in);
    if (mask) {                                                   // wave-uniform
      int base = 0;
      if (lane == 0) base = atomicAdd(&s_count, __popcll(mask));
      base = __shfl(base, 0, 64);
      const int pos = base + __popcll(mask & ((1ull << lane) - 1ull));
      if (in && pos < K2) {
        lv[pos] = v;
        li[pos] = id[q];
      }
    }
  }
  __syncthreads();
  int count = s_count;
  if (count > K2) {
    if (tid == 0 && overflow) atomicOr(overflow, 1);
    count = K2;
  }
  for (int e = count + tid; e < K2; e += 256) {
    lv[e] = INFINITY;
    li[e] = -1;
  }
}

// Sharded matching, between the two stages: thr[row] = the k-th smallest filter distance over ALL shards' lists; an
// entry of this shard's list stays a member iff D_h <= thr + 2 eps_any, eps_any from the largest |y| / residual over
// all shards (ymax_all [W][2]); the others get index -1 (the rescoring skips them).
__global__ __launch_bounds__(256) void f16r_keep_members_kernel(const float* __restrict__ lval, int32_t* __restrict__ lidx,
                                                                int m, int K2, const float* __restrict__ thr,
                                                                const float* __restrict__ xn,
                                                                const float4* __restrict__ xaux,
                                                                const float* __restrict__ ymax_all, int W, float gamma) {
  const long e = (long)blockIdx.x * 256 + threadIdx.x;
  if (e >= (long)m * K2) return;
  const int row = (int)(e / K2);
  float ymx = 0.f, rmx = 0.f;
  for (int w = 0; w < W; ++w) {
    ymx = fmaxf(ymx, ymax_all[2 * w]);
    rmx = fmaxf(rmx, ymax_all[2 * w + 1]);
  }
  const float4 xa = xaux[row];
  const float nx = xa.y, rx = xa.z;
  const float A = 2.0f * (rx + gamma * (nx + rx)), B = 2.0f * (nx + rx) * (1.0f + gamma);
  const float eps_any = fmaf(A, ymx, B * rmx) + 1e-6f * (xn[row] + ymx * ymx);
  if (lval[e] > thr[row] + 2.0f * eps_any) lidx[e] = -1;        // (a NaN keeps the entry)
}

struct F16rRescoreParams {
  const void* xsrc;       // [m][d] stored rows (OIBL_ST_* of the instantiation)
  const void* ysrc;       // [n][d] stored rows
  const float* xn;
  const float* yn;
  const int32_t* lidx;    // [m][K2] members of the rescore set (global indices; -1 = padding), any order
  int m, d, k, K2, index_base;
  float* out_val;         // [m][k]
  int32_t* out_idx;
};

__device__ static inline double wave_sum_f64(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned lo = __shfl_xor((unsigned)__double2loint(v), o, 64);
    const unsigned hi = __shfl_xor((unsigned)__double2hiint(v), o, 64);
    v += __hiloint2double((int)hi, (int)lo);
  }
  return v;
}

// one workgroup (4 waves; 16 for the wide member windows of k > 16) per query: D of every member (one wave per pair:
// fp64 accumulation over the resident stored rows, widened exactly), then the k smallest (D, index)
template <int XST, int YST>
__global__ __launch_bounds__(1024) void f16r_rescore_kernel(F16rRescoreParams p) {
  __shared__ float s_val[F16R_MAX_K2];
  __shared__ int s_idx[F16R_MAX_K2];
  __shared__ short s_list[F16R_MAX_K2];   // the occupied slots, compacted (any order): the waves share them evenly
  __shared__ int s_nmem;
  const int q = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nthr = (int)blockDim.x, nwave = nthr >> 6;
  const int K2 = p.K2, k = p.k;
  if (tid == 0) s_nmem = 0;
  __syncthreads();
  for (int e = tid; e < K2; e += nthr) {
    const int id = p.lidx[(size_t)q * K2 + e];
    s_idx[e] = id;
    if (id >= 0) s_list[atomicAdd(&s_nmem, 1)] = (short)e;
  }
  __syncthreads();
  const int nmem = s_nmem;
  const float xn = p.xn[q];
  const char* xr = static_cast<const char*>(p.xsrc) + (size_t)q * p.d * (XST == OIBL_ST_F32 ? 4 : 2);
  for (int c = wave; c < nmem; c += nwave) {
    const int e = s_list[c];                       // wave-uniform
    const int id = s_idx[e];
    const char* yr = static_cast<const char*>(p.ysrc) + (size_t)(id - p.index_base) * p.d * (YST == OIBL_ST_F32 ? 4 : 2);
    // four 16-byte (8-byte: 16-bit storage) loads of the gallery row in flight per lane — a member is one 16 KB row
    // nobody else reads: with one load in flight the kernel ran at 3.8 TB/s on 315 queries (round 6)
    double acc0 = 0.0, acc1 = 0.0;
#pragma unroll 4
    for (int i = lane * 4; i < p.d; i += 256) {
      const float4 a = load4_widen<XST>(xr, i);
      const float4 b = load4_widen<YST>(yr, i);
      acc0 = fma((double)a.x, (double)b.x, acc0);
      acc1 = fma((double)a.y, (double)b.y, acc1);
      acc0 = fma((double)a.z, (double)b.z, acc0);
      acc1 = fma((double)a.w, (double)b.w, acc1);
    }
    const double acc = wave_sum_f64(acc0 + acc1);
    if (lane == 0) {
      const float yn = p.yn[id - p.index_base];
      s_val[e] = (float)((double)(xn + yn) - 2.0 * acc);
    }
  }
  __syncthreads();
  // rank of every member among the members by (D, index): keys are unique (indices are)
  for (int e = tid; e < K2; e += nthr) {
    const int id = s_idx[e];
    if (id < 0) continue;
    const unsigned long long key = ((unsigned long long)ordered_bits(s_val[e]) << 32) | (unsigned)id;
    int rank = 0;
    for (int f = 0; f < K2; ++f) {
      const int idf = s_idx[f];
      const unsigned long long kf = ((unsigned long long)ordered_bits(s_val[f]) << 32) | (unsigned)idf;
      rank += (idf >= 0 && kf < key) ? 1 : 0;
    }
    if (rank < k) {
      p.out_val[(size_t)q * k + rank] = s_val[e];
      p.out_idx[(size_t)q * k + rank] = id;
    }
  }
  for (int r = nmem + tid; r < k; r += nthr) {     // fewer members than k (a gallery shorter than k): pad
    p.out_val[(size_t)q * k + r] = INFINITY;
    p.out_idx[(size_t)q * k + r] = -1;
  }
}

}  // namespace oibl
