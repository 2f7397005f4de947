// NetVLAD aggregation + intra-/L2 normalisation on gfx950.
// Reference behaviour: NetVLAD.forward (ibl/models/netvlad.py:44-61) and the normalisation the
// Embed* modules apply to its output (netvlad.py:78-80, 100-102, 202-204).
//
// The reference materialises residual[N][K][C][P] (157 MB / image).  Here the same numbers come
// from two contractions that read the NHWC feature map once each:
//   logits[p][k] = (x_p . w_k) / max(|x_p|, eps)             NT GEMM on the MFMA core, softmax
//                                                            over the K = 64 columns in the epilogue
//   vlad[k][c]   = sum_p a[p][k] * xh[p][c]  -  (sum_p a[p][k]) * centroids[k][c]
//                                                            exact-fp32 MFMA (v_mfma_f32_32x32x2_f32,
//                                                            one operand element per lane, so both
//                                                            a[P][K] and x[P][C] are read in their
//                                                            natural layouts), xh formed on the fly
// followed by a per-image normalisation pass.  All NetVLAD arithmetic after the logits is fp32 in
// both precisions; `precision` only selects the element type of the feature map and the MFMA used
// for the logits.
#include "gemm_core.h"

namespace oibl {

// 1 / max(||x_p||_2, 1e-12) per position (F.normalize(x, p=2, dim=1), netvlad.py:46-47).
template <typename T>
__global__ void row_invnorm_kernel(const T* __restrict__ feat, float* __restrict__ inv, long rows,
                                   int C, int normalize) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= rows) return;
  float s = 0.f;
  if (normalize) {
    const T* xr = feat + row * C;
    for (int i = lane; i < C; i += 64) {
      const float v = Elem<T>::load(xr + i);
      s = fmaf(v, v, s);
    }
    s = wave_sum(s);
  }
  if (lane == 0) inv[row] = normalize ? 1.0f / fmaxf(sqrtf(s), 1e-12f) : 1.0f;
}

// soft-assignment: a[m][0..63] = softmax_k(inv[m] * (x_m . w_k))
template <typename Cfg, bool GLDS>
__global__ __launch_bounds__(Cfg::NTHREADS) void netvlad_assign_kernel(
    const void* feat, const void* w, const float* __restrict__ inv, float* __restrict__ a,
    long rows, int C) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = typename Cfg::T;
  static_assert(Cfg::BN == 64 && Cfg::BM == 128 && Cfg::NTHREADS == 256, "assign tile is 128x64");
  const WaveCoord c = wave_coord<Cfg>();
  const long m0 = (long)blockIdx.x * Cfg::BM;

  RowLoader<Cfg, Cfg::A_LOADS> la;
  RowLoader<Cfg, Cfg::B_LOADS> lb;
  la.init(c, feat, m0, rows, (long)C * sizeof(T));
  lb.init(c, w, 0, 64, (long)C * sizeof(T));

  f32x16_t acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gemm_nt_mainloop<Cfg, GLDS>(acc, smem, c, la, lb, C / Cfg::BK);

  constexpr int LP = 65;  // padded row pitch (floats)
  float* L = reinterpret_cast<float*>(smem);
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (c.wm * Cfg::TM + i) * 32 + acc_row(r, c.lane);
        const int col = (c.wn * Cfg::TN + j) * 32 + (c.lane & 31);
        long m = m0 + row;
        if (m > rows - 1) m = rows - 1;
        L[row * LP + col] = acc[i][j][r] * inv[m];
      }
  __syncthreads();
  // two threads per row, 32 columns each
  const int row = threadIdx.x >> 1, half = threadIdx.x & 1;
  const float* lr = L + row * LP + half * 32;
  float mx = -INFINITY;
#pragma unroll
  for (int k = 0; k < 32; ++k) mx = fmaxf(mx, lr[k]);
  mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
  float e[32], s = 0.f;
#pragma unroll
  for (int k = 0; k < 32; ++k) {
    e[k] = expf(lr[k] - mx);
    s += e[k];
  }
  s += __shfl_xor(s, 1, 64);
  const float is = 1.0f / s;
  const long m = m0 + row;
  if (m < rows) {
    float4* dst = reinterpret_cast<float4*>(a + m * 64 + half * 32);
#pragma unroll
    for (int k = 0; k < 8; ++k)
      dst[k] = make_float4(e[4 * k] * is, e[4 * k + 1] * is, e[4 * k + 2] * is, e[4 * k + 3] * is);
  }
}

// vlad_raw[n][k][c0..c0+63] for one image n and one 64-channel slice; K = 64 clusters.
// 4 waves as 2 (clusters) x 2 (channels), one 32x32 fp32 accumulator tile each.
// gridDim.z > 1 (few images: N x C / 64 workgroups walking all P pixels one chunk after the other is a
// latency chain on 8 CUs): workgroup z takes the pixels [z pchunk, (z + 1) pchunk) and writes its partial
// sum — the expression is linear in the pixels — to slab z of `vlad_raw` (slab stride N K C);
// netvlad_rowstats_kernel adds the slabs in fixed order.
template <typename T>
__global__ __launch_bounds__(256) void netvlad_aggregate_kernel(
    const T* __restrict__ feat, const float* __restrict__ inv, const float* __restrict__ a,
    const float* __restrict__ centroids, float* __restrict__ vlad_raw, int P, int C, int pchunk) {
  __shared__ __attribute__((aligned(16))) float a_s[32][64];
  __shared__ __attribute__((aligned(16))) float x_s[32][64];
  __shared__ float s_sum[64];
  const int n = blockIdx.x, c0 = blockIdx.y * 64;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int pz = blockIdx.z * pchunk;                       // first pixel of this workgroup
  const T* fbase = feat + ((size_t)n * P + pz) * C + c0;
  const float* abase = a + ((size_t)n * P + pz) * 64;
  const float* ibase = inv + (size_t)n * P + pz;
  vlad_raw += (size_t)blockIdx.z * gridDim.x * 64 * C;
  P = P - pz < pchunk ? P - pz : pchunk;                   // pixels of this workgroup (<= 0: a zero slab)

  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float colsum = 0.f;  // threads 0..63: sum_p a[p][tid]

  // staging roles: a chunk = 32 x 64 floats = 512 float4 (2 per thread);
  //                x chunk = 32 pixels x 64 channels (8 threads per pixel, 8 channels each).
  // The next chunk's global loads are issued before the current chunk's MFMAs (register
  // prefetch): one workgroup per CU has nothing else to hide the load latency behind.
  const int xp = threadIdx.x >> 3, xc = (threadIdx.x & 7) * 8;
  float4 pa[2];
  uint4 px0, px1;  // bf16: px0 only (8 elements); fp32: px0, px1
  float psc;
  auto prefetch = [&](int p0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int idx = threadIdx.x + q * 256;  // float4 index
      const int pr = idx >> 4, cq = (idx & 15) * 4;
      pa[q] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (p0 + pr < P) pa[q] = *reinterpret_cast<const float4*>(abase + (size_t)(p0 + pr) * 64 + cq);
    }
    px0 = make_uint4(0, 0, 0, 0);
    px1 = make_uint4(0, 0, 0, 0);
    psc = 0.f;
    if (p0 + xp < P) {
      psc = ibase[p0 + xp];
      const T* src = fbase + (size_t)(p0 + xp) * C + xc;
      px0 = *reinterpret_cast<const uint4*>(src);
      if constexpr (sizeof(T) == 4) px1 = *reinterpret_cast<const uint4*>(src + 4);
    }
  };
  if (P > 0) prefetch(0);
  for (int p0 = 0; p0 < P; p0 += 32) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      const int idx = threadIdx.x + q * 256;
      *reinterpret_cast<float4*>(&a_s[idx >> 4][(idx & 15) * 4]) = pa[q];
    }
    {
      float xv[8];
      if constexpr (sizeof(T) == 2) {
        const uint32_t wds[4] = {px0.x, px0.y, px0.z, px0.w};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          xv[2 * k] = bf16_bits_to_f32((uint16_t)(wds[k] & 0xffffu)) * psc;
          xv[2 * k + 1] = bf16_bits_to_f32((uint16_t)(wds[k] >> 16)) * psc;
        }
      } else {
        const uint32_t wds[8] = {px0.x, px0.y, px0.z, px0.w, px1.x, px1.y, px1.z, px1.w};
#pragma unroll
        for (int k = 0; k < 8; ++k) xv[k] = __uint_as_float(wds[k]) * psc;
      }
      *reinterpret_cast<float4*>(&x_s[xp][xc]) = make_float4(xv[0], xv[1], xv[2], xv[3]);
      *reinterpret_cast<float4*>(&x_s[xp][xc + 4]) = make_float4(xv[4], xv[5], xv[6], xv[7]);
    }
    __syncthreads();
    if (p0 + 32 < P) prefetch(p0 + 32);
    if (threadIdx.x < 64) {
#pragma unroll
      for (int p = 0; p < 32; ++p) colsum += a_s[p][threadIdx.x];
    }
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const int p = 2 * s + (lane >> 5);
      const float av = a_s[p][wm * 32 + (lane & 31)];
      const float bv = x_s[p][wn * 32 + (lane & 31)];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    __syncthreads();
  }
  if (threadIdx.x < 64) s_sum[threadIdx.x] = colsum;
  __syncthreads();
  const int ch = c0 + wn * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int k = wm * 32 + acc_row(r, lane);
    const float v = acc[r] - s_sum[k] * centroids[(size_t)k * C + ch];
    vlad_raw[((size_t)n * 64 + k) * C + ch] = v;
  }
}

// intra-normalise every cluster row, then L2-normalise the flattened K*C vector (k-major).
// Two launches with one wave per (image, cluster) row — N*K/4 workgroups instead of N:
//   rowstats: iv = 1 / max(|r|, eps) and s2 = |r * iv|^2 of every row;
//   apply   : ginv = 1 / max(sqrt(sum_k s2[n][k]), eps) (fixed-order wave reduction), out = r * iv * ginv.
// slabs > 1: `parts` holds that many partial sums of the raw rows (netvlad_aggregate_kernel with
// gridDim.z > 1); this kernel adds them in slab order, writes the row to `raw` and goes on with it.
__global__ __launch_bounds__(256) void netvlad_rowstats_kernel(float* __restrict__ raw,
                                                               const float* __restrict__ parts, int slabs,
                                                               float* __restrict__ stats, long rows,
                                                               int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  float* r = raw + row * C;
  float s = 0.f;
  if (slabs > 1) {
    for (int i = lane; i < C; i += 64) {
      float v = parts[row * C + i];
      for (int z = 1; z < slabs; ++z) v += parts[((size_t)z * rows + row) * C + i];
      r[i] = v;   // read back below by this same thread
    }
  }
  for (int i = lane; i < C; i += 64) s = fmaf(r[i], r[i], s);
  s = wave_sum(s);
  const float iv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
  float s2 = 0.f;
  for (int i = lane; i < C; i += 64) {
    const float v = r[i] * iv;
    s2 = fmaf(v, v, s2);
  }
  s2 = wave_sum(s2);
  if (lane == 0) {
    stats[2 * row] = iv;
    stats[2 * row + 1] = s2;
  }
}

__global__ __launch_bounds__(256) void netvlad_apply_kernel(const float* __restrict__ raw,
                                                            const float* __restrict__ stats,
                                                            float* __restrict__ out, long rows, int K,
                                                            int C) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const long n = row / K;
  float t = 0.f;
  for (int k = lane; k < K; k += 64) t += stats[2 * (n * K + k) + 1];
  t = wave_sum(t);
  const float iv = stats[2 * row];
  const float ginv = 1.0f / fmaxf(sqrtf(t), 1e-12f);
  const float* r = raw + row * C;
  float* o = out + row * C;
  for (int i = lane * 4; i < C; i += 256) {  // C % 64 == 0 and 16-byte aligned rows
    const float4 v = *reinterpret_cast<const float4*>(r + i);
    *reinterpret_cast<float4*>(o + i) =
        make_float4(v.x * iv * ginv, v.y * iv * ginv, v.z * iv * ginv, v.w * iv * ginv);
  }
}


// ---------------------------------------------------------------------------------------------
// The FUSED NetVLAD layer (fp32 feature map: the head of the fp32 / bf16x3 / f16mx arithmetics) — north_star's
// "soft-assignment softmax, residual accumulation and intra-/L2-norm as a fused kernel with coalesced reads of the
// H x W x 512 feature map" (ibl/models/netvlad.py:44-61, 100-102), three launches instead of five, the map read ONCE:
//
//   netvlad_fused_kernel     one workgroup per (image, slab of pixels).  Per chunk of 32 pixels: the chunk's 32 x
//                            512 values go to LDS with coalesced 16-byte loads (64.5 KB, rows 4 floats apart in the
//                            banks); 1 / |x_p| from LDS; logits = x . w^T on v_mfma_f32_32x32x2_f32 — every wave
//                            contracts its 128 channels, the four partial [32 x 64] tiles are added through LDS —
//                            scaled by 1 / |x_p|; softmax over the 64 clusters; aggregation
//                            sum_p a[p][k] / |x_p| * x[p][c] on the same instruction from the same LDS chunk, 64
//                            clusters x 128 channels of accumulators per wave (128 VGPRs), kept across the slab's
//                            chunks; at the end acc - (sum_p a[p][k]) centroids[k][c] -> parts[slab][n][k][c].
//                            The assignment weights of a wave's channels live in 128 registers for the whole slab,
//                            the next chunk is prefetched into registers under the current one's matrix work.
//   netvlad_rowstats_kernel  adds the slabs in slab order, one wave per (image, cluster) row, + the row statistics;
//   netvlad_apply_kernel     intra-norm + L2 over the 32768-vector (netvlad.py:100-102) — the two normalising
//                            launches of the five-launch path (a single-workgroup-per-image finalize was tried:
//                            132 us at batch 32 against 112 for five launches — its serial slab sums — and dropped).
// Three launches, the map read ONCE (five launches: three times).  Slab = 160 pixels: 8 slabs of a 30 x 40 map, 256
// workgroups at batch 32: 97 us against 112 (the kernel's phases — norms, logits, softmax, aggregation — are
// serialised by workgroup barriers with one workgroup per CU: ~45 % of its time is on the matrix pipe).  Batches
// below 16 images keep the five-launch path (79 against 86 us at batch 8; 51 us for one image with its 4-slab
// aggregation).  Exact fp32 throughout; against the five-launch path the sums differ by association only.
constexpr int NVF_XP = 516;          // floats per LDS row of the chunk: 16-byte aligned, +4 banks per pixel
constexpr int NVF_LP = 65;           // pitch of the [32][64] logit / assignment tiles
constexpr int NVF_LDS = (32 * NVF_XP + 4 * 32 * NVF_LP + 2 * 32 * NVF_LP + 32 + 64) * 4;

__global__ __launch_bounds__(256) void netvlad_fused_kernel(const float* __restrict__ feat, const float* __restrict__ w,
                                                            const float* __restrict__ centroids,
                                                            float* __restrict__ parts, int P, int slab_px,
                                                            int normalize) {
  constexpr int C = 512;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  float* const x_s = reinterpret_cast<float*>(smem);             // [32][NVF_XP]
  float* const lp_s = x_s + 32 * NVF_XP;                          // [4 waves][32][NVF_LP] partial logits
  float* const a_s = lp_s + 4 * 32 * NVF_LP;                      // [32][NVF_LP] a[p][k]
  float* const a2_s = a_s + 32 * NVF_LP;                          // [32][NVF_LP] a[p][k] / |x_p|
  float* const inv_s = a2_s + 32 * NVF_LP;                        // [32]
  float* const cs_s = inv_s + 32;                                 // [64] sum_p a[p][k] of the slab
  const int n = blockIdx.x, slab = blockIdx.y;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int l31 = lane & 31, kh = lane >> 5;
  const int p_lo = slab * slab_px;
  int p_hi = p_lo + slab_px;
  if (p_hi > P) p_hi = P;
  const float* fimg = feat + (size_t)n * P * C;

  f32x16_t acc[2][4];          // [cluster tile][channel tile of this wave's 128 channels]
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[kt][ct][r] = 0.f;
  float colsum = 0.f;          // threads 0..63
  // the assignment weights of this wave's 128 channels stay in registers for the whole slab (128 VGPRs: the kernel
  // runs one wave per SIMD): lane (cluster l31 / 32 + l31, k half kh) holds w[cluster][128 wave + 8 j + 4 kh ..+3]
  float4 wr[2][16];
  {
    const float* wb = w + (size_t)l31 * C + 128 * wave + 4 * kh;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      wr[0][j] = *reinterpret_cast<const float4*>(wb + 8 * j);
      wr[1][j] = *reinterpret_cast<const float4*>(wb + (size_t)32 * C + 8 * j);
    }
  }
  // register prefetch of the NEXT chunk (one workgroup per CU has nothing else to hide the load latency behind)
  float4 pf[16];
  auto prefetch = [&](int p0) __attribute__((always_inline)) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = (int)threadIdx.x + 256 * q;          // float4 index inside the chunk
      const int px = idx >> 7, c4 = (idx & 127) * 4;
      pf[q] = make_float4(0.f, 0.f, 0.f, 0.f);             // a pixel beyond the slab reads as zeros (a = 0 below)
      if (p0 + px < p_hi) pf[q] = *reinterpret_cast<const float4*>(fimg + (size_t)(p0 + px) * C + c4);
    }
  };
  prefetch(p_lo);

  for (int p0 = p_lo; p0 < p_hi; p0 += 32) {
    // ---- the chunk -> LDS
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int idx = (int)threadIdx.x + 256 * q;
      *reinterpret_cast<float4*>(x_s + (idx >> 7) * NVF_XP + (idx & 127) * 4) = pf[q];
    }
    __syncthreads();
    if (p0 + 32 < p_hi) prefetch(p0 + 32);
    // ---- 1 / |x_p|: eight threads per pixel, interleaved float4s
    {
      const int px = (int)threadIdx.x >> 3, sub = (int)threadIdx.x & 7;
      float ss = 0.f;
      if (normalize) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          const float4 v = *reinterpret_cast<const float4*>(x_s + px * NVF_XP + 4 * (sub + 8 * j));
          ss = fmaf(v.x, v.x, ss);
          ss = fmaf(v.y, v.y, ss);
          ss = fmaf(v.z, v.z, ss);
          ss = fmaf(v.w, v.w, ss);
        }
        ss += __shfl_xor(ss, 1, 64);
        ss += __shfl_xor(ss, 2, 64);
        ss += __shfl_xor(ss, 4, 64);
      }
      if (sub == 0) inv_s[px] = normalize ? 1.0f / fmaxf(sqrtf(ss), 1e-12f) : 1.0f;
    }
    // ---- partial logits of this wave's 128 channels: [32 pixels] x [64 clusters]
    {
      f32x16_t lg[2];
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) lg[ct][r] = 0.f;
      const float* xa = x_s + l31 * NVF_XP + 128 * wave + 4 * kh;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float4 a = *reinterpret_cast<const float4*>(xa + 8 * j);
        const float4 b0 = wr[0][j], b1 = wr[1][j];
        lg[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b0.x, lg[0], 0, 0, 0);
        lg[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b1.x, lg[1], 0, 0, 0);
        lg[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b0.y, lg[0], 0, 0, 0);
        lg[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b1.y, lg[1], 0, 0, 0);
        lg[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b0.z, lg[0], 0, 0, 0);
        lg[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b1.z, lg[1], 0, 0, 0);
        lg[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b0.w, lg[0], 0, 0, 0);
        lg[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b1.w, lg[1], 0, 0, 0);
      }
      float* lw = lp_s + wave * 32 * NVF_LP;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct)
#pragma unroll
        for (int r = 0; r < 16; ++r) lw[acc_row(r, lane) * NVF_LP + 32 * ct + l31] = lg[ct][r];
    }
    __syncthreads();
    // ---- softmax over the 64 clusters: eight threads per pixel, eight clusters each
    {
      const int px = (int)threadIdx.x >> 3, sub = (int)threadIdx.x & 7;
      const float iv = inv_s[px];
      float l[8], mx = -INFINITY;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int o = px * NVF_LP + sub * 8 + k;
        l[k] = (lp_s[o] + lp_s[32 * NVF_LP + o] + lp_s[2 * 32 * NVF_LP + o] + lp_s[3 * 32 * NVF_LP + o]) * iv;
        mx = fmaxf(mx, l[k]);
      }
      mx = fmaxf(mx, __shfl_xor(mx, 1, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 2, 64));
      mx = fmaxf(mx, __shfl_xor(mx, 4, 64));
      float ssum = 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        l[k] = expf(l[k] - mx);
        ssum += l[k];
      }
      ssum += __shfl_xor(ssum, 1, 64);
      ssum += __shfl_xor(ssum, 2, 64);
      ssum += __shfl_xor(ssum, 4, 64);
      const float is = (p0 + px < p_hi) ? 1.0f / ssum : 0.f;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const float a = l[k] * is;
        a_s[px * NVF_LP + sub * 8 + k] = a;
        a2_s[px * NVF_LP + sub * 8 + k] = a * iv;
      }
    }
    __syncthreads();
    if (threadIdx.x < 64) {
#pragma unroll
      for (int p = 0; p < 32; ++p) colsum += a_s[p * NVF_LP + threadIdx.x];
    }
    // ---- aggregation: acc[k][c] += sum_p (a[p][k] / |x_p|) x[p][c]
#pragma unroll 2
    for (int s2 = 0; s2 < 16; ++s2) {
      const int p = 2 * s2 + kh;
      const float av0 = a2_s[p * NVF_LP + l31], av1 = a2_s[p * NVF_LP + 32 + l31];
#pragma unroll
      for (int ct = 0; ct < 4; ++ct) {
        const float bv = x_s[p * NVF_XP + 128 * wave + 32 * ct + l31];
        acc[0][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av0, bv, acc[0][ct], 0, 0, 0);
        acc[1][ct] = __builtin_amdgcn_mfma_f32_32x32x2f32(av1, bv, acc[1][ct], 0, 0, 0);
      }
    }
    __syncthreads();
  }
  if (threadIdx.x < 64) cs_s[threadIdx.x] = colsum;
  __syncthreads();
  float* out = parts + ((size_t)slab * gridDim.x + n) * 64 * C;
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
      const int ch = 128 * wave + 32 * ct + l31;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int k = 32 * kt + acc_row(r, lane);
        out[(size_t)k * C + ch] = acc[kt][ct][r] - cs_s[k] * centroids[(size_t)k * C + ch];
      }
    }
}

}  // namespace oibl

using namespace oibl;

OIBL_HOOK(int, g_nv_slabs, 1);   // test hook: 1 = default (fp32 maps: the fused kernel; bf16 maps: five launches, the
                                 // aggregation split over the pixels for few images), 0 = five launches, never split,
                                 // 2 = five launches with the pixel split for fp32 maps too

extern "C" {

static size_t nv_off_a(int N, int P) { return align_up((size_t)N * P * sizeof(float), 256); }
static size_t nv_off_raw(int N, int P) {
  return nv_off_a(N, P) + align_up((size_t)N * P * 64 * sizeof(float), 256);
}
static size_t nv_off_w(int N, int P, int K, int C) {
  return nv_off_raw(N, P) + align_up((size_t)N * K * C * sizeof(float), 256);
}

static size_t nv_off_stats(int N, int P, int K, int C) {
  return nv_off_w(N, P, K, C) + align_up((size_t)K * C * sizeof(uint16_t), 256);
}

// pixel split of the aggregation for few images (the same for every N it applies to: a row's result must
// not depend on its batch mates within a kernel selection)
static int nv_pixel_slabs(int N, int P) { return (N <= 4 && P >= 256) ? 4 : 1; }
static size_t nv_off_parts(int N, int P, int K, int C) {
  return nv_off_stats(N, P, K, C) + align_up((size_t)N * K * 2 * sizeof(float), 256);
}

// the fused kernel's slabs: 160 pixels; it serves batches of 16 images and more (8 slabs of a 30 x 40 map: from 128
// workgroups on; measured 97 against 112 us for five launches at batch 32, 86 against 79 at batch 8)
// Round 6 ("fused NetVLAD for N < 16 with the pixel-slab split inside the kernel"): built — the slab
// shrinks with the batch (96 pixels below 16 images, 64 below 8, 32 = one chunk below 4: N = 1 is 38 workgroups) — and
// MEASURED (tests/gpu_head_bench.py, profiles/r06_h_head_bench.txt; fused | five launches, us): N = 32: 97 | 113,
// 16: 87 | 84, 12: 79 | 79, 8: 78 | 78, 4: 80 | 54, 2: 110 | 52, 1: 100 | 51.  Every workgroup of the fused kernel
// first loads its waves' 128 KB of assignment weights into registers and ends in a 128 KB partial: fixed costs a
// 32-pixel slab does not amortise, while the five launches' pixel-split aggregation has none.  So the fused kernel
// keeps serving batches of 16 images and more, the five launches everything below (hook 3 forces the fused kernel at
// any N, hook 2 the five launches).  The slab size is a function of N alone (a row's result must not depend on its
// batch mates within a kernel selection).
constexpr int NVF_MIN_N = 16;
static int nvf_slab_px(int N) { return N >= 16 ? 160 : N >= 8 ? 96 : N >= 4 ? 64 : 32; }
static int nvf_slabs(int N, int P) { return (P + nvf_slab_px(N) - 1) / nvf_slab_px(N); }

size_t oibl_netvlad_workspace_bytes(int N, int P, int K, int C) {
  if (N <= 0 || P <= 0 || K <= 0 || C <= 0) return 0;
  int slabs = nv_pixel_slabs(N, P);
  if (nvf_slabs(N, P) > slabs) slabs = nvf_slabs(N, P);
  return nv_off_parts(N, P, K, C) + align_up((size_t)slabs * N * K * C * sizeof(float), 256);
}

int oibl_netvlad_forward(const void* feat, int N, int P, int K, int C, int precision,
                         const float* assign_w, const float* centroids, int normalize_input,
                         float* vlad_raw, float* vlad_norm, void* ws, size_t ws_bytes,
                         void* stream) {
  OIBL_REQUIRE(feat && assign_w && centroids && ws, "netvlad: null pointer");
  OIBL_REQUIRE(vlad_raw || vlad_norm, "netvlad: no output requested");
  OIBL_REQUIRE(precision == OIBL_BF16 || precision == OIBL_F32, "netvlad: bad precision %d",
               precision);
  OIBL_REQUIRE(K == 64, "netvlad: kernels are built for num_clusters = 64 (got %d)", K);
  OIBL_REQUIRE(C > 0 && C % 64 == 0, "netvlad: dim must be a multiple of 64 (got %d)", C);
  OIBL_REQUIRE(N > 0 && P > 0, "netvlad: bad shape N=%d P=%d", N, P);
  OIBL_REQUIRE((uintptr_t)ws % 256 == 0 && (uintptr_t)feat % 16 == 0,
               "netvlad: workspace must be 256-byte and feat 16-byte aligned");
  const size_t need = oibl_netvlad_workspace_bytes(N, P, K, C);
  if (ws_bytes < need) {
    set_error("netvlad: workspace %zu < required %zu bytes", ws_bytes, need);
    return OIBL_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* wsb = (char*)ws;
  float* inv = (float*)wsb;
  float* a = (float*)(wsb + nv_off_a(N, P));
  float* raw = vlad_raw ? vlad_raw : (float*)(wsb + nv_off_raw(N, P));
  void* w_t = wsb + nv_off_w(N, P, K, C);
  const long rows = (long)N * P;
  if (precision == OIBL_F32 && ((g_nv_slabs == 1 && N >= NVF_MIN_N) || g_nv_slabs == 3) && C == 512 &&
      (uintptr_t)assign_w % 16 == 0) {
    // the fused layer: the map read once (netvlad_fused_kernel), then the slab sum + the two normalising launches
    const int ns = nvf_slabs(N, P);
    float* parts = ns > 1 ? (float*)(wsb + nv_off_parts(N, P, K, C)) : raw;   // (one slab: it IS the raw output)
    OIBL_SET_MAX_LDS(netvlad_fused_kernel, NVF_LDS);
    hipLaunchKernelGGL(netvlad_fused_kernel, dim3((unsigned)N, (unsigned)ns), dim3(256), NVF_LDS, st, (const float*)feat,
                       assign_w, centroids, parts, P, nvf_slab_px(N), normalize_input);
    OIBL_LAUNCH_CHECK();
    float* stats = (float*)(wsb + nv_off_stats(N, P, K, C));
    const long vrows = (long)N * K;
    const unsigned fgrid = (unsigned)((vrows + 3) / 4);
    hipLaunchKernelGGL(netvlad_rowstats_kernel, dim3(fgrid), dim3(256), 0, st, raw, (const float*)parts, ns, stats,
                       vrows, C);
    OIBL_LAUNCH_CHECK();
    if (vlad_norm) {
      hipLaunchKernelGGL(netvlad_apply_kernel, dim3(fgrid), dim3(256), 0, st, raw, stats, vlad_norm, vrows, K, C);
      OIBL_LAUNCH_CHECK();
    }
    return OIBL_OK;
  }
  // few images: the aggregation is split over the pixels (only when the normalised output is wanted: the
  // kernel that normalises is the one that adds the slabs)
  const int slabs = (vlad_norm && g_nv_slabs) ? nv_pixel_slabs(N, P) : 1;
  float* agg_out = slabs > 1 ? (float*)(wsb + nv_off_parts(N, P, K, C)) : raw;
  const int pchunk = slabs > 1 ? (((P + slabs - 1) / slabs + 31) / 32) * 32 : P;

  const unsigned inv_grid = (unsigned)((rows + 3) / 4);
  const unsigned asg_grid = (unsigned)((rows + 127) / 128);
  if (precision == OIBL_BF16) {
    int rc = oibl_cast_f32_to_bf16(assign_w, (uint16_t*)w_t, (size_t)K * C, stream);
    if (rc) return rc;
    hipLaunchKernelGGL(row_invnorm_kernel<bf16_t>, dim3(inv_grid), dim3(256), 0, st,
                       (const bf16_t*)feat, inv, rows, C, normalize_input);
    OIBL_LAUNCH_CHECK();
    using Cfg = GemmCfg<bf16_t, 2, 2, 2, 1>;
    if (g_regstage)
      hipLaunchKernelGGL((netvlad_assign_kernel<Cfg, false>), dim3(asg_grid), dim3(256),
                         Cfg::MAIN_LDS_BYTES, st, feat, (const void*)w_t, inv, a, rows, C);
    else
      hipLaunchKernelGGL((netvlad_assign_kernel<Cfg, true>), dim3(asg_grid), dim3(256),
                         Cfg::MAIN_LDS_BYTES, st, feat, (const void*)w_t, inv, a, rows, C);
    OIBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(netvlad_aggregate_kernel<bf16_t>, dim3(N, C / 64, slabs), dim3(256), 0, st,
                       (const bf16_t*)feat, inv, a, centroids, agg_out, P, C, pchunk);
    OIBL_LAUNCH_CHECK();
  } else {
    OIBL_REQUIRE((uintptr_t)assign_w % 16 == 0, "netvlad: assign_w must be 16-byte aligned");
    hipLaunchKernelGGL(row_invnorm_kernel<float>, dim3(inv_grid), dim3(256), 0, st,
                       (const float*)feat, inv, rows, C, normalize_input);
    OIBL_LAUNCH_CHECK();
    using Cfg = GemmCfg<float, 2, 2, 2, 1>;
    if (g_regstage)
      hipLaunchKernelGGL((netvlad_assign_kernel<Cfg, false>), dim3(asg_grid), dim3(256),
                         Cfg::MAIN_LDS_BYTES, st, feat, (const void*)assign_w, inv, a, rows, C);
    else
      hipLaunchKernelGGL((netvlad_assign_kernel<Cfg, true>), dim3(asg_grid), dim3(256),
                         Cfg::MAIN_LDS_BYTES, st, feat, (const void*)assign_w, inv, a, rows, C);
    OIBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(netvlad_aggregate_kernel<float>, dim3(N, C / 64, slabs), dim3(256), 0, st,
                       (const float*)feat, inv, a, centroids, agg_out, P, C, pchunk);
    OIBL_LAUNCH_CHECK();
  }
  if (vlad_norm) {
    float* stats = (float*)(wsb + nv_off_stats(N, P, K, C));
    const long vrows = (long)N * K;
    const unsigned fgrid = (unsigned)((vrows + 3) / 4);
    hipLaunchKernelGGL(netvlad_rowstats_kernel, dim3(fgrid), dim3(256), 0, st, raw, (const float*)agg_out, slabs,
                       stats, vrows, C);
    OIBL_LAUNCH_CHECK();
    hipLaunchKernelGGL(netvlad_apply_kernel, dim3(fgrid), dim3(256), 0, st, raw, stats, vlad_norm,
                       vrows, K, C);
    OIBL_LAUNCH_CHECK();
  }
  return OIBL_OK;
}

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_netvlad_slabs(int on) {
  g_nv_slabs = on < 0 ? 0 : (on > 3 ? 3 : on);
  return OIBL_OK;
}
#endif

}  // extern "C"
