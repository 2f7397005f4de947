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

}  // namespace oibl

using namespace oibl;

OIBL_HOOK(int, g_nv_slabs, 1);   // test hook: 0 = never split the aggregation over the pixels

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

size_t oibl_netvlad_workspace_bytes(int N, int P, int K, int C) {
  if (N <= 0 || P <= 0 || K <= 0 || C <= 0) return 0;
  const int slabs = nv_pixel_slabs(N, P);
  return nv_off_parts(N, P, K, C) + (slabs > 1 ? align_up((size_t)slabs * N * K * C * sizeof(float), 256) : 0);
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
  g_nv_slabs = on ? 1 : 0;
  return OIBL_OK;
}
#endif

}  // extern "C"
