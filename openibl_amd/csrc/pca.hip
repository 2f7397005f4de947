// PCA-whitening projection y = normalize(W v + b), 32768 -> 4096, on gfx950.
// Reference behaviour: EmbedNetPCA.forward's pca_layer + F.normalize (ibl/models/netvlad.py:105-108)
// and PCA.infer (ibl/pca.py:108-123) — a 1x1 convolution over a [N][D][1][1] tensor, i.e. a GEMM
// with M = batch.
//
// At batch <= 64 the contraction is bound by streaming W once from HBM (268 MB bf16 / 537 MB
// fp32); the kernel is the shared NT core with a 32 x 128 tile (one MFMA row-tile of batch rows,
// four waves side by side over 128 output dims), split along K so that every CU streams a
// disjoint [128 x D/S] panel of W; fp32 partial sums go to the workspace and a second kernel
// reduces them in fixed order, adds the bias and L2-normalises each row.
#include "gemm_core.h"

namespace oibl {

struct PcaParams {
  const void* v;  // [N][D] T
  const void* w;  // [d][D] T
  float* part;    // [splits][N][d]
  int N, D, d;
  int tiles_n, splits, ksteps_per_split;
};

template <typename Cfg, bool GLDS>
__global__ __launch_bounds__(Cfg::NTHREADS) void pca_partial_kernel(PcaParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = typename Cfg::T;
  const WaveCoord c = wave_coord<Cfg>();
  // block -> (split, m tile, n tile); splits of one output tile are adjacent
  const unsigned bid = blockIdx.x;
  const int split = bid % p.splits;
  const unsigned t = bid / p.splits;
  const int tn = t % p.tiles_n, tm = t / p.tiles_n;
  const long m0 = (long)tm * Cfg::BM, n0 = (long)tn * Cfg::BN;
  const long k0_bytes = (long)split * p.ksteps_per_split * 128;

  RowLoader<Cfg, Cfg::A_LOADS> la;
  RowLoader<Cfg, Cfg::B_LOADS> lb;
  la.init(c, reinterpret_cast<const char*>(p.v) + k0_bytes, m0, p.N, (long)p.D * sizeof(T));
  lb.init(c, reinterpret_cast<const char*>(p.w) + k0_bytes, n0, p.d, (long)p.D * sizeof(T));

  f32x16_t acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gemm_nt_mainloop<Cfg, GLDS>(acc, smem, c, la, lb, p.ksteps_per_split);

  float* out = p.part + (size_t)split * p.N * p.d;
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + (c.wm * Cfg::TM + i) * 32 + acc_row(r, c.lane);
        const long n = n0 + (c.wn * Cfg::TN + j) * 32 + (c.lane & 31);
        if (m < p.N) out[m * p.d + n] = acc[i][j][r];
      }
}

// out[n][:] = normalize(sum_s part[s][n][:] + b), two launches of N x ceil(d / 256) workgroups:
//   reduce: fixed-order sum of the split-K partials + bias, block partial of the squared norm;
//   scale : fixed-order sum of the block partials, out *= 1 / max(sqrt(.), eps).
__global__ __launch_bounds__(256) void pca_reduce_kernel(const float* __restrict__ part,
                                                         const float* __restrict__ bias,
                                                         float* __restrict__ out,
                                                         float* __restrict__ ss_part, int N, int d,
                                                         int splits) {
  __shared__ float red[4];
  const int n = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  float ss = 0.f;
  if (j < d) {
    float v = 0.f;
    for (int s = 0; s < splits; ++s) v += part[((size_t)s * N + n) * d + j];
    v += bias[j];
    out[(size_t)n * d + j] = v;
    ss = v * v;
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  if (threadIdx.x == 0) ss_part[(size_t)n * gridDim.x + blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

__global__ __launch_bounds__(256) void pca_scale_kernel(float* __restrict__ out,
                                                        const float* __restrict__ ss_part, int d) {
  const int n = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  float t = 0.f;
  for (unsigned b = 0; b < gridDim.x; ++b) t += ss_part[(size_t)n * gridDim.x + b];
  const float inv = 1.0f / fmaxf(sqrtf(t), 1e-12f);
  if (j < d) out[(size_t)n * d + j] *= inv;
}

static int pca_splits(int N, int D, int d, int precision) {
  const int bk = precision == OIBL_BF16 ? 64 : 32;
  const int ksteps = D / bk;
  const int tiles = ((N + 31) / 32) * (d / 128);
  int s = 1;
  // aim for >= 512 workgroups (2 per CU) while keeping >= 16 K-steps per split
  while (tiles * s < 512 && ksteps % (s * 2) == 0 && ksteps / (s * 2) >= 16) s *= 2;
  return s;
}

}  // namespace oibl

using namespace oibl;

extern "C" {

size_t oibl_pca_workspace_bytes(int N, int D, int d, int precision) {
  if (N <= 0 || D <= 0 || d <= 0) return 0;
  const int s = pca_splits(N, D, d, precision);
  return align_up((size_t)N * D * oibl_elem_size(precision), 256) +
         align_up((size_t)s * N * d * sizeof(float), 256) +
         align_up((size_t)N * ((d + 255) / 256) * sizeof(float), 256);
}

int oibl_pca_forward(const float* v, int N, int D, const void* w, const float* b, int d,
                     int precision, int l2norm, float* out, void* ws, size_t ws_bytes,
                     void* stream) {
  OIBL_REQUIRE(v && w && b && out && ws, "pca: null pointer");
  OIBL_REQUIRE(precision == OIBL_BF16 || precision == OIBL_F32, "pca: bad precision %d", precision);
  const int bk = precision == OIBL_BF16 ? 64 : 32;
  OIBL_REQUIRE(N <= 65535, "pca: at most 65535 rows per call (got %d)", N);
  OIBL_REQUIRE(N > 0 && D > 0 && d > 0 && D % bk == 0 && d % 128 == 0,
               "pca: unsupported shape N=%d D=%d d=%d", N, D, d);
  OIBL_REQUIRE((uintptr_t)ws % 256 == 0 && (uintptr_t)w % 16 == 0 && (uintptr_t)v % 16 == 0,
               "pca: workspace must be 256-byte, w and v 16-byte aligned");
  const size_t need = oibl_pca_workspace_bytes(N, D, d, precision);
  if (ws_bytes < need) {
    set_error("pca: workspace %zu < required %zu bytes", ws_bytes, need);
    return OIBL_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* wsb = (char*)ws;
  const void* v_t = v;
  if (precision == OIBL_BF16) {
    int rc = oibl_cast_f32_to_bf16(v, (uint16_t*)wsb, (size_t)N * D, stream);
    if (rc) return rc;
    v_t = wsb;
  }
  PcaParams p;
  p.v = v_t;
  p.w = w;
  p.part = (float*)(wsb + align_up((size_t)N * D * oibl_elem_size(precision), 256));
  p.N = N;
  p.D = D;
  p.d = d;
  p.tiles_n = d / 128;
  p.splits = pca_splits(N, D, d, precision);
  p.ksteps_per_split = D / bk / p.splits;
  const unsigned grid = (unsigned)(((N + 31) / 32) * p.tiles_n * p.splits);
  if (precision == OIBL_BF16) {
    using Cfg = GemmCfg<bf16_t, 1, 4, 1, 1>;
    if (g_regstage)
      hipLaunchKernelGGL((pca_partial_kernel<Cfg, false>), dim3(grid), dim3(Cfg::NTHREADS),
                         Cfg::MAIN_LDS_BYTES, st, p);
    else
      hipLaunchKernelGGL((pca_partial_kernel<Cfg, true>), dim3(grid), dim3(Cfg::NTHREADS),
                         Cfg::MAIN_LDS_BYTES, st, p);
  } else {
    using Cfg = GemmCfg<float, 1, 4, 1, 1>;
    if (g_regstage)
      hipLaunchKernelGGL((pca_partial_kernel<Cfg, false>), dim3(grid), dim3(Cfg::NTHREADS),
                         Cfg::MAIN_LDS_BYTES, st, p);
    else
      hipLaunchKernelGGL((pca_partial_kernel<Cfg, true>), dim3(grid), dim3(Cfg::NTHREADS),
                         Cfg::MAIN_LDS_BYTES, st, p);
  }
  OIBL_LAUNCH_CHECK();
  float* ss_part = (float*)((char*)p.part + align_up((size_t)p.splits * N * d * sizeof(float), 256));
  const dim3 rgrid((unsigned)((d + 255) / 256), (unsigned)N);
  hipLaunchKernelGGL(pca_reduce_kernel, rgrid, dim3(256), 0, st, p.part, b, out, ss_part, N, d,
                     p.splits);
  OIBL_LAUNCH_CHECK();
  if (l2norm) {
    hipLaunchKernelGGL(pca_scale_kernel, rgrid, dim3(256), 0, st, out, ss_part, d);
    OIBL_LAUNCH_CHECK();
  }
  return OIBL_OK;
}

}  // extern "C"
