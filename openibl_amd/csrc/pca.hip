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

// Reduction, bias, and the row's L2 normalisation in ONE launch for the packed stream (round 6; N <= 32 rows, d <= 4096):
// a workgroup of 1024 threads owns a row, a thread four of its dimensions, all 4 x 16 partials of the thread in flight
// at once (summed in a loop they were a chain of loads N workgroups cannot hide: 10 us slower than the two launches).
constexpr int PRN_PER = 4;
template <int SPLITS>
__global__ __launch_bounds__(1024) void pca_reduce_norm_kernel(const float* __restrict__ part,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ out, int N, int d, int l2norm) {
  __shared__ float red[16];
  const int n = blockIdx.x;
  float v[PRN_PER], ss = 0.f;
  float p[PRN_PER][SPLITS];       // every partial of the thread in flight at once: N workgroups cannot hide a chain
#pragma unroll
  for (int i = 0; i < PRN_PER; ++i) {
    const int j = min(threadIdx.x + 1024 * i, d - 1);
#pragma unroll
    for (int s = 0; s < SPLITS; ++s) p[i][s] = __builtin_nontemporal_load(part + ((size_t)s * N + n) * d + j);
  }
#pragma unroll
  for (int i = 0; i < PRN_PER; ++i) {
    const int j = threadIdx.x + 1024 * i;
    v[i] = 0.f;
    if (j < d) {
      float t = 0.f;
#pragma unroll
      for (int s = 0; s < SPLITS; ++s) t += p[i][s];
      v[i] = t + bias[j];
      ss = fmaf(v[i], v[i], ss);
    }
  }
  ss = wave_sum(ss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ss;
  __syncthreads();
  float inv = 1.f;
  if (l2norm) {
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < 16; ++w) t += red[w];
    inv = 1.0f / fmaxf(sqrtf(t), 1e-12f);
  }
#pragma unroll
  for (int i = 0; i < PRN_PER; ++i) {
    const int j = threadIdx.x + 1024 * i;
    if (j < d) out[(size_t)n * d + j] = v[i] * inv;
  }
}

__global__ __launch_bounds__(256) void pca_scale_kernel(float* __restrict__ out,
                                                        const float* __restrict__ ss_part, int d) {
  const int n = blockIdx.y, j = blockIdx.x * 256 + threadIdx.x;
  float t = 0.f;
  for (unsigned b = 0; b < gridDim.x; ++b) t += ss_part[(size_t)n * gridDim.x + b];
  const float inv = 1.0f / fmaxf(sqrtf(t), 1e-12f);
  if (j < d) out[(size_t)n * d + j] *= inv;
}

// Few rows (N <= 8, fp32): a streaming kernel instead of the MFMA tile — with a handful of images the
// contraction is a matrix-VECTOR product, 2 flop per 4 bytes of W, and the only thing that matters is
// how many bytes of W a CU keeps in flight.  A wave owns PS_ROWS output dims over one K range of W:
// every lane streams 16-byte pieces of those rows (non-temporal: W is read once) against the matching
// piece of the N input rows (L2-resident), 16 + 2 N loads in flight per lane; per-lane fp32 sums in a
// fixed order, one wave reduction at the end, partials [split][n][j] for pca_reduce_kernel.
// 537 MB of fp32 W: 125 us through the MFMA tile (4.3 TB/s), 86 us here (6.2 TB/s) for N = 1, 2.  From
// N = 4 on the input pieces — the same addresses from every workgroup of a split — cost more than the
// tile's LDS staging (tests/gpu_pca_bench.py: 140 us at N = 4): the tile keeps N > 2.
constexpr int PS_ROWS = 8, PS_SPLITS = 8, PS_MAXN = 2;
template <int NB>
__global__ __launch_bounds__(256) void pca_small_kernel(const float* __restrict__ v, const float* __restrict__ w,
                                                        float* __restrict__ part, int N, int D, int d, int kper) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j0 = (blockIdx.x * 4 + wave) * PS_ROWS;
  const long k0 = (long)blockIdx.y * kper;
  const float* wp = w + (long)j0 * D + k0 + lane * 4;
  const float* vp = v + k0 + lane * 4;
  float acc[PS_ROWS][NB];
#pragma unroll
  for (int r = 0; r < PS_ROWS; ++r)
#pragma unroll
    for (int n = 0; n < NB; ++n) acc[r][n] = 0.f;
  for (int k = 0; k < kper; k += 512) {
    f4 xv[2][NB], wv[2][PS_ROWS];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
#pragma unroll
      for (int r = 0; r < PS_ROWS; ++r)
        wv[u][r] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(wp + (long)r * D + k + u * 256));
#pragma unroll
      for (int n = 0; n < NB; ++n)
        xv[u][n] = n < N ? *reinterpret_cast<const f4*>(vp + (long)n * D + k + u * 256) : (f4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int r = 0; r < PS_ROWS; ++r)
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[r][n] = fmaf(wv[u][r][e], xv[u][n][e], acc[r][n]);
  }
#pragma unroll
  for (int r = 0; r < PS_ROWS; ++r)
#pragma unroll
    for (int n = 0; n < NB; ++n) {
      const float t = wave_sum(acc[r][n]);
      if (lane == 0 && n < N) part[((size_t)blockIdx.y * N + n) * d + j0 + r] = t;
    }
}

OIBL_HOOK(int, g_pca_small, 1);   // test hook: 0 = the MFMA tile for every N
static bool pca_small_ok(int N, int D, int d, int precision) {
  return precision == OIBL_F32 && N <= PS_MAXN && d % (4 * PS_ROWS) == 0 && D % (PS_SPLITS * 512) == 0;
}

// Streaming form for 3 .. 32 rows (round 5).  W as the B operand of v_mfma_f32_32x32x2_f32 read from the
// row-major matrix is 32 rows x 32 bytes per wave instruction — not a streaming pattern for the vector memory
// path (a kernel doing that ran 178 us = 3.0 TB/s).  So W is RE-PACKED ONCE (oibl_pca_pack_weight, a second
// 537 MB copy) into 1 KB tiles [32 output dims][8 k] laid out lane by lane as the operand registers want them
// (lane = (k half << 5) | dim: the four floats W[dim][8 kb + 4 half + 0..3]), the tiles of one group of 32
// output dims consecutive along k: a wave streams ONE contiguous 128 KB run (its 1/32 of K) with eight non-temporal
// 16-byte loads per lane in flight, four waves per SIMD.  A workgroup is 8 waves = 8 such groups over the same 1/32 of K; the matching piece of
// the input rows goes through LDS in the same lane order (chunks of 256 k, double-buffered, one barrier per
// chunk; rows beyond N repeat row N - 1 and are not stored).  fp32 products and sums (the matrix pipe needs
// 55 us of the 90 the stream takes); partials [32 splits][N][d] for pca_reduce_kernel.  With the reduction and
// the scaling: 93 us for 3 rows (5.8 TB/s of W), 94 for 8, 102 for 16, 110 for 32 (4.9 TB/s; the row-major tile:
// 137 us) — the growth with N is the partials (16.7 MB written and read at 32 rows).
constexpr int PK_WAVES = 8, PK_SPLITS = 32, PK_CHUNK = 256, PK_TILES = PK_CHUNK / 8, PK_MAXN = 32;
// KP = 2 (round 6, the default): a workgroup of 16 waves is TWO such workgroups over neighbouring K parts — waves 8-15
// the second part, with their own input chunks in LDS — and the second part's sums reach the first through LDS at the
// end: 16 partials per output instead of 32 (8.4 MB written and read at 32 rows instead of 16.7).  With the one-launch
// reduction below: 98-99 us at 32 rows (5.4-5.5 TB/s; 109 before), 94.6 at 16 (102), 93 at 3-8 rows (unchanged).
template <int PK_DEPTH, int OCC, int KP>
__global__ __launch_bounds__(KP * PK_WAVES * 64, OCC) void pca_stream_kernel(const float* __restrict__ v,
                                                                           const float* __restrict__ wp,
                                                                           float* __restrict__ part, int N, int D,
                                                                           int d, int kper) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  extern __shared__ __attribute__((aligned(16))) char pk_smem[];
  const int lane = threadIdx.x & 63;
  const int kp = KP == 1 ? 0 : __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 9));
  const int wave = (threadIdx.x >> 6) & 7;
  char* const smem = pk_smem + kp * (2 * PK_TILES * 1024);
  const int nb = blockIdx.x * PK_WAVES + wave;
  const long k0 = (long)(blockIdx.y * KP + kp) * kper;
  const int chunks = kper / PK_CHUNK;
  const f4* wt = reinterpret_cast<const f4*>(wp) + ((size_t)nb * (D / 8) + k0 / 8) * 64 + lane;   // tile t: wt[64 t]
  const int m = min(lane & 31, N - 1);
  const float* vsrc = v + (size_t)m * D + k0 + 4 * (lane >> 5) + wave * 8;   // tile (wave + 8 i) of a chunk: + 64 i
  char* const lds_w = smem + wave * 1024 + lane * 16;                        // ... its place: + 8192 i
  const char* const lds_r = smem + lane * 16;                                // tile t of a chunk: + 1024 t

  f4 ring[PK_DEPTH];
#pragma unroll
  for (int u = 0; u < PK_DEPTH; ++u) ring[u] = __builtin_nontemporal_load(wt + (size_t)u * 64);
  f4 vs[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) vs[i] = *reinterpret_cast<const f4*>(vsrc + 64 * i);
#pragma unroll
  for (int i = 0; i < 4; ++i) *reinterpret_cast<f4*>(lds_w + 8192 * i) = vs[i];
  f32x16_t acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  __syncthreads();

  auto chunk = [&](int c, auto last_tag) __attribute__((always_inline)) {
    constexpr bool LAST = decltype(last_tag)::value;
    const char* cur = lds_r + (c & 1) * (PK_TILES * 1024);
    if (!LAST) {
#pragma unroll
      for (int i = 0; i < 4; ++i) vs[i] = *reinterpret_cast<const f4*>(vsrc + (long)(c + 1) * PK_CHUNK + 64 * i);
    }
    const f4* wn = wt + ((size_t)c * PK_TILES + PK_DEPTH) * 64;
    f4 a = *reinterpret_cast<const f4*>(cur);
#pragma unroll
    for (int t = 0; t < PK_TILES; ++t) {
      const f4 w = ring[t % PK_DEPTH];
      if (!LAST || t + PK_DEPTH < PK_TILES) ring[t % PK_DEPTH] = __builtin_nontemporal_load(wn + (size_t)t * 64);
      f4 an = a;
      if (t + 1 < PK_TILES) an = *reinterpret_cast<const f4*>(cur + (t + 1) * 1024);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], w[j], acc, 0, 0, 0);
      a = an;
      __builtin_amdgcn_sched_barrier(0);       // keep the written order: the scheduler sinks the refill otherwise
    }
    if (!LAST) {
      char* nxt = lds_w + ((c + 1) & 1) * (PK_TILES * 1024);
#pragma unroll
      for (int i = 0; i < 4; ++i) *reinterpret_cast<f4*>(nxt + 8192 * i) = vs[i];
      __syncthreads();
    }
  };
  for (int c = 0; c + 1 < chunks; ++c) chunk(c, std::false_type{});
  chunk(chunks - 1, std::true_type{});

  if constexpr (KP == 2) {
    // the second K part hands its sums over through its own (now idle) chunk buffers: [wave][register][lane]
    __syncthreads();
    float* const xch = reinterpret_cast<float*>(pk_smem + 2 * PK_TILES * 1024) + (wave * 16) * 64 + lane;
    if (kp == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r) xch[r * 64] = acc[r];
    }
    __syncthreads();
    if (kp == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += xch[r * 64];
  }
  float* out = part + (size_t)blockIdx.y * N * d + nb * 32 + (lane & 31);
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = acc_row(r, lane);
    if (row < N) out[(size_t)row * d] = acc[r];
  }
}

// packed[((nb * D/8 + kb) * 64 + lane) * 4 + j] = w[nb * 32 + (lane & 31)][8 kb + 4 (lane >> 5) + j]
__global__ __launch_bounds__(256) void pca_pack_kernel(const float* __restrict__ w, float* __restrict__ packed, int D,
                                                       size_t units) {
  typedef __attribute__((ext_vector_type(4))) float f4;
  for (size_t u = (size_t)blockIdx.x * 256 + threadIdx.x; u < units; u += (size_t)gridDim.x * 256) {
    const int lane = (int)(u & 63);
    const size_t tile = u >> 6;
    const size_t kb = tile % (size_t)(D / 8), nb = tile / (size_t)(D / 8);
    reinterpret_cast<f4*>(packed)[u] =
        *reinterpret_cast<const f4*>(w + (nb * 32 + (lane & 31)) * (size_t)D + kb * 8 + 4 * (lane >> 5));
  }
}

OIBL_HOOK(int, g_pca_stream, 1);   // test hook: 0 = oibl_pca_forward_packed refuses (callers fall back to the tile)
static bool pca_stream_ok(int N, int D, int d) {
  return N >= 1 && N <= PK_MAXN && d % (32 * PK_WAVES) == 0 && D % (PK_SPLITS * PK_CHUNK) == 0;
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
  int s = pca_splits(N, D, d, precision);
  if (pca_small_ok(N, D, d, precision) && s < PS_SPLITS) s = PS_SPLITS;
  if (precision == OIBL_F32 && pca_stream_ok(N, D, d) && s < PK_SPLITS) s = PK_SPLITS;
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
  if (pca_small_ok(N, D, d, precision) && g_pca_small && !g_regstage) {
    p.splits = PS_SPLITS;
    const dim3 sg((unsigned)(d / (4 * PS_ROWS)), PS_SPLITS);
    const int kper = D / PS_SPLITS;
    const float* vf = (const float*)v_t;
    const float* wf = (const float*)w;
    if (N == 1) hipLaunchKernelGGL(pca_small_kernel<1>, sg, dim3(256), 0, st, vf, wf, p.part, N, D, d, kper);
    else hipLaunchKernelGGL(pca_small_kernel<2>, sg, dim3(256), 0, st, vf, wf, p.part, N, D, d, kper);
  } else if (precision == OIBL_BF16) {
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

int oibl_pca_pack_weight(const float* w, int D, int d, float* packed, void* stream) {
  OIBL_REQUIRE(w && packed, "pca_pack_weight: null pointer");
  OIBL_REQUIRE(D > 0 && d > 0 && D % 8 == 0 && d % 32 == 0, "pca_pack_weight: unsupported shape D=%d d=%d", D, d);
  OIBL_REQUIRE((uintptr_t)w % 16 == 0 && (uintptr_t)packed % 16 == 0, "pca_pack_weight: pointers must be 16-byte aligned");
  const size_t units = (size_t)d * D / 4;
  const unsigned grid = (unsigned)std::min<size_t>((units + 255) / 256, 65536);
  hipLaunchKernelGGL(pca_pack_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, w, packed, D, units);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_pca_packed_supported(int N, int D, int d) { return pca_stream_ok(N, D, d) && g_pca_stream ? 1 : 0; }

int oibl_pca_forward_packed(const float* v, int N, int D, const float* w_packed, const float* b, int d, int l2norm,
                            float* out, void* ws, size_t ws_bytes, void* stream) {
  OIBL_REQUIRE(v && w_packed && b && out && ws, "pca_packed: null pointer");
  OIBL_REQUIRE(pca_stream_ok(N, D, d) && g_pca_stream,
               "pca_packed: unsupported shape N=%d D=%d d=%d (1 <= N <= %d, d %% %d == 0, D %% %d == 0; "
               "oibl_pca_packed_supported)", N, D, d, PK_MAXN, 32 * PK_WAVES, PK_SPLITS * PK_CHUNK);
  OIBL_REQUIRE((uintptr_t)ws % 256 == 0 && (uintptr_t)w_packed % 16 == 0 && (uintptr_t)v % 16 == 0,
               "pca_packed: workspace must be 256-byte, w_packed and v 16-byte aligned");
  const size_t need = oibl_pca_workspace_bytes(N, D, d, OIBL_F32);
  if (ws_bytes < need) {
    set_error("pca_packed: workspace %zu < required %zu bytes", ws_bytes, need);
    return OIBL_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* wsb = (char*)ws;
  float* part = (float*)(wsb + align_up((size_t)N * D * sizeof(float), 256));
  const dim3 grid((unsigned)(d / (32 * PK_WAVES)), PK_SPLITS);
  constexpr int lds1 = 2 * PK_TILES * 1024;
  int nparts = PK_SPLITS;
  // 8 loads in flight x 4 waves per SIMD (104 VGPRs) against 16 x 2 (173): 109.8 against 112.8 us at 32 rows, 92.7
  // against 95.5 at 3 (tests/gpu_pca_bench.py; all three launches)
  if (g_pca_stream == 2) {
    auto kern = pca_stream_kernel<16, 2, 1>;
    OIBL_SET_MAX_LDS(kern, lds1);
    hipLaunchKernelGGL(kern, grid, dim3(PK_WAVES * 64), lds1, st, v, w_packed, part, N, D, d, D / PK_SPLITS);
  } else if (g_pca_stream == 3) {      // rounds 5's launch: 32 partials
    auto kern = pca_stream_kernel<8, 4, 1>;
    OIBL_SET_MAX_LDS(kern, lds1);
    hipLaunchKernelGGL(kern, grid, dim3(PK_WAVES * 64), lds1, st, v, w_packed, part, N, D, d, D / PK_SPLITS);
  } else {
    auto kern = pca_stream_kernel<8, 1, 2>;
    OIBL_SET_MAX_LDS(kern, 2 * lds1);
    nparts = PK_SPLITS / 2;
    hipLaunchKernelGGL(kern, dim3(grid.x, nparts), dim3(2 * PK_WAVES * 64), 2 * lds1, st, v, w_packed, part, N, D, d,
                       D / PK_SPLITS);
  }
  OIBL_LAUNCH_CHECK();
  if (d <= 1024 * PRN_PER && g_pca_stream == 1) {
    hipLaunchKernelGGL(pca_reduce_norm_kernel<PK_SPLITS / 2>, dim3((unsigned)N), dim3(1024), 0, st, part, b, out, N, d, l2norm);
    OIBL_LAUNCH_CHECK();
    return OIBL_OK;
  }
  float* ss_part = (float*)((char*)part + align_up((size_t)PK_SPLITS * N * d * sizeof(float), 256));
  const dim3 rgrid((unsigned)((d + 255) / 256), (unsigned)N);
  hipLaunchKernelGGL(pca_reduce_kernel, rgrid, dim3(256), 0, st, part, b, out, ss_part, N, d, nparts);
  OIBL_LAUNCH_CHECK();
  if (l2norm) {
    hipLaunchKernelGGL(pca_scale_kernel, rgrid, dim3(256), 0, st, out, ss_part, d);
    OIBL_LAUNCH_CHECK();
  }
  return OIBL_OK;
}

#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_pca_small(int on) {
  g_pca_small = on ? 1 : 0;
  return OIBL_OK;
}
int oibl_debug_set_pca_stream(int on) {
  g_pca_stream = on;
  return OIBL_OK;
}
#endif

}  // extern "C"
