// Query x gallery squared-L2 distance matrix and per-row top-k on gfx950.
// Reference behaviour: pairwise_distance (ibl/evaluators.py:105-130) and the argsort consumed by
// evaluate_all (ibl/evaluators.py:142-159).
#include "gemm_core.h"

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

__global__ __launch_bounds__(256) void row_topk_kernel(const float* __restrict__ vals,
                                                       const int32_t* __restrict__ idx_in, int n,
                                                       size_t ld, int k, int index_base,
                                                       float* __restrict__ out_val,
                                                       int32_t* __restrict__ out_idx) {
  __shared__ unsigned long long cand[TOPK_CAP];
  __shared__ int cnt;
  __shared__ unsigned long long thr;
  const int row = blockIdx.x, tid = threadIdx.x;
  const float* vr = vals + (size_t)row * ld;
  const int32_t* ir = idx_in ? idx_in + (size_t)row * ld : nullptr;
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
  for (int i = have + tid; i < TOPK_CAP; i += 256) cand[i] = TOPK_INF;
  __syncthreads();
  bitonic_sort_lds(cand, TOPK_CAP, tid, 256);
  for (int i = tid; i < k; i += 256) {
    const unsigned long long key = cand[i];
    const bool valid = i < have;
    out_val[(size_t)row * k + i] = valid ? from_ordered_bits((uint32_t)(key >> 32)) : INFINITY;
    out_idx[(size_t)row * k + i] = valid ? (int32_t)(uint32_t)(key & 0xffffffffu) : -1;
  }
}

}  // namespace oibl

using namespace oibl;

extern "C" {

static size_t pw_off_yn(int m) { return align_up((size_t)m * sizeof(float), 256); }
static size_t pw_off_xt(int m, int n) { return pw_off_yn(m) + align_up((size_t)n * sizeof(float), 256); }

size_t oibl_pairwise_workspace_bytes(int m, int n, int d, int precision) {
  if (m <= 0 || n <= 0 || d <= 0) return 0;
  size_t b = pw_off_xt(m, n);
  if (precision == OIBL_BF16)
    b += align_up((size_t)m * d * 2, 256) + align_up((size_t)n * d * 2, 256);
  return b;
}

int oibl_pairwise_sqdist(const float* x, int m, const float* y, int n, int d, int precision,
                         float* dist, size_t ldd, void* ws, size_t ws_bytes, void* stream) {
  OIBL_REQUIRE(x && y && dist && ws, "pairwise: null pointer");
  OIBL_REQUIRE(precision == OIBL_BF16 || precision == OIBL_F32, "pairwise: bad precision %d",
               precision);
  OIBL_REQUIRE(m > 0 && n > 0 && d > 0 && d % 64 == 0 && ldd >= (size_t)n,
               "pairwise: unsupported shape m=%d n=%d d=%d ldd=%zu", m, n, d, ldd);
  OIBL_REQUIRE((uintptr_t)ws % 256 == 0 && (uintptr_t)x % 16 == 0 && (uintptr_t)y % 16 == 0,
               "pairwise: workspace must be 256-byte, x and y 16-byte aligned");
  const size_t need = oibl_pairwise_workspace_bytes(m, n, d, precision);
  if (ws_bytes < need) {
    set_error("pairwise: workspace %zu < required %zu bytes", ws_bytes, need);
    return OIBL_E_WORKSPACE;
  }
  hipStream_t st = (hipStream_t)stream;
  char* wsb = (char*)ws;
  PairParams p;
  p.xn = (float*)wsb;
  p.yn = (float*)(wsb + pw_off_yn(m));
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3((m + 3) / 4), dim3(256), 0, st, x, (float*)p.xn, m, d);
  OIBL_LAUNCH_CHECK();
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3((n + 3) / 4), dim3(256), 0, st, y, (float*)p.yn, n, d);
  OIBL_LAUNCH_CHECK();
  p.x = x;
  p.y = y;
  if (precision == OIBL_BF16) {
    uint16_t* xt = (uint16_t*)(wsb + pw_off_xt(m, n));
    uint16_t* yt = (uint16_t*)((char*)xt + align_up((size_t)m * d * 2, 256));
    int rc = oibl_cast_f32_to_bf16(x, xt, (size_t)m * d, stream);
    if (rc) return rc;
    rc = oibl_cast_f32_to_bf16(y, yt, (size_t)n * d, stream);
    if (rc) return rc;
    p.x = xt;
    p.y = yt;
  }
  p.dist = dist;
  p.ldd = ldd;
  p.m = m;
  p.n = n;
  p.d = d;
  p.tiles_n = (n + 127) / 128;
  const long grid = (long)((m + 127) / 128) * p.tiles_n;
  OIBL_REQUIRE(grid <= 0x7fffffffL, "pairwise: grid too large");
  if (precision == OIBL_BF16) {
    using Cfg = GemmCfg<bf16_t, 2, 2, 2, 2>;
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

int oibl_row_topk(const float* vals, const int32_t* idx_in, int m, int n, size_t ld, int k,
                  int index_base, float* out_val, int32_t* out_idx, void* stream) {
  OIBL_REQUIRE(vals && out_val && out_idx, "row_topk: null pointer");
  OIBL_REQUIRE(m > 0 && n > 0 && ld >= (size_t)n, "row_topk: bad shape m=%d n=%d ld=%zu", m, n, ld);
  OIBL_REQUIRE(k >= 1 && k <= 1024, "row_topk: k=%d outside [1, 1024]", k);
  hipLaunchKernelGGL(row_topk_kernel, dim3(m), dim3(256), 0, (hipStream_t)stream, vals, idx_in, n,
                     ld, k, index_base, out_val, out_idx);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

}  // extern "C"
