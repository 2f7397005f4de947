// Library-level entry points: error string, casts, row normalisation and the bare NT GEMM used
// by the tests to validate the MFMA core.  See include/openibl_amd.h.
#include <stdarg.h>
#include <string.h>

#include "gemm_core.h"

namespace oibl {

static thread_local char g_err[512] = "";
#ifdef OIBL_DEBUG_HOOKS
int g_regstage = 0;
#endif

void set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

__device__ uint4 g_zero_line[8];  // 128 B, zero-initialised device memory

const void* zero_line_device_ptr() {
  void* p = nullptr;
  if (hipGetSymbolAddress(&p, HIP_SYMBOL(g_zero_line)) != hipSuccess) return nullptr;
  return p;
}

// ---- casts ---------------------------------------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                     size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x * 8;
  for (size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8; i < n; i += stride) {
    if (i + 8 <= n) {
      const float4 a = *reinterpret_cast<const float4*>(src + i);
      const float4 b = *reinterpret_cast<const float4*>(src + i + 4);
      uint4 o;
      o.x = f32_to_bf16_bits(a.x) | ((uint32_t)f32_to_bf16_bits(a.y) << 16);
      o.y = f32_to_bf16_bits(a.z) | ((uint32_t)f32_to_bf16_bits(a.w) << 16);
      o.z = f32_to_bf16_bits(b.x) | ((uint32_t)f32_to_bf16_bits(b.y) << 16);
      o.w = f32_to_bf16_bits(b.z) | ((uint32_t)f32_to_bf16_bits(b.w) << 16);
      *reinterpret_cast<uint4*>(dst + i) = o;
    } else {
      for (size_t k = i; k < n; ++k) dst[k] = f32_to_bf16_bits(src[k]);
    }
  }
}

__global__ void cast_bf16_f32_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst,
                                     size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = bf16_bits_to_f32(src[i]);
}

__global__ void cast_f32_f16_kernel(const float* __restrict__ src, uint16_t* __restrict__ dst,
                                    size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = f32_to_f16_bits(src[i]);
}

__global__ void cast_f16_f32_kernel(const uint16_t* __restrict__ src, float* __restrict__ dst,
                                    size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
    dst[i] = f16_bits_to_f32(src[i]);
}

// ---- bilinear resize (F.interpolate, align_corners=False) ------------------------------------
// One thread per output pixel, consecutive threads along W2 (stores coalesce, the two source rows
// are read as near-contiguous runs).  Arithmetic in the order ATen's upsample_bilinear2d uses:
// src = scale * (dst + 0.5) - 0.5 clamped at 0; i1 = i0 + (i0 < in - 1); l1 = src - i0, l0 = 1 - l1;
// out = l0y * (l0x * v00 + l1x * v01) + l1y * (l0x * v10 + l1x * v11).
__global__ void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ out, int H,
                                       int W, int H2, int W2, float sy, float sx) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x;
  const int oy = blockIdx.y;
  if (ox >= W2) return;
  const size_t plane = blockIdx.z;
  float fy = sy * ((float)oy + 0.5f) - 0.5f;
  float fx = sx * ((float)ox + 0.5f) - 0.5f;
  fy = fy < 0.f ? 0.f : fy;
  fx = fx < 0.f ? 0.f : fx;
  int y0 = (int)fy, x0 = (int)fx;
  y0 = y0 > H - 1 ? H - 1 : y0;
  x0 = x0 > W - 1 ? W - 1 : x0;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
  const float ly0 = 1.f - ly1, lx0 = 1.f - lx1;
  const float* p = x + plane * H * W;
  const float v00 = p[(size_t)y0 * W + x0], v01 = p[(size_t)y0 * W + x1];
  const float v10 = p[(size_t)y1 * W + x0], v11 = p[(size_t)y1 * W + x1];
  out[(plane * H2 + oy) * W2 + ox] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
}

// ---- row L2 normalise -------------------------------------------------------------------
// one wave per row; x / max(||x||, 1e-12) as F.normalize does.
__global__ void l2_normalize_rows_kernel(const float* __restrict__ x, float* __restrict__ out,
                                         int N, int D) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= N) return;
  const float* xr = x + (size_t)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) s += xr[i] * xr[i];
  s = wave_sum(s);
  const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
  float* orow = out + (size_t)row * D;
  for (int i = lane; i < D; i += 64) orow[i] = xr[i] * inv;
}

// out[row] = normalize(xs[0][row] + xs[1][row] + ... ) — descriptors of S scales fused into one
__global__ void sum_l2_normalize_kernel(const float* __restrict__ xs, int S, int N, int D,
                                        float* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (row >= N) return;
  const size_t plane = (size_t)N * D;
  const float* xr = xs + (size_t)row * D;
  float* orow = out + (size_t)row * D;
  float s = 0.f;
  for (int i = lane; i < D; i += 64) {
    float a = xr[i];
    for (int k = 1; k < S; ++k) a += xr[i + k * plane];
    orow[i] = a;
    s += a * a;
  }
  s = wave_sum(s);
  const float inv = 1.0f / fmaxf(sqrtf(s), 1e-12f);
  for (int i = lane; i < D; i += 64) orow[i] *= inv;   // each lane re-reads its own writes
}

// ---- bare NT GEMM (diagnostic) ------------------------------------------------------------
template <typename Cfg, bool GLDS>
__global__ __launch_bounds__(Cfg::NTHREADS) void gemm_nt_kernel(const void* A, int M,
                                                                const void* B, int N, int K,
                                                                float* C, size_t ldc) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  using T = typename Cfg::T;
  const WaveCoord c = wave_coord<Cfg>();
  const int tiles_n = N / Cfg::BN;
  const unsigned tile = xcd_remap(blockIdx.x, gridDim.x);
  const int tn = tile % tiles_n, tm = tile / tiles_n;
  const long m0 = (long)tm * Cfg::BM, n0 = (long)tn * Cfg::BN;

  RowLoader<Cfg, Cfg::A_LOADS> la;
  RowLoader<Cfg, Cfg::B_LOADS> lb;
  la.init(c, A, m0, M, (long)K * sizeof(T));
  lb.init(c, B, n0, N, (long)K * sizeof(T));

  f32x16_t acc[Cfg::TM][Cfg::TN];
#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  gemm_nt_mainloop<Cfg, GLDS>(acc, smem, c, la, lb, K / Cfg::BK);

#pragma unroll
  for (int i = 0; i < Cfg::TM; ++i)
#pragma unroll
    for (int j = 0; j < Cfg::TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + (c.wm * Cfg::TM + i) * 32 + acc_row(r, c.lane);
        const long n = n0 + (c.wn * Cfg::TN + j) * 32 + (c.lane & 31);
        if (m < M) C[m * ldc + n] = acc[i][j][r];
      }
}

template <typename T, bool GLDS>
static int launch_gemm_nt(const void* A, int M, const void* B, int N, int K, float* C,
                          size_t ldc, hipStream_t st) {
  using Cfg = GemmCfg<T, 2, 2, 2, 2>;
  using Cfg64 = GemmCfg<T, 2, 2, 2, 1>;
  if (N % 128 == 0) {
    const int grid = ((M + Cfg::BM - 1) / Cfg::BM) * (N / Cfg::BN);
    hipLaunchKernelGGL((gemm_nt_kernel<Cfg, GLDS>), dim3(grid), dim3(Cfg::NTHREADS),
                       Cfg::MAIN_LDS_BYTES, st, A, M, B, N, K, C, ldc);
  } else {
    const int grid = ((M + Cfg64::BM - 1) / Cfg64::BM) * (N / Cfg64::BN);
    hipLaunchKernelGGL((gemm_nt_kernel<Cfg64, GLDS>), dim3(grid), dim3(Cfg64::NTHREADS),
                       Cfg64::MAIN_LDS_BYTES, st, A, M, B, N, K, C, ldc);
  }
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

#ifdef OIBL_DEBUG_HOOKS
// Diagnostic: the matrix pipe with nothing else to do.  Every wave keeps four independent
// v_mfma_f32_32x32x16_bf16 chains going on register operands (no LDS, no memory in the loop): what
// the chip sustains here, at the clock its power management settles on, is the ceiling every
// MFMA-bound kernel of this library is measured against in DESIGN.md §6 (tests/gpu_mfma_peak.py).
__global__ __launch_bounds__(512) void mfma_peak_kernel(long iters, float* out) {
  // four operand pairs of pseudo-random bf16 values in (-2, 2) per lane, rotated MFMA by MFMA:
  // constant operands would toggle nothing in the multipliers and flatter the power figure
  bf16x8_t a[4], b[4];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll
  for (int v = 0; v < 4; ++v)
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u;
      a[v][e] = (short)(((h >> 16) & 0x807f) | (0x3f00 + ((h >> 8) & 0x0080)));   // sign, 7 mantissa bits, exponent 126 / 127
      h = h * 1664525u + 1013904223u;
      b[v][e] = (short)(((h >> 16) & 0x807f) | (0x3f00 + ((h >> 8) & 0x0080)));
    }
  f32x16_t acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (long i = 0; i < iters; ++i) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
      for (int c = 0; c < 4; ++c)
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + c) & 3], b[(u + 2 * c + 1) & 3], acc[c], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 12345.678f) out[0] = s;   // keeps the chains alive
}
#endif

// n 32-bit words src -> dst by a KERNEL (one workgroup): for the f16mx range flag's trip to pinned host memory.
// A hipMemcpyAsync of the same 4 bytes queues on a DMA engine behind whatever that engine is moving — next
// to a 118 MB input copy of the following batch it held the lane up by milliseconds (-7 % through
// extract_features from fp32 host batches); a kernel's store to host-coherent memory does not.
__global__ void copy_words_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) dst[i] = src[i];
}

}  // namespace oibl

using namespace oibl;

extern "C" {

// diagnostic (not in the public header): `blocks` workgroups x 8 waves x `iters` x 16 MFMAs of
// 32x32x16 bf16 (32768 flop each) on register operands
#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_mfma_peak(long iters, int blocks, void* scratch, void* stream) {
  OIBL_REQUIRE(iters > 0 && blocks > 0 && scratch, "mfma_peak: bad arguments");
  hipLaunchKernelGGL(mfma_peak_kernel, dim3(blocks), dim3(512), 0, (hipStream_t)stream, iters, (float*)scratch);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}
#endif

// test hook (not in the public header): 1 = register-staged main loop, 0 = global_load_lds
#ifdef OIBL_DEBUG_HOOKS
int oibl_debug_set_regstage(int on) {
  g_regstage = on ? 1 : 0;
  return OIBL_OK;
}
#endif

int oibl_copy_words(const void* src, void* dst, int n, void* stream) {
  OIBL_REQUIRE(src && dst && n > 0, "copy_words: bad arguments");
  OIBL_REQUIRE((uintptr_t)src % 4 == 0 && (uintptr_t)dst % 4 == 0, "copy_words: unaligned pointer");
  hipLaunchKernelGGL(copy_words_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, (const uint32_t*)src,
                     (uint32_t*)dst, n);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_host) {
  OIBL_REQUIRE(ev_start && ev_stop && ms_host, "event_elapsed: null pointer");
  OIBL_HIP_CHECK(hipEventElapsedTime(ms_host, (hipEvent_t)ev_start, (hipEvent_t)ev_stop));
  return OIBL_OK;
}

int oibl_abi_version(void) { return 3; }
const char* oibl_last_error(void) { return g_err; }
const char* oibl_target_arch(void) { return "gfx950"; }
size_t oibl_elem_size(int precision) {
  return precision == OIBL_BF16 ? 2
         : (precision == OIBL_F32 || precision == OIBL_BF16X3 || precision == OIBL_F16MX) ? 4 : 0;
}

int oibl_cast_f32_to_bf16(const float* src, uint16_t* dst, size_t n, void* stream) {
  OIBL_REQUIRE(src && dst, "cast_f32_to_bf16: null pointer");
  if (n == 0) return OIBL_OK;
  OIBL_REQUIRE(((uintptr_t)src % 16 == 0) && ((uintptr_t)dst % 16 == 0),
               "cast_f32_to_bf16: pointers must be 16-byte aligned");
  size_t blocks = (n / 8 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks == 0) blocks = 1;
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, src, dst, n);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_cast_bf16_to_f32(const uint16_t* src, float* dst, size_t n, void* stream) {
  OIBL_REQUIRE(src && dst, "cast_bf16_to_f32: null pointer");
  if (n == 0) return OIBL_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cast_bf16_f32_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, src, dst, n);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_cast_f32_to_f16(const float* src, uint16_t* dst, size_t n, void* stream) {
  OIBL_REQUIRE(src && dst, "cast_f32_to_f16: null pointer");
  if (n == 0) return OIBL_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cast_f32_f16_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, src, dst, n);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_cast_f16_to_f32(const uint16_t* src, float* dst, size_t n, void* stream) {
  OIBL_REQUIRE(src && dst, "cast_f16_to_f32: null pointer");
  if (n == 0) return OIBL_OK;
  size_t blocks = (n + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(cast_f16_f32_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, src, dst, n);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_resize_bilinear_nchw(const float* x, int N, int C, int H, int W, float* out, int H2, int W2,
                              void* stream) {
  OIBL_REQUIRE(x && out, "resize_bilinear: null pointer");
  OIBL_REQUIRE(N >= 0 && C > 0 && H > 0 && W > 0 && H2 > 0 && W2 > 0,
               "resize_bilinear: bad shape N=%d C=%d %dx%d -> %dx%d", N, C, H, W, H2, W2);
  OIBL_REQUIRE((long)N * C <= 65535 && H2 <= 65535, "resize_bilinear: N*C=%ld or H2=%d above 65535",
               (long)N * C, H2);
  if (N == 0) return OIBL_OK;
  // ATen: scale = in / out in fp32 (area_pixel_compute_scale, align_corners = false)
  const float sy = (float)H / (float)H2, sx = (float)W / (float)W2;
  hipLaunchKernelGGL(resize_bilinear_kernel, dim3((W2 + 255) / 256, H2, N * C), dim3(256), 0,
                     (hipStream_t)stream, x, out, H, W, H2, W2, sy, sx);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_l2_normalize_rows(const float* x, int N, int D, float* out, void* stream) {
  OIBL_REQUIRE(x && out, "l2_normalize_rows: null pointer");
  OIBL_REQUIRE(N >= 0 && D > 0, "l2_normalize_rows: bad shape N=%d D=%d", N, D);
  if (N == 0) return OIBL_OK;
  hipLaunchKernelGGL(l2_normalize_rows_kernel, dim3((N + 3) / 4), dim3(256), 0,
                     (hipStream_t)stream, x, out, N, D);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

int oibl_sum_l2_normalize(const float* xs, int S, int N, int D, float* out, void* stream) {
  OIBL_REQUIRE(xs && out, "sum_l2_normalize: null pointer");
  OIBL_REQUIRE(S >= 1 && N >= 0 && D > 0, "sum_l2_normalize: bad shape S=%d N=%d D=%d", S, N, D);
  if (N == 0) return OIBL_OK;
  hipLaunchKernelGGL(sum_l2_normalize_kernel, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                     xs, S, N, D, out);
  OIBL_LAUNCH_CHECK();
  return OIBL_OK;
}

// precision bit 8 (0x100) selects the register-staged variant of the main loop (test hook).
int oibl_gemm_nt(const void* A, int M, const void* B, int N, int K, int precision, float* C,
                 size_t ldc, void* stream) {
  const bool regstage = (precision & 0x100) != 0;
  precision &= 0xff;
  OIBL_REQUIRE(A && B && C, "gemm_nt: null pointer");
  OIBL_REQUIRE(precision == OIBL_BF16 || precision == OIBL_F32, "gemm_nt: bad precision %d",
               precision);
  const int bk = precision == OIBL_BF16 ? 64 : 32;
  OIBL_REQUIRE(M > 0 && N > 0 && K > 0 && K % bk == 0 && N % 64 == 0 && ldc >= (size_t)N,
               "gemm_nt: unsupported shape M=%d N=%d K=%d", M, N, K);
  OIBL_REQUIRE((uintptr_t)A % 16 == 0 && (uintptr_t)B % 16 == 0, "gemm_nt: unaligned operand");
  hipStream_t st = (hipStream_t)stream;
  if (precision == OIBL_BF16)
    return regstage ? launch_gemm_nt<bf16_t, false>(A, M, B, N, K, C, ldc, st)
                    : launch_gemm_nt<bf16_t, true>(A, M, B, N, K, C, ldc, st);
  return regstage ? launch_gemm_nt<float, false>(A, M, B, N, K, C, ldc, st)
                  : launch_gemm_nt<float, true>(A, M, B, N, K, C, ldc, st);
}

}  // extern "C"
