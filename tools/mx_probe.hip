// Hardware probe (diagnostic, not product code): semantics of the gfx950 MX matrix instruction
// v_mfma_scale_f32_32x32x64_f8f6f4 with e2m3 operands, of the fp6 pack/convert instructions, and what
// the matrix pipe sustains on the mixed instruction stream of the f16mx mode
// (2 x v_mfma_f32_32x32x16_f16 + 1 x MX fp6 K=64 per 32 real K).
//   hipcc --offload-arch=gfx950 -O3 tools/mx_probe.hip -o build/mx_probe && build/mx_probe [seconds]
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef unsigned v6u __attribute__((ext_vector_type(6)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef _Float16 v32h __attribute__((ext_vector_type(32)));
typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef short v8s __attribute__((ext_vector_type(8)));

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

static float e2m3_value(unsigned c) {
  const int s = (c >> 5) & 1, e = (c >> 3) & 3, m = c & 7;
  const float v = e == 0 ? m / 8.0f : (1.0f + m / 8.0f) * (float)(1 << (e - 1));
  return s ? -v : v;
}

// ---- part 1: one MX MFMA per wave --------------------------------------------------------------
template <int FMT>
__global__ void mx_one(const v8i* a, const v8i* b, const int* sa, const int* sb, v16f* out) {
  const int l = threadIdx.x;
  v16f c = {};
  c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[l], b[l], c, FMT, FMT, 0, sa[l], 0, sb[l]);
  out[l] = c;
}

// ---- part 2: conversions ------------------------------------------------------------------------
__global__ void cvt_f32(const v16f* x, v6u* out, float scale) {
  const int l = threadIdx.x;
  out[l] = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(x[2 * l], x[2 * l + 1], scale);
}
__global__ void cvt_f16(const v32h* x, v6u* out, float scale) {
  const int l = threadIdx.x;
  out[l] = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(x[l], scale);
}

// ---- part 3: throughput -------------------------------------------------------------------------
// MODE 0: bf16 x16 per iteration; 1: f16 x16; 2: MX fp6 x16; 3: MX fp8 x16;
// 4: f16mx units: per accumulator (f16, f16, fp6) -> 4 units = 12 MFMAs per iteration;
// 5: bf16x3 units: per accumulator 6 bf16 MFMAs -> 4 units = 24 MFMAs per iteration
template <int MODE>
__global__ __launch_bounds__(512) void peak(long iters, float* out) {
  v8s a[4], b[4];
  v8i ma[2], mb[2];
  unsigned h = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int v = 0; v < 4; ++v)
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u;
      // bf16: sign, 7 mantissa bits, exponent 126/127; f16: sign, 10 mantissa bits, exponent 14/15
      a[v][e] = MODE == 0 || MODE == 5 ? (short)(((h >> 16) & 0x807f) | (0x3f00 + ((h >> 8) & 0x0080)))
                                       : (short)(((h >> 16) & 0x83ff) | (0x3800 + ((h >> 8) & 0x0400)));
      h = h * 1664525u + 1013904223u;
      b[v][e] = MODE == 0 || MODE == 5 ? (short)(((h >> 16) & 0x807f) | (0x3f00 + ((h >> 8) & 0x0080)))
                                       : (short)(((h >> 16) & 0x83ff) | (0x3800 + ((h >> 8) & 0x0400)));
    }
  for (int v = 0; v < 2; ++v)
    for (int e = 0; e < 8; ++e) {
      h = h * 1664525u + 1013904223u;
      ma[v][e] = (int)(MODE == 3 ? (h & 0xb7b7b7b7u) : h);   // fp8: keep exponents off the NaN code
      h = h * 1664525u + 1013904223u;
      mb[v][e] = (int)(MODE == 3 ? (h & 0xb7b7b7b7u) : h);
    }
  const int sc = 0x7f7f7f7f - (int)(threadIdx.x & 3);
  v16f acc[4];
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  for (long i = 0; i < iters; ++i) {
    if constexpr (MODE == 0) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + c) & 3], b[(u + 2 * c + 1) & 3], acc[c], 0, 0, 0);
    } else if constexpr (MODE == 1) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a[(u + c) & 3]),
                                                          __builtin_bit_cast(v8h, b[(u + 2 * c + 1) & 3]), acc[c], 0, 0, 0);
    } else if constexpr (MODE == 2) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ma[(u + c) & 1], mb[(u + (c >> 1)) & 1], acc[c], 2, 2, 0, sc, 0, sc);
    } else if constexpr (MODE == 3) {
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int c = 0; c < 4; ++c)
          acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ma[(u + c) & 1], mb[(u + (c >> 1)) & 1], acc[c], 0, 0, 0, sc, 0, sc);
    } else if constexpr (MODE == 4) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a[c]), __builtin_bit_cast(v8h, b[(c + 1) & 3]), acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(v8h, a[(c + 2) & 3]), __builtin_bit_cast(v8h, b[(c + 3) & 3]), acc[c], 0, 0, 0);
        acc[c] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(ma[c & 1], mb[(c >> 1) & 1], acc[c], 2, 2, 0, sc, 0, sc);
      }
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int u = 0; u < 6; ++u)
          acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u + c) & 3], b[(u + 2 * c + 1) & 3], acc[c], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int c = 0; c < 4; ++c)
    for (int r = 0; r < 16; ++r) s += acc[c][r];
  if (s == 12345.678f) out[0] = s;
}

template <int MODE>
static void run_peak(const char* name, double seconds, int mfma_per_iter, double flop_per_iter, int units_per_iter, float* scratch) {
  const long iters = 20000;
  const int blocks = 512;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int w = 0; w < 3; ++w) hipLaunchKernelGGL(peak<MODE>, dim3(blocks), dim3(512), 0, 0, iters, scratch);
  CK(hipDeviceSynchronize());
  double total_ms = 0, last = 0;
  int reps = 0;
  while (total_ms < seconds * 1e3) {
    CK(hipEventRecord(e0));
    for (int r = 0; r < 4; ++r) hipLaunchKernelGGL(peak<MODE>, dim3(blocks), dim3(512), 0, 0, iters, scratch);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    total_ms += ms;
    last = ms / 4;
    ++reps;
  }
  const double waves = blocks * 8.0;
  const double t = last * 1e-3;
  printf("peak %-28s last launch %.3f ms: %7.1f TFLOP/s of its own operand type, %.3e MFMA/s, %.3e K32-tile-units/s%s\n", name, last,
         waves * iters * flop_per_iter / t / 1e12, waves * iters * mfma_per_iter / t,
         units_per_iter ? waves * iters * units_per_iter / t : 0.0, units_per_iter ? "" : " (n/a)");
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? atof(argv[1]) : 2.0;
  // ---------------- part 1 -------------------------------------------------------------------
  {
    std::vector<unsigned> Acode(32 * 64), Bcode(64 * 32);   // A[i][k], B[k][j] as e2m3 codes
    std::vector<int> sA(64), sB(64);                        // scale byte per (row, kblock) / (col, kblock)
    srand(7);
    for (auto& c : Acode) c = rand() & 63;
    for (auto& c : Bcode) c = rand() & 63;
    for (int l = 0; l < 64; ++l) { sA[l] = 120 + rand() % 14; sB[l] = 121 + rand() % 12; }
    std::vector<v8i> ha(64), hb(64);
    std::vector<int> hsa(64), hsb(64);
    for (int l = 0; l < 64; ++l) {
      const int r = l & 31, kb = l >> 5;
      unsigned wa[8] = {0}, wb[8] = {0};
      for (int e = 0; e < 32; ++e) {
        const unsigned ca = Acode[r * 64 + kb * 32 + e], cb = Bcode[(kb * 32 + e) * 32 + r];
        const int bit = 6 * e;
        wa[bit >> 5] |= ca << (bit & 31);
        if ((bit & 31) > 26) wa[(bit >> 5) + 1] |= ca >> (32 - (bit & 31));
        wb[bit >> 5] |= cb << (bit & 31);
        if ((bit & 31) > 26) wb[(bit >> 5) + 1] |= cb >> (32 - (bit & 31));
      }
      for (int w = 0; w < 8; ++w) { ha[l][w] = (int)wa[w]; hb[l][w] = (int)wb[w]; }
      ha[l][6] = 0x55aa55aa; ha[l][7] = 0x12345678;   // garbage beyond the 6 dwords must be ignored
      hsa[l] = sA[l] | 0x11223300; hsb[l] = sB[l] | 0x44556600;   // only byte 0 (opsel 0) may matter
    }
    v8i *da, *db; int *dsa, *dsb; v16f* dout;
    CK(hipMalloc(&da, 64 * sizeof(v8i))); CK(hipMalloc(&db, 64 * sizeof(v8i)));
    CK(hipMalloc(&dsa, 256)); CK(hipMalloc(&dsb, 256)); CK(hipMalloc(&dout, 64 * sizeof(v16f)));
    CK(hipMemcpy(da, ha.data(), 64 * sizeof(v8i), hipMemcpyHostToDevice));
    CK(hipMemcpy(db, hb.data(), 64 * sizeof(v8i), hipMemcpyHostToDevice));
    CK(hipMemcpy(dsa, hsa.data(), 256, hipMemcpyHostToDevice));
    CK(hipMemcpy(dsb, hsb.data(), 256, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(mx_one<2>, dim3(1), dim3(64), 0, 0, da, db, dsa, dsb, dout);
    CK(hipDeviceSynchronize());
    std::vector<v16f> ho(64);
    CK(hipMemcpy(ho.data(), dout, 64 * sizeof(v16f), hipMemcpyDeviceToHost));
    double maxerr = 0, maxref = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0;
        for (int k = 0; k < 64; ++k)
          ref += (double)e2m3_value(Acode[i * 64 + k]) * ldexp(1.0, sA[i + 32 * (k >> 5)] - 127) *
                 (double)e2m3_value(Bcode[k * 32 + j]) * ldexp(1.0, sB[j + 32 * (k >> 5)] - 127);
        // C/D layout: col = lane & 31, row = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5)
        const int lane = j + 32 * ((i >> 2) & 1), reg = (i & 3) + 4 * (i >> 3);
        const double got = ho[lane][reg];
        if (fabs(got - ref) > maxerr) maxerr = fabs(got - ref);
        if (fabs(ref) > maxref) maxref = fabs(ref);
      }
    printf("part1 MX e2m3 32x32x64, hypothesis (lane = row + 32 kblock, element e at bits 6e.., scale byte 0 = 2^(b-127)): max |err| %.3e of max |ref| %.3e -> %s\n",
           maxerr, maxref, maxerr <= 1e-5 * maxref ? "CONFIRMED" : "MISMATCH");
  }
  // ---------------- part 2 -------------------------------------------------------------------
  {
    std::vector<float> x(64 * 32);
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 32; ++e) x[l * 32 + e] = (float)(e + 1) * 0.0625f * (l & 1 ? -1.f : 1.f) * (float)(1 + (l >> 1) % 3);
    float* dx; v6u* dq;
    CK(hipMalloc(&dx, x.size() * 4)); CK(hipMalloc(&dq, 64 * sizeof(v6u)));
    CK(hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    for (float scale : {1.0f, 4.0f, 0.25f, 3.0f}) {
      hipLaunchKernelGGL(cvt_f32, dim3(1), dim3(64), 0, 0, (const v16f*)dx, dq, scale);
      CK(hipDeviceSynchronize());
      std::vector<v6u> q(64);
      CK(hipMemcpy(q.data(), dq, 64 * sizeof(v6u), hipMemcpyDeviceToHost));
      for (int l : {0, 1, 2}) {
        printf("part2 cvt_2xpk16_fp6_f32 scale %.2f lane %d:", scale, l);
        for (int e = 0; e < 32; ++e) {
          const int bit = 6 * e;
          unsigned long long w = q[l][bit >> 5];
          if ((bit >> 5) + 1 < 6) w |= (unsigned long long)q[l][(bit >> 5) + 1] << 32;
          printf(" %g>%g", x[l * 32 + e], e2m3_value((unsigned)(w >> (bit & 31)) & 63));
        }
        printf("\n");
      }
    }
    std::vector<_Float16> xh(64 * 32);
    for (size_t i = 0; i < xh.size(); ++i) xh[i] = (_Float16)x[i];
    _Float16* dxh;
    CK(hipMalloc(&dxh, xh.size() * 2));
    CK(hipMemcpy(dxh, xh.data(), xh.size() * 2, hipMemcpyHostToDevice));
    for (float scale : {1.0f, 4.0f}) {
      hipLaunchKernelGGL(cvt_f16, dim3(1), dim3(64), 0, 0, (const v32h*)dxh, dq, scale);
      CK(hipDeviceSynchronize());
      std::vector<v6u> q(64);
      CK(hipMemcpy(q.data(), dq, 64 * sizeof(v6u), hipMemcpyDeviceToHost));
      printf("part2 cvt_pk32_fp6_f16 scale %.2f lane 0:", scale);
      for (int e = 0; e < 32; ++e) {
        const int bit = 6 * e;
        unsigned long long w = q[0][bit >> 5];
        if ((bit >> 5) + 1 < 6) w |= (unsigned long long)q[0][(bit >> 5) + 1] << 32;
        printf(" %g>%g", x[e], e2m3_value((unsigned)(w >> (bit & 31)) & 63));
      }
      printf("\n");
    }
  }
  // ---------------- part 3 -------------------------------------------------------------------
  {
    float* scratch;
    CK(hipMalloc(&scratch, 256));
    run_peak<0>("bf16 32x32x16", seconds, 16, 16 * 32768.0, 0, scratch);
    run_peak<1>("f16 32x32x16", seconds, 16, 16 * 32768.0, 0, scratch);
    run_peak<2>("MX fp6(e2m3) 32x32x64", seconds, 16, 16 * 131072.0, 0, scratch);
    run_peak<3>("MX fp8(e4m3) 32x32x64", seconds, 16, 16 * 131072.0, 0, scratch);
    run_peak<5>("bf16x3 unit (6 bf16)", seconds, 24, 24 * 32768.0, 4, scratch);
    run_peak<4>("f16mx unit (2 f16 + 1 fp6)", seconds, 12, 8 * 32768.0 + 4 * 131072.0, 4, scratch);
    run_peak<5>("bf16x3 unit (6 bf16) again", seconds, 24, 24 * 32768.0, 4, scratch);
  }
  return 0;
}
