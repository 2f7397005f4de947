// Shared host/device helpers for the gfx950 kernels.  Internal; the public surface is
// include/openibl_amd.h.
#pragma once

#include <hip/hip_runtime.h>

#include <atomic>
#include <stdint.h>
#include <stdio.h>

#include "openibl_amd.h"

namespace oibl {

// ---- error plumbing (host) ---------------------------------------------------------
void set_error(const char* fmt, ...);

#define OIBL_REQUIRE(cond, ...)                 \
  do {                                          \
    if (!(cond)) {                              \
      ::oibl::set_error(__VA_ARGS__);           \
      return OIBL_E_INVALID;                    \
    }                                           \
  } while (0)

#define OIBL_HIP_CHECK(expr)                                                          \
  do {                                                                                \
    hipError_t _e = (expr);                                                           \
    if (_e != hipSuccess) {                                                           \
      ::oibl::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                        __LINE__);                                                    \
      return OIBL_E_HIP;                                                              \
    }                                                                                 \
  } while (0)

// hipFuncSetAttribute applies to the CURRENT device: bit d of a per-kernel mask remembers that the
// kernel's dynamic-LDS limit has been raised on device d (a process may drive several devices).
// The bit is set only after the attribute call succeeded (a transient failure is retried by the
// next launch) and atomically (two host threads may launch the same kernel).
static inline unsigned long long oibl_device_bit() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return 0;
  return 1ull << (dev & 63);
}
#define OIBL_SET_MAX_LDS(kern, lds)                                                              \
  do {                                                                                           \
    static std::atomic<unsigned long long> seen_{0};                                             \
    const unsigned long long bit_ = ::oibl::oibl_device_bit();                                   \
    if (!(seen_.load(std::memory_order_relaxed) & bit_)) {                                       \
      OIBL_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),                    \
                                         hipFuncAttributeMaxDynamicSharedMemorySize, (lds)));    \
      seen_.fetch_or(bit_, std::memory_order_relaxed);                                           \
    }                                                                                            \
  } while (0)

#define OIBL_LAUNCH_CHECK()                                                          \
  do {                                                                               \
    hipError_t _e = hipGetLastError();                                               \
    if (_e != hipSuccess) {                                                          \
      ::oibl::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(_e),    \
                        __FILE__, __LINE__);                                         \
      return OIBL_E_HIP;                                                             \
    }                                                                                \
  } while (0)

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- element types -------------------------------------------------------------------
// bf16 is carried as raw 16-bit patterns; arithmetic always happens in fp32.
struct bf16_t {
  uint16_t bits;
};

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

__host__ __device__ static inline uint16_t f32_to_bf16_bits(float f) {
  // round-to-nearest-even; NaN kept quiet.  Device code: one v_cvt_pk_bf16_f32 (same rounding).
#if defined(__HIP_DEVICE_COMPILE__)
  return __builtin_bit_cast(uint16_t, (__bf16)f);
#endif
  union {
    float f;
    uint32_t u;
  } v;
  v.f = f;
  uint32_t u = v.u;
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x0040u);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (uint16_t)(u >> 16);
}

// IEEE binary16 <-> fp32 (round-to-nearest-even; v_cvt_f16_f32 / v_cvt_f32_f16 on the device)
__host__ __device__ static inline uint16_t f32_to_f16_bits(float f) {
  return __builtin_bit_cast(uint16_t, (_Float16)f);
}
__host__ __device__ static inline float f16_bits_to_f32(uint16_t b) {
  return (float)__builtin_bit_cast(_Float16, b);
}

__host__ __device__ static inline float bf16_bits_to_f32(uint16_t b) {
  union {
    float f;
    uint32_t u;
  } v;
  v.u = ((uint32_t)b) << 16;
  return v.f;
}

// bf16x3 ("split bf16", OIBL_BF16X3): a value v travels as the pair hi = bf16(v), lo = bf16(v - hi)
// (v - hi is exact in fp32; hi + lo carries 16 significant bits), and a product is evaluated as
// a_hi b_hi + a_hi b_lo + a_lo b_hi on the bf16 matrix cores with fp32 accumulation (the dropped
// a_lo b_lo term is ~2^-18 relative).  Storage: rows of elements in groups of 32 —
// [32 x hi | 32 x lo] = one 128-byte line per group, i.e. exactly one K-step of the GEMM cores, whose
// four 16-byte k-chunks per lane half are then hi[0:16], hi[16:32], lo[0:16], lo[16:32].
// The struct is only a tag (sizeof = 4 bytes per element); data is addressed through x3_off().
struct bf16x3_t {
  uint16_t hi, lo;
};
// byte offset of element c's hi half inside a row of x3 elements (its lo half sits 64 bytes further)
__host__ __device__ static inline size_t x3_off(size_t c) { return (c >> 5) * 128 + (c & 31) * 2; }
__device__ static inline void x3_split(float v, uint16_t& hi, uint16_t& lo) {
  hi = f32_to_bf16_bits(v);
  lo = f32_to_bf16_bits(v - bf16_bits_to_f32(hi));
}
__device__ static inline void x3_store(void* row, size_t c, float v) {
  uint16_t hi, lo;
  x3_split(v, hi, lo);
  uint16_t* p = reinterpret_cast<uint16_t*>(static_cast<char*>(row) + x3_off(c));
  p[0] = hi;
  p[32] = lo;
}
__device__ static inline float x3_load(const void* row, size_t c) {
  const uint16_t* p = reinterpret_cast<const uint16_t*>(static_cast<const char*>(row) + x3_off(c));
  return bf16_bits_to_f32(p[0]) + bf16_bits_to_f32(p[32]);
}

// f16mx (OIBL_F16MX): a value v travels as hi = fp16(v) plus MX-fp6 (e2m3, one e8m0 scale per 32
// elements) images of hi and of lo = v - hi, and a product is evaluated as
//     a.b ~= hi(a).hi(b)                        2 x v_mfma_f32_32x32x16_f16 per 32 K
//          + q6(hi(a)).q6(lo(b)) + q6(lo(a)).q6(hi(b))   1 x v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64:
//                                                the two cross terms concatenated along K)
// with fp32 accumulation.  The cross terms are 2^-11 of the product and carry 4 significant bits, the
// dropped lo.lo term is 2^-22: ~2^-15 relative per product (tools/f16mx_numerics.py: 3.9e-5 on the
// conv5_3 map after 13 layers against fp64; bf16x3 1.5e-5, plain bf16 8e-3).  Half the matrix-pipe
// time of bf16x3 per product (96 instead of 192 cycles per 32x32x32 block; 1.77x measured under the
// power cap, profiles/r03_a_mx_probe.txt).
// Storage: rows of elements in groups of 32 = one 128-byte line = one K-step of the GEMM cores:
//   bytes   0.. 63  hi[0:32] fp16                       (k-chunks 0, 1: the two f16 MFMAs)
//   bytes  64.. 79  first 16 bytes of q6(hi)            (slot 4)
//   bytes  80.. 95  first 16 bytes of q6(lo)            (slot 5)
//   bytes  96..111  last 8 bytes of q6(hi) | 4 x 0 | scale byte of q6(hi) | 3 x 0     (slot 6)
//   bytes 112..127  last 8 bytes of q6(lo) | 4 x 0 | scale byte of q6(lo) | 3 x 0     (slot 7)
// (the scale sits 12 bytes into its slot, not 8: the kernels fetch the tail as ds_read_b64 + ds_read_b32,
//  and adjacent loads would be merged into one ds_read_b96 — twice the LDS cycles, and a merged load
//  carries no alias information, so the compiler drains the LDS-DMA queue (vmcnt(0)) in front of it)
// q6 = 32 x e2m3 packed 6 bits each (element e at bits 6e..6e+5, the order of
// v_cvt_scalef32_pk32_fp6_f16), value = code * 2^(scale byte - 127).  A lane of the MX instruction
// holds one row's 32 K elements of one 32-block (lanes 0-31: block 0, lanes 32-63: block 1): the A
// side reads slots (4, 6) in its lower half and (5, 7) in its upper half — K = [q6(hi) | q6(lo)] —
// the B side the other way round — K = [q6(lo) | q6(hi)] — so ONE stored format serves both operands.
// The struct is only a tag (sizeof = 4 bytes per element).
struct f16mx_t {
  uint16_t a, b;
};
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(8))) int i32x8_t;

// scale byte (e8m0) of a block whose largest |hi| is amax: the smallest power of two s with
// amax / s <= 7.5 (the e2m3 maximum), never below 12 so that the byte of the lo block (11 less:
// |lo| <= 2^-11 |hi| elementwise) stays a valid exponent
__host__ __device__ static inline int mx_scale_byte(float amax) {
  const uint32_t bits = __builtin_bit_cast(uint32_t, amax);
  const int b = (int)((bits + 0x00100000u) >> 23) - 2;   // + 1 when the mantissa is >= 1.875
  return b < 12 ? 12 : (b > 254 ? 254 : b);
}

// Range guard of the f16mx arithmetic.  hi = fp16(v) exists only for |v| <= 65504: beyond that the packers
// saturate hi and the excess goes to lo, whose e2m3 image saturates too — the line no longer carries v (one
// 256 -> 256 layer: rel-L2 2.4e-1 with activations up to 2e5 against 1.5e-5 up to 3e4).  Every packer
// therefore RAISES A STICKY FLAG — one 32-bit word the caller zeroes before the pass and reads after it —
// when a group's largest |v| is beyond fp16; the host mirror re-runs such a batch in bf16x3
// (openibl_amd/models.py).  The branch is wave-uniform and off the common path; all writers store 1.
__device__ static inline void mx_raise_range_flag(unsigned* flag) {
  if (flag != nullptr) __hip_atomic_store(flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// 32 fp32 values (one pixel's / row's 32-element group) -> the 128-byte f16mx line
__device__ static inline void mx_pack_line(const float (&v)[32], uint4 (&out)[8], unsigned* range_flag = nullptr) {
#if defined(__HIP_DEVICE_COMPILE__)   // (the host pass only needs the declaration)
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  typedef __attribute__((ext_vector_type(32))) _Float16 h32;
  typedef __attribute__((ext_vector_type(6))) unsigned u6;
  typedef __attribute__((ext_vector_type(16))) unsigned u16v_;
  u16v_ hw_;           // hi[2i], hi[2i + 1] as one dword each
  f32x16_t le, lo;     // remainders of the even / odd elements (the fp32 convert interleaves its inputs)
  // Largest |v| of the group (16 x v_max3_f32 with |.| modifiers); the largest |hi| is its fp16 rounding
  // (rounding is monotonic).  ~70 vector instructions per line instead of ~135: the epilogues of the f16mx
  // kernels spend most of their time here.
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 32; e += 2) asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v[e]), "v"(v[e + 1]));
  // beyond fp16: hi saturates at +-65504, the excess goes to lo, and the range flag is raised (the line is
  // then NOT a 1e-4 image of v: see mx_raise_range_flag).  Wave-uniform branch, so that the 32 clamps are
  // not if-converted into the common path.
  float c[32];
#pragma unroll
  for (int e = 0; e < 32; ++e) c[e] = v[e];
  if (__builtin_amdgcn_ballot_w64(amax > 65504.f) != 0) {
#pragma unroll
    for (int e = 0; e < 32; ++e) c[e] = __builtin_amdgcn_fmed3f(c[e], -65504.f, 65504.f);
    if (amax > 65504.f) mx_raise_range_flag(range_flag);
  }
#pragma unroll
  for (int e = 0; e < 32; e += 2) {
    const h2 p = __builtin_convertvector((f2){c[e], c[e + 1]}, h2);
    const unsigned pw = __builtin_bit_cast(unsigned, p);
    hw_[e >> 1] = pw;
    // v - hi in ONE instruction: v_fma_mix_f32 reads the fp16 half of its first operand directly (op_sel
    // picks the half); exact like convert + subtract.  (Written as fmaf the compiler folds the -1 into a
    // subtraction and converts first.)
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(pw), "v"(v[e]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(pw), "v"(v[e + 1]));
    le[e >> 1] = r0;
    lo[e >> 1] = r1;
  }
  const h32 h = __builtin_bit_cast(h32, hw_);
  amax = (float)(_Float16)fminf(amax, 65504.f);
  const int bh = mx_scale_byte(amax), bl = bh - 11;
  const u6 h6 = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(h, __builtin_bit_cast(float, (uint32_t)bh << 23));
  const u6 l6 = __builtin_amdgcn_cvt_scalef32_2xpk16_fp6_f32(le, lo, __builtin_bit_cast(float, (uint32_t)bl << 23));
#pragma unroll
  for (int s = 0; s < 4; ++s) out[s] = make_uint4(hw_[4 * s], hw_[4 * s + 1], hw_[4 * s + 2], hw_[4 * s + 3]);
  out[4] = make_uint4(h6[0], h6[1], h6[2], h6[3]);
  out[5] = make_uint4(l6[0], l6[1], l6[2], l6[3]);
  out[6] = make_uint4(h6[4], h6[5], 0u, (unsigned)bh);
  out[7] = make_uint4(l6[4], l6[5], 0u, (unsigned)bl);
#endif
}

// The same for ONE LANE'S HALF of a group: 16 elements here, the other 16 in lane ^ 32 — the accumulator
// layout of a 32x32 MFMA with the group along the rows (lane half hp, registers 4g + r <-> row 8g + 4hp + r).
// Out: the fp16 hi parts (8 dwords), the e2m3 images of hi and lo (3 dwords each: element e at bits 6e), the
// two scale bytes (the block's: both lanes compute the same).  ONE convert instruction makes both images:
// lo goes in as fp16(lo * 2^11) beside hi (|lo| 2^11 <= |hi|: same block scale, and a full 16-register
// source instead of two half-used ones — these callers have no registers to give away).  lo is thereby
// rounded twice (fp16, then e2m3; mx_pack_line converts it from fp32): a code can move by one step where
// the fp16 rounding crosses an e2m3 midpoint — 2^-12 of a term that is itself 2^-11 of the product.
// CLAMP = false: the caller guarantees |v| <= 65504 (the stems fold the bound into their ReLU).
// Range guard: `seen` accumulates the largest group maximum the caller has packed (one v_max per call — no
// branch, no memory operation: these callers count their own lgkmcnt / vmcnt waits, and a kernel-argument or
// flag access inside their loops would disturb the counts); the caller raises the flag ONCE, behind its loops,
// when seen > 65504 (CLAMP) / seen >= 65504 (!CLAMP: a value clamped to the bound, or — harmlessly — exactly it).
template <bool CLAMP = true>
__device__ static inline void mx_pack_half(const float (&v)[16], unsigned (&h16)[8], unsigned (&h6)[3],
                                           unsigned (&l6)[3], unsigned& bh, unsigned& bl, float& seen) {
#if defined(__HIP_DEVICE_COMPILE__)
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) _Float16 h2;
  typedef __attribute__((ext_vector_type(32))) _Float16 h32;
  typedef __attribute__((ext_vector_type(6))) unsigned u6;
  typedef __attribute__((ext_vector_type(16))) unsigned u16v_;
  float amax = 0.f;
#pragma unroll
  for (int e = 0; e < 16; e += 2) asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(amax) : "v"(v[e]), "v"(v[e + 1]));
  {  // the other half's maximum.  v_permlane32_swap exchanges the upper half of its first register with the
     // lower half of its second: with the maximum in both, every lane finds its partner's in one of them.
     // (Inline asm keeps the instruction count exact; with __builtin_amdgcn_permlane32_swap mind that
     //  __builtin_bit_cast(float, r[1]) of the returned vector reads element 0 — bit_cast of a vector-element
     //  lvalue — which cost this function a GPU run.  s_nop: the VALU-write -> permlane hazard is the
     //  compiler's to handle for the builtin, ours here.)
    unsigned a = __builtin_bit_cast(unsigned, amax), b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(a), "+v"(b));
    asm("v_max3_f32 %0, %0, %1, %2" : "+v"(amax) : "v"(a), "v"(b));
  }
  float c[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) c[e] = v[e];
  seen = fmaxf(seen, amax);
  if constexpr (CLAMP) {
    if (__builtin_amdgcn_ballot_w64(amax > 65504.f) != 0) {   // wave-uniform, off the common path
#pragma unroll
      for (int e = 0; e < 16; ++e) c[e] = __builtin_amdgcn_fmed3f(c[e], -65504.f, 65504.f);
    }
  }
  u16v_ w_;   // dwords 0..7: hi pairs, 8..15: fp16(lo * 2^11) pairs
#pragma unroll
  for (int e = 0; e < 16; e += 2) {
    const h2 p = __builtin_convertvector((f2){c[e], c[e + 1]}, h2);
    const unsigned pw = __builtin_bit_cast(unsigned, p);
    w_[e >> 1] = pw;
    h16[e >> 1] = pw;
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(pw), "v"(v[e]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(pw), "v"(v[e + 1]));
    const h2 q = __builtin_convertvector((f2){r0, r1} * 2048.f, h2);
    w_[8 + (e >> 1)] = __builtin_bit_cast(unsigned, q);
  }
  amax = (float)(_Float16)(CLAMP ? fminf(amax, 65504.f) : amax);
  const int b = mx_scale_byte(amax);
  bh = (unsigned)b;
  bl = (unsigned)(b - 11);
  const u6 o = __builtin_amdgcn_cvt_scalef32_pk32_fp6_f16(__builtin_bit_cast(h32, w_),
                                                          __builtin_bit_cast(float, (uint32_t)b << 23));
  h6[0] = o[0];
  h6[1] = o[1];
  h6[2] = o[2];
  l6[0] = o[3];
  l6[1] = o[4];
  l6[2] = o[5];
#endif
}

// value of e2m3 code c (sign, 2 exponent bits, 3 mantissa bits; bias 1)
__host__ __device__ static inline float mx_e2m3_value(unsigned c) {
  const int e = (c >> 3) & 3, m = c & 7;
  const float v = e == 0 ? (float)m * 0.125f : (1.0f + (float)m * 0.125f) * (float)(1 << (e - 1));
  return (c & 32) ? -v : v;
}
// element c (0..31) of a stored line as the kernels see it: hi (exact fp16) and the two fp6 images
__host__ __device__ static inline void mx_line_decode(const void* line, int c, float& hi, float& hi6, float& lo6) {
  const unsigned char* b = static_cast<const unsigned char*>(line);
  hi = f16_bits_to_f32((uint16_t)(b[2 * c] | (b[2 * c + 1] << 8)));
  const int bit = 6 * c;
  auto code = [&](int first, int last) {   // the 24 packed bytes are split 16 + 8 over two slots
    unsigned long long w = 0;
    const int byte0 = bit >> 3;
    for (int k = 0; k < 2; ++k) {
      const int by = byte0 + k;
      if (by < 24) w |= (unsigned long long)b[by < 16 ? first + by : last + by - 16] << (8 * k);
    }
    return (unsigned)(w >> (bit & 7)) & 63u;
  };
  const float sh = __builtin_ldexpf(1.0f, (int)b[108] - 127), sl = __builtin_ldexpf(1.0f, (int)b[124] - 127);
  hi6 = mx_e2m3_value(code(64, 96)) * sh;
  lo6 = mx_e2m3_value(code(80, 112)) * sl;
}

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  __device__ static inline float load(const float* p) { return *p; }
  __device__ static inline void store(float* p, float v) { *p = v; }
  __device__ static inline float from_f32(float v) { return v; }
  __device__ static inline float to_f32(float v) { return v; }
};
template <>
struct Elem<bf16_t> {
  __device__ static inline float load(const bf16_t* p) { return bf16_bits_to_f32(p->bits); }
  __device__ static inline void store(bf16_t* p, float v) { p->bits = f32_to_bf16_bits(v); }
  __device__ static inline bf16_t from_f32(float v) {
    bf16_t r;
    r.bits = f32_to_bf16_bits(v);
    return r;
  }
  __device__ static inline float to_f32(bf16_t v) { return bf16_bits_to_f32(v.bits); }
};

// Test hooks.  The shipping library (libopenibl_amd.so) is compiled WITHOUT OIBL_DEBUG_HOOKS: every hook
// variable is then a compile-time constant holding its default — no mutable process-wide state on any
// launch path, the experiment kernels behind the hooks are not even instantiated — and no oibl_debug_*
// function exists.  libopenibl_amd_dbg.so (same sources, -DOIBL_DEBUG_HOOKS) carries the hooks for the
// variant tests and the tests/gpu_* diagnostics; they are plain process-wide ints there: not
// thread-safe, not stream-scoped, test infrastructure only.
#ifdef OIBL_DEBUG_HOOKS
#define OIBL_HOOK(type, name, dflt) static type name = dflt
extern int g_regstage;   // 1 = stage GEMM tiles through registers instead of global_load_lds
#else
#define OIBL_HOOK(type, name, dflt) static constexpr type name = dflt
constexpr int g_regstage = 0;
#endif

// 128 zero bytes: the source every out-of-image im2col tap points its load at.
const void* zero_line_device_ptr();

// wave-level reductions (wave = 64 lanes on gfx950)
__device__ static inline float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ static inline float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Bijective XCD-aware remap of a 1-D block id: the hardware deals block b to XCD b % 8;
// give every XCD a contiguous range of logical tiles so neighbouring tiles share an L2.
__device__ static inline unsigned xcd_remap(unsigned bid, unsigned nblk) {
  const unsigned xcd = bid & 7u, idx = bid >> 3;
  const unsigned q = nblk >> 3, r = nblk & 7u;
  const unsigned start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
  return start + idx;
}

// XCD-aware tile rasterisation for a tiles_m x tiles_n grid of output tiles (block b runs on XCD
// b % 8, each XCD has its own 4 MiB L2).
//   raster 0: every XCD gets a contiguous range of tile ids, N-tile fastest — the tiles of an XCD
//             share their M panels (activations) but cycle through ALL N panels (weights);
//   raster 1 (tiles_n divides 8): an XCD serves ONE N-tile — its weight panel (2.36 MB for a
//             512-channel layer at BN = 256) stays resident in that L2 next to a contiguous range of
//             M panels; an M panel is then fetched by tiles_n XCDs instead of one.
// Both are bijections between blocks and tiles (the per-XCD block counts of the hardware's
// round-robin deal equal the per-XCD tile counts of the split: nblk = tiles_m * tiles_n,
// 8 = G * tiles_n  =>  nblk / 8 = tiles_m / G and nblk % 8 = (tiles_m % G) * tiles_n).
__device__ static inline void xcd_tile(unsigned bid, unsigned tiles_m, unsigned tiles_n, int raster,
                                       int& tm, int& tn) {
  if (raster == 1 && tiles_n > 1 && tiles_n <= 8 && (8u % tiles_n) == 0u) {
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned G = 8u / tiles_n, g = xcd / tiles_n;
    const unsigned q = tiles_m / G, r = tiles_m % G;
    tn = (int)(xcd % tiles_n);
    tm = (int)(g * q + (g < r ? g : r) + idx);
    return;
  }
  const unsigned tile = xcd_remap(bid, tiles_m * tiles_n);
  tn = (int)(tile % tiles_n);
  tm = (int)(tile / tiles_n);
}

}  // namespace oibl
