/*
 * openibl_amd.h — C ABI of the MI355X (gfx950) descriptor + matching hot path.
 *
 * This is the drop-in boundary.  The reference (yxgeee/OpenIBL) has no
 * FFI of its own: its hot path is the Python call surface ibl.models / ibl.pca /
 * ibl.evaluators.  Every entry point below replaces the device work of one reference
 * function (cited per function as path:line under the reference tree) and is what a
 * ctypes binding added to the reference would bind.
 *
 * Conventions
 *   - plain C: pointers and sizes only, no torch / C++ types.
 *   - every pointer argument is a DEVICE pointer valid on the current HIP device unless
 *     the name ends in _host.
 *   - `stream` is a hipStream_t passed as void*; every call is asynchronous and
 *     stream-ordered; re-entrant across streams (no hidden global state besides the
 *     thread-local error string: the shipping library has no mutable statics on any launch path — the
 *     test hooks exist only in libopenibl_amd_dbg.so, built from the same sources with
 *     -DOIBL_DEBUG_HOOKS).
 *   - no hidden allocation: outputs and scratch are caller-provided; scratch size comes
 *     from the matching *_workspace_bytes() query.  Workspace pointers must be 256-byte
 *     aligned.
 *   - return value: 0 = OK, negative = error (OIBL_E_*); message via oibl_last_error().
 *   - `precision` selects the arithmetic of the contraction:
 *       OIBL_BF16 : bf16 operands, fp32 accumulate on v_mfma_f32_32x32x16_bf16
 *       OIBL_F32  : exact fp32 on v_mfma_f32_32x32x2_f32 (parity mode)
 *       OIBL_BF16X3 : "split bf16" — every operand travels as hi = bf16(v), lo = bf16(v - hi) and a
 *                   product is hi.hi + hi.lo + lo.hi on the bf16 matrix cores, fp32 accumulate
 *                   (~2^-17 relative per product: fp32-class descriptors at 3x the bf16 MFMA work
 *                   instead of 16x).  Element = 4 bytes; a row of C elements (C % 32 == 0) is stored
 *                   as C/32 groups of [32 x hi | 32 x lo] (128 bytes), see oibl_x3_split_rows.
 *       OIBL_F16MX : fp16 main term + MX-fp6 cross terms — every operand travels as hi = fp16(v)
 *                   plus block-scaled e2m3 images of hi and of lo = v - hi (one e8m0 scale per 32
 *                   elements); a product is hi.hi on v_mfma_f32_32x32x16_f16 plus
 *                   q6(hi).q6(lo) + q6(lo).q6(hi) on ONE v_mfma_scale_f32_32x32x64_f8f6f4 (the two
 *                   cross terms concatenated along K), fp32 accumulate: ~2^-15 relative per product —
 *                   inside north_star's 1e-4 on the descriptor — at HALF the matrix-pipe time of
 *                   OIBL_BF16X3.  Element = 4 bytes; a row of C elements (C % 32 == 0) is C/32 lines
 *                   of 128 bytes: [32 x fp16 | q6(hi) 16 B | q6(lo) 16 B | q6(hi) 8 B, scale, pad |
 *                   q6(lo) 8 B, scale, pad], see oibl_mx_split_rows.
 *                   RANGE: hi = fp16(v) exists only for |v| <= 65504.  A producer of f16mx lines that meets
 *                   a larger value saturates it (the line is then NOT a 1e-4 image of the value) and raises
 *                   the caller's RANGE FLAG — a uint32 in device memory that the caller zeroes before the
 *                   pass and reads after it: non-zero = the pass must be repeated in OIBL_BF16X3 (same
 *                   tolerance class, no range limit).  oibl_vgg16_conv5_forward* keep the flag in the first
 *                   word of their workspace; oibl_conv3x3_nhwc_flagged / oibl_mx_split_rows_flagged take it
 *                   as an argument; the matching entry points mark an out-of-range descriptor row with a
 *                   +inf norm (all its distances are +inf).  The host mirror (openibl_amd/models.py) reads
 *                   the flag once per batch and re-runs flagged batches in OIBL_BF16X3.
 *     and with it the element type of activation / packed-weight buffers ("T" below:
 *     uint16 bf16 bits, float, or the 4-byte split element).
 */
#ifndef OPENIBL_AMD_H
#define OPENIBL_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OIBL_OK 0
#define OIBL_E_INVALID (-1)   /* bad argument (shape, alignment, null pointer)      */
#define OIBL_E_WORKSPACE (-2) /* workspace too small                                */
#define OIBL_E_HIP (-3)       /* a HIP runtime call / kernel launch failed           */
#define OIBL_E_UNSUPPORTED (-4)

#define OIBL_BF16 0
#define OIBL_F32 1
#define OIBL_BF16X3 2
#define OIBL_F16MX 3

/* Storage type of a descriptor matrix handed to the *_st matching entry points. */
#define OIBL_ST_F32 0
#define OIBL_ST_F16 1  /* IEEE binary16 */
#define OIBL_ST_BF16 2

#define OIBL_VGG16_NUM_CONV 13

/* ---- library ---------------------------------------------------------------------- */

/* ABI version of this header (bumped on any signature change). */
int oibl_abi_version(void);
/* Last error message of the calling thread (never NULL). */
const char* oibl_last_error(void);
/* Name of the gfx target the kernels were compiled for ("gfx950"). */
const char* oibl_target_arch(void);
/* sizeof(T) for a precision code, 0 if unknown. */
size_t oibl_elem_size(int precision);

/* ---- element conversion helpers --------------------------------------------------- */

/* fp32 -> bf16 (round-to-nearest-even), n elements. */
int oibl_cast_f32_to_bf16(const float* src, uint16_t* dst, size_t n, void* stream);
/* bf16 -> fp32, n elements. */
int oibl_cast_bf16_to_f32(const uint16_t* src, float* dst, size_t n, void* stream);
/* fp32 <-> IEEE binary16 (round-to-nearest-even), n elements: 16-bit descriptor storage
 * (BASELINE.json configs[4], "fp16 descriptors"; no counterpart in the reference). */
int oibl_cast_f32_to_f16(const float* src, uint16_t* dst, size_t n, void* stream);
int oibl_cast_f16_to_f32(const uint16_t* src, float* dst, size_t n, void* stream);

/* fp32 rows [rows][C] <-> OIBL_BF16X3 rows (C/32 groups of [32 hi | 32 lo] per row; C % 32 == 0):
 * the activation / operand layout of the OIBL_BF16X3 kernels.  join returns hi + lo (exact). */
int oibl_x3_split_rows(const float* src, void* dst, size_t rows, int C, void* stream);
int oibl_x3_join_rows(const void* src, float* dst, size_t rows, int C, void* stream);
/* fp32 rows [rows][C] <-> OIBL_F16MX rows (C/32 lines of 128 bytes per row; C % 32 == 0).  join
 * returns what the kernels see of every element: which = 0: hi + q6(lo) (the stored value to ~2^-15
 * of the line's largest element), 1: hi (fp16, exact), 2: q6(hi), 3: q6(lo). */
int oibl_mx_split_rows(const float* src, void* dst, size_t rows, int C, void* stream);
/* The same with a range flag (may be NULL): *range_flag is set to 1 when a group holds a value beyond
 * +-65504 (see OIBL_F16MX above); never cleared by the call. */
int oibl_mx_split_rows_flagged(const float* src, void* dst, size_t rows, int C, uint32_t* range_flag,
                               void* stream);
int oibl_mx_join_rows(const void* src, float* dst, size_t rows, int C, int which, void* stream);

/* ---- bilinear resize --------------------------------------------------------------- *
 * x [N][C][H][W] fp32 -> out [N][C][H2][W2] fp32 with the arithmetic of
 * torch.nn.functional.interpolate(x, size=(H2, W2), mode="bilinear", align_corners=False)
 * (source index (dst + 0.5) * in/out - 0.5 clamped at 0, no antialiasing).  Used by the
 * multi-scale extraction of BASELINE.json configs[4]; the reference itself resizes PIL images in
 * its loader (ibl/utils/data/__init__.py:37-42) and has no multi-scale path. */
int oibl_resize_bilinear_nchw(const float* x, int N, int C, int H, int W, float* out, int H2, int W2,
                              void* stream);

/* ---- VGG16 conv1_1 .. conv5_3 backbone -------------------------------------------- *
 * Replaces VGG.forward's `self.base(x)`  (ibl/models/vgg.py:61-62; layer list built at
 * vgg.py:40-42 = torchvision vgg16.features[:-2]: 13 conv3x3(pad 1)+bias, ReLU after all
 * but the last, 2x2/2 max-pool after conv1_2, conv2_2, conv3_3, conv4_3).                */

/* Re-pack one conv weight from the state-dict layout [Cout][Cin][3][3] fp32 into the
 * kernel layout [tap=ky*3+kx][Cout][Cin_pad] of T (Cin_pad = Cin rounded up to the K-step:
 * 64 elements bf16 / 32 fp32; pad is zero).  packed must hold
 * oibl_conv3x3_packed_bytes(cout, cin, precision) bytes. */
size_t oibl_conv3x3_packed_bytes(int cout, int cin, int precision);
int oibl_pack_conv3x3_weights(const float* w_oihw, int cout, int cin, int precision,
                              void* packed, void* stream);

/* One 3x3 / pad 1 / stride 1 convolution + bias (+ReLU) (+2x2/2 max-pool, floor) on NHWC
 * activations: in [N][H][W][Cin] T -> out [N][Ho][Wo][Cout] T, (Ho,Wo) = (H,W) or
 * (H/2,W/2) when pool != 0.  Cin % 64 == 0 (bf16) / % 32 (fp32); Cout % 64 == 0.
 * (nn.Conv2d + nn.ReLU + nn.MaxPool2d modules of vgg.py:41-42.) */
int oibl_conv3x3_nhwc(const void* in, int N, int H, int W, int cin, const void* packed_w,
                      const float* bias, int cout, int relu, int pool, int precision,
                      void* out, void* stream);
/* The same with a range flag for OIBL_F16MX outputs (may be NULL; ignored by the other precisions):
 * *range_flag is set to 1 when an output of the layer is beyond +-65504 and was saturated (see OIBL_F16MX
 * above); never cleared by the call. */
int oibl_conv3x3_nhwc_flagged(const void* in, int N, int H, int W, int cin, const void* packed_w,
                              const float* bias, int cout, int relu, int pool, int precision,
                              void* out, uint32_t* range_flag, void* stream);
/* The same with scratch: a layer whose tiling would leave most of the chip idle — conv4 / conv5 of a few
 * images — is then contracted split-K (several workgroups per output tile, fp32 partial tiles in `ws`, a
 * fixed-order reduction that also applies bias / ReLU / pool: deterministic, equal to the one-pass result up
 * to fp32 association), in every precision.  oibl_conv3x3_workspace_bytes may return 0; ws == NULL runs the
 * layer in one pass whatever its size.  This is what oibl_vgg16_conv5_forward runs per layer. */
size_t oibl_conv3x3_workspace_bytes(int N, int H, int W, int cin, int cout, int pool, int precision);
int oibl_conv3x3_nhwc_ws(const void* in, int N, int H, int W, int cin, const void* packed_w,
                         const float* bias, int cout, int relu, int pool, int precision, void* out,
                         void* ws, size_t ws_bytes, uint32_t* range_flag, void* stream);

/* First layer: reads the reference's input tensor directly — x [N][3][H][W] fp32 NCHW
 * (already mean/std normalised, ibl/utils/data/__init__.py:40-41) — conv1_1 + bias + ReLU,
 * writes NHWC T [N][H][W][64].  w is the plain state-dict tensor [64][3][3][3] fp32. */
int oibl_conv1_1_nchw(const float* x_nchw, int N, int H, int W, const float* w_oihw,
                      const float* bias, int precision, void* out, void* stream);

/* Global max-pool over positions of an NHWC feature map -> pool_x [N][C] fp32
 * (nn.AdaptiveMaxPool2d(1), vgg.py:43,67-68). */
int oibl_global_maxpool_nhwc(const void* feat, int N, int P, int C, int precision,
                             float* out, void* stream);

/* NHWC T -> NCHW fp32 (the layout VGG.forward returns `x` in, vgg.py:70). */
int oibl_nhwc_to_nchw_f32(const void* feat, int N, int P, int C, int precision, float* out,
                          void* stream);
/* NCHW fp32 -> NHWC T (to feed NetVLAD.forward from a reference-layout tensor). */
int oibl_nchw_f32_to_nhwc(const float* x, int N, int C, int P, int precision, void* out,
                          void* stream);

/* The same backbone fed with the loader's RAW image: x_nhwc [N][H][W][3] uint8 (what
 * PIL / cv2 decode to), ToTensor + Normalize (ibl/utils/data/__init__.py:37-42:
 * (u / 255 - mean) / std, fp32) folded into the first kernel.  mean3_host / std3_host: 3 floats
 * each, HOST pointers.  The host -> device copy shrinks 4x (0.9 MB instead of 3.7 MB per 480x640 image).
 * OIBL_BF16: the fused stem looks the normalised bf16 operand up in a 3 x 257 table built with
 * exactly that arithmetic — bit-identical to oibl_vgg16_conv5_forward on the normalised fp32 tensor.
 * OIBL_BF16X3 / OIBL_F16MX: the fused stems gather the bytes themselves (three 12-byte loads per window)
 * and evaluate Normalize as ONE fma per value, u * 1/(255 std) - mean/std: within 2^-16 of the loader's
 * three rounded operations on values up to 151 (identical bf16 hi parts of conv1_1's operand for all 768
 * (channel, byte) pairs, 42 lo parts one unit apart) — the conv5_3 map is within ~1e-6 (bf16x3) / ~1e-5
 * (f16mx: the size of its own rounding) of the fp32-input result, not bit-identical.  These stems read the image
 * through dword-aligned loads over a descriptor rounded up to whole dwords (a partly covered last dword reads as
 * zero, nothing beyond the rounded size is touched): x_nhwc that is NOT 4-byte aligned takes the normalising
 * pass of OIBL_F32 below instead (same results as the fp32-input route), it is never read misaligned.
 * OIBL_F32 (and shapes the stems do not take): a normalising uint8 -> fp32 NCHW pass into the workspace,
 * then the regular path (bit-identical).  ev_*: optional hipEvent_t recorded around the matrix-core
 * launches (as oibl_vgg16_conv5_forward_ev), may be NULL.      */
size_t oibl_vgg16_u8_workspace_bytes(int N, int H, int W, int precision);
int oibl_vgg16_conv5_forward_u8(const uint8_t* x_nhwc, int N, int H, int W, const float* mean3_host,
                                const float* std3_host, const void* const* packed_w_host,
                                const float* const* bias_host, int precision, void* feat, void* ws,
                                size_t ws_bytes, void* stream, void* ev_igemm_begin,
                                void* ev_igemm_end);

/* Fused VGG stem, bf16: conv1_1 + ReLU + conv1_2 + ReLU + 2x2/2 max-pool in one launch —
 * replaces modules 0-4 of VGG.base (ibl/models/vgg.py:40-42; forward :61-62):
 *   x_nchw [N][3][H][W] fp32 -> out [N][H/2][W/2][64] bf16 (NHWC).
 * w1_oihw / b1: conv1_1 in the state-dict layout (fp32); packed_w2: conv1_2 packed by
 * oibl_pack_conv3x3_weights(..., OIBL_BF16).  Bit-identical to oibl_conv1_1_nchw followed by
 * oibl_conv3x3_nhwc(relu=1, pool=1) in OIBL_BF16; the conv1_1 activations never reach HBM.
 * oibl_vgg16_conv5_forward uses it automatically in OIBL_BF16. */
int oibl_vgg16_stem_bf16(const float* x_nchw, int N, int H, int W, const float* w1_oihw,
                         const float* b1, const void* packed_w2, const float* b2, void* out,
                         void* stream);

/* The same fusion in OIBL_BF16X3: out [N][H/2][W/2][64] as (hi, lo) split elements; packed_w2 from
 * oibl_pack_conv3x3_weights(..., OIBL_BF16X3).  A workgroup serves half of conv1_2's output channels
 * and consumes a tile in two passes over halves of its input channels, so that the split weights
 * (72 KiB) and two 42.5 KiB halo buffers fit LDS.  Bit-identical to oibl_conv1_1_nchw followed by
 * oibl_conv3x3_nhwc(relu=1, pool=1) in OIBL_BF16X3 up to the summation order of conv1_2
 * (channel chunk outer, tap inner); used automatically by oibl_vgg16_conv5_forward. */
int oibl_vgg16_stem_x3(const float* x_nchw, int N, int H, int W, const float* w1_oihw,
                       const float* b1, const void* packed_w2, const float* b2, void* out,
                       void* stream);

/* The same fusion in OIBL_F16MX: out [N][H/2][W/2][64] as f16mx lines; packed_w2 from
 * oibl_pack_conv3x3_weights(..., OIBL_F16MX).  conv1_1 (K = 27) is computed in split bf16, its output is
 * packed into f16mx lines inside LDS, conv1_2 runs in the f16mx arithmetic (2 fp16 + 1 scaled-fp6 MFMA per 32
 * channels) and the pooled map leaves as f16mx lines.  The e2m3 image of a lo part is rounded through fp16
 * here (the other f16mx producers convert it from fp32): its codes can differ by one step from
 * oibl_mx_split_rows of the same values.  H >= 2, W >= 3.  Used automatically by oibl_vgg16_conv5_forward
 * in OIBL_F16MX. */
int oibl_vgg16_stem_mx(const float* x_nchw, int N, int H, int W, const float* w1_oihw,
                       const float* b1, const void* packed_w2, const float* b2, void* out,
                       void* stream);

/* Whole backbone: x [N][3][H][W] fp32 -> feat [N][P][512] T, P = (H/16)*(W/16) (floor at
 * every pool).  packed_w_host / bias_host are HOST arrays of 13 DEVICE pointers: entry 0
 * is the plain [64][3][3][3] fp32 conv1_1 weight, entries 1..12 are packed by
 * oibl_pack_conv3x3_weights; bias entries are [Cout] fp32.
 * OIBL_BF16X3: activations between the layers are (hi, lo) split elements, but `feat` is written as
 * plain fp32 (the head consumes it with OIBL_F32).
 * OIBL_F16MX: every entry packed with OIBL_F16MX (conv1_1 + conv1_2 + pool = oibl_vgg16_stem_mx; the
 * mode has no unfused front and 32-bit-offset kernels only: a batch whose fp32 input or whose largest
 * activation — conv2_2's input, N (H/2) (W/2) 128 x 4 bytes: 95 images of 480x640 — reaches 3.5 GB is
 * refused with OIBL_E_INVALID); `feat` is plain fp32.  The FIRST
 * 32-BIT WORD OF THE WORKSPACE is the pass's range flag: zeroed (stream-ordered) when the call starts,
 * 1 afterwards if any activation between the layers was beyond the format's range — `feat` is then not a 1e-4
 * result and the batch has to be repeated in OIBL_BF16X3 (see OIBL_F16MX at the top).  The whole-backbone
 * entry points store their intermediate activations multiplied by 1/8 (exact: a power of two, undone in the
 * layer that writes `feat`), so that the bound is 8 x 65504 = 5.2e5 in activation units; the stand-alone
 * layer entries (oibl_conv3x3_nhwc*) store what they compute, bound 65504.  The other precisions leave the
 * word untouched.
 * The workspace holds the range flag, the two ping-pong activation buffers and the fp32 partial tiles of
 * the layers that run split-K: a layer whose tiling leaves most of the chip idle (small batches), and in
 * OIBL_F16MX also the tiles of a nearly empty LAST ROUND of a big layer (conv5_x at batch 32: 300 tiles on
 * 256 CUs), are contracted by several workgroups per tile and reduced in a fixed order — deterministic,
 * equal to the one-pass kernels up to fp32 association. */
size_t oibl_vgg16_workspace_bytes(int N, int H, int W, int precision);
int oibl_vgg16_conv5_forward(const float* x_nchw, int N, int H, int W,
                             const void* const* packed_w_host,
                             const float* const* bias_host, int precision, void* feat,
                             void* ws, size_t ws_bytes, void* stream);
/* Same, with two optional hipEvent_t handles (may be NULL) recorded on `stream` right before the
 * first and right after the last implicit-GEMM convolution (conv1_2 .. conv5_3, 12 launches of
 * the dominant kernel): bench.py brackets the kernel it reports a roofline for with these. */
int oibl_vgg16_conv5_forward_ev(const float* x_nchw, int N, int H, int W,
                                const void* const* packed_w_host,
                                const float* const* bias_host, int precision, void* feat,
                                void* ws, size_t ws_bytes, void* stream, void* ev_igemm_begin,
                                void* ev_igemm_end);

/* ---- NetVLAD + intra-norm + L2 ---------------------------------------------------- *
 * Replaces NetVLAD.forward (ibl/models/netvlad.py:44-61) and the normalisation that
 * every Embed* module applies to it (netvlad.py:78-80 / 100-102 / 202-204):
 *   xh = x / max(|x|_2, 1e-12)  per position; a = softmax_k(assign_w . xh);
 *   vlad[k][c] = sum_p a[p][k] * (xh[p][c] - centroids[k][c]).
 * feat [N][P][C] T (NHWC feature map), assign_w [K][C] fp32 (conv.weight[:, :, 0, 0]),
 * centroids [K][C] fp32.  K = 64, C = 512 are what the kernels are built for.
 *   vlad_raw  (optional, may be NULL): [N][K][C] fp32 un-normalised — NetVLAD.forward's
 *             return value.
 *   vlad_norm (optional, may be NULL): [N][K*C] fp32, intra-normalised per cluster then
 *             L2-normalised over K*C, k-major (index k*C + c).
 *   normalize_input: NetVLAD(normalize_input=...) (netvlad.py:46-47).                    */
size_t oibl_netvlad_workspace_bytes(int N, int P, int K, int C);
int oibl_netvlad_forward(const void* feat, int N, int P, int K, int C, int precision,
                         const float* assign_w, const float* centroids, int normalize_input,
                         float* vlad_raw, float* vlad_norm, void* ws, size_t ws_bytes,
                         void* stream);

/* ---- PCA-whitening projection + L2 ------------------------------------------------ *
 * Replaces EmbedNetPCA.pca_layer + F.normalize (netvlad.py:105-108) and PCA.infer
 * (ibl/pca.py:108-123):  y = normalize(W v + b).
 * v [N][D] fp32; w [d][D] T (row-major, i.e. pca_layer.weight[:, :, 0, 0], cast with
 * oibl_cast_f32_to_bf16 for OIBL_BF16); b [d] fp32; out [N][d] fp32.
 * D % 64 == 0, d % 128 == 0.  l2norm != 0 applies the final F.normalize.               */
size_t oibl_pca_workspace_bytes(int N, int D, int d, int precision);
int oibl_pca_forward(const float* v, int N, int D, const void* w, const float* b, int d,
                     int precision, int l2norm, float* out, void* ws, size_t ws_bytes,
                     void* stream);
/* The same projection in fp32 from a RE-PACKED copy of the weight (round 5): for 1 <= N <= 32 rows the weight
 * stream is the whole cost (537 MB for 32768 -> 4096), and the row-major matrix does not stream as an MFMA
 * operand.  oibl_pca_pack_weight writes w [d][D] fp32 once as 1 KB tiles ([32 output dims][8 k] in operand
 * lane order, the tiles of a 32-dim group consecutive along k; d * D floats, 16-byte aligned);
 * oibl_pca_forward_packed streams it (csrc/pca.hip, pca_stream_kernel).  Same arguments and workspace
 * (oibl_pca_workspace_bytes(N, D, d, OIBL_F32)) as oibl_pca_forward; results differ from it only by the
 * association of the fp32 sums.  oibl_pca_packed_supported: 1 where the packed form serves the shape
 * (N <= 32, d % 256 == 0, D % 8192 == 0), else the caller keeps oibl_pca_forward.                      */
int oibl_pca_pack_weight(const float* w, int D, int d, float* packed, void* stream);
int oibl_pca_packed_supported(int N, int D, int d);
int oibl_pca_forward_packed(const float* v, int N, int D, const float* w_packed, const float* b, int d,
                            int l2norm, float* out, void* ws, size_t ws_bytes, void* stream);

/* Row-wise L2 normalisation x / max(|x|_2, 1e-12)  (F.normalize(dim=-1), e.g. the extra
 * one in extract_cnn_feature, ibl/evaluators.py:29-33).  In place allowed. */
int oibl_l2_normalize_rows(const float* x, int N, int D, float* out, void* stream);
/* out[i] = normalize(xs[0][i] + xs[1][i] + ... + xs[S-1][i]), xs [S][N][D] fp32 (summed in that
 * order): fuses the per-scale descriptors of the multi-scale extraction (BASELINE.json configs[4];
 * no counterpart in the reference) into one unit-norm descriptor. */
int oibl_sum_l2_normalize(const float* xs, int S, int N, int D, float* out, void* stream);

/* ---- query x gallery squared-L2 --------------------------------------------------- *
 * Replaces the arithmetic of pairwise_distance (ibl/evaluators.py:122-129):
 *   dist[i][j] = |x_i|^2 + |y_j|^2 - 2 x_i . y_j      (no clamp, no sqrt)
 * x [m][d] fp32, y [n][d] fp32, dist [m][ldd] fp32 (ldd >= n).  d % 64 == 0.
 * OIBL_BF16 rounds x,y to bf16 for the dot product only (norms stay fp32).              */
size_t oibl_pairwise_workspace_bytes(int m, int n, int d, int precision);
int oibl_pairwise_sqdist(const float* x, int m, const float* y, int n, int d,
                         int precision, float* dist, size_t ldd, void* ws, size_t ws_bytes,
                         void* stream);

/* The same with descriptors stored as fp32, IEEE half or bf16 (OIBL_ST_*): the result is that of
 * oibl_pairwise_sqdist on the stored values widened to fp32 (norms from the widened values; in
 * OIBL_BF16 a bf16-stored operand is read in place, no copy).  x [m][d], y [n][d] of their types. */
size_t oibl_pairwise_st_workspace_bytes(int m, int n, int d, int precision, int x_st, int y_st);
int oibl_pairwise_sqdist_st(const void* x, int x_st, int m, const void* y, int y_st, int n, int d,
                            int precision, float* dist, size_t ldd, void* ws, size_t ws_bytes,
                            void* stream);

/* Fused distance + top-k: the k nearest gallery rows of every query without materialising the
 * [m][n] matrix — what Evaluator.evaluate / evaluate_all need from pairwise_distance + argsort
 * (ibl/evaluators.py:122-129, 143-159) when only ranks are consumed.
 *   out_val [m][k] fp32 ascending, out_idx [m][k] int32 = index_base + gallery row; ties broken
 *   by lowest index; n < k pads with (+inf, -1).  Same arithmetic as oibl_pairwise_sqdist, so the
 *   result equals oibl_row_topk applied to its matrix.
 * OIBL_BF16, large galleries: thresholds from a strided gallery sample, one filtered pass of the
 * distance kernel that appends only candidates, exact selection over the candidates.  A candidate
 * list that outgrows its capacity (possible only for degenerate data, e.g. thousands of duplicate
 * rows) sets *overflow (device int32, may be NULL) to 1 and leaves that call's outputs undefined:
 * repeat the call with exact = 1, which materialises distance tiles inside the workspace (always
 * used for OIBL_F32 and small problems; *overflow stays 0).                                   */
size_t oibl_sqdist_topk_workspace_bytes(int m, int n, int d, int k, int precision);
int oibl_sqdist_topk(const float* x, int m, const float* y, int n, int d, int k, int index_base,
                     int precision, int exact, float* out_val, int32_t* out_idx, int32_t* overflow,
                     void* ws, size_t ws_bytes, void* stream);
/* Storage-typed variant (see oibl_pairwise_sqdist_st). */
size_t oibl_sqdist_topk_st_workspace_bytes(int m, int n, int d, int k, int precision, int x_st,
                                           int y_st);
int oibl_sqdist_topk_st(const void* x, int x_st, int m, const void* y, int y_st, int n, int d, int k,
                        int index_base, int precision, int exact, float* out_val, int32_t* out_idx,
                        int32_t* overflow, void* ws, size_t ws_bytes, void* stream);

/* Prepared operands: a descriptor matrix that is matched many times (the resident gallery shard of
 * a retrieval service, the query set that meets every shard) pays the norm / operand pass once.
 *   oibl_match_operand_bytes : size of the operand buffer for (rows, d, precision, storage); 0 means
 *       the contraction reads the stored rows in place (OIBL_BF16 on bf16 storage, OIBL_F32 on fp32
 *       storage) and `operand` may be NULL.
 *   oibl_match_prepare : norms[rows] = fp32 squared norms of the rows widened to fp32 (the
 *       |x|^2 / |y|^2 terms of ibl/evaluators.py:127-128); operand = what the contraction reads:
 *       bf16 rows (OIBL_BF16), [32 hi | 32 lo] groups (OIBL_BF16X3), fp32 rows (OIBL_F32).
 *   oibl_sqdist_topk_prepared : oibl_sqdist_topk_st behind that pass, bit-identical results; xo / yo
 *       are the prepared operands (or the stored rows where operand_bytes is 0). */
size_t oibl_match_operand_bytes(int rows, int d, int precision, int st);
int oibl_match_prepare(const void* x, int x_st, int rows, int d, int precision, float* norms,
                       void* operand, void* stream);
size_t oibl_sqdist_topk_prepared_workspace_bytes(int m, int n, int d, int k, int precision);
int oibl_sqdist_topk_prepared(const void* xo, const float* xn, int m, const void* yo, const float* yn,
                              int n, int d, int k, int index_base, int precision, int exact,
                              float* out_val, int32_t* out_idx, int32_t* overflow, void* ws,
                              size_t ws_bytes, void* stream);

/* ---- "f16r": fp16 filter pass + exact rescoring ---------------------------------------------- *
 * The k nearest gallery rows per query by fp32 squared-L2 (pairwise_distance + the argsort prefix evaluate_all
 * reads, ibl/evaluators.py:105-130, 142-159) with EXACT lists at the cost of a 2-byte operand stream: every pair
 * is contracted once in fp16 (v_mfma_f32_32x32x16_f16, power-of-two row scales), kept if its distance could
 * belong to a member of the true top-k given a rigorous per-pair error bound (Cauchy-Schwarz on the fp16
 * rounding residuals + the fp32 accumulation slack d 2^-24), and the survivors near the k-th distance (k + a few
 * per query) are recomputed from the resident fp32 rows with fp64 accumulation.  out_val are
 * fl32((|x|^2 + |y|^2) - 2 x.y) with correctly rounded dot products; ties: lowest index first.  The lists do not
 * depend on the fp16 pass (csrc/match_f16r.h has the derivation).
 *   oibl_match_prepare_f16r : x [rows][d] fp32 (d % 64 == 0) -> norms[rows] (the fp32 squared norms of
 *       oibl_match_prepare), rows_f16 [rows][d] (IEEE binary16 of x 2^e, e per row), aux [rows][4] fp32 =
 *       {2^-e, |x| rounded up, |x - rows_f16 2^-e| rounded up, 0}.  The fp32 rows themselves are the fourth
 *       part of a prepared operand: they are read again by the rescoring.
 *   oibl_sqdist_topk_f16r : k <= 1024 (fused path: k <= 496); problems too small for the fused path, exact != 0,
 *       and the repeat after *overflow (device int32, may be NULL: a candidate list outgrew its capacity, or more
 *       than 32 (k <= 16; 2k + 32 otherwise) candidates sat within the bound of the k-th distance — near-duplicate
 *       galleries) take their member set from fp32 distance tiles + oibl_row_topk on the fp32 rows (what OIBL_F32
 *       runs: the oibl_f16r_members(k) nearest) and rescore it like the fused path does.                      */
int oibl_match_prepare_f16r(const float* x, int rows, int d, float* norms, float* aux, void* rows_f16,
                            void* stream);
/* The two stages of oibl_sqdist_topk_f16r on their own — for a caller that puts something between them: gallery-
 * sharded matching exchanges the FILTER lists first, takes the threshold from all shards' lists, and lets every rank
 * rescore only its members of the GLOBAL rescore set (openibl_amd/sharded.py: the rescoring work then divides by the
 * number of ranks instead of being repeated on each).
 *   oibl_f16r_members(k)        member slots per query: K2 = 32 (k <= 16), min(2k + 32, 1024) otherwise
 *   oibl_f16r_fused(m, n, d, k) 1 when the problem takes the fused path (else only oibl_sqdist_topk_f16r serves it)
 *   oibl_f16r_filter_select     stage 1 -> lval / lidx [m][K2]: filter distances and global indices of the candidates
 *       that can belong to the top-k of this gallery (any order; (+inf, -1) paddings); ymax_out (device, 2 floats, may
 *       be NULL): the gallery's largest |y| and largest fp16 residual norm — the pair bound of a query row i towards
 *       ANY row of it is eps_i = A_i ymax[0] + B_i ymax[1] + 1e-6 (|x_i|^2 + ymax[0]^2), A_i = 2 (r_i + g (n_i +
 *       r_i)), B_i = 2 (n_i + r_i)(1 + g), (n_i, r_i) = aux[i][1], aux[i][2], g = d 2^-24.  Workspace as
 *       oibl_sqdist_topk_f16r.
 *   oibl_f16r_rescore           stage 2: exact distances of the members listed in lidx [m][K2] (-1 = none), the k
 *       smallest (distance, index) per query, (+inf, -1) beyond the member count.                              */
int oibl_f16r_members(int k);
int oibl_f16r_fused(int m, int n, int d, int k);
int oibl_f16r_filter_select(const void* xh, const float* xaux, const float* xn, int m, const void* yh,
                            const float* yaux, const float* yn, int n, int d, int k, int index_base, float* lval,
                            int32_t* lidx, float* ymax_out, int32_t* overflow, void* ws, size_t ws_bytes,
                            void* stream);
int oibl_f16r_rescore(const float* xsrc, const float* xn, int m, const float* ysrc, const float* yn, int d, int k,
                      int index_base, const int32_t* lidx, float* out_val, int32_t* out_idx, void* stream);
/* Storage-typed f16r (descriptors stored as fp32, IEEE half or bf16: OIBL_ST_*, as oibl_pairwise_sqdist_st — the lists
 * are those of the stored values widened to fp32) and the whole ranked prefix the reference ever reads: k up to 496
 * on the fused path (spatial NMS looks at max(recall_topk) * 12 = 120 ranks, ibl/evaluators.py:152-153,
 * examples/test.py:130; the k-th filter distance of a list comes from a bisection instead of k register rounds,
 * csrc/match_f16r.h).  An fp16-stored row is its own fp16 image (residual 0): the filter bound of such a gallery is
 * the fp32 accumulation slack alone.  The rescoring reads the STORED rows (8 KB instead of 16 KB per 4096-d member).
 *   oibl_match_prepare_f16r_st        oibl_match_prepare_f16r on stored rows (norms of the widened rows)
 *   oibl_sqdist_topk_f16r_st          xsrc / ysrc = the stored rows; workspace from the _st query.  EVERY path ends in
 *       the same rescoring: the exact path (small problems, exact != 0, the repeat after *overflow) takes the
 *       oibl_f16r_members(k) nearest by fp32 distance tiles and rescores those, so values and tie order do not depend
 *       on the path that produced the member set.
 *   oibl_f16r_rescore_st              stage 2 with an explicit member-slot count (lidx [m][members], <= 1024)      */
int oibl_match_prepare_f16r_st(const void* x, int x_st, int rows, int d, float* norms, float* aux, void* rows_f16,
                               void* stream);
size_t oibl_sqdist_topk_f16r_st_workspace_bytes(int m, int n, int d, int k, int x_st, int y_st);
int oibl_sqdist_topk_f16r_st(const void* xh, const float* xaux, const float* xn, const void* xsrc, int x_st, int m,
                             const void* yh, const float* yaux, const float* yn, const void* ysrc, int y_st, int n,
                             int d, int k, int index_base, int exact, float* out_val, int32_t* out_idx,
                             int32_t* overflow, void* ws, size_t ws_bytes, void* stream);
int oibl_f16r_rescore_st(const void* xsrc, int x_st, const float* xn, int m, const void* ysrc, int y_st,
                         const float* yn, int d, int k, int members, int index_base, const int32_t* lidx,
                         float* out_val, int32_t* out_idx, void* stream);
/* Between the stages, sharded: thr [m] (device) = the k-th smallest filter distance over the lists of ALL shards,
 * ymax_all [shards][2] (device) = every shard's ymax_out; entries of lidx [m][K2] whose lval exceeds thr + 2 eps (eps
 * from the maxima over all shards) are set to -1. */
int oibl_f16r_keep_members(const float* lval, int32_t* lidx, int m, int k, const float* thr, const float* xn,
                           const float* xaux, const float* ymax_all, int shards, int d, void* stream);
size_t oibl_sqdist_topk_f16r_workspace_bytes(int m, int n, int d, int k);
int oibl_sqdist_topk_f16r(const void* xh, const float* xaux, const float* xn, const float* xsrc, int m,
                          const void* yh, const float* yaux, const float* yn, const float* ysrc, int n, int d,
                          int k, int index_base, int exact, float* out_val, int32_t* out_idx, int32_t* overflow,
                          void* ws, size_t ws_bytes, void* stream);

/* ---- top-k ------------------------------------------------------------------------ *
 * Replaces np.argsort(distmat, axis=1) (ibl/evaluators.py:143), of which evaluate_all only
 * consumes the first max(recall_topk) (or 12x that with nms) entries per row.
 * Per row of vals [m][ld] keep the k smallest, ascending, ties broken by lowest index:
 *   out_val [m][k] fp32, out_idx [m][k] int32.
 * idx_in == NULL: element j of a row has index index_base + j (gallery shard offset);
 * idx_in != NULL ([m][ld] int32): element j has index idx_in[row][j] (cross-shard merge).
 * 1 <= k <= 1024; if n < k the tail is filled with (+inf, -1).                          */
int oibl_row_topk(const float* vals, const int32_t* idx_in, int m, int n, size_t ld, int k,
                  int index_base, float* out_val, int32_t* out_idx, void* stream);

/* Full-row ranking: out_idx [m][n] int32 = stable ascending argsort of every row of vals [m][ld]
 * (ties: lowest index first), out_val [m][n] the sorted values (may be NULL).  Replaces
 * torch.argsort(distmat, dim=1) of the hard-negative mining samplers
 * (ibl/utils/data/sampler.py:46-54, 126-135) and serves evaluate_all when more than 1024 ranks per
 * query are needed (np.argsort, ibl/evaluators.py:143).  Per-row radix sort, any n. */
size_t oibl_row_argsort_workspace_bytes(int m, int n);
int oibl_row_argsort(const float* vals, int m, int n, size_t ld, int32_t* out_idx, float* out_val,
                     void* ws, size_t ws_bytes, void* stream);

/* ---- recall counting ---------------------------------------------------------------- *
 * Replaces the per-query Python loop of evaluate_all and spatial_nms
 * (ibl/evaluators.py:132-140, 149-160).  For every query, the rank (0-based, inside the
 * prediction list the reference builds) of the first prediction that is a ground-truth
 * neighbour, -1 if there is none; Recall@N = #(0 <= rank < N) / m for any N.
 *   topk_idx [m][k] int32 ranked gallery positions (-1 = padding), k <= 1024
 *   gt_offsets [m+1], gt_values [gt_offsets[m]] int32: ground truth in CSR form
 *   gallery_pids [n] int32 or NULL.  Non-NULL switches on spatial NMS: only the first
 *   nms_window (= 12 * max(recall_topk) in the reference) predictions count and a prediction
 *   whose pid occurred earlier in the list is dropped.                                        */
int oibl_first_hit_rank(const int32_t* topk_idx, int m, int k, const int32_t* gt_offsets,
                        const int32_t* gt_values, const int32_t* gallery_pids, int nms_window,
                        int32_t* out_rank, void* stream);

/* ---- k-means centroid initialisation (f4) -------------------------------------------- *
 * examples/cluster.py:110-115 runs scikit-learn's KMeans(num_clusters, max_iter = 100,
 * random_state = seed) on 50 000 L2-normalised conv5 descriptors.  The assignment step of a Lloyd
 * iteration is oibl_sqdist_topk(k = 1) against the centres; this is the update step:
 *   centers[c] <- mean of the rows x[i] with labels[i] == c   (fp64 accumulation in point order,
 *                 correctly rounded fp32 result), untouched when no row carries the label;
 *   counts[c]  <- number of such rows (the caller relocates empty clusters as scikit-learn does).
 *   x [n][d] fp32, labels [n] int32 in [0, num_clusters), centers [num_clusters][d] fp32.          */
int oibl_cluster_means(const float* x, const int32_t* labels, int n, int d, int num_clusters,
                       float* centers, int32_t* counts, void* stream);

/* n 32-bit words src -> dst, by a kernel (one workgroup) on `stream`: dst may be pinned (host-coherent)
 * memory mapped into the device — how the f16mx range flag reaches the host behind a replayed graph without a
 * DMA-engine copy that would queue behind the next batch's input transfer. */
int oibl_copy_words(const void* src, void* dst, int n, void* stream);

/* ---- diagnostics ------------------------------------------------------------------ */

/* Elapsed milliseconds between two recorded hipEvent_t (HOST pointer ms_host) — also when the events
 * were recorded by event nodes of a replayed hipGraph, which torch's Event.elapsed_time refuses: how
 * bench.py measures the span of the matrix-core launches inside the replayed forward. */
int oibl_event_elapsed_ms(void* ev_start, void* ev_stop, float* ms_host);

/* Plain C = A . B^T on the shared MFMA GEMM core (used by tests to validate the core and
 * its fragment layout independently of the operators above):
 * A [M][K] T, B [N][K] T, C [M][ldc] fp32; K % (128/sizeof(T)) == 0, N % 64 == 0. */
int oibl_gemm_nt(const void* A, int M, const void* B, int N, int K, int precision, float* C,
                 size_t ldc, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OPENIBL_AMD_H */
