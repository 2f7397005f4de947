"""uint8 input of the bf16x3 / f16mx stems (csrc/conv.hip, vgg_stem_x3_kernel<MX, U8>), the parts that can be
checked without a GPU:

1. Normalize as ONE fma, v = u * a_c + b_c, against the loader's three rounded fp32 operations
   (u / 255 - mean) / std (ibl/utils/data/__init__.py:40-41) — exhaustively over the 768 (channel, byte) pairs:
   within 2^-16 absolute (the rounding the loader's own intermediate carries), identical bf16 hi parts, a few lo
   parts one unit apart.
2. The byte plumbing of the producers, restated instruction for instruction (v_alignbyte_b32, v_perm_b32 with the
   kernel's selectors): three aligned 12-byte loads -> the 16 K slots of each lane half in (ky, kx, c) order, for
   every byte alignment, and the 27-bit tap-validity mask of border pixels."""
import numpy as np
import torch

from openibl_amd import ops

MEAN, STD = ops.REF_MEAN, ops.REF_STD


def _split(v):
    t = torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32))
    hi = t.to(torch.bfloat16).float()
    lo = (t - hi).to(torch.bfloat16).float()
    return hi.numpy(), lo.numpy()


def test_fma_normalisation_is_within_an_ulp_of_the_loaders_arithmetic():
    u = np.arange(256, dtype=np.float32)
    worst, lo_diff = 0.0, 0
    for c in range(3):
        m, s = np.float32(MEAN[c]), np.float32(STD[c])
        ref = ((u / np.float32(255.0)) - m) / s                          # three rounded fp32 operations
        a = np.float32(1.0 / (255.0 * float(s)))
        b = np.float32(-float(m) / float(s))
        fast = (u.astype(np.float64) * float(a) + float(b)).astype(np.float32)   # one fma: exact product, one rounding
        worst = max(worst, float(np.abs(ref.astype(np.float64) - fast.astype(np.float64)).max()))
        # the loader rounds u / 255 - mean at magnitude <= 1 (ulp 6e-8) and then divides by 1 / 255: ITS result
        # carries up to 255 x 6e-8 = 1.5e-5 of rounding; the fma's single rounding is the more exact of the two
        assert (np.abs(ref.astype(np.float64) - fast.astype(np.float64)) <= 2.0 ** -16).all()
        h1, l1 = _split(ref)
        h2, l2 = _split(fast)
        assert np.array_equal(h1, h2)                                       # the main operand of conv1_1: identical
        d = l1 != l2
        lo_diff += int(d.sum())
        # a differing lo part moved by what v moved (<= 2^-16) plus at most one unit of its own bf16 grid
        assert (np.abs(l1[d] - l2[d]) <= 2.0 ** -16 + np.maximum(np.abs(l1[d]), np.abs(l2[d])) * 2.0 ** -7).all()
    print(f"fma Normalize vs the loader's: max |diff| {worst:.2e}, {lo_diff} of 768 lo parts differ")
    assert worst < 2e-5 and lo_diff < 64


def _alignbyte(hi, lo, sh):
    return ((int(hi) << 32 | int(lo)) >> (8 * sh)) & 0xFFFFFFFF


def _perm(s0, s1, sel):
    src = int(s0) << 32 | int(s1)
    out = 0
    for i in range(4):
        k = (sel >> (8 * i)) & 0xFF
        assert k < 8
        out |= ((src >> (8 * k)) & 0xFF) << (8 * i)
    return out


def _lane_slots(img, y, x, half):
    """What one producer lane (halo pixel (y, x) inside the image) hands conv1_1: 16 bytes in slot order, and
    the 27-bit validity mask, following vgg_stem_x3_kernel<., true> step by step."""
    H, W, _ = img.shape
    flat = np.concatenate([img.reshape(-1), np.full(16, 77, dtype=np.uint8)])   # (whatever follows the tensor)

    records = (img.size + 3) & ~3         # the descriptor covers the tensor rounded up to whole dwords

    def load96(off):                      # raw_buffer_load_b96 at a dword-aligned offset: a dword that is not
        if off < 0:                       # wholly inside num_records reads as zero
            return [0, 0, 0]
        out = []
        for i in range(3):
            o = off + 4 * i
            b = flat[o:o + 4].astype(np.uint64) if o + 4 <= records else np.zeros(4, dtype=np.uint64)
            out.append(int(b[0] | b[1] << 8 | b[2] << 16 | b[3] << 24))
        return out

    b0 = ((y - 1) * W + (x - 1)) * 3
    xsh = b0 & 3
    ya, yc, xa, xc = y > 0, y + 1 < H, x > 0, x + 1 < W
    rows = (0x1FF if ya else 0) | 0x3FE00 | (0x7FC0000 if yc else 0)
    cols = (0x0040201 * 7 if xa else 0) | (0x0040201 * 7 << 3) | (0x0040201 * 7 << 6 if xc else 0)
    mask = rows & cols
    xr = []
    for ky in range(3):
        ok = ya if ky == 0 else yc if ky == 2 else True
        off = (b0 + ky * 3 * W) & ~3
        if ok and off < 0:                # the tensor's very first pixel: fetch from 0, one dword late
            d = load96(0)
            xr.append([0, d[0], d[1]])
        else:
            xr.append(load96(off) if ok else [0, 0, 0])
    w3 = (3 * W) & 3
    w = []
    for ky in range(3):
        sh = (xsh + ky * w3) & 3
        w.append([_alignbyte(xr[ky][1], xr[ky][0], sh), _alignbyte(xr[ky][2], xr[ky][1], sh), xr[ky][2] >> (8 * sh)])
    a2 = _perm(w[1][0], w[0][2], 0x06050400)
    a3 = _alignbyte(w[1][1], w[1][0], 3)
    t = _perm(w[1][2], w[1][1], 0x00000403)
    b0_ = _perm(w[2][0], t, 0x05040100)
    b1 = _alignbyte(w[2][1], w[2][0], 2)
    b2 = _alignbyte(w[2][2], w[2][1], 2) & 0x00FFFFFF
    dw = [b0_, b1, b2, 0] if half else [w[0][0], w[0][1], a2, a3]
    slots = [(dw[j >> 2] >> (8 * (j & 3))) & 0xFF for j in range(16)]
    return slots, (mask >> (16 if half else 0)) & 0xFFFF


def test_window_bytes_reach_their_k_slots_for_every_alignment_and_border():
    rng = np.random.default_rng(7)
    for (H, W) in ((9, 13), (8, 32), (5, 6), (12, 7), (3, 5)):   # 3 * W mod 4 = 3, 0, 2, 1; sizes 3, 0, 2, 0, 1 mod 4
        img = rng.integers(1, 256, size=(H, W, 3), dtype=np.uint8)   # no zero bytes: a zeroed tap is visible
        for y in range(H):
            for x in range(W):
                for half in (0, 1):
                    slots, mask = _lane_slots(img, y, x, half)
                    for j in range(16):
                        idx = j + 16 * half
                        if idx >= 27:
                            continue                               # zero weights: any finite value
                        ky, kx, c = idx // 9, (idx % 9) // 3, idx % 3
                        yy, xx = y + ky - 1, x + kx - 1
                        inside = 0 <= yy < H and 0 <= xx < W
                        assert ((mask >> j) & 1) == int(inside), (H, W, y, x, half, j)
                        if inside:
                            assert slots[j] == int(img[yy, xx, c]), (H, W, y, x, half, j)
