"""CPU restatement of the index logic of the f16mx stem's producers (openibl_amd/csrc/conv.hip,
vgg_stem_x3_kernel<true>): conv1_1's 3 x 3 x 3 window travels as nine ROWS (c, ky) of three consecutive
pixels — the lower lane half owns rows 0-4, the upper half rows 5-8 (and row 8 once more under zero weights) —
and K slot idx = 8 s + e of a half is element kx = idx % 3 of its row idx / 3.  At the image's left / right
edge the 12-byte fetch starts one pixel later / earlier and the elements are moved into place.  This test
replays exactly that arithmetic in numpy against a plain zero-padded convolution (ibl/models/vgg.py:40-42)."""
import numpy as np


def _kernel_conv1_1(x, w, ch, y, xx):
    C, H, W = x.shape
    flat = x.reshape(-1)
    plane = H * W
    total = 0.0
    for half in (0, 1):
        edge = 1 if xx == 0 else 2 if xx + 1 >= W else 0
        shift = 0 if edge == 1 else -2 if edge == 2 else -1
        xv = [0.0] * 16
        for i in range(5):
            r = (i + 5 if i + 5 < 9 else 8) if half else i
            ky = r % 3
            ok = (y > 0) if ky == 0 else (y + 1 < H) if ky == 2 else True
            base = (r // 3) * plane + (y + ky - 1) * W + xx + shift
            a, b, c = (flat[base], flat[base + 1], flat[base + 2]) if ok else (0.0, 0.0, 0.0)
            if edge == 1:
                a, b, c = 0.0, a, b
            elif edge == 2:
                a, b, c = b, c, 0.0
            xv[3 * i: 3 * i + 3] = [a, b, c]
        for s in range(2):
            for e in range(8):
                idx = 8 * s + e
                if idx < (12 if half else 15):
                    k = (idx // 3 + (5 if half else 0)) * 3 + idx % 3
                    total += w[ch].reshape(-1)[k] * xv[idx]
    return total


def test_row_gather_reproduces_the_padded_convolution():
    rng = np.random.default_rng(3)
    H, W = 5, 7
    x = rng.standard_normal((3, H, W))
    w = rng.standard_normal((2, 3, 3, 3))
    pad = np.pad(x, ((0, 0), (1, 1), (1, 1)))
    for y in range(H):
        for xx in range(W):
            want = float((w[1] * pad[:, y:y + 3, xx:xx + 3]).sum())
            assert abs(_kernel_conv1_1(x, w, 1, y, xx) - want) < 1e-12, (y, xx)


def test_no_fetch_leaves_the_tensor():
    """Every 12-byte fetch of an in-image pixel stays inside [0, 3 H W) for W >= 3 (oibl_vgg16_stem_mx's
    requirement): rows above / below the image are not fetched at all."""
    for H, W in ((2, 3), (4, 5), (3, 8)):
        plane = H * W
        for y in range(H):
            for xx in range(W):
                shift = 0 if xx == 0 else -2 if xx + 1 >= W else -1
                for r in range(9):
                    ky = r % 3
                    if (ky == 0 and y == 0) or (ky == 2 and y + 1 >= H):
                        continue
                    base = (r // 3) * plane + (y + ky - 1) * W + xx + shift
                    assert 0 <= base and base + 2 < 3 * plane, (H, W, y, xx, r)
                    assert (base % W) + 2 <= W - 1, (H, W, y, xx, r)      # and inside ONE image row
