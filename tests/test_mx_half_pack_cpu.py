"""What the one-convert pack of the fused f16mx stem (mx_pack_half: lo through fp16(lo * 2^11) on hi's block
scale) costs against the two-convert pack of every other f16mx producer (mx_pack_line: lo from fp32), on the
host emulation of both (tests/helpers/mx_emul.py): the e2m3 code of lo moves by at most ONE step, in a small
fraction of the elements, and the value a line carries (hi + q6(lo)) stays inside the format's own bound."""
import torch

from helpers import mx_emul


def test_double_rounded_lo_is_one_code_step_at_most():
    g = torch.Generator().manual_seed(11)
    x = torch.relu(torch.randn((4096, 64), generator=g)) * torch.logspace(-2, 2, 64)[None, :] * 3.0
    x[0, :4] = torch.tensor([0.0, 65504.0, 1e-6, 300.0])
    hi_a, hi6_a, lo6_a = mx_emul.split(x)
    hi_b, hi6_b, lo6_b = mx_emul.split_half_pack(x)
    assert torch.equal(hi_a, hi_b) and torch.equal(hi6_a, hi6_b)
    # one e2m3 step of the lo block: 1/8 of its scale below 2, up to 1/2 of it in the top binade
    xb = x.reshape(-1, 2, 32)
    bh = mx_emul.scale_byte(xb.clamp(-65504, 65504).half().float().abs().amax(-1, keepdim=True))
    sl = torch.pow(2.0, (bh - 11 - 127).double()).expand(-1, -1, 32).reshape(x.shape)
    d = (lo6_a - lo6_b).abs()
    assert (d <= 0.5 * sl + 1e-300).all()
    frac = float((d > 0).double().mean())
    print(f"lo codes that differ between the two packs: {frac:.4%}")
    assert frac < 0.01
    # the carried value: |v - (hi + q6(lo))| within 2^-14 of the group's largest element, as for mx_pack_line
    gmax = xb.abs().amax(-1, keepdim=True).expand(-1, -1, 32).reshape(x.shape).double()
    ok = x.abs() <= 65504
    assert ((hi_b + lo6_b - x.double()).abs()[ok] <= 2.0 ** -14 * gmax[ok] + 1e-30).all()
