"""Host-side weight preparation for the three-term bf16 kernels (mirx.model._*_split3): every layout is checked
by rebuilding the fp32 weight from its three terms (h + m + l == w exactly for normal fp32 values: 3 x 8 mantissa
bits) at the position the kernel's header documents."""
import torch

import mirx.model as mm


def _sum3(t, term_axis):
    return t.float().sum(dim=term_axis)


def test_split3_weights_linear_layout_and_padding():
    torch.manual_seed(0)
    w = torch.randn(200, 48)                                   # 200 outputs -> padded to 256
    w3 = mm._split3_weights(w)
    assert w3.shape == (2, 3, 3, 128, 16) and w3.dtype == torch.bfloat16
    full = _sum3(w3, 2)                                        # [n / 128, k / 16, 128, 16]
    rebuilt = full.permute(0, 2, 1, 3).reshape(256, 48)
    assert torch.equal(rebuilt[:200], w)
    assert torch.count_nonzero(rebuilt[200:]) == 0             # zero rows of the last output tile


def test_split3_terms_are_exact_and_ordered():
    torch.manual_seed(1)
    w = torch.randn(128, 16) * torch.logspace(-6, 3, 128)[:, None]
    w3 = mm._split3_weights(w)[0, 0]                           # [3, 128, 16]
    h, m, lo = (w3[i].float() for i in range(3))
    assert torch.equal(h + m + lo, w)
    assert bool((m.abs() <= h.abs() * 2.0 ** -8 + 1e-45).all()) and bool((lo.abs() <= h.abs() * 2.0 ** -16 + 1e-45).all())


def test_winograd_split3_layout():
    torch.manual_seed(2)
    w = torch.randn(32, 128, 3, 3)
    u = mm._winograd_weights(w)                                # [stage 16][xi][c % 8][oc] fp32
    u3 = mm._winograd_weights_split3(w)                        # [stage 8][xi][term][oc][c % 16]
    assert u3.shape == (8, 16, 3, 32, 16)
    full = _sum3(u3, 2)                                        # [8, 16, 32, 16]
    ref = u.reshape(8, 2, 16, 8, 32).permute(0, 2, 4, 1, 3).reshape(8, 16, 32, 16)
    assert torch.equal(full, ref)


def test_conv3x3_direct_split3_layout():
    torch.manual_seed(3)
    w = torch.randn(32, 128, 3, 3)
    d3 = mm._conv3x3_weights_split3(w)                         # [stage][tap][term][oc][c % 16]
    assert d3.shape == (8, 9, 3, 32, 16)
    full = _sum3(d3, 2)
    for st, tap, oc, c in ((0, 0, 0, 0), (3, 4, 17, 9), (7, 8, 31, 15)):
        assert float(full[st, tap, oc, c]) == float(w[oc, 16 * st + c, tap // 3, tap % 3])


def test_stem_split3_layout():
    torch.manual_seed(4)
    w = torch.randn(64, 3, 7, 7)
    w3 = mm._stem_weights_split3(w)                            # [block][step][term][oc][k]
    assert w3.shape == (2, 11, 3, 32, 16)
    full = _sum3(w3, 2)
    for oc in (0, 33, 63):
        for rho in (0, 6, 7, 20):
            c, ky = divmod(rho, 7)
            s, g = divmod(rho, 2)
            row = full[oc // 32, s, oc % 32, 8 * g: 8 * g + 8]
            assert torch.equal(row[0:4], w[oc, c, ky, 0::2]) and torch.equal(row[4:7], w[oc, c, ky, 1::2])
            assert float(row[7]) == 0.0                        # kx = 7 does not exist
    assert torch.count_nonzero(full[:, 10, :, 8:]) == 0        # row 21 does not exist
