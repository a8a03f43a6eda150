"""Conv-epilogue GroupNorm statistics + split GroupNorm (finalize + elementwise apply) vs torch."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

import tfmq_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous().to(DEV)


def nchw(y):
    return y.permute(0, 3, 1, 2).contiguous().cpu()


@pytest.mark.parametrize("B,H,W,cin,cout,k,narrow", [(3, 16, 16, 64, 128, 3, False), (4, 8, 8, 64, 96, 3, False),
                                                    (8, 4, 4, 64, 256, 1, False), (2, 32, 32, 64, 32, 3, True),
                                                    (5, 8, 4, 128, 64, 1, False)])
def test_conv_stats_and_split_groupnorm(B, H, W, cin, cout, k, narrow):
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    import tfmq_dm_amd.ops as ops
    gen = torch.Generator().manual_seed(B * 100 + cout)
    x = torch.randn(B, cin, H, W, generator=gen)
    w = torch.randn(cout, cin, k, k, generator=gen) * 0.05
    b = torch.randn(cout, generator=gen) * 0.1
    res = torch.randn(B, cout, H, W, generator=gen)
    rowadd = torch.randn(B, cout, generator=gen)
    ref = F.conv2d(x, w, b, padding=k // 2) + res + rowadd[:, :, None, None]
    pf = ops.pack_w_f16(w.to(DEV), b.to(DEV))
    pad = (k // 2,) * 4
    y = ops.conv2d_f16(nhwc(x), pf, pad=pad, residual=nhwc(res), rowadd=rowadd.to(DEV), want_stats=True)
    assert float((nchw(y) - ref).abs().max() / ref.abs().max()) <= 2e-3
    st, seg = y._tfmq_stats
    assert seg == ops.stats_segment(H * W) and st.shape == (B * H * W // seg, cout, 2)
    yy = y.reshape(B * H * W // seg, seg, cout).cpu().double()
    np.testing.assert_allclose(st[..., 0].cpu().numpy(), yy.sum(1).numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(st[..., 1].cpu().numpy(), (yy * yy).sum(1).numpy(), rtol=1e-5, atol=1e-4)
    # split GroupNorm (+SiLU +quant) on that tensor == fused kernel == torch
    gamma, beta = torch.randn(cout, generator=gen), torch.randn(cout, generator=gen) * 0.3
    yref = O.swish(F.group_norm(nchw(y), 32, gamma, beta, 1e-6))
    ad, az = O.minmax(yref, 256)
    qt = torch.tensor([[float(ad), float(az)]], device=DEV)
    yq, yf, _ = ops.groupnorm(y, gamma.to(DEV), beta.to(DEV), 1e-6, True, ops.qsel(qt), want_f32=True)
    assert float((nchw(yf) - yref).abs().max() / yref.abs().max()) <= 1e-5
    assert torch.equal(nchw(yq.float()) + 128, O.quant_index(nchw(yf), ad, az, 256))
    # concat of two conv outputs with statistics
    y2 = ops.conv2d_f16(nhwc(x), pf, pad=pad, want_stats=True)
    g2, b2 = torch.randn(2 * cout, generator=gen), torch.randn(2 * cout, generator=gen)
    cref = F.group_norm(torch.cat([nchw(y), nchw(y2)], 1), 32, g2, b2, 1e-6)
    _, cf, xcat = ops.groupnorm(y, g2.to(DEV), b2.to(DEV), 1e-6, False, None, x2=y2, want_f32=True, want_cat=True)
    assert float((nchw(cf) - cref).abs().max() / cref.abs().max()) <= 1e-5
    assert torch.equal(xcat, torch.cat([y, y2], -1))
