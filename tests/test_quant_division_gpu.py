"""The activation quantizer's quotient: q = clamp(rint(x / delta) + zp, 0, 255) with TRUE IEEE division
(quant/quant_layer.py:225 `torch.round(x / delta)`).  The kernels compute RN(x / delta) as reciprocal + two FMA residual
corrections (common.hpp div_rn_f); this file pins it bit for bit against the float32 division of torch on the CPU: random
inputs, inputs a few ulps around every rounding boundary (k + 0.5) * delta -- where a quotient off by one ulp would flip
the code -- and divisors whose significand is all ones (the case the theorem excludes, routed to the real division)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import tfmq_dm_amd.ops as ops
    return ops


def _codes_cpu(x, delta, zp):
    q = torch.round(x / torch.tensor(delta, dtype=torch.float32)) + zp        # float32 IEEE division, half-to-even
    return torch.clamp(q, 0, 255)


def _codes_gpu(ops, x, delta, zp):
    qt = torch.tensor([[float(delta), float(zp)]], dtype=torch.float32, device=DEV)
    return ops.quantize_act(x.to(DEV), ops.qsel(qt)).cpu().float() + 128


def _deltas():
    rng = np.random.default_rng(5)
    ds = list(np.exp(rng.uniform(np.log(1e-4), np.log(3.0), 40)).astype(np.float32))
    ds += [np.float32(0.1), np.float32(1.0 / 3.0), np.float32(0.0078125), np.float32(2.0 ** -7 * 1.9999999)]
    # all-ones significands (excluded by the theorem -> real division in the kernel)
    ds += [np.frombuffer(np.uint32(0x3DFFFFFF).tobytes(), dtype=np.float32)[0], np.frombuffer(np.uint32(0x3C7FFFFF).tobytes(), dtype=np.float32)[0]]
    return [float(d) for d in ds]


def test_quotient_is_the_ieee_division(ops):
    g = torch.Generator().manual_seed(11)
    bad = 0
    for delta in _deltas():
        zp = float(torch.randint(0, 256, (1,), generator=g))
        d32 = np.float32(delta)
        # every rounding boundary of the clamp range, +- 0..3 ulps
        k = np.arange(-zp - 2, 258 - zp, dtype=np.float64) + 0.5
        centre = (k * float(d32)).astype(np.float32)
        pts = [centre]
        for _ in range(3):
            pts.append(np.nextafter(pts[-1], np.float32(np.inf)))
        lo = centre
        for _ in range(3):
            lo = np.nextafter(lo, np.float32(-np.inf))
            pts.append(lo)
        edge = torch.from_numpy(np.concatenate(pts))
        rnd = (torch.rand(1 << 18, generator=g) - 0.5) * float(d32) * 300.0
        tiny = torch.tensor([0.0, 1e-38, -1e-38, 1e-45, 3e-39, float(d32) * 1e-30], dtype=torch.float32)
        x = torch.cat([edge, rnd, tiny]).float()
        ref = _codes_cpu(x, d32, zp)
        got = _codes_gpu(ops, x, delta, zp)
        bad += int((ref != got).sum())
    assert bad == 0
