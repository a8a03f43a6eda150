"""K15b (round 5): k_gemm_bx3 -- the bf16x3 reconstruction GEMM with the hi / lo split done once per block on the way into LDS (128 x 128
tiles, K-step 32; csrc/gemm_f32_mfma.hip).  It serves tfmq_gemm_f32 whenever the operand mode is bf16x3 (the reconstruction iterations'
default, engine/recon.py) and both operands take 16-byte loads.  Against float64: every product is a b ~ hi hi' + hi lo' + lo hi' with
relative error <= 2^-16, accumulated in fp32 -- the bound below is that of tests/test_recon_precision_gpu.py (which also runs the reference's
400-iteration loss curve, fixture F8b, through this kernel).  The reference runs these products as fp32 torch.matmul / F.conv2d backward
(quant/reconstruction.py:63-78,182-198 via autograd)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def ops():
    import tfmq_dm_amd.ops as ops
    return ops


def _ref(A, B, ta, tb):
    return (A.T if ta else A).double() @ (B.T if tb else B).double()


# (M, N, K, ta, tb): loader modes (A, B) = (1, 1) k-contiguous both, (1, 2), (2, 1), (2, 2); ragged M / N (not multiples of 128 / 32),
# K tails (K % 32 = 4 ... 28), one K-step only, a K long enough for split-K, shapes of SD units (conv forward, dgrad, wgrad)
SHAPES = [(512, 320, 2880, False, True), (1000, 200, 644, False, True), (129, 72, 36, False, True),
          (512, 2880, 320, False, False), (300, 96, 640, False, False), (260, 132, 100, False, False),
          (2880, 320, 4096, True, False), (324, 196, 1028, True, False), (320, 320, 32768, True, False),
          (640, 328, 520, True, True), (132, 68, 92, True, True)]


@pytest.mark.parametrize("M,N,K,ta,tb", SHAPES)
def test_bx3_gemm_vs_float64(ops, M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M * 3 + N * 5 + K)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    ref = _ref(A, B, ta, tb)
    scale = float(ref.abs().max())
    Ad, Bd = A.to(DEV), B.to(DEV)
    with ops.gemm_precision("bf16x3"):
        C = ops.gemm(Ad, Bd, trans_a=ta, trans_b=tb)
        C2 = ops.gemm(Ad, Bd, trans_a=ta, trans_b=tb)
    assert torch.equal(C, C2)                                   # deterministic (split-K slices are added in a fixed order)
    err = float((C.cpu().double() - ref).abs().max()) / scale
    exact = ops.gemm(Ad, Bd, trans_a=ta, trans_b=tb)
    err32 = float((exact.cpu().double() - ref).abs().max()) / scale
    print(f"{M}x{N}x{K} ta={int(ta)} tb={int(tb)}: max-normalised error bf16x3 {err:.2e} (exact fp32 products {err32:.2e})")
    assert err <= 2e-5
    # a transposed or shifted tile would be wrong by O(1): every row and column must also agree on its own scale
    rowerr = (C.cpu().double() - ref).abs().amax(dim=1) / ref.abs().amax(dim=1).clamp_min(1e-30)
    colerr = (C.cpu().double() - ref).abs().amax(dim=0) / ref.abs().amax(dim=0).clamp_min(1e-30)
    assert float(rowerr.max()) <= 2e-4 and float(colerr.max()) <= 2e-4


def test_bx3_epilogue_bias_rowadd_residual_accumulate(ops):
    M, N, K, rpi = 384, 192, 256, 96
    g = torch.Generator().manual_seed(11)
    A, B = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g)
    bias, rowadd = torch.randn(N, generator=g), torch.randn(M // rpi, N, generator=g)
    res, C0 = torch.randn(M, N, generator=g), torch.randn(M, N, generator=g)
    ref = 0.25 * (A.double() @ B.double().T) + bias.double() + rowadd.double().repeat_interleave(rpi, dim=0) + res.double()
    with ops.gemm_precision("bf16x3"):
        y = ops.gemm(A.to(DEV), B.to(DEV), trans_b=True, alpha=0.25, bias=bias.to(DEV), rowadd=rowadd.to(DEV), rows_per_img=rpi, residual=res.to(DEV))
        out = C0.to(DEV).clone()
        ops.gemm(A.to(DEV), B.to(DEV), trans_b=True, alpha=0.25, bias=bias.to(DEV), rowadd=rowadd.to(DEV), rows_per_img=rpi, residual=res.to(DEV),
                 out=out, accumulate=True)
    scale = float(ref.abs().max())
    assert float((y.cpu().double() - ref).abs().max()) / scale <= 2e-5
    assert float((out.cpu().double() - ref - C0.double()).abs().max()) / scale <= 2e-5


def test_bx3_batched_items_are_independent_of_the_batch(ops):
    """A batched launch equals its items launched alone, bit for bit (the slicing -- hence the summation order -- is decided per item)."""
    nb, M, N, K = 3, 256, 160, 1536
    g = torch.Generator().manual_seed(3)
    A, B = torch.randn(nb, M, K, generator=g).to(DEV), torch.randn(nb, N, K, generator=g).to(DEV)
    with ops.gemm_precision("bf16x3"):
        Cb = ops.gemm(A, B, trans_b=True)
        for z in range(nb):
            Cz = ops.gemm(A[z].contiguous(), B[z].contiguous(), trans_b=True)
            assert torch.equal(Cb[z], Cz)
    ref = torch.einsum("bmk,bnk->bmn", A.cpu().double(), B.cpu().double())
    assert float((Cb.cpu().double() - ref).abs().max()) / float(ref.abs().max()) <= 2e-5


def test_bx3_large_values_and_exact_zeros(ops):
    """hi + lo reproduces 16 of the 24 significand bits of each operand: operands that ARE bf16 values multiply exactly (lo = 0), zero rows /
    columns stay exactly zero, and magnitudes far from 1 keep the relative bound (the split is scale free)."""
    M, N, K = 256, 256, 512
    g = torch.Generator().manual_seed(5)
    A = torch.randn(M, K, generator=g).bfloat16().float()
    B = torch.randn(N, K, generator=g).bfloat16().float()
    A[7] = 0.0
    B[200] = 0.0
    with ops.gemm_precision("bf16x3"):
        C = ops.gemm(A.to(DEV), B.to(DEV), trans_b=True).cpu()
    ref = A.double() @ B.double().T
    assert float(C[7].abs().max()) == 0.0 and float(C[:, 200].abs().max()) == 0.0
    assert float((C.double() - ref).abs().max()) / float(ref.abs().max()) <= 2e-6       # products exact, fp32 accumulation only
    As, Bs = (A * 3.1e4).to(DEV), (B * 2.7e-5).to(DEV)
    A2 = torch.randn(M, K, generator=g) * 3.1e4
    B2 = torch.randn(N, K, generator=g) * 2.7e-5
    with ops.gemm_precision("bf16x3"):
        C2 = ops.gemm(A2.to(DEV), B2.to(DEV), trans_b=True).cpu()
    ref2 = A2.double() @ B2.double().T
    assert float((C2.double() - ref2).abs().max()) / float(ref2.abs().max()) <= 2e-5
    del As, Bs


def test_bx3_forced_for_short_reductions_in_a_fresh_process():
    """The launcher gives k_gemm_bx3 the launches with K >= 1024 (where it measured faster); the short-K shapes of the table above reach it
    only under TFMQ_GEMM_BX3=2, which is read once per process: the whole table again in a child process with the kernel forced."""
    for mode in ("2", "3"):       # 2: the 128 x 128 form for every K >= 64; 3: the 128 x 64 short-K form below K = 1024
        env = dict(os.environ, TFMQ_GEMM_BX3=mode)
        r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "vs_float64 or epilogue or batched"],
                           env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
        assert r.returncode == 0, (mode, r.stdout[-3000:] + r.stderr[-2000:])
