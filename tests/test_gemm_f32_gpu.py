"""K15 strided fp32 GEMM of the reconstruction units (tfmq_gemm_f32): FMA tile, fp32-MFMA tiles and the split-K path
(weight gradients dW = X^T dY with K = batch * tokens; context-side gradients of cross attention, 77 x 40 with
K = 4096), against a float64 reference.  The reference framework runs these as torch.matmul / F.linear backward in
fp32 (quant/reconstruction.py:180-200 via autograd); fp32 products with fp32 accumulation differ from float64 by
O(sqrt(K) * 2^-24) relative to the row / column norms -- that is the tolerance written below."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def ops():
    import tfmq_dm_amd.ops as ops
    return ops


def _tol(A64, B64, K):
    # |sum a_k b_k - fl(...)| <= ~K eps * sum |a_k b_k| worst case; typical sqrt(K): use 8 sqrt(K) eps |a|.|b|
    return 8.0 * (K ** 0.5) * 2.0 ** -24 * float((A64.abs() @ B64.abs()).max())


@pytest.mark.parametrize("M,N,K,ta,tb", [(320, 320, 32768, True, False),     # dW, 9 tiles -> split-K
                                          (320, 2560, 8192, True, False),
                                          (130, 70, 4099, False, True),       # ragged K, not a multiple of 4
                                          (77, 40, 4096, True, False),        # skinny (below the 96-row rule)
                                          (512, 384, 96, False, False),       # plain MFMA tiles
                                          (40, 16, 50, False, True)])         # FMA tile
def test_gemm_shapes(ops, M, N, K, ta, tb):
    g = torch.Generator().manual_seed(M * 7 + N)
    A = torch.randn((K, M) if ta else (M, K), generator=g)
    B = torch.randn((N, K) if tb else (K, N), generator=g)
    bias = torch.randn(N, generator=g)
    C0 = torch.randn(M, N, generator=g)
    A64 = (A.T if ta else A).double()
    B64 = (B.T if tb else B).double()
    ref = 0.5 * (A64 @ B64) + bias.double()
    y = ops.gemm(A.to(DEV), B.to(DEV), trans_a=ta, trans_b=tb, alpha=0.5, bias=bias.to(DEV))
    tol = _tol(A64, B64, K)
    assert (y.cpu().double() - ref).abs().max() <= tol
    # accumulate into an existing C, twice: deterministic (slices are added in a fixed order)
    out1 = C0.to(DEV).clone()
    ops.gemm(A.to(DEV), B.to(DEV), trans_a=ta, trans_b=tb, alpha=0.5, bias=bias.to(DEV), out=out1, accumulate=True)
    out2 = C0.to(DEV).clone()
    ops.gemm(A.to(DEV), B.to(DEV), trans_a=ta, trans_b=tb, alpha=0.5, bias=bias.to(DEV), out=out2, accumulate=True)
    assert torch.equal(out1, out2)
    assert (out1.cpu().double() - (ref + C0.double())).abs().max() <= tol


def test_gemm_strided_batched_skinny_splitk(ops):
    # dK of cross attention on the [B, T, heads*d] layout: dK[b,h] (77 x 40) = dS[b,h]^T (77 x 4096) Q[b,h] (4096 x 40)
    Bb, heads, T, L, d = 2, 8, 4096, 77, 40
    g = torch.Generator().manual_seed(1)
    dS = torch.randn(Bb * heads, T, L, generator=g)
    Q = torch.randn(Bb, T, heads * d, generator=g)
    out = torch.zeros(Bb, L, heads * d, device=DEV)
    dSd, Qd = dS.to(DEV), Q.to(DEV)
    for b in range(Bb):     # batch over heads: A(m=l, k=t) = dS[b*heads+h][t][l]; B(k=t, n) = Q[b][t][h*d + n]
        ops.gemm_strided(dSd, b * heads * T * L, 1, L, T * L, Qd, b * T * heads * d, heads * d, 1, d,
                         out, b * L * heads * d, heads * d, d, L, d, T, batch=heads)
    ref = torch.einsum("bhtl,bthd->blhd", dS.view(Bb, heads, T, L).double(), Q.view(Bb, T, heads, d).double()).reshape(Bb, L, heads * d)
    tol = 8.0 * (T ** 0.5) * 2.0 ** -24 * float(torch.einsum("bhtl,bthd->blhd", dS.view(Bb, heads, T, L).abs().double(),
                                                              Q.view(Bb, T, heads, d).abs().double()).max())
    assert (out.cpu().double() - ref).abs().max() <= tol
