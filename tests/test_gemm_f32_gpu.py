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


@pytest.mark.parametrize("B,H,T,L,d", [(2, 8, 256, 256, 160), (2, 8, 256, 77, 160), (3, 4, 64, 40, 24), (1, 2, 96, 1030, 72)])
def test_heads_batch_is_the_per_head_loop(ops, B, H, T, L, d):
    """tfmq_gemm_f32_heads (two-level batch: all heads of an attention product in one launch) against one launch per head,
    bit for bit (the split-K slicing is decided per batch item), for the four operand layouts the attention backward
    uses, and against float64."""
    g = torch.Generator().manual_seed(T + L)
    C = H * d
    q = torch.randn(B, T, C, generator=g).to(DEV)
    k = torch.randn(B, L, C, generator=g).to(DEV)
    P = torch.randn(B, H, T, L, generator=g).to(DEV)
    # S_h = q_h k_h^T
    S1 = torch.empty(B, H, T, L, device=DEV)
    S2 = torch.empty_like(S1)
    for h in range(H):
        ops.gemm_strided(q, h * d, C, 1, T * C, k, h * d, 1, C, L * C, S1, h * T * L, L, H * T * L, T, L, d, B)
    ops.gemm_strided(q, 0, C, 1, T * C, k, 0, 1, C, L * C, S2, 0, L, H * T * L, T, L, d, B, heads=H, hsa=d, hsb=d, hsc=T * L)
    assert torch.equal(S1, S2)
    qh = q.double().reshape(B, T, H, d).permute(0, 2, 1, 3)
    kh = k.double().reshape(B, L, H, d).permute(0, 2, 1, 3)
    ref = qh @ kh.transpose(-1, -2)
    assert float((S2.double() - ref).abs().max()) <= 8.0 * d ** 0.5 * 2.0 ** -24 * float((qh.abs() @ kh.abs().transpose(-1, -2)).max())
    # o_h = P_h v_h  and  dK_h = P_h^T q_h (transposed A operand, long reduction over T)
    o1 = torch.empty(B, T, C, device=DEV)
    o2 = torch.empty_like(o1)
    dk1 = torch.empty(B, L, C, device=DEV)
    dk2 = torch.empty_like(dk1)
    for h in range(H):
        ops.gemm_strided(P, h * T * L, L, 1, H * T * L, k, h * d, C, 1, L * C, o1, h * d, C, T * C, T, d, L, B)
        ops.gemm_strided(P, h * T * L, 1, L, H * T * L, q, h * d, C, 1, T * C, dk1, h * d, C, L * C, L, d, T, B)
    ops.gemm_strided(P, 0, L, 1, H * T * L, k, 0, C, 1, L * C, o2, 0, C, T * C, T, d, L, B, heads=H, hsa=T * L, hsb=d, hsc=d)
    ops.gemm_strided(P, 0, 1, L, H * T * L, q, 0, C, 1, T * C, dk2, 0, C, L * C, L, d, T, B, heads=H, hsa=T * L, hsb=d, hsc=d)
    assert torch.equal(o1, o2)
    assert torch.equal(dk1, dk2)
    refo = (P.double() @ kh).permute(0, 2, 1, 3).reshape(B, T, C)
    assert float((o2.double() - refo).abs().max()) <= 8.0 * L ** 0.5 * 2.0 ** -24 * float((P.double().abs() @ kh.abs()).max())
