"""How far does the REFERENCE's own quantised forward move when every quantizer input carries rounding-level noise?

The engine's exact-fp32 mode (TFMQ_EXACT_FP=1) computes every layer in fp32 but in its own summation order, so the tensor reaching a
quantizer differs from the reference's by a few fp32 ulps.  A value sitting on a rounding boundary then lands in the neighbouring bin,
that changes the layer's output by delta_a * w, which moves more values across boundaries downstream: an avalanche.  The oracle -- pinned
bit-for-bit to the reference on these fixtures (tests/test_oracle_r03_fixtures.py) -- measures the size of that avalanche on the same
model, tables and input: it is the resolution limit of ANY end-to-end comparison that is not bit-identical in summation order, and the
yardstick the exact-mode bars are set against.  Test infrastructure only (imports the oracle)."""
import torch

import tfmq_oracle as O


class NoisySpec(O.QuantSpec):
    """QuantSpec whose activation-quantizer inputs are multiplied by (1 + rel * N(0,1)) first."""

    def __init__(self, *a, rel=1e-6, seed=0, **k):
        super().__init__(*a, **k)
        self.rel, self.gen = rel, torch.Generator().manual_seed(seed)

    def act(self, name, x):
        return super().act(name, x * (1.0 + self.rel * torch.randn(x.shape, generator=self.gen)))


def avalanche(forward, spec_kwargs, rel=1e-6, seeds=(0, 1, 2)):
    """forward(spec) -> eps.  Returns (clean eps, [(eps rel-L2 change, fraction of activation bins moved)] per seed)."""
    clean = O.QuantSpec(**spec_kwargs)
    clean.trace = {}
    with torch.no_grad():
        e0 = forward(clean)
    out = []
    for s in seeds:
        noisy = NoisySpec(rel=rel, seed=s, **spec_kwargs)
        noisy.trace = {}
        with torch.no_grad():
            e1 = forward(noisy)
        common = [n for n in clean.trace if n in noisy.trace]
        moved = sum(int((clean.trace[n] != noisy.trace[n]).sum()) for n in common)
        total = sum(clean.trace[n].numel() for n in common)
        out.append((float((e1 - e0).norm() / e0.norm()), moved / max(total, 1)))
    return e0, out


def tie_distance(x: torch.Tensor, delta: float, moved: torch.Tensor) -> float:
    """Largest distance (in bins) from a rounding boundary among the elements of x whose bin differs from the reference's: a flip caused
    by summation order alone has x / delta within rounding error of k + 0.5."""
    u = (x.double() / float(delta))[moved]
    if u.numel() == 0:
        return 0.0
    return float(((u - torch.floor(u)) - 0.5).abs().max())


def first_divergence(rates: dict, ties: dict):
    """rates / ties: quantizer name -> fraction of moved bins / tie_distance, in the reference's call order.
    Returns (name of the first quantizer with a moved bin | None, number of bit-identical quantizers before it, its tie distance)."""
    clean = 0
    for n, r in rates.items():
        if r > 0.0:
            return n, clean, ties[n]
        clean += 1
    return None, clean, 0.0


def engine_bins_vs_trace(eng, args, qtable, act_names, trace, dev="cuda:0"):
    """Run the engine once in its 'record' calibration mode and compare the bins of every observed quantizer input with the oracle's
    trace (name -> bins, the reference's call order).  Returns (rates, ties, overall moved fraction)."""
    import tfmq_dm_amd.ops as ops
    eng.set_calibration("record", 0)
    eng.forward(*args)
    eng.set_calibration(None)
    qid = {n: i for i, n in enumerate(act_names)}
    rates, ties, flips, total = {}, {}, 0, 0
    for n in trace:
        i = qid.get(n)
        if i is None or i not in eng.observed:
            continue
        xe = eng.observed[i].float().contiguous()
        be = (ops.quantize_act(xe, ops.qsel(qtable[:, i:i + 1].contiguous().to(dev))).to(torch.int32) + 128).cpu()
        bo = trace[n].to(torch.int32)
        if bo.dim() == 4:
            bo = bo.permute(0, 2, 3, 1)
        if bo.numel() == 4 * be.numel():        # the engine observes the stride-2 conv's input after the fused 2x subsampling
            bo = bo[:, ::2, ::2, :]
        moved = (be - bo.reshape(be.shape)) != 0
        rates[n] = float(moved.float().mean())
        ties[n] = tie_distance(xe.cpu(), float(qtable[0, i, 0]), moved)
        flips += int(moved.sum())
        total += moved.numel()
    return rates, ties, flips / max(total, 1)
