"""Loss bookkeeping of the reconstruction (reference quant/reconstruction_util.py).

The arithmetic (lp_loss, rounding regulariser, their gradients) lives in the K13/K12 kernels and is
driven by engine.recon; these classes keep the reference's names, constructor signature, iteration
counter, temperature schedule and logging cadence."""
from __future__ import annotations

import logging
from enum import Enum

logger = logging.getLogger(__name__)

RLOSS = Enum("RLOSS", ("RELAXATION", "MSE", "FISHER_DIAG", "FISHER_FULL", "NONE"))
print_freq = 2000


class LinearTempDecay:
    """b: start_b during warm-up, then linear to end_b (the reference's docstring says cosine; its code
    is linear, :194-198)."""

    def __init__(self, t_max: int, rel_start_decay: float = 0.2, start_b: int = 10, end_b: int = 2) -> None:
        self.t_max = t_max
        self.start_decay = rel_start_decay * t_max
        self.start_b, self.end_b = start_b, end_b

    def __call__(self, t) -> float:
        if t < self.start_decay:
            return self.start_b
        rel_t = (t - self.start_decay) / (self.t_max - self.start_decay)
        return self.end_b + (self.start_b - self.end_b) * max(0.0, (1 - rel_t))


class LossFunc:
    """total = rec + round; `rec`/`round` are produced on the device by the reconstruction unit, this
    object tracks count / b / warm-up and logs like the reference (:50-91)."""

    def __init__(self, o, round_loss: RLOSS = RLOSS.RELAXATION, w: float = 1.0, rec_loss: RLOSS = RLOSS.MSE,
                 max_count: int = 2000, b_range: tuple = (10, 2), decay_start: float = 0.0, warmup: float = 0.0,
                 p: float = 2.0) -> None:
        if rec_loss not in (RLOSS.MSE, RLOSS.FISHER_DIAG, RLOSS.FISHER_FULL):
            raise ValueError("Not supported reconstruction loss function: {}".format(rec_loss))
        if p != 2.0:
            raise NotImplementedError("lp_loss with p != 2 is not on the hot path")
        self.o, self.round_loss, self.w, self.rec_loss, self.p = o, round_loss, w, rec_loss, p
        self.loss_start = max_count * warmup
        self.temp_decay = LinearTempDecay(t_max=max_count, rel_start_decay=warmup + (1 - warmup) * decay_start,
                                          start_b=b_range[0], end_b=b_range[1])
        self.count = 0

    def tick(self):
        """-> (b, regulariser_active) for the next iteration (count is 1-based like the reference)."""
        self.count += 1
        b = self.temp_decay(self.count)
        active = not (self.count < self.loss_start or self.round_loss == RLOSS.NONE)
        return (b if active else 0.0), active

    def __call__(self, pred, tgt, grad=None):
        """The reference's call signature (reconstruction_util.py:36-91), VALUE ONLY: total = rec + round as a device scalar, count advanced, the
        reference's log line every `print_freq` calls.  layer_ / block_reconstruction do not come through here -- their units compute the same
        two terms and their gradients in the fused kernels (K12 / K13) and use tick() / log() -- but code that drove the class like the
        reference does keeps working.  rec: tfmq_recon_loss (p = 2) / tfmq_fisher_loss on the device; the rounding term is the reference's
        expression over `wqtizer.get_soft_tgt()` of every eligible layer of `o`."""
        import torch
        from tfmq_dm_amd import ops
        self.count += 1
        pred, tgt = pred.detach().float().contiguous(), tgt.detach().float().contiguous()
        if self.rec_loss == RLOSS.MSE:
            rec = ops.recon_loss(pred, tgt, pred.numel() // pred.shape[1], want_grad=False)[0][0]
        else:
            if grad is None:
                raise ValueError("LossFunc: the Fisher-weighted losses need `grad`")
            mode = fisher_mode(self.rec_loss)
            denom = pred.numel() // pred.shape[1]        # (FISHER_FULL: the kernel takes its own mean / 100)
            fg = grad.detach().float().contiguous()
            # the kernel takes the cached weights |dL/d out| (DIAG squares them, FULL uses them as they are) -- reference :53-59
            rec = ops.fisher_loss(pred, tgt, fg.abs().contiguous(), mode, denom, want_grad=False)[0][0]
        b = self.temp_decay(self.count)
        rnd = torch.zeros((), device=pred.device)
        if self.count < self.loss_start or self.round_loss == RLOSS.NONE:
            b = 0
        elif self.round_loss == RLOSS.RELAXATION:
            from .quant_layer import QuantLayer
            layers = [self.o] if isinstance(self.o, QuantLayer) else [m for _, m in self.o.named_modules()
                                                                      if isinstance(m, QuantLayer) and not m.quant_emb and not m.ignore_recon]
            from .quant_layer import QMODE
            for m in layers:
                term = (1 - ((m.wqtizer.get_soft_tgt() - 0.5).abs() * 2).pow(b)).sum()
                split = int(getattr(m, "split", 0) or 0)
                if not isinstance(self.o, QuantLayer) and split != 0 and QMODE.QDIFF.value in m.aq_mode and getattr(m, "wqtizer1", None) is not None:
                    # a QDIFF-split layer inside a block (reference :72-79): its two weight quantizers' terms weighted by their share of the input channels
                    cin = m.w.shape[1]
                    term1 = (1 - ((m.wqtizer1.get_soft_tgt() - 0.5).abs() * 2).pow(b)).sum()
                    term = (term * split + term1 * (cin - split)) / cin
                rnd = rnd + self.w * term
        else:
            raise NotImplementedError
        total = rec + rnd
        self.log(total, rec, rnd, b)
        return total

    def log(self, total, rec, rnd, b, rank0: bool = True):
        if self.count % print_freq == 0 and rank0:
            logger.info("Total loss:\t{:.8f} (rec:{:.8f}, round:{:.8f})\tb={:.2f}\tcount={}".format(
                float(total), float(rec), float(rnd), b, self.count))


def fisher_mode(rec_loss: RLOSS):
    """RLOSS -> the device loss kernel's mode (ops.FISHER_DIAG / FISHER_FULL), None for the plain lp_loss."""
    from tfmq_dm_amd import ops
    return {RLOSS.MSE: None, RLOSS.FISHER_DIAG: ops.FISHER_DIAG, RLOSS.FISHER_FULL: ops.FISHER_FULL}[rec_loss]


class LossFuncTimeEmbedding(LossFunc):
    """TIB variant: `rec` is the sum of lp_loss over the projections (reference :94-173)."""

    def __call__(self, preds, tgts):
        """Value only, the reference's signature (:117-173): sum of lp_loss over the (prediction, target) pairs of the TIB's projections + the
        rounding term over the block's QuantLayers, walked exactly as the reference walks them (named_modules(), then emb_layers / temb_projs
        again -- what is registered twice is counted twice there too)."""
        import torch
        from tfmq_dm_amd import ops
        from .quant_layer import QuantLayer
        if self.rec_loss != RLOSS.MSE:
            raise ValueError("Not supported reconstruction loss function: {}".format(self.rec_loss))
        self.count += 1
        rec = torch.zeros((), device=preds[0].device)
        for pred, tgt in zip(preds, tgts):
            pr, tg = pred.detach().float().contiguous(), tgt.detach().float().contiguous()
            rec = rec + ops.recon_loss(pr, tg, pr.numel() // pr.shape[1], want_grad=False)[0][0]
        b = self.temp_decay(self.count)
        rnd = torch.zeros((), device=preds[0].device)
        if self.count < self.loss_start or self.round_loss == RLOSS.NONE:
            b = 0
        elif self.round_loss == RLOSS.RELAXATION:
            def term(m):
                rv = m.wqtizer.get_soft_tgt()
                return self.w * (1 - ((rv - 0.5).abs() * 2).pow(b)).sum()
            for _, m in self.o.named_modules():
                if isinstance(m, QuantLayer) and not m.ignore_recon:
                    rnd = rnd + term(m)
            if hasattr(self.o, "emb_layers"):
                for emb in self.o.emb_layers:
                    for _, m in emb.named_modules():
                        if isinstance(m, QuantLayer) and not m.ignore_recon:
                            rnd = rnd + term(m)
            else:
                for m in getattr(self.o, "temb_projs", ()):
                    if not m.ignore_recon:
                        rnd = rnd + term(m)
        else:
            raise NotImplementedError
        total = rec + rnd
        self.log(total, rec, rnd, b)
        return total
