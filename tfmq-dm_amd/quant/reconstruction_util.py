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
