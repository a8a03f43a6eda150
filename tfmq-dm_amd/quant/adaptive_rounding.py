"""AdaRound quantizer (reference quant/adaptive_rounding.py) on the K12 kernels."""
from __future__ import annotations

from enum import Enum

import torch
import torch.nn as nn

from tfmq_dm_amd import ops
from .quant_layer import UniformAffineQuantizer

RMODE = Enum("RMODE", ("LEARNED_ROUND_SIGMOID", "NEAREST", "NEAREST_STE", "STOCHASTIC", "LEARNED_HARD_SIGMOID"))


class AdaRoundQuantizer(nn.Module):
    """Learned rounding: w_hat = delta*(clamp(floor(w/delta) + h(alpha) + zp, 0, L-1) - zp) with
    h = clamp(sigmoid(alpha)*1.2 - 0.1, 0, 1) while `soft_tgt`, else [alpha >= 0] (reference :12-74).
    delta / zero_point are frozen copies of the uniform quantizer's; only alpha is trained."""

    def __init__(self, uaqtizer: UniformAffineQuantizer, w: torch.Tensor, rmode: RMODE = RMODE.LEARNED_ROUND_SIGMOID) -> None:
        super().__init__()
        if rmode != RMODE.LEARNED_HARD_SIGMOID:
            raise NotImplementedError("only RMODE.LEARNED_HARD_SIGMOID is used by the reference's calibration")
        self.level = uaqtizer.level
        self.symmetric = uaqtizer.symmetric
        self.delta = uaqtizer.delta
        self.zero_point = uaqtizer.zero_point
        self.rmode = rmode
        self.soft_tgt = False
        self.gamma, self.zeta = -0.1, 1.1
        self.alpha = None
        self._version_ = 0
        self.init_alpha(x=w.clone())

    def init_alpha(self, x: torch.Tensor) -> None:
        """alpha = -log(1.2 / (w/delta - floor(w/delta) + 0.1) - 1) (reference :31-38)."""
        self.delta = self.delta.detach().to(x.device) if torch.is_tensor(self.delta) else self.delta
        self.alpha = nn.Parameter(ops.adaround_init(x.detach().float().contiguous(), self.delta.float().contiguous()))

    def get_soft_tgt(self) -> torch.Tensor:
        """h(alpha) = clamp(sigmoid(alpha) (zeta - gamma) + gamma, 0, 1) (reference :40-41).  The reconstruction kernels evaluate it inside
        adaround_soft_fwd / adaround_bwd_adam; this accessor serves callers of the reference's API (LossFunc.__call__)."""
        return torch.clamp(torch.sigmoid(self.alpha.detach()) * (self.zeta - self.gamma) + self.gamma, 0, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        d = self.delta.detach().float().contiguous()
        z = self.zero_point.detach().float().contiguous()
        x = x.detach().float().contiguous()
        return ops.adaround_soft_fwd(x, self.alpha.detach().contiguous(), d, z, self.level, hard=not self.soft_tgt)

    def extra_repr(self) -> str:
        return f"level={self.level}, symmetric={self.symmetric}, rmode={self.rmode}"
