"""Drop-in for the reference's train/loss.py: `LossOutput`, `DPOLoss`, `create_loss_strategy` with the same
argument meaning and error behaviour (train/loss.py:15-22,25-121,124-155), computed by the fused HIP reduction
(videogpa_amd/csrc/dpo_loss.hip)."""
from dataclasses import dataclass

import torch
import torch.nn as nn

from . import ops


@dataclass
class LossOutput:
    loss: torch.Tensor
    reward_margin: torch.Tensor
    winner_reward: torch.Tensor
    loser_reward: torch.Tensor
    accuracy: torch.Tensor


class DPOLoss(nn.Module):
    def __init__(self, beta: float = 500.0, label_smoothing: float = 0.0, loss_type: str = "sigmoid"):
        super().__init__()
        self.beta = beta
        self.label_smoothing = label_smoothing
        self.loss_type = loss_type

    def forward(self, v_win, v_lose, v_win_ref, v_lose_ref, v_win_target, v_lose_target) -> LossOutput:
        if self.loss_type not in ("sigmoid", "hinge"):
            raise ValueError(f"Unknown loss type: {self.loss_type}")
        # under bf16 autocast the reference subtracts in bf16 before the fp32 square (train/loss.py:73-77)
        round_diff = v_win.dtype == torch.bfloat16 and torch.is_autocast_enabled()
        loss, margin, wr, lr, acc, _ = ops.dpo_loss(v_win, v_lose, v_win_ref, v_lose_ref, v_win_target, v_lose_target,
                                                    beta=self.beta, label_smoothing=self.label_smoothing,
                                                    loss_type=self.loss_type, round_diff=round_diff)
        return LossOutput(loss=loss, reward_margin=margin.detach(), winner_reward=wr.detach(), loser_reward=lr.detach(),
                          accuracy=acc.detach())


class SFTLoss(nn.Module):
    def forward(self, v_pred, v_target, **kwargs):
        z = torch.tensor(0.0)
        return LossOutput(loss=torch.nn.functional.mse_loss(v_pred, v_target), reward_margin=z, winner_reward=z,
                          loser_reward=z, accuracy=z)


def create_loss_strategy(strategy: str = "dpo", beta: float = 1.0, label_smoothing: float = 0.0) -> nn.Module:
    if strategy == "dpo":
        return DPOLoss(beta=beta, label_smoothing=label_smoothing)
    elif strategy == "sft":
        return SFTLoss()
    raise ValueError(f"Unknown strategy: {strategy}")
