"""Diffusion-DPO loss, CPU oracle.  PINNED against train/loss.py (tests/golden/dpo_loss.pt).

Follows train/loss.py:53-121:
  e_*      = mean_{F,C,H,W}((v - tgt)^2)                       (:73-77)
  logits   = beta * ((e_ref_w - e_w) - (e_ref_l - e_l))         (:82-93)
  sigmoid  : mean(-logsigmoid(logits))                           (:105)
             or BCE-with-logits against 1 - label_smoothing      (:97-103)
  hinge    : mean(relu(1 - logits))                              (:108)
  reward_margin = mean(e_l - e_w); winner/loser reward = -e      (:86-88,116-119)
  accuracy = mean(e_w < e_l)                                     (:113)
"""
import torch
import torch.nn.functional as F


def dpo_loss(v_win, v_lose, v_win_ref, v_lose_ref, tgt_win, tgt_lose, beta=1.0,
             label_smoothing=0.0, loss_type="sigmoid"):
    dims = list(range(1, v_win.dim()))
    e_w = (v_win - tgt_win).pow(2).mean(dim=dims)
    e_l = (v_lose - tgt_lose).pow(2).mean(dim=dims)
    e_rw = (v_win_ref - tgt_win).pow(2).mean(dim=dims)
    e_rl = (v_lose_ref - tgt_lose).pow(2).mean(dim=dims)
    logits = beta * ((e_rw - e_w) - (e_rl - e_l))
    if loss_type == "sigmoid":
        if label_smoothing > 0:
            loss = F.binary_cross_entropy_with_logits(logits, torch.full_like(logits, 1.0 - label_smoothing))
        else:
            loss = -F.logsigmoid(logits).mean()
    elif loss_type == "hinge":
        loss = F.relu(1.0 - logits).mean()
    else:
        raise ValueError(f"Unknown loss type: {loss_type}")
    wr, lr = -e_w, -e_l
    return {
        "loss": loss,
        "reward_margin": (wr - lr).mean(),
        "winner_reward": wr.mean(),
        "loser_reward": lr.mean(),
        "accuracy": (wr > lr).float().mean(),
        "logits": logits,
    }
