"""Optimizer side of the DPO step, MI355X-first (replaces Lightning's AdamW / cosine-warmup / clip / DDP reducer:
train/CogVideoX-5B/03_train.py:208-213,257-266).

All trainable (LoRA) parameters live as views into ONE flat fp32 buffer, and so do their gradients and Adam
moments.  That makes the data-parallel exchange a single RCCL all-reduce of the flat gradient (264 MB at r=64; one
message keeps every xGMI link busy instead of 11 latency-bound 25 MB buckets), and the optimizer a two-launch
fused HIP kernel (global-norm + clip + AdamW) with no host synchronisation.
"""
import math
import os

import torch
import torch.distributed as dist

from . import ops


def cosine_schedule_with_warmup(step, warmup_steps, total_steps, num_cycles=0.5):
    """transformers.get_cosine_schedule_with_warmup lr multiplier (train/CogVideoX-5B/03_train.py:210-212)."""
    if step < warmup_steps:
        return float(step) / float(max(1, warmup_steps))
    progress = float(step - warmup_steps) / float(max(1, total_steps - warmup_steps))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * float(num_cycles) * 2.0 * progress)))


class FlatParams:
    """Re-homes `params` (fp32, requires_grad) into one flat buffer; .grad of each is a view of one flat grad."""

    TAIL = 4   # (loss, reward_margin, reward_accuracy, micro-step count) summed over ranks by the gradient all-reduce

    def __init__(self, params):
        self.params = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        if dt != torch.float32:
            raise ValueError("adapter parameters are kept in fp32")
        # 4-element alignment per tensor keeps float4 paths aligned
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += (p.numel() + 3) // 4 * 4
        self.numel = n
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        # gradient exchange buffer = [flat gradient | TAIL logged scalars]: the scalars the reference reduces with
        # `self.log(..., sync_dist=True)` (train/CogVideoX-5B/03_train.py:164-173) ride the ONE all-reduce of the step
        self.buf = torch.zeros(n + self.TAIL, dtype=dt, device=dev)
        self.grad = self.buf[:n]
        self.tail = self.buf[n:]
        for p, o in zip(self.params, self.offsets):
            self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + p.numel()].view_as(p)
            p.grad = self.grad[o:o + p.numel()].view_as(p)

    def zero_grad(self):
        self.buf.zero_()
        for p, o in zip(self.params, self.offsets):   # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)


class FlatAdamW:
    """AdamW + clip-by-global-norm + (optional) data-parallel mean all-reduce over a FlatParams buffer."""

    def __init__(self, flat: FlatParams, lr=5e-6, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0,
                 warmup_steps=500, total_steps=10000, process_group=None):
        self.flat = flat
        self.base_lr, self.betas, self.eps, self.wd = lr, betas, eps, weight_decay
        self.max_grad_norm = max_grad_norm
        self.warmup_steps, self.total_steps = warmup_steps, total_steps
        self.exp_avg = torch.zeros_like(flat.flat)
        self.exp_avg_sq = torch.zeros_like(flat.flat)
        self.step_count = 0
        self.pg = process_group
        self.total_norm = torch.zeros(1, dtype=torch.float32, device=flat.flat.device)
        self._comm_stream = None

    @property
    def lr(self):
        # LambdaLR semantics: the lr used for optimizer step k (0-based) is base * lambda(k)
        return self.base_lr * cosine_schedule_with_warmup(self.step_count, self.warmup_steps, self.total_steps)

    def world(self):
        if dist.is_available() and dist.is_initialized():
            return dist.get_world_size(self.pg)
        return 1

    def all_reduce_grads(self):
        """SUM all-reduce of the flat gradient on a side stream; the 1/world mean is folded into the step."""
        if self.world() == 1 and os.environ.get("VGPA_FORCE_DIST") != "1":
            return None
        g = self.flat.buf          # gradient + logged-scalar tail in one message
        if g.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                work = dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
            return work
        return dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def step(self, pending=None):
        if pending is not None:
            pending.wait()
            if self._comm_stream is not None:
                torch.cuda.current_stream().wait_stream(self._comm_stream)
        scale = 1.0 / self.world()
        lr = self.lr
        self.step_count += 1
        self._apply_update(lr, scale)
        ops.bump_adapter_epoch()      # the kernel wrote the adapters through raw pointers: refresh their cached bf16 copies (ops.LoraExt)
        return lr

    def _apply_update(self, lr, scale):
        """global-norm clip + AdamW on the flat buffers (two HIP launches, clip coefficient stays on the device)."""
        f = self.flat
        if not f.flat.is_cuda:
            raise RuntimeError("FlatAdamW.step needs the HIP kernels (GPU tensors); there is no CPU fallback")
        if self.max_grad_norm and self.max_grad_norm > 0:
            ops.grad_norm(f.grad, scale, out=self.total_norm)
        ops.adamw_step(f.flat, f.grad, self.exp_avg, self.exp_avg_sq, lr, self.betas[0], self.betas[1], self.eps, self.wd,
                       self.step_count, scale, self.max_grad_norm or 0.0, self.total_norm)

    # ------------------------------------------------------------------ resume state (fit.py checkpoints)
    def state_dict(self):
        return {"exp_avg": self.exp_avg, "exp_avg_sq": self.exp_avg_sq, "step_count": self.step_count, "param": self.flat.flat}

    def load_state_dict(self, sd):
        self.exp_avg.copy_(sd["exp_avg"])
        self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        self.flat.flat.copy_(sd["param"])
        self.step_count = int(sd["step_count"])
        ops.bump_adapter_epoch()

    def zero_grad(self):
        self.flat.zero_grad()
