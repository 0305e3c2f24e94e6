"""Optimizer side of the DPO step, MI355X-first (replaces Lightning's AdamW / cosine-warmup / clip / DDP reducer:
train/CogVideoX-5B/03_train.py:208-213,257-266).

All trainable (LoRA) parameters live as views into ONE flat fp32 buffer, and so do their gradients and Adam
moments.  That makes the data-parallel exchange a single RCCL all-reduce of the flat gradient (264 MB at r=64; one
message keeps every xGMI link busy instead of 11 latency-bound 25 MB buckets), and the optimizer a two-launch
fused HIP kernel (global-norm + clip + AdamW) with no host synchronisation.

The exchange is selectable (VGPA_DP_COLLECTIVE, or FlatAdamW(collective=...)):
  all_reduce  (default) one dist.all_reduce(SUM) of [gradient | logged scalars];
  rs_ag       reduce_scatter_tensor + all_gather_into_tensor of the same buffer (on point-to-point xGMI the two halves each run
              over all 7 links, where a ring all-reduce is bound by one); every element is still the sum of the same `world` values, so the
              result is bit-identical to all_reduce whenever the backend adds the ranks' contributions in one order (world 2: always;
              tests/test_dp_gloo.py) and equal to fp32 summation order otherwise.
Every exchange is bracketed by events on the communication stream and its wait by events on the compute stream: comm_report() gives the
mean collective duration and the time the compute stream really stalled for it -- what bench.py prints at N > 1.
"""
import collections
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
    PAD_WORLD = 64   # largest world size the reduce-scatter form can split the message for

    def message(self, world=1):
        """the exchange buffer as a view whose length is a multiple of `world` (the slack behind the tail is zero on every rank and stays zero)"""
        m = -(-self.buf.numel() // world) * world
        if world > self.PAD_WORLD or m > self.storage.numel():
            raise ValueError(f"world size {world} > {self.PAD_WORLD}: FlatParams.PAD_WORLD limits the reduce-scatter split")
        return self.storage[:m]

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
        # (+ up to PAD_WORLD - 1 zero elements of slack, so that a reduce-scatter can cut the message into `world` equal pieces)
        self.storage = torch.zeros(n + self.TAIL + self.PAD_WORLD - 1, dtype=dt, device=dev)
        self.buf = self.storage[:n + self.TAIL]
        self.grad = self.buf[:n]
        self.tail = self.buf[n:]
        for p, o in zip(self.params, self.offsets):
            self.flat[o:o + p.numel()].copy_(p.data.reshape(-1))
            p.data = self.flat[o:o + p.numel()].view_as(p)
            p.grad = self.grad[o:o + p.numel()].view_as(p)

    def zero_grad(self):
        self.storage.zero_()
        for p, o in zip(self.params, self.offsets):   # re-attach if something replaced .grad
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view_as(p)


class FlatAdamW:
    """AdamW + clip-by-global-norm + (optional) data-parallel mean all-reduce over a FlatParams buffer."""

    COMM_EVENT_RING = 256

    def __init__(self, flat: FlatParams, lr=5e-6, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01, max_grad_norm=1.0,
                 warmup_steps=500, total_steps=10000, process_group=None, collective=None):
        self.flat = flat
        self.collective = (collective or os.environ.get("VGPA_DP_COLLECTIVE", "all_reduce")).lower()
        if self.collective not in ("all_reduce", "rs_ag"):
            raise ValueError(f"collective / VGPA_DP_COLLECTIVE: all_reduce or rs_ag, got {self.collective!r}")
        # (start, end) on the communication stream, (before, after) the compute stream's wait: one 4-tuple per exchange.  A ring of the last
        # COMM_EVENT_RING exchanges: only comm_report() (bench.py) drains it, a 10 000-step training run never does and must not collect event handles
        self._comm_events = collections.deque(maxlen=self.COMM_EVENT_RING)
        self._wait_open = None
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

    class _Works:
        """the handles of one exchange (reduce-scatter + all-gather are two)"""

        def __init__(self, works):
            self.works = works

        def wait(self):
            for w in self.works:
                w.wait()

    def _exchange(self, g):
        """SUM over ranks of the message g (in place), asynchronously -> handle"""
        if self.collective == "all_reduce":
            return dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        w = self.world()
        m = self.flat.message(w)
        piece = m.numel() // w
        r = dist.get_rank(self.pg)
        mine = m[r * piece:(r + 1) * piece]
        # reduce_scatter_tensor may not alias its output with its input: the own piece is summed into a scratch chunk, then gathered back in place
        if getattr(self, "_rs_piece", None) is None or self._rs_piece.numel() != piece:
            self._rs_piece = torch.empty(piece, dtype=m.dtype, device=m.device)
        w1 = dist.reduce_scatter_tensor(self._rs_piece, m, op=dist.ReduceOp.SUM, group=self.pg, async_op=True)
        if not m.is_cuda:
            w1.wait()          # host backends do not order two async collectives of one group by themselves
        w2 = dist.all_gather_into_tensor(m, self._rs_piece, group=self.pg, async_op=True)
        return FlatAdamW._Works([w1, w2])

    def all_reduce_grads(self):
        """SUM over ranks of the flat gradient (+ logged-scalar tail, one message) on a side stream; the 1/world mean is folded into the step."""
        if self.world() == 1 and os.environ.get("VGPA_FORCE_DIST") != "1":
            return None
        g = self.flat.buf
        if g.is_cuda:
            if self._comm_stream is None:
                self._comm_stream = torch.cuda.Stream()
            self._comm_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._comm_stream):
                ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                ev0.record()
                work = self._exchange(g)
                work.wait()            # stream-orders the collective on the communication stream (no host block): ev1 marks its end there
                ev1.record()
            self._wait_open = (ev0, ev1)
            return work
        return self._exchange(g)

    def comm_report(self):
        """{"collective", "bytes", "exchanges", "allreduce_ms", "exposed_wait_ms"}: means over the exchanges since the last call (synchronises).
        allreduce_ms = duration of the collective on the communication stream; exposed_wait_ms = how long the compute stream stood still at the
        optimizer step waiting for it (0 when the collective had finished under the next micro-step's reference pass)."""
        evs, self._comm_events = list(self._comm_events), collections.deque(maxlen=self.COMM_EVENT_RING)
        if not evs:
            return None
        torch.cuda.synchronize()
        n = len(evs)
        msg = self.flat.message(self.world()) if self.collective == "rs_ag" else self.flat.buf
        return {"collective": self.collective, "bytes": int(msg.numel() * msg.element_size()), "exchanges": n, "allreduce_ms": sum(a.elapsed_time(b) for a, b, _, _ in evs) / n,
                "exposed_wait_ms": sum(c.elapsed_time(d) for _, _, c, d in evs) / n}

    def step(self, pending=None):
        if pending is not None:
            cur = torch.cuda.current_stream() if self._comm_stream is not None else None
            if cur is not None and self._wait_open is not None:
                w0 = torch.cuda.Event(enable_timing=True)
                w0.record(cur)
            pending.wait()
            if self._comm_stream is not None:
                cur.wait_stream(self._comm_stream)
                if self._wait_open is not None:
                    w1 = torch.cuda.Event(enable_timing=True)
                    w1.record(cur)
                    self._comm_events.append(self._wait_open + (w0, w1))
                    self._wait_open = None
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
