"""Data-parallel exchange of the flat LoRA-gradient buffer, world_size 2 over gloo on CPU:
N-rank mean gradient == single-process gradient on the concatenated batch; every rank ends identical."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from videogpa_amd.dataset import shard_indices
from videogpa_amd.optim import FlatAdamW, FlatParams


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _make_params(seed=0):
    g = torch.Generator().manual_seed(seed)
    # two "LoRA pairs" with odd sizes (exercises the 4-element alignment padding of the flat buffer)
    return [torch.nn.Parameter(torch.randn(s, generator=g)) for s in ((4, 37), (37, 4), (3, 10), (10, 3))]


def _pair_loss(params, x):
    A1, B1, A2, B2 = params
    h = x @ A1.t() @ B1.t()                      # [n, 37]
    h2 = h[:, :10] @ A2.t() @ B2.t()
    return (h.pow(2).mean(dim=1) + h2.pow(2).mean(dim=1))   # one loss value per preference pair


def _data(n=8):
    g = torch.Generator().manual_seed(123)
    return torch.randn(n, 37, generator=g)


def _worker(rank, world, port, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = _make_params(seed=rank)               # ranks start from DIFFERENT values ...
    flat = FlatParams(params)
    dist.broadcast(flat.flat, src=0)               # ... and take rank 0's, as DPOEngine does at construction
    assert all(torch.equal(p.detach(), q.detach()) for p, q in zip(params, _make_params(seed=0)))
    opt = FlatAdamW(flat, lr=1e-3, max_grad_norm=1.0, warmup_steps=0, total_steps=10)
    data = _data()
    idx = shard_indices(len(data), rank, world, shuffle=False)
    opt.zero_grad()
    # two micro-steps of one pair each, mean over the local accumulation (Lightning divides by accumulate_grad_batches)
    for i in idx[:2]:
        (_pair_loss(params, data[i:i + 1]).sum() / 2).backward()
    work = opt.all_reduce_grads()
    work.wait()
    g = flat.grad.clone() / world
    torch.save({"grad": g.clone(), "idx": idx[:2], "views_ok": all(p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o for p, o in zip(params, flat.offsets))},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


class _CpuStepAdamW(FlatAdamW):
    """TEST-ONLY: the product optimizer runs its update as HIP kernels and refuses CPU tensors; to exercise the
    data-parallel protocol (all-reduce of [grad | scalar tail], 1/world folded into the step, LR schedule, zero_grad)
    under gloo on CPU, the update itself is restated here with torch ops (clip_grad_norm_ + AdamW semantics)."""

    def _apply_update(self, lr, scale):
        f = self.flat
        g = f.grad * scale
        if self.max_grad_norm and self.max_grad_norm > 0:
            norm = g.norm()
            self.total_norm.copy_(norm.reshape(1))
            g = g * torch.clamp(self.max_grad_norm / (norm + 1e-6), max=1.0)
        b1, b2 = self.betas
        f.flat.mul_(1 - lr * self.wd)
        self.exp_avg.mul_(b1).add_(g, alpha=1 - b1)
        self.exp_avg_sq.mul_(b2).addcmul_(g, g, value=1 - b2)
        k = self.step_count
        denom = (self.exp_avg_sq / (1 - b2 ** k)).sqrt_().add_(self.eps)
        f.flat.addcdiv_(self.exp_avg / (1 - b1 ** k), denom, value=-lr)


def _train_worker(rank, world, port, out_dir, collective="all_reduce", tag="train"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params = _make_params(seed=rank)
    flat = FlatParams(params)
    dist.broadcast(flat.flat, src=0)
    opt = _CpuStepAdamW(flat, lr=1e-2, weight_decay=0.01, max_grad_norm=1.0, warmup_steps=1, total_steps=10, collective=collective)
    data = _data(8)
    synced = []
    for step in range(3):                                      # three optimizer steps, one pair per rank per step
        opt.zero_grad()
        i = step * world + rank
        loss = _pair_loss(params, data[i:i + 1]).sum()
        loss.backward()
        flat.tail[:3].add_(torch.stack([loss.detach(), loss.detach() * 2, (loss.detach() > 0).float()]))
        flat.tail[3:4].add_(1.0)
        opt.step(opt.all_reduce_grads())
        synced.append((flat.tail[:3] / flat.tail[3:4]).clone())
    torch.save({"param": flat.flat.clone(), "synced": torch.stack(synced), "slack": flat.storage[flat.buf.numel():].clone()},
               os.path.join(out_dir, f"{tag}{rank}.pt"))
    dist.destroy_process_group()


def test_two_ranks_three_optimizer_steps_match_single_process(tmp_path):
    """Ranks step together: final parameters equal (bit-identical across ranks) and equal, to 1e-6, to ONE process running
    torch.optim.AdamW + clip_grad_norm_ on the mean loss of the concatenated per-step batches; the logged scalars that
    rode the all-reduce equal the mean over ranks."""
    world = 2
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"train{i}.pt") for i in range(world)]
    assert torch.equal(r[0]["param"], r[1]["param"])
    assert torch.equal(r[0]["synced"], r[1]["synced"])
    params = _make_params(0)
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.01)
    from videogpa_amd.optim import cosine_schedule_with_warmup
    data = _data(8)
    ref_losses = []
    for step in range(3):
        for gq in opt.param_groups:
            gq["lr"] = 1e-2 * cosine_schedule_with_warmup(step, 1, 10)
        opt.zero_grad()
        losses = _pair_loss(params, data[step * world:(step + 1) * world])
        losses.mean().backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        ref_losses.append(losses.mean().item())
    flat = FlatParams(_make_params(0))
    got = r[0]["param"]
    for p, o in zip(params, flat.offsets):
        assert torch.allclose(got[o:o + p.numel()].view_as(p), p.detach(), rtol=0, atol=1e-6)
    assert torch.allclose(r[0]["synced"][:, 0], torch.tensor(ref_losses), rtol=1e-5, atol=1e-6)
    assert torch.allclose(r[0]["synced"][:, 1], 2 * torch.tensor(ref_losses), rtol=1e-5, atol=1e-6)


def test_flat_allreduce_matches_single_process(tmp_path):
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"rank{i}.pt") for i in range(world)]
    assert all(x["views_ok"] for x in r)
    assert torch.equal(r[0]["grad"], r[1]["grad"])
    used = r[0]["idx"] + r[1]["idx"]
    assert sorted(used) == sorted(set(used)) and len(used) == 4
    params = _make_params()
    flat = FlatParams(params)
    flat.zero_grad()
    _pair_loss(params, _data()[used]).mean().backward()          # single process, concatenated batch
    assert torch.allclose(flat.grad, r[0]["grad"], rtol=1e-4, atol=1e-3)   # fp32 summation order differs


def test_flatparams_keeps_values_and_accumulates():
    params = _make_params(1)
    before = [p.detach().clone() for p in params]
    flat = FlatParams(params)
    for p, b in zip(params, before):
        assert torch.equal(p.detach(), b)
    flat.zero_grad()
    x = _data(3)
    _pair_loss(params, x).sum().backward()
    g1 = flat.grad.clone()
    _pair_loss(params, x).sum().backward()
    assert torch.allclose(flat.grad, 2 * g1)
    assert flat.numel % 4 == 0


# ---------------------------------------------------------------- DPOEngine itself under gloo (fake trainer, CPU update stub)
class _FakeTrainer(torch.nn.Module):
    """Stands for CogVideoXDPOTrainer in DPOEngine: same training_step / configure_optimizers protocol on a toy loss."""

    def __init__(self, accumulate):
        super().__init__()
        self.params = torch.nn.ParameterList(_make_params(seed=int(os.environ.get("RANK", "0"))))   # ranks start DIFFERENT
        self.config = {"accumulate_grad_batches": accumulate}
        self.global_step = 0

    def training_step(self, batch, idx=0):
        loss = _pair_loss(list(self.params), batch).mean()
        return loss, {"train/loss": loss.detach(), "train/reward_margin": loss.detach() * 3, "train/reward_accuracy": (loss.detach() > 0).float()}

    def configure_optimizers(self, process_group=None):
        return _CpuStepAdamW(FlatParams(self.parameters()), lr=1e-2, weight_decay=0.0, max_grad_norm=1.0, warmup_steps=0, total_steps=10,
                             process_group=process_group)


def _engine_worker(rank, world, port, out_dir, overlap=None, tag="eng"):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from videogpa_amd import ops
    from videogpa_amd.trainer import DPOEngine
    tr = _FakeTrainer(accumulate=2)
    e0 = ops.ADAPTER_EPOCH
    eng = DPOEngine(tr, overlap=overlap)                        # broadcasts rank 0's parameters
    assert eng.overlap == (True if overlap is None else overlap)    # more than one rank: the optimizer step is overlapped by default
    assert ops.ADAPTER_EPOCH == e0 + 1
    assert all(torch.equal(p.detach(), q.detach()) for p, q in zip(tr.params, _make_params(seed=0)))
    data = _data(8)
    synced, lrs = [], []
    for micro in range(4):                                      # 2 optimizer steps of 2 micro-steps, one pair per rank and micro-step
        i = micro * world + rank
        logs = eng.micro_step(data[i:i + 1])
        if "lr" in logs:
            synced.append(logs["sync"].clone())
            lrs.append(logs["lr"])
    if eng.overlap:                                             # the second window's step is still in flight: it lands at flush()
        assert tr.global_step == 1 and len(synced) == 1 and eng._pending is not None
        last = eng.flush()
        synced.append(last["sync"].clone())
        lrs.append(last["lr"])
    assert eng._pending is None and eng.flush() is eng.last      # idempotent
    assert tr.global_step == 2 and len(synced) == 2 and ops.ADAPTER_EPOCH == e0 + 3
    torch.save({"param": eng.opt.flat.flat.clone(), "synced": torch.stack(synced), "lrs": torch.tensor(lrs)}, os.path.join(out_dir, f"{tag}{rank}.pt"))
    dist.destroy_process_group()


def test_dpo_engine_two_ranks_accumulation_and_synced_scalars(tmp_path):
    world = 2
    mp.spawn(_engine_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r = [torch.load(tmp_path / f"eng{i}.pt") for i in range(world)]
    assert torch.equal(r[0]["param"], r[1]["param"]) and torch.equal(r[0]["synced"], r[1]["synced"])
    # single process: each optimizer step sees the mean loss over its 4 pairs (2 ranks x 2 micro-steps)
    params = _make_params(0)
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=0.0)
    data = _data(8)
    ref = []
    from videogpa_amd.optim import cosine_schedule_with_warmup
    for step in range(2):
        for gq in opt.param_groups:
            gq["lr"] = 1e-2 * cosine_schedule_with_warmup(step, 0, 10)
        opt.zero_grad()
        losses = _pair_loss(params, data[step * 4:(step + 1) * 4])
        losses.mean().backward()
        torch.nn.utils.clip_grad_norm_(params, 1.0)
        opt.step()
        ref.append(losses.mean().item())
    flat = FlatParams(_make_params(0))
    for p, o in zip(params, flat.offsets):
        assert torch.allclose(r[0]["param"][o:o + p.numel()].view_as(p), p.detach(), rtol=0, atol=1e-6)
    assert torch.allclose(r[0]["synced"][:, 0], torch.tensor(ref), rtol=1e-5, atol=1e-6)            # loss: mean over ranks and micro-steps
    assert torch.allclose(r[0]["synced"][:, 1], 3 * torch.tensor(ref), rtol=1e-5, atol=1e-6)        # reward margin rides the same tail


def test_overlapped_optimizer_step_is_bit_identical_to_the_immediate_one(tmp_path):
    """DPOEngine(overlap=True) applies the optimizer step of a window inside the NEXT micro-step (after its reference pass, or right
    after training_step for trainers without that hook) so that the all-reduce hides under compute: parameters, rank-mean scalars
    and learning rates must equal the immediate form bit for bit, on both ranks."""
    world = 2
    for overlap, tag in ((True, "ov"), (False, "im")):
        mp.spawn(_engine_worker, args=(world, _free_port(), str(tmp_path), overlap, tag), nprocs=world, join=True)
    for rank in range(world):
        a, b = torch.load(tmp_path / f"ov{rank}.pt"), torch.load(tmp_path / f"im{rank}.pt")
        assert torch.equal(a["param"], b["param"]) and torch.equal(a["synced"], b["synced"]) and torch.equal(a["lrs"], b["lrs"])


# ---------------------------------------------------------------- the after_reference hook protocol (single process, overlap forced on)
class _HookTrainer(_FakeTrainer):
    """A trainer that CARRIES the `after_reference` attribute (like CogVideoXDPOTrainer / WanDPOTrainer).  mode "calls": training_step calls the
    hook between its (pretend) reference pass and its policy forward; "never": it never does; "stops": it calls it on the first two micro-steps only."""

    def __init__(self, accumulate, mode):
        super().__init__(accumulate)
        self.after_reference = None
        self.mode, self.calls = mode, 0

    def training_step(self, batch, idx=0):
        self.calls += 1
        if self.after_reference is not None and (self.mode == "calls" or (self.mode == "stops" and self.calls <= 2)):
            self.after_reference()
        return super().training_step(batch, idx)         # the policy forward reads the parameters only here


def _run_engine(mode, overlap, steps=6):
    import pytest  # noqa: F401
    from videogpa_amd.trainer import DPOEngine
    os.environ.pop("RANK", None)
    tr = _HookTrainer(1, mode) if mode else _FakeTrainer(1)
    eng = DPOEngine(tr, overlap=overlap)
    data = _data(8)
    for i in range(steps):
        eng.micro_step(data[i:i + 1])
    eng.flush()
    return eng.opt.flat.flat.clone(), tr, eng


def test_engine_never_steps_between_a_policy_forward_and_its_backward():
    """with a trainer that has `after_reference` but never calls it, the old engine applied the pending update AFTER the policy forward
    and BEFORE backward (new A / B against old activations).  Now: a trainer that never calls the hook is detected on the first micro-step (nothing
    is pending yet) and the engine falls back to stepping BEFORE each forward -- bit-identical to the immediate engine; a trainer that stops calling
    the hook while a step is pending is an error, not a silent mis-step."""
    import pytest
    want, _, _ = _run_engine(None, overlap=False)
    for mode in ("calls", "never"):
        got, tr, eng = _run_engine(mode, overlap=True)
        assert torch.equal(got, want), mode
        assert eng._hooked == (mode == "calls") and tr.global_step == 6
        assert (tr.after_reference is None) == (mode == "never")
    with pytest.raises(RuntimeError, match="after_reference"):
        _run_engine("stops", overlap=True)


def test_reduce_scatter_all_gather_exchange_is_bit_identical_to_the_all_reduce(tmp_path):
    """VGPA_DP_COLLECTIVE=rs_ag (reduce-scatter + all-gather of the flat [gradient | scalars] message, which on point-to-point xGMI uses every
    link where a ring all-reduce is bound by one): three optimizer steps on two ranks end in the SAME bits as with the all-reduce -- parameters and the
    rank-mean scalars -- on both ranks; the message length (odd here: 4-aligned tensors + 4 scalars) is padded to a multiple of the world size with slack
    that stays zero."""
    world = 2
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path), "all_reduce", "ar"), nprocs=world, join=True)
    mp.spawn(_train_worker, args=(world, _free_port(), str(tmp_path), "rs_ag", "rs"), nprocs=world, join=True)
    ar = [torch.load(tmp_path / f"ar{i}.pt") for i in range(world)]
    rs = [torch.load(tmp_path / f"rs{i}.pt") for i in range(world)]
    for i in range(world):
        assert torch.equal(rs[i]["param"], ar[i]["param"]) and torch.equal(rs[i]["synced"], ar[i]["synced"])
        assert rs[i]["slack"].abs().max().item() == 0.0
    assert torch.equal(rs[0]["param"], rs[1]["param"])


def test_exchange_message_is_padded_for_any_world_size():
    flat = FlatParams(_make_params(0))
    n = flat.buf.numel()
    for w in (1, 2, 3, 7, 8, 64):
        m = flat.message(w)
        assert m.numel() % w == 0 and n <= m.numel() < n + w and m.data_ptr() == flat.buf.data_ptr()
    try:
        flat.message(65)
        raise AssertionError("expected a ValueError")
    except ValueError:
        pass
    try:
        FlatAdamW(flat, collective="ring")
        raise AssertionError("expected a ValueError")
    except ValueError:
        pass
