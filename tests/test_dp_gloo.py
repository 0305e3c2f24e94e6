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
    g = flat.grad / world
    torch.save({"grad": g.clone(), "idx": idx[:2], "views_ok": all(p.grad.data_ptr() == flat.grad.data_ptr() + 4 * o for p, o in zip(params, flat.offsets))},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.destroy_process_group()


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
