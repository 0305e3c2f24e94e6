"""Two data-parallel ranks driving the REAL HIP path through DPOEngine.  A gpurun box has one GPU and RCCL refuses two ranks on
one device, so both ranks run on cuda:0 and the flat [gradients | scalars] buffer travels over gloo (which stages CUDA tensors through the
host) -- everything else is what an 8-GPU run executes: broadcast of rank 0's adapters at construction, per-rank pairs and per-rank (t, eps)
streams (seed + rank), the SUM all-reduce issued on the side stream after backward, the optimizer step deferred to the hook between the
next micro-step's reference and policy pass, 1 / world folded into the fused clip + AdamW kernel.

Checked: (i) both ranks end with bit-identical adapters and optimizer state; (ii) they equal ONE process that feeds the same pairs with the
same (t, eps) streams through an engine with accumulate_grad_batches = world (mean of rank gradients == accumulated gradient: the scaling
by 1/2 is exact in binary floating point; the summation order differs, hence 1e-6 instead of bit equality); (iii) the rank-mean scalars
the reference logs with sync_dist=True arrive through the same buffer."""
import os
import socket

import pytest
import torch

pytestmark = pytest.mark.gpu

KW = dict(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=32, text_embed_dim=48, sample_width=8, sample_height=8,
          sample_frames=9, max_text_seq_length=6)
CFG = {"beta": 1.0, "accumulate_grad_batches": 1, "learning_rate": 1e-3, "warmup_steps": 0, "max_steps": 10, "seed": 2}
STEPS = 3


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _batches(rank):
    g = torch.Generator().manual_seed(100 + rank)
    return [{"x_pair": (0.7 * torch.randn(1, 2, 3, 16, 8, 8, generator=g)).to(torch.bfloat16).cuda(),
             "prompt_emb": (0.5 * torch.randn(1, 6, 48, generator=g)).to(torch.bfloat16).cuda()} for _ in range(STEPS)]


def _trainer(cfg, seed_b):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from oracle import cogvideox as ocv                    # test infrastructure: the seeded state dict of the tiny configuration
    from videogpa_amd.lora import LoraConfig, get_peft_model
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    from videogpa_amd.transformer import CogVideoXTransformer3DModel
    sd = {k: v.to(torch.bfloat16) for k, v in ocv.init_state_dict(ocv.CogVideoXConfig(**KW), seed=0, std=0.05, mod_std=0.2).items()}
    model = CogVideoXTransformer3DModel(use_rotary_positional_embeddings=True, **KW)
    model.load_state_dict(sd, strict=True)
    torch.manual_seed(11)                                  # PEFT's kaiming-uniform lora_A
    pm = get_peft_model(model.to(device="cuda", dtype=torch.bfloat16), LoraConfig(r=4, lora_alpha=8, target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    with torch.no_grad():
        gb = torch.Generator(device="cuda").manual_seed(seed_b)
        for n, p in pm.named_parameters():
            if ".lora_B." in n:
                p.normal_(0.0, 0.05, generator=gb)
    tr = CogVideoXDPOTrainer(dict(cfg), transformer=pm)
    tr.train()
    return tr


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from videogpa_amd.trainer import DPOEngine
    tr = _trainer(CFG, seed_b=5 + rank)                    # ranks start from DIFFERENT lora_B values: the engine must take rank 0's
    eng = DPOEngine(tr)
    assert eng.overlap and eng._hooked
    synced = []
    for b in _batches(rank):
        logs = eng.micro_step(b)
        if "sync" in logs:
            synced.append(logs["sync"].clone().cpu())
    synced.append(eng.flush()["sync"].clone().cpu())
    torch.save({"flat": eng.opt.flat.flat.cpu(), "m": eng.opt.exp_avg.cpu(), "v": eng.opt.exp_avg_sq.cpu(), "sync": torch.stack(synced), "steps": tr.global_step},
               os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_on_the_hip_path_match_one_process_accumulating(tmp_path):
    import torch.multiprocessing as mp
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    world = 2
    env = {"HSA_ENABLE_IPC_MODE_LEGACY": "0"}
    os.environ.update({k: os.environ.get(k, v) for k, v in env.items()})
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(os.path.join(tmp_path, f"rank{r}.pt")) for r in range(world))
    assert r0["steps"] == r1["steps"] == STEPS
    for k in ("flat", "m", "v", "sync"):
        assert torch.equal(r0[k], r1[k]), k                # every rank holds the same adapters, moments and rank-mean scalars
    assert r0["sync"].shape == (STEPS, 3) and torch.isfinite(r0["sync"]).all()

    # one process, accumulate_grad_batches = world, fed rank 0's and rank 1's pair of every step with THEIR (t, eps) streams
    from videogpa_amd.trainer import DPOEngine
    tr = _trainer(dict(CFG, accumulate_grad_batches=world), seed_b=5)      # rank 0's starting values
    eng = DPOEngine(tr, overlap=False)
    dev = torch.device("cuda", 0)
    gens = [torch.Generator(device=dev).manual_seed(CFG["seed"] + r) for r in range(world)]
    per_rank = [_batches(r) for r in range(world)]
    losses = []
    for s in range(STEPS):
        for r in range(world):
            tr._rng = gens[r]
            logs = eng.micro_step(per_rank[r][s])
            losses.append(float(logs["train/loss"]))
    eng.flush()
    assert tr.global_step == STEPS
    one = eng.opt.flat.flat.cpu()
    scale = one.abs().max().item()
    assert (one - r0["flat"]).abs().max().item() <= 1e-6 * scale, (one - r0["flat"]).abs().max().item()
    assert (eng.opt.exp_avg.cpu() - r0["m"]).abs().max().item() <= 1e-6 * r0["m"].abs().max().item()
    # the synced loss of step s is the mean over the two ranks' losses of that step
    want = torch.tensor([(losses[2 * s] + losses[2 * s + 1]) / 2 for s in range(STEPS)])
    assert (r0["sync"][:, 0] - want).abs().max().item() <= 1e-5
