"""The full 42-block models of BASELINE configs[1] (CogVideoX-5B T2V) and configs[2] (CogVideoX-5B-I2V) at S = 17 776 through the
engine, checked by what holds regardless of the random weights (SURVEY 8c known answers): with the adapters' B = 0 (the PEFT
initialisation) policy == reference bit for bit, so the Diffusion-DPO loss is ln 2 to 1e-6 after 42 layers, the gradient reaches only
lora_B, everything is finite -- plus the memory the reference's I2V batch size needs, and one engine step through a real RCCL
communicator; and BASELINE configs[3] (CogVideoX1.5-5B, S = 41 026) at full depth with lean activations.  -m gpu only; these tests take the whole
GPU (up to ~270 GB)."""
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


@pytest.fixture(autouse=True)
def _release_gpu_memory_between_tests():
    """every test here takes most of the GPU: drop what the previous one left in the caching allocator (autograd cycles included) before and after"""
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    yield
    gc.collect()
    torch.cuda.empty_cache()


def _build(model_cfg, layers=42, seed=0):
    from videogpa_amd import transformer as vtr
    from videogpa_amd.transformer import CogVideoXTransformer3DModel
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("cuda"):
            model = CogVideoXTransformer3DModel(**dict(getattr(vtr, model_cfg), num_layers=layers))
    finally:
        torch.set_default_dtype(prev)
    g = torch.Generator(device="cuda").manual_seed(seed)
    with torch.no_grad():
        for n, p in model.named_parameters():
            if n.endswith("norm.weight") or n == "norm_final.weight" or ".norm_q.weight" in n or ".norm_k.weight" in n:
                p.fill_(1.0)
            elif n.endswith(".bias"):
                p.zero_()
            else:
                p.normal_(0.0, 0.02, generator=g)
    return model


def _ln2_step(trainer, batch, n_lora):
    torch.cuda.reset_peak_memory_stats()
    out = trainer._shared_step(batch)
    assert abs(out.loss.item() - math.log(2.0)) < 1e-6, out.loss.item()          # B = 0: policy == reference through all 42 blocks
    out.loss.backward()
    torch.cuda.synchronize()
    named = {n: p for n, p in trainer.transformer.named_parameters() if p.requires_grad}
    assert len(named) == n_lora and all(p.grad is not None and torch.isfinite(p.grad).all() for p in named.values())
    assert all(float(p.grad.abs().max()) == 0.0 for n, p in named.items() if ".lora_A." in n)      # dA = dT^T x with dT = dy B = 0
    nz = sum(float(p.grad.abs().max()) > 0.0 for n, p in named.items() if ".lora_B." in n)
    assert nz >= (n_lora // 2) - 4, nz              # every lora_B (a few may underflow to exact zero in bf16 at this depth)
    return torch.cuda.max_memory_allocated() / 2 ** 30


def test_cfg2_full_42_block_pair_step_is_ln2_at_b0():
    """BASELINE configs[1]: CogVideoX-5B T2V, 42 blocks, paired latents [1,2,13,16,60,90] (S = 17 776), LoRA r = 64 on q/k/v/out."""
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    tr = CogVideoXDPOTrainer({"lora_rank": 64, "lora_alpha": 128, "beta": 1.0, "seed": 7}, transformer=_build("COGVIDEOX_5B"))
    tr.train()
    g = torch.Generator(device="cuda").manual_seed(1234)
    batch = {"x_pair": (0.7 * torch.randn(1, 2, 13, 16, 60, 90, generator=g, device="cuda")).to(torch.bfloat16),
             "prompt_emb": (0.2 * torch.randn(1, 226, 4096, generator=g, device="cuda")).to(torch.bfloat16)}
    gb = _ln2_step(tr, batch, n_lora=42 * 8)
    assert gb < 200.0, gb
    print(f"cfg2 pair-step peak memory {gb:.1f} GB")


def test_cfg2_full_depth_on_trained_like_qk_norm_gains_stays_finite_and_ln2_at_b0():
    """Round 6: the whole 42-block pair step with the QK-norm gains of a TRAINED model's shape (gain 2.5 +- 20 %, three 3 x outlier channels, biases: what
    `bench.py --weights trained_like` runs) instead of the gains of 1 a random init gives -- sharp attention rows, scores spread over +-100 log2 units.  This is the
    run that found the forward's overflow window (a row whose true maximum lies 112-128 above the sampled shift: finite row sum, O = inf, NaN loss from block 38
    on; tools/trained_like_diag.py, tools/attn_fault_repro.py).  B = 0: loss = ln 2 to 1e-6 and finite lora_B gradients; afterwards every layer reports its
    redone-strip fraction (measured <= 0.8 %) and none has left the bound-shifted forward."""
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    model = _build("COGVIDEOX_5B")
    gq = torch.Generator(device="cuda").manual_seed(7)
    with torch.no_grad():
        for blk in model.transformer_blocks:
            for nrm in (blk.attn1.norm_q, blk.attn1.norm_k):
                w = 2.5 * (1 + 0.2 * torch.randn(64, generator=gq, device="cuda"))
                w[:3] *= 3.0
                nrm.weight.copy_(w.to(nrm.weight.dtype))
                nrm.bias.copy_((0.25 * torch.randn(64, generator=gq, device="cuda")).to(nrm.bias.dtype))
    tr = CogVideoXDPOTrainer({"lora_rank": 64, "lora_alpha": 128, "beta": 1.0, "seed": 7}, transformer=model)
    tr.train()
    g = torch.Generator(device="cuda").manual_seed(1234)
    batch = {"x_pair": (0.7 * torch.randn(1, 2, 13, 16, 60, 90, generator=g, device="cuda")).to(torch.bfloat16),
             "prompt_emb": (0.2 * torch.randn(1, 226, 4096, generator=g, device="cuda")).to(torch.bfloat16)}
    _ln2_step(tr, batch, n_lora=42 * 8)
    rep = tr.transformer.get_base_model().attention_forward_report()
    fr = [r["redo_fraction"] for r in rep]
    assert all(f is not None and 0.0 <= f <= 0.05 for f in fr), max(f for f in fr if f is not None)
    assert all(r["mode"] == "bound" for r in rep)
    print(f"trained-like gains: redone-strip fraction max {max(fr):.4f}, mean {sum(fr) / len(fr):.5f}")


def test_cfg3_i2v_batch_2_fits_and_is_ln2_at_b0():
    """BASELINE configs[2] at the reference's I2V batch size (train/CogVideoX-I2V-5B/03_train.py:59-60: batch_size 2, no accumulation):
    TWO pairs per step = four 17 776-token sequences through 42 blocks with NO block recomputed.  Must fit the 288 GB part with room to spare: with the
    full activation set + the precise-delta residuals the step peaks at 269 GB, which leaves the caching allocator no slack (it fragmented into an
    out-of-memory error on one box in round 4), so two pairs per step run with lean activations (the LN output and the normalised q / k are made again
    in the backward, bit-identical: DESIGN section 3; the output's res8 bytes for the backward's delta are kept) -- chosen by the trainer's own memory
    policy (lean_activations "auto", the default), which is what this test exercises."""
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    tr = CogVideoXDPOTrainer({"lora_rank": 64, "lora_alpha": 128, "beta": 1.0, "batch_size": 2, "accumulate_grad_batches": 1, "seed": 7},
                             transformer=_build("COGVIDEOX_5B_I2V"))
    assert tr.config["lean_activations"] == "auto"
    tr.train()
    g = torch.Generator(device="cuda").manual_seed(4321)
    batch = {"x_pair": (0.7 * torch.randn(2, 2, 13, 16, 60, 90, generator=g, device="cuda")).to(torch.bfloat16),
             "prompt_emb": (0.2 * torch.randn(2, 226, 4096, generator=g, device="cuda")).to(torch.bfloat16),
             "image_latent": (0.7 * torch.randn(2, 1, 16, 60, 90, generator=g, device="cuda")).to(torch.bfloat16)}
    gb = _ln2_step(tr, batch, n_lora=42 * 8)
    assert tr.memory_policy_log["lean_activations"] is True and tr.transformer.get_base_model().lean_activations, tr.memory_policy_log
    assert gb < 250.0, gb
    print(f"cfg3 batch-2 pair-step peak memory {gb:.1f} GB; policy {tr.memory_policy_log}")


def test_cfg4_full_depth_lean_step_is_ln2_at_b0():
    """BASELINE configs[3]: CogVideoX1.5-5B T2V (patch_size_t = 2, no patch bias), 81f x 768x1360 -> paired latents [1,2,21,16,96,170], even-cropped to
    20 frames by the 1.5 step (train/CogVideoX1.5-5B/03_train.py:118-186) -> S = 226 + 10 x 48 x 85 = 41 026 tokens, ALL 42 blocks, LoRA r = 64, lean
    activations and NO recompute -- exactly what `bench.py --config cfg4` times (230 GB of the 288).  B = 0: policy == reference through 42 layers at
    S = 41 026, so the loss is ln 2 to 1e-6, gradients reach only lora_B and are finite."""
    import gc
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    gc.collect(); torch.cuda.empty_cache()
    model = _build("COGVIDEOX_1_5_5B")
    tr = CogVideoXDPOTrainer({"lora_rank": 64, "lora_alpha": 128, "beta": 1.0, "seed": 7, "weight_decay": 1e-3, "lean_activations": True,
                              "enable_gradient_checkpointing": False}, transformer=model)
    assert model.lean_activations and not model.gradient_checkpointing
    tr.train()
    g = torch.Generator(device="cuda").manual_seed(2468)
    batch = {"x_pair": (0.7 * torch.randn(1, 2, 21, 16, 96, 170, generator=g, device="cuda")).to(torch.bfloat16),
             "prompt_emb": (0.2 * torch.randn(1, 226, 4096, generator=g, device="cuda")).to(torch.bfloat16)}
    seen = {}
    orig = model.forward

    def spy(hs, *a, **k):
        seen["tokens"] = 226 + (hs.shape[1] // 2) * (hs.shape[3] // 2) * (hs.shape[4] // 2)
        return orig(hs, *a, **k)
    model.forward = spy
    gb = _ln2_step(tr, batch, n_lora=42 * 8)
    model.forward = orig
    assert seen["tokens"] == 41026, seen
    assert gb < 260.0, gb
    print(f"cfg4 full-depth lean pair-step peak memory {gb:.1f} GB")
    del tr, model, batch
    gc.collect(); torch.cuda.empty_cache()


def test_engine_micro_steps_through_a_real_rccl_communicator(monkeypatch):
    """DPOEngine with VGPA_FORCE_DIST=1 on one rank: the flat [gradients | scalars] SUM exchange really goes through RCCL (backend
    "nccl" on ROCm) on the side stream -- as ONE all-reduce and as reduce-scatter + all-gather (VGPA_DP_COLLECTIVE=rs_ag) -- the optimizer
    step is applied overlapped (after the next micro-step's reference pass), and the result equals the engine without any communicator bit for bit.
    FlatAdamW.comm_report() (what bench.py prints at N > 1) has timed both exchanges: bytes of the message, duration on the communication stream,
    and the stall of the compute stream at the optimizer step."""
    import torch.distributed as dist
    from oracle import cogvideox as ocv
    from videogpa_amd.lora import LoraConfig, get_peft_model
    from videogpa_amd.trainer import CogVideoXDPOTrainer, DPOEngine
    from videogpa_amd.transformer import CogVideoXTransformer3DModel
    kw = dict(num_attention_heads=2, attention_head_dim=64, num_layers=2, time_embed_dim=32, text_embed_dim=48, sample_width=8, sample_height=8,
              sample_frames=9, max_text_seq_length=6)
    cfg = ocv.CogVideoXConfig(**kw)
    sd = {k: v.to(torch.bfloat16) for k, v in ocv.init_state_dict(cfg, seed=0, std=0.05, mod_std=0.2).items()}
    g = torch.Generator().manual_seed(3)
    batches = [{"x_pair": (0.7 * torch.randn(1, 2, 3, 16, 8, 8, generator=g)).to(torch.bfloat16).cuda(),
                "prompt_emb": (0.5 * torch.randn(1, 6, 48, generator=g)).to(torch.bfloat16).cuda()} for _ in range(4)]

    reports = {}

    def run(force, collective="all_reduce"):
        monkeypatch.setenv("VGPA_FORCE_DIST", "1" if force else "0")
        monkeypatch.setenv("VGPA_DP_COLLECTIVE", collective)
        model = CogVideoXTransformer3DModel(use_rotary_positional_embeddings=True, **kw)
        model.load_state_dict(sd, strict=True)
        torch.manual_seed(11)      # PEFT's kaiming-uniform lora_A
        pm = get_peft_model(model.to(device="cuda", dtype=torch.bfloat16), LoraConfig(r=4, lora_alpha=8, target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
        with torch.no_grad():
            gb = torch.Generator(device="cuda").manual_seed(5)
            for n, p in pm.named_parameters():
                if ".lora_B." in n:
                    p.normal_(0.0, 0.05, generator=gb)
        tr = CogVideoXDPOTrainer({"beta": 1.0, "accumulate_grad_batches": 2, "learning_rate": 1e-3, "warmup_steps": 0, "max_steps": 10, "seed": 2}, transformer=pm)
        tr.train()
        eng = DPOEngine(tr)
        assert eng.overlap == force
        seen = []
        for b in batches:
            logs = eng.micro_step(b)
            if "sync" in logs:
                seen.append(logs["sync"].clone())
        last = eng.flush()
        if force:
            seen.append(last["sync"].clone())
        assert tr.global_step == 2 and len(seen) == 2
        if force:
            rep = eng.opt.comm_report()
            assert rep["collective"] == collective and rep["exchanges"] == 2 and rep["bytes"] == (eng.opt.flat.numel + 4) * 4
            assert rep["allreduce_ms"] > 0 and rep["exposed_wait_ms"] >= 0 and eng.opt.comm_report() is None      # a report drains the events
            reports[collective] = rep
        return eng.opt.flat.flat.clone(), torch.stack(seen)

    plain = run(False)
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        rccl = run(True)
        rsag = run(True, "rs_ag")
    finally:
        dist.destroy_process_group()
    assert torch.equal(plain[0], rccl[0]) and torch.equal(plain[1], rccl[1])
    assert torch.equal(plain[0], rsag[0]) and torch.equal(plain[1], rsag[1])
    print({"comm_report": reports})


def test_bench_line_under_a_forced_process_group_carries_the_comm_attribution():
    """`bench.py` with VGPA_FORCE_DIST=1 (the N > 1 code path on the one GPU of this box: RCCL group of one rank, side-stream exchange, deferred optimizer
    step) on a 2-block debug shape: the JSON line carries what a scaling loss would be attributed from -- ranks_seen, per-rank ms_per_step (min / max) and
    comm = {collective, bytes, exchanges, allreduce_ms, exposed_wait_ms (+ max over ranks)} -- for both collectives; since round 6 ONE invocation
    carries both forms (`collective_ab`: a second short timed region on the other exchange) and the per-rank `preflight` table (device, free memory, RCCL version)."""
    import json
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for coll in ("all_reduce", "rs_ag"):
        with socket.socket() as sk:          # a free port per run (a fixed one collides under parallel test runs: ADVICE r5)
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ, VGPA_FORCE_DIST="1", VGPA_DP_COLLECTIVE=coll, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--layers", "2", "--frames", "3", "--height", "16", "--width", "16", "--steps", "3",
                            "--warmup", "1", "--no-cpu-baseline", "--no-other-configs"], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        j = json.loads(next(ln for ln in reversed(r.stdout.splitlines()) if ln.startswith("{")))
        assert j["ranks_seen"] == 1 and j["n_gpus"] == 1 and len(j["ms_per_step_by_rank"]) == 1
        assert 0 < j["ms_per_step_min"] <= j["ms_per_step_max"] <= j["ms_per_step"] * 1.001
        c = j["comm"]
        assert c["collective"] == coll and c["exchanges"] == 3 and c["bytes"] > 4 * 2 * 4 * 64 * 3072 and c["allreduce_ms"] > 0
        assert 0 <= c["exposed_wait_ms"] <= c["exposed_wait_ms_max_over_ranks"] + 1e-9
        ab = j["collective_ab"]
        assert set(ab) == {"all_reduce", "rs_ag"} and all(v["ms_per_step"] > 0 and v["comm"]["collective"] == k and v["comm"]["exchanges"] == 3 for k, v in ab.items())
        pre = j["preflight"]
        assert len(pre) == 1 and pre[0]["rank"] == 0 and pre[0]["free_gb"] > 1 and pre[0]["rccl_version"][0].isdigit()
        assert "preflight: rank 0" in r.stderr
        print({coll: c, "ab": {k: v["ms_per_step"] for k, v in ab.items()}})


@pytest.mark.parametrize("recompute,mem_gb", [(True, 80), (False, 230)])
def test_cfg5_wan22_ti2v_5b_full_size_pair_step(recompute, mem_gb):
    """BASELINE configs[4]: Wan2.2-TI2V-5B at 81 f x 704 x 1280 (latent 48 x 21 x 44 x 80 -> 18 480 tokens, 30 blocks, 24 heads of 128, text 512),
    LoRA r = 64 on q/k/v/o, e4m3 feed-forward; with the reference's per-block checkpointing (48 GB) and with every activation resident (the
    bench.py --config cfg5 setting: 168-206 GB of the 288).  Random weights: with B = 0 the policy equals the reference, so the loss is ln 2 after 30
    layers (the fp8 and bf16 paths are deterministic per input), gradients reach exactly the 240 lora_B tensors and are finite; after one optimizer
    step the policy has moved and the loss is still finite."""
    from videogpa_amd.wan import WanDPOTrainer
    from videogpa_amd.wan_model import WanModel
    import gc
    gc.collect(); torch.cuda.empty_cache(); torch.cuda.reset_peak_memory_stats()
    torch.manual_seed(0)
    prev = torch.get_default_dtype()
    torch.set_default_dtype(torch.bfloat16)
    try:
        with torch.device("cuda"):
            m = WanModel()                                             # defaults = the TI2V-5B configuration
    finally:
        torch.set_default_dtype(prev)
    assert sum(p.numel() for p in m.parameters()) > 4.9e9
    with torch.no_grad():
        torch.nn.init.normal_(m.head.head.weight, std=0.02)
    m.enable_fp8(True)
    tr = WanDPOTrainer({"enable_gradient_checkpointing": recompute}, m)
    assert m.gradient_checkpointing == recompute
    opt = tr.configure_optimizers()
    g = torch.Generator(device="cuda").manual_seed(1)
    batch = {"x_win": torch.randn(1, 48, 21, 44, 80, device="cuda", generator=g).bfloat16(), "x_lose": torch.randn(1, 48, 21, 44, 80, device="cuda", generator=g).bfloat16(),
             "prompt_emb": torch.randn(1, 300, 4096, device="cuda", generator=g).bfloat16(), "image_latent": torch.randn(1, 48, 1, 44, 80, device="cuda", generator=g).bfloat16()}
    loss, logs = tr.training_step(batch)
    assert abs(loss.item() - math.log(2.0)) < 1e-6
    loss.backward()
    named = dict(tr.transformer.named_parameters())
    gb = [p.grad for n, p in named.items() if ".lora_B." in n]
    ga = [p.grad for n, p in named.items() if ".lora_A." in n]
    assert len(gb) == 30 * 8 and all(x is not None and torch.isfinite(x).all() for x in gb) and sum(x.abs().sum().item() for x in gb) > 0
    assert all(x is None or x.abs().max().item() == 0 for x in ga)            # B = 0: dA = 0
    opt.step(); opt.zero_grad()
    loss2, _ = tr.training_step(batch)
    assert torch.isfinite(loss2) and abs(loss2.item() - math.log(2.0)) < 0.05
    assert torch.cuda.max_memory_allocated() / 2 ** 30 < mem_gb
    del tr, opt, m, loss, loss2, named, gb, ga
    gc.collect(); torch.cuda.empty_cache()
