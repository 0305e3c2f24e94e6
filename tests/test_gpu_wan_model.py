"""videogpa_amd.wan_model.WanModel (HIP kernels, bf16 operands, fp32 stream) against oracle/wan.py (plain torch restatement of the published
Wan2.2 WanModel; parity UNPINNED -- the Wan2.2 source is not vendored by the reference) at a small configuration that keeps every structural
feature of TI2V-5B: head_dim 128, per-token timesteps with first-frame tokens at t = 0, cross-attention over a padded text, affine norm3,
LoRA on q/k/v/o of both attentions, block checkpointing.  The oracle runs in fp64 on the same bf16-representable parameters.
Outputs and every LoRA gradient: cosine >= 0.995 and max error within 4 % of the tensor's range (bf16 activations through 2 blocks)."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu

CFG = dict(model_type="ti2v", patch_size=(1, 2, 2), text_len=32, in_dim=4, dim=256, ffn_dim=512, freq_dim=32, text_dim=64, out_dim=4, num_heads=2,
           num_layers=2, cross_attn_norm=True, eps=1e-6)


def _build(seed=0):
    from videogpa_amd.lora import LoraConfig, get_peft_model
    from videogpa_amd.wan_model import WanModel
    torch.manual_seed(seed)
    m = WanModel(**CFG)
    with torch.no_grad():
        torch.nn.init.normal_(m.head.head.weight, std=0.05)               # upstream zero-inits the output layer: give the test a signal
        for blk in m.blocks:
            blk.norm3.weight.add_(0.1 * torch.randn_like(blk.norm3.weight)); blk.norm3.bias.add_(0.1 * torch.randn_like(blk.norm3.bias))
            for a in (blk.self_attn, blk.cross_attn):
                a.norm_q.weight.add_(0.1 * torch.randn_like(a.norm_q.weight)); a.norm_k.weight.add_(0.1 * torch.randn_like(a.norm_k.weight))
        for p in m.parameters():
            if p.dim() == 1 and p.abs().max() == 0:
                p.add_(0.02 * torch.randn_like(p))                           # biases
    m = m.to(device="cuda", dtype=torch.bfloat16)
    state = {k: v.detach().clone() for k, v in m.state_dict().items()}
    pm = get_peft_model(m, LoraConfig(r=8, lora_alpha=16.0, lora_dropout=0.0, target_modules=["q", "k", "v", "o"]))
    g = torch.Generator(device="cuda").manual_seed(seed + 1)
    lora = {}
    for name, mod in pm.get_base_model().named_modules():
        if type(mod).__name__ == "LoraLinear":
            with torch.no_grad():
                mod.lora_B["default"].weight.copy_(torch.randn(mod.lora_B["default"].weight.shape, device="cuda", generator=g) * 0.05)
            lora[name] = mod
    return pm, state, lora


def _inputs(B=2, seed=5):
    g = torch.Generator(device="cuda").manual_seed(seed)
    C, Fr, H, W = CFG["in_dim"], 3, 8, 12
    x = [torch.randn(C, Fr, H, W, device="cuda", generator=g).bfloat16().float() for _ in range(B)]
    L = Fr * (H // 2) * (W // 2)
    n0 = (H // 2) * (W // 2)
    t = torch.tensor([417.0, 902.0][:B], device="cuda")[:, None].expand(B, L).clone()
    t[:, :n0] = 0.0                                                          # TI2V: first latent frame is clean (03_train.py:119-125)
    ctx = [torch.randn(n, CFG["text_dim"], device="cuda", generator=g).bfloat16() for n in (20, 32)[:B]]
    gout = [torch.randn(CFG["out_dim"], Fr, H, W, device="cuda", generator=g) for _ in range(B)]
    return x, t, ctx, L, gout


def _oracle(state, lora, x, t, ctx, L, gout, enabled=True, fp8_ffn=False, **kw):
    from oracle import wan as ow
    ldict, leaves = {}, {}
    if enabled:
        for name, mod in lora.items():
            A = mod.lora_A["default"].weight.detach().double().requires_grad_(True)
            Bm = mod.lora_B["default"].weight.detach().double().requires_grad_(True)
            ldict[name] = (A, Bm, mod.scaling["default"])
            leaves[name] = (A, Bm)
    P = ow.Params(state, ldict, dtype=torch.float64, fp8_ffn=fp8_ffn, **kw)
    out = ow.forward(P, CFG, [u.double() for u in x], t.double(), [c.double() for c in ctx], L)
    if enabled:
        sum((o * g.double()).sum() for o, g in zip(out, gout)).backward()
    return [o.detach() for o in out], leaves


def _close(got, ref, what, tol=0.04, cos_min=0.995):
    got, ref = got.detach().double().flatten(), ref.detach().double().flatten()
    err = (got - ref).abs().max().item()
    cos = float(got @ ref / (got.norm() * ref.norm()).clamp_min(1e-300))
    assert err <= tol * ref.abs().max().item() and cos >= cos_min, (what, err, ref.abs().max().item(), cos)


@pytest.mark.parametrize("ckpt", [False, True])
def test_wan_model_forward_and_lora_gradients_vs_oracle(ckpt):
    pm, state, lora = _build()
    pm.get_base_model().enable_gradient_checkpointing(ckpt)
    x, t, ctx, L, gout = _inputs()
    out = pm(x, t=t, context=ctx, seq_len=L)
    assert len(out) == 2 and out[0].shape == gout[0].shape and out[0].dtype == torch.float32
    sum((o * g).sum() for o, g in zip(out, gout)).backward()
    ref, leaves = _oracle(state, lora, x, t, ctx, L, gout)
    for b in range(2):
        _close(out[b], ref[b], f"out[{b}]")
    for name, mod in lora.items():
        _close(mod.lora_A["default"].weight.grad, leaves[name][0].grad, name + ".A", tol=0.06)
        _close(mod.lora_B["default"].weight.grad, leaves[name][1].grad, name + ".B", tol=0.06)


def test_wan_model_reference_pass_and_scalar_timestep():
    """adapter switched off = the frozen reference (03_train.py:164-168); t given as [B] expands to every token (upstream t.dim() == 1 branch)"""
    pm, state, lora = _build(seed=3)
    x, t, ctx, L, gout = _inputs(seed=9)
    with torch.no_grad(), pm.disable_adapter():
        out = pm(x, t=t[:, -1].contiguous(), context=ctx, seq_len=L)
    ref, _ = _oracle(state, lora, x, t[:, -1].contiguous(), ctx, L, gout, enabled=False)
    for b in range(2):
        _close(out[b], ref[b], f"ref out[{b}]")
    with pytest.raises(NotImplementedError):
        pm(x, t=t, context=ctx, seq_len=L + 8)


def test_wan_from_pretrained_gives_the_identical_forward(tmp_path):
    """train/Wan2.2-TI2V-5B/03_train.py:140-141: `WanModel.from_pretrained(path)` then `.to(torch.bfloat16)`.  A random model saved in the upstream
    layout (sharded) and loaded back runs the bit-identical forward on the HIP path."""
    from videogpa_amd.wan_model import WanModel
    torch.manual_seed(4)
    m = WanModel(**CFG)
    with torch.no_grad():
        torch.nn.init.normal_(m.head.head.weight, std=0.05)
    m.to(torch.bfloat16).save_pretrained(str(tmp_path), max_shard_size=300_000)
    a = m.to("cuda")
    b = WanModel.from_pretrained(str(tmp_path))
    b.to(torch.bfloat16)
    b = b.to("cuda")
    x, t, ctx, L, _ = _inputs()
    with torch.no_grad():
        oa, ob = a(x, t=t, context=ctx, seq_len=L), b(x, t=t, context=ctx, seq_len=L)
    assert all(torch.equal(u, v) and torch.isfinite(u).all() for u, v in zip(oa, ob)) and oa[0].abs().max() > 0


def test_wan_unfrozen_modulation_raises_instead_of_training_without_its_gradient():
    """one policy for every modulation consumer (blocks and head): a bare WanModel (parameters require grad) refuses a grad-enabled forward, because the row
    kernels return no gradient for shift / scale / gate; frozen -- what get_peft_model does -- or under no_grad it runs"""
    from videogpa_amd.wan_model import WanModel
    torch.manual_seed(2)
    m = WanModel(**CFG).to(device="cuda", dtype=torch.bfloat16)
    x, t, ctx, L, _ = _inputs()
    with pytest.raises(NotImplementedError, match="modulation table"):
        m(x, t=t, context=ctx, seq_len=L)
    with torch.no_grad():
        assert torch.isfinite(m(x, t=t, context=ctx, seq_len=L)[0]).all()
    m.requires_grad_(False)
    assert torch.isfinite(m(x, t=t, context=ctx, seq_len=L)[0]).all()


def test_wan_dpo_trainer_step_runs_on_the_hip_model():
    """WanDPOTrainer (train/Wan2.2-TI2V-5B/03_train.py:130-242) driving the HIP WanModel: one pair step, finite loss, every LoRA B gets a gradient"""
    from videogpa_amd.wan import WanDPOTrainer
    from videogpa_amd.wan_model import WanModel
    torch.manual_seed(0)
    m = WanModel(**CFG)
    with torch.no_grad():
        torch.nn.init.normal_(m.head.head.weight, std=0.05)
    m = m.to(device="cuda", dtype=torch.bfloat16)
    m.enable_gradient_checkpointing(True)
    tr = WanDPOTrainer({"lora_rank": 8, "lora_alpha": 16.0}, m)
    g = torch.Generator(device="cuda").manual_seed(1)
    batch = {"x_win": torch.randn(1, 4, 3, 8, 12, device="cuda", generator=g).bfloat16(), "x_lose": torch.randn(1, 4, 3, 8, 12, device="cuda", generator=g).bfloat16(),
             "prompt_emb": torch.randn(1, 24, CFG["text_dim"], device="cuda", generator=g).bfloat16()}
    loss, logs = tr.training_step(batch)
    loss.backward()
    assert torch.isfinite(loss) and abs(loss.item() - 0.6931) < 0.05          # B = 0 at init: policy == reference, loss = log 2
    grads = [p.grad for n, p in tr.transformer.named_parameters() if ".lora_B." in n]
    assert len(grads) == 16 and all(gr is not None and torch.isfinite(gr).all() for gr in grads) and sum(gr.abs().sum().item() for gr in grads) > 0


def test_wan_model_fp8_feed_forward_stays_close_to_the_oracle():
    """enable_fp8: e4m3 feed-forward operands, against TWO oracles.
    (a) the fp64 oracle with the e4m3 roundings INJECTED (oracle/wan.py::_Fp8Ffn: per-row e4m3 of the LN output, GELU output, gate-backward and
        GELU-backward results, per-row e4m3 of W and W^T, bf16 GEMM outputs): same arithmetic type at the same places, so the bounds are those of the
        bf16 test -- outputs and EVERY LoRA gradient, lora_A included: cosine >= 0.995, max error within 4 % / 6 % of the tensor's range.
    (b) the plain fp64 oracle, as context for what e4m3 itself costs: output cosine >= 0.99, lora_A / lora_B gradients cosine >= 0.98."""
    pm, state, lora = _build()
    pm.get_base_model().enable_fp8(True)
    x, t, ctx, L, gout = _inputs()
    out = pm(x, t=t, context=ctx, seq_len=L)
    sum((o * g).sum() for o, g in zip(out, gout)).backward()
    ref8, leaves8 = _oracle(state, lora, x, t, ctx, L, gout, fp8_ffn=True)
    ref, leaves = _oracle(state, lora, x, t, ctx, L, gout)
    for b in range(2):
        _close(out[b], ref8[b], f"out[{b}] vs e4m3-injected oracle")
        _close(out[b], ref[b], f"out[{b}]", tol=0.08, cos_min=0.99)
    for name, mod in lora.items():
        for which, idx in (("A", 0), ("B", 1)):
            got = (mod.lora_A if idx == 0 else mod.lora_B)["default"].weight.grad
            _close(got, leaves8[name][idx].grad, f"{name}.{which} vs e4m3-injected oracle", tol=0.06)
            _close(got, leaves[name][idx].grad, f"{name}.{which}", tol=0.15, cos_min=0.98)


def test_wan_model_e4m3_self_attention_stays_close_to_the_oracle():
    """enable_fp8(False, attention=True): the self-attention FORWARD on e4m3 operands (csrc/attention_hd128.hip attn128_fwd_f8_kernel) in a model long
    enough for it (1152 tokens >= ops.ATTN128_F8_MIN_KEYS; the 32-key cross-attention stays bf16), backward = the bf16 kernels on the forward's own
    dequantised operands (the straight-through gradient of the e4m3 forward).  Two oracles, as for the feed-forward:
    (a) the fp64 oracle with the attention's roundings INJECTED (oracle/wan.py::_F8Attn inside the activation-rounded mode: e4m3 q c / k / v with one power
        of two per head, per-tile-and-row scaled e4m3 softmax weights, unquantised row sums; backward on the dequantised operands): the same arithmetic type
        at the same places, so the bounds are those of the bf16 test of this file -- outputs and EVERY LoRA gradient cosine >= 0.995, max error within 4 % /
        6 % of the tensor's range;
    (b) the plain fp64 oracle, as context for what e4m3 itself costs: outputs cosine >= 0.995 / 6 % of range, gradients cosine >= 0.98 / 12 % of range --
        e4m3 scores carry ~0.05 of noise each (tests/test_gpu_wan_kernels.py::test_attention128_e4m3_forward_vs_fp64).
    The policy and the adapter-off reference pass use the same kernel, so at B = 0 they agree bit for bit (loss = ln 2: tests/test_gpu_fullmodel.py)."""
    from videogpa_amd import ops
    pm, state, lora = _build()
    base = pm.get_base_model()
    base.enable_fp8(False, attention=True)
    assert all(b.self_attn.fp8_attn and not b.fp8_ffn for b in base.blocks)
    g = torch.Generator(device="cuda").manual_seed(5)
    C, Fr, H, W = CFG["in_dim"], 3, 32, 48
    x = [torch.randn(C, Fr, H, W, device="cuda", generator=g).bfloat16().float() for _ in range(2)]
    L = Fr * (H // 2) * (W // 2)
    assert L >= ops.ATTN128_F8_MIN_KEYS
    t = torch.tensor([417.0, 902.0], device="cuda")[:, None].expand(2, L).clone()
    t[:, : (H // 2) * (W // 2)] = 0.0
    ctx = [torch.randn(n, CFG["text_dim"], device="cuda", generator=g).bfloat16() for n in (20, 32)]
    gout = [torch.randn(CFG["out_dim"], Fr, H, W, device="cuda", generator=g) for _ in range(2)]
    seen = []
    orig = ops.attention128_fwd_raw

    def spy(q, k, v, scale, o_pad=0, f8=False, **kw):
        seen.append((k.shape[2], f8))
        return orig(q, k, v, scale, o_pad, f8=f8, **kw)
    ops.attention128_fwd_raw = spy
    try:
        out = pm(x, t=t, context=ctx, seq_len=L)
    finally:
        ops.attention128_fwd_raw = orig
    assert (L, True) in seen and all(f8 == (n == L) for n, f8 in seen)          # self-attention e4m3, cross-attention (32 keys) bf16
    sum((o * g_).sum() for o, g_ in zip(out, gout)).backward()
    ref, leaves = _oracle(state, lora, x, t, ctx, L, gout)
    ref8, leaves8 = _oracle(state, lora, x, t, ctx, L, gout, round_activations=True, exact_delta=True, f8_attn=True)
    worst = {"out_cos": 1.0, "grad_cos": 1.0}
    for b in range(2):
        _close(out[b], ref8[b], f"out[{b}] vs e4m3-injected oracle")
        _close(out[b], ref[b], f"out[{b}]", tol=0.06, cos_min=0.995)
    for name, mod in lora.items():
        for which, idx in (("A", 0), ("B", 1)):
            got = (mod.lora_A if idx == 0 else mod.lora_B)["default"].weight.grad
            _close(got, leaves8[name][idx].grad, f"{name}.{which} vs e4m3-injected oracle", tol=0.06)
            _close(got, leaves[name][idx].grad, f"{name}.{which}", tol=0.12, cos_min=0.98)
            a, r = got.double().flatten().cpu(), leaves[name][idx].grad.double().flatten().cpu()
            worst["grad_cos"] = min(worst["grad_cos"], float(a @ r / (a.norm() * r.norm())))
    print("e4m3 self-attention, worst LoRA gradient cosine vs fp64 oracle:", worst["grad_cos"])


def test_wan_enable_fp8_attention_auto_keeps_layers_with_a_wide_score_range_on_bf16():
    """enable_fp8(attention="auto") (ops.F8AttnPolicy): every self-attention layer decides at its first call whether e4m3 scores are accurate enough on its data.
    With the model's own QK-norm gains (1) every layer stays on the e4m3 forward; with the gains of block 0 raised to 4 (a row bound of ~260 log2 units: an e4m3
    score would be off by ~1 log2 unit rms) THAT layer runs the bf16 forward, in the reference pass and in the policy pass alike, and the others keep e4m3."""
    from videogpa_amd import ops
    pm, state, lora = _build()
    base = pm.get_base_model()
    base.enable_fp8(False, attention="auto")
    assert all(b.self_attn.fp8_attn and b.self_attn.f8_policy is not None for b in base.blocks)
    with pytest.raises(ValueError):
        base.enable_fp8(False, attention="maybe")
    g = torch.Generator(device="cuda").manual_seed(5)
    C, Fr, H, W = CFG["in_dim"], 3, 32, 48
    x = [torch.randn(C, Fr, H, W, device="cuda", generator=g).bfloat16().float() for _ in range(2)]
    L = Fr * (H // 2) * (W // 2)
    t = torch.tensor([417.0, 902.0], device="cuda")[:, None].expand(2, L).clone()
    ctx = [torch.randn(n, CFG["text_dim"], device="cuda", generator=g).bfloat16() for n in (20, 32)]

    def run():
        seen = []
        orig = ops.attention128_fwd_raw

        def spy(q, k, v, scale, o_pad=0, f8=False, **kw):
            if k.shape[2] == L:
                seen.append(bool(f8))
            return orig(q, k, v, scale, o_pad, f8=f8, **kw)
        ops.attention128_fwd_raw = spy
        try:
            with torch.no_grad():
                out = pm(x, t=t, context=ctx, seq_len=L)
        finally:
            ops.attention128_fwd_raw = orig
        assert all(torch.isfinite(o).all() for o in out)
        return seen
    assert run() == [True] * len(base.blocks)
    rep = base.fp8_attention_report()
    assert all(r["fp8_attn"] and 0.0 < r["estimated_score_error_log2"] < 0.5 for r in rep), rep
    with torch.no_grad():
        for nrm in (base.blocks[0].self_attn.norm_q, base.blocks[0].self_attn.norm_k):
            nrm.weight.mul_(4.0)
    base.enable_fp8(False, attention="auto")                   # fresh policies: the decision is taken once per layer
    first = run()
    assert first == [False] + [True] * (len(base.blocks) - 1), first
    assert run() == first                                      # sticky: the policy pass runs what the reference pass ran
    rep = base.fp8_attention_report()
    assert not rep[0]["fp8_attn"] and rep[0]["estimated_score_error_log2"] > 0.5 and all(r["fp8_attn"] for r in rep[1:]), rep


def test_wan_adapter_mount_scale_merge_as_the_generate_script_does(tmp_path):
    """generate/Wan2.2-TI2V-5B.py:53-71: PeftModel.from_pretrained on the engine's model, scaling *= lora_weight, merge_and_unload.  The merged
    plain model must reproduce the LoRA-active model at that weight (bf16 weight rounding of the merged delta: 3 % of range, cos >= 0.999)."""
    from videogpa_amd.lora import PeftModel
    from videogpa_amd.wan_model import WanModel
    pm, state, lora = _build(seed=7)
    pm.save_pretrained(str(tmp_path))
    x, t, ctx, L, gout = _inputs(seed=11)
    weight = 0.5
    for mod in lora.values():
        mod.scaling["default"] *= weight
    with torch.no_grad():
        want = pm(x, t=t, context=ctx, seq_len=L)
    fresh = WanModel(**CFG).to(device="cuda", dtype=torch.bfloat16)
    fresh.load_state_dict(state)
    eng = PeftModel.from_pretrained(fresh, str(tmp_path), adapter_name="default", torch_dtype=torch.bfloat16)
    for module in eng.modules():                                              # the script's own loop (:64-67)
        if hasattr(module, "scaling") and isinstance(module.scaling, dict):
            for adapter in module.scaling:
                module.scaling[adapter] *= weight
    merged = eng.merge_and_unload()
    assert not any(type(m).__name__ == "LoraLinear" for m in merged.modules())
    merged.eval()
    with torch.no_grad():
        got = merged(x, t=t, context=ctx, seq_len=L)
    for b in range(2):
        _close(got[b], want[b], f"merged out[{b}]", tol=0.03, cos_min=0.999)


def test_wan_trainer_under_the_dpo_engine_deferred_step_is_bit_identical():
    """DPOEngine (flat AdamW, all-reduce slot, optimizer step deferred to the hook between the reference and the policy pass) driving WanDPOTrainer
    without block recompute: three micro-steps with the deferred step land on exactly the adapters of the immediate-step engine"""
    from videogpa_amd.trainer import DPOEngine
    from videogpa_amd.wan import WanDPOTrainer
    from videogpa_amd.wan_model import WanModel
    g = torch.Generator(device="cuda").manual_seed(1)
    batch = {"x_win": torch.randn(1, 4, 3, 8, 12, device="cuda", generator=g).bfloat16(), "x_lose": torch.randn(1, 4, 3, 8, 12, device="cuda", generator=g).bfloat16(),
             "prompt_emb": torch.randn(1, 24, CFG["text_dim"], device="cuda", generator=g).bfloat16(),
             "image_latent": torch.randn(1, 4, 1, 8, 12, device="cuda", generator=g).bfloat16()}
    finals = []
    for overlap in (False, True):
        torch.manual_seed(0)
        m = WanModel(**CFG)
        with torch.no_grad():
            torch.nn.init.normal_(m.head.head.weight, std=0.05)
        m = m.to(device="cuda", dtype=torch.bfloat16)
        m.enable_fp8(True)
        tr = WanDPOTrainer({"lora_rank": 8, "lora_alpha": 16.0, "accumulate_grad_batches": 1, "learning_rate": 1e-3, "warmup_steps": 0,
                            "enable_gradient_checkpointing": False}, m)
        assert not m.gradient_checkpointing
        with torch.no_grad():
            gb = torch.Generator(device="cuda").manual_seed(2)
            for n, p in tr.transformer.named_parameters():
                if ".lora_B." in n:
                    p.normal_(0.0, 0.02, generator=gb)
        eng = DPOEngine(tr, overlap=overlap)
        for _ in range(3):
            logs = eng.micro_step(batch)
            assert torch.isfinite(logs["train/loss"])
        eng.flush()
        assert tr.global_step == 3
        finals.append(eng.opt.flat.flat.clone())
    assert torch.equal(finals[0], finals[1])
    assert (finals[0] != 0).any()


def test_wan_fit_end_to_end_from_disk(tmp_path):
    """main_train of train/Wan2.2-TI2V-5B/03_train.py:309-387 without Lightning: latents / conditions on disk (x [C,F,H,W], encoder_hidden_states,
    image_latent) + meta json -> DPODataset -> the reference's collate_fn keys -> prefetching loader -> DPOEngine(WanDPOTrainer) -> periodic
    checkpoint + final_lora; resume continues from the checkpoint's step."""
    import json
    from videogpa_amd.fit import fit
    from videogpa_amd.wan_model import WanModel
    g = torch.Generator().manual_seed(3)
    groups = []
    for gi in range(6):
        vids = []
        for vi in range(2):
            torch.save(torch.randn(4, 3, 8, 12, generator=g).to(torch.bfloat16), tmp_path / f"lat_{gi}_{vi}.pt")
            torch.save({"encoder_hidden_states": torch.randn(20, CFG["text_dim"], generator=g).to(torch.bfloat16),
                        "image_latent": torch.randn(4, 1, 8, 12, generator=g).to(torch.bfloat16)}, tmp_path / f"cond_{gi}_{vi}.pt")
            vids.append({"video_path": f"v{gi}_{vi}.mp4", "consistency_score": 0.2 + 0.5 * vi, "motion_norm": 1.0,
                         "latent_path": f"lat_{gi}_{vi}.pt", "condition_path": f"cond_{gi}_{vi}.pt"})
        groups.append({"group_id": f"g{gi}", "prompt": "p", "videos": vids})
    (tmp_path / "meta_wan_data.json").write_text(json.dumps({"groups": groups}))

    def model():
        torch.manual_seed(0)
        m = WanModel(**CFG)
        with torch.no_grad():
            torch.nn.init.normal_(m.head.head.weight, std=0.05)
        m = m.to(device="cuda", dtype=torch.bfloat16)
        m.enable_fp8(True)
        return m

    base = {"base_path": str(tmp_path), "metadata_path": str(tmp_path / "meta_wan_data.json"), "accumulate_grad_batches": 2, "batch_size": 1, "num_workers": 0,
            "learning_rate": 1e-3, "warmup_steps": 1, "lora_rank": 8, "lora_alpha": 16.0}
    logs = []
    tr = fit(dict(base, max_steps=3, log_every_n_steps=1, checkpoint_every_n_steps=2, output_dir=str(tmp_path / "out")), transformer=model(), log=logs.append, model="wan")
    assert type(tr).__name__ == "WanDPOTrainer" and tr.global_step == 3 and len(logs) >= 3
    assert (tmp_path / "out" / "final_lora" / "adapter_model.safetensors").exists()
    assert any(float(p.detach().abs().max()) > 0 for n, p in tr.transformer.named_parameters() if "lora_B" in n)     # B starts at zero: trained
    ck = tmp_path / "out" / "checkpoints" / "step=2"
    assert (ck / "adapter_model.safetensors").exists() and (ck / "optimizer.pt").exists() and (ck / "rng_rank0.pt").exists()
    tr2 = fit(dict(base, max_steps=4, checkpoint_every_n_steps=0, resume_from=str(ck)), transformer=model(), log=lambda m: None, model="wan")
    assert tr2.global_step == 4
    with pytest.raises(ValueError):
        fit(dict(base, max_steps=1), transformer=model(), model="dit")
