"""Full-width gradient parity of the Wan2.2-TI2V denoiser (BASELINE.json configs[4]; train/Wan2.2-TI2V-5B/03_train.py:189-242 runs it four times per
pair): dim 3072, 24 heads x 128, ffn 14 336, text 512 x 4096, 2 blocks, 7 680 tokens (5 latent frames of 32 x 48 patches, the first one clean: t = 0 on its
1 536 tokens), batch 2 (win / lose on one timestep, as WanDPOTrainer batches a pair), LoRA r = 64 / alpha 128 on q / k / v / o of BOTH attentions --
videogpa_amd.wan_model.WanModel on the HIP kernels against oracle/wan.py run in fp32 on the same device.  The small-model tests (tests/test_gpu_wan_model.py:
dim 256, 2 heads) cannot see what this one is for: gradients that are small by cancellation (the q / k adapters of the last block) react to COHERENT
errors of the attention backward -- round 4 found 37-87 % at head_dim 64 that way -- and cancellation needs width.

Every LoRA tensor (32 of them), three device configurations, each against the oracle that makes the SAME roundings at the same places
(oracle/wan.py "activation-rounded mode"; what is left is accumulation order and the independent realisation of the rounding noise):
    bf16         default path ("Precise delta": eight further mantissa bits of the attention output)      vs  round_activations, exact_delta
    bf16_plain   precise_delta None: the textbook flash-attention backward                                vs  round_activations, delta from the bf16 output
    fp8          enable_fp8(True): e4m3 feed-forward GEMMs + the hand-written e4m3 self-attention forward,
                 backward on the forward's own (dequantised) operands                                      vs  + fp8_ffn, f8_attn
  bound: relative error <= 10 % AND cosine >= 0.995 on every tensor (the cfg1 bound of tests/test_gpu_cfg1.py); output within 3 % of its range.
Reported next to it (gpurun_out/wan_parity_<mode>.json -> profiles/r05_wan_parity_*.json): each tensor's distance from the PLAIN fp32 oracle, i.e. what
the arithmetic type itself costs, and for `fp8` the distance from the oracle that restates the round-4 device backward (f8_attn="r4": bf16 q / k / v against
the e4m3 forward's lse2 and output) -- the before / after of VERDICT r4 items 1 and 2."""
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
CFG = dict(model_type="ti2v", patch_size=(1, 2, 2), text_len=512, in_dim=48, dim=3072, ffn_dim=14336, freq_dim=256, text_dim=4096, out_dim=48,
           num_heads=24, num_layers=2, cross_attn_norm=True, eps=1e-6)
GRID = (5, 64, 96)              # latent frames x height x width -> 5 x 32 x 48 = 7 680 tokens
RANK, ALPHA = 64, 128.0
REL, COS = 0.10, 0.995


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    torch.cuda.empty_cache()
    yield
    _CACHE.clear()
    torch.cuda.empty_cache()


def _state():
    """random weights with the structure of a trained checkpoint where it matters for cancellation: xavier linears (upstream init_weights), a non-zero
    output layer, perturbed norm weights, small biases, and -- the part that matters -- modulation tables of ORDER ONE: upstream initialises them at
    dim^-1/2 (shift, scale, gate ~ 0.02: every token's normalised row then has no component in common with the others), a trained denoiser does not; with a
    common shift the keys and values of a sequence share a large mean, attention outputs sit near that mean, and the q / k gradients become the small
    difference of large sums that a coherent error in delta swamps.  Built on the CPU from one seed, bf16-representable."""
    from videogpa_amd.wan_model import WanModel
    torch.manual_seed(11)
    m = WanModel(**CFG)
    with torch.no_grad():
        torch.nn.init.normal_(m.head.head.weight, std=0.02)
        for blk in m.blocks:
            blk.modulation.copy_(torch.randn_like(blk.modulation) * torch.tensor([0.5, 0.3, 0.5, 0.5, 0.3, 0.5]).view(1, 6, 1))     # shift, scale, gate x 2
            blk.norm3.weight.add_(0.1 * torch.randn_like(blk.norm3.weight))
            blk.norm3.bias.add_(0.05 * torch.randn_like(blk.norm3.bias))
            for a in (blk.self_attn, blk.cross_attn):
                a.norm_q.weight.add_(0.1 * torch.randn_like(a.norm_q.weight))
                a.norm_k.weight.add_(0.1 * torch.randn_like(a.norm_k.weight))
        for p in m.parameters():
            if p.dim() == 1 and p.abs().max() == 0:
                p.add_(0.02 * torch.randn_like(p))
    return m.to(torch.bfloat16)


def _inputs():
    g = torch.Generator().manual_seed(12)
    Fr, H, W = GRID
    L = Fr * (H // 2) * (W // 2)
    x = [(0.8 * torch.randn(CFG["in_dim"], Fr, H, W, generator=g)).bfloat16().float() for _ in range(2)]
    t = torch.full((2, L), 613.0)
    t[:, : (H // 2) * (W // 2)] = 0.0                                        # TI2V: the first latent frame is clean (03_train.py:119-125)
    ctx = [(0.5 * torch.randn(n, CFG["text_dim"], generator=g)).bfloat16() for n in (512, 377)]
    tgt = [torch.randn(CFG["out_dim"], Fr, H, W, generator=g) for _ in range(2)]
    return x, t, ctx, L, tgt


def _lora_init(names):
    g = torch.Generator().manual_seed(13)
    out = {}
    for n in names:
        A = (torch.rand(RANK, CFG["dim"], generator=g) * 2 - 1) / CFG["dim"] ** 0.5          # PEFT's kaiming_uniform(a = sqrt 5) bound; every target maps dim -> dim
        Bm = torch.randn(CFG["dim"], RANK, generator=g) * 0.01
        out[n] = (A, Bm)
    return out


def _loss(outs, tgt):
    return sum(((o - t.to(o.device, o.dtype)) ** 2).mean() for o, t in zip(outs, tgt))


_CACHE = {}


def _hip(mode):
    """one forward + backward of the HIP model -> (outputs fp32 on the CPU, {module path: (dA, dB)} fp32 on the CPU)"""
    if ("hip", mode) in _CACHE:
        return _CACHE[("hip", mode)]
    from videogpa_amd.lora import LoraConfig, get_peft_model
    m = _state().to("cuda")
    if mode == "fp8":
        m.enable_fp8(True)
    m.set_precise_delta(None if mode == "bf16_plain" else "int8")
    pm = get_peft_model(m, LoraConfig(r=RANK, lora_alpha=ALPHA, lora_dropout=0.0, target_modules=["q", "k", "v", "o"]))
    mods = {n: mod for n, mod in pm.get_base_model().named_modules() if type(mod).__name__ == "LoraLinear"}
    assert len(mods) == 16
    init = _lora_init(sorted(mods))
    with torch.no_grad():
        for n, mod in mods.items():
            mod.lora_A["default"].weight.copy_(init[n][0])
            mod.lora_B["default"].weight.copy_(init[n][1])
    x, t, ctx, L, tgt = _inputs()
    out = pm([u.cuda() for u in x], t=t.cuda(), context=[c.cuda() for c in ctx], seq_len=L)
    loss = _loss(out, tgt)
    loss.backward()
    torch.cuda.synchronize()
    res = ([o.detach().float().cpu() for o in out], {n: (mod.lora_A["default"].weight.grad.float().cpu(), mod.lora_B["default"].weight.grad.float().cpu())
                                                     for n, mod in mods.items()}, float(loss))
    del pm, m, mods, out, loss
    torch.cuda.empty_cache()
    _CACHE[("hip", mode)] = res
    return res


def _oracle(**kw):
    key = ("oracle",) + tuple(sorted(kw.items()))
    if key in _CACHE:
        return _CACHE[key]
    from oracle import wan as ow
    state = {k: v.detach().float().cuda() for k, v in _state().state_dict().items()}
    names = sorted(f"blocks.{i}.{a}.{p}" for i in range(CFG["num_layers"]) for a in ("self_attn", "cross_attn") for p in "qkvo")
    init = _lora_init(names)
    leaves = {n: (init[n][0].cuda().requires_grad_(True), init[n][1].cuda().requires_grad_(True)) for n in names}
    P = ow.Params(state, {n: (A, Bm, ALPHA / RANK) for n, (A, Bm) in leaves.items()}, dtype=torch.float32, **kw)
    x, t, ctx, L, tgt = _inputs()
    out = ow.forward(P, CFG, [u.cuda() for u in x], t.cuda(), [c.float().cuda() for c in ctx], L)
    loss = _loss(out, tgt)
    loss.backward()
    res = ([o.detach().float().cpu() for o in out], {n: (A.grad.float().cpu(), Bm.grad.float().cpu()) for n, (A, Bm) in leaves.items()}, float(loss))
    del P, state, leaves, out, loss
    torch.cuda.empty_cache()
    _CACHE[key] = res
    return res


def _rel(a, r):
    return float((a.double() - r.double()).norm() / r.double().norm().clamp_min(1e-300))


def _cos(a, r):
    a, r = a.double().flatten(), r.double().flatten()
    return float((a @ r) / (a.norm() * r.norm()).clamp_min(1e-300))


ORACLE_FOR = {"bf16": dict(round_activations=True, exact_delta=True),
              "bf16_plain": dict(round_activations=True, exact_delta=False),
              "fp8": dict(round_activations=True, exact_delta=True, fp8_ffn=True, f8_attn=True)}


@pytest.mark.parametrize("mode", ["bf16", "bf16_plain", "fp8"])
def test_wan_full_width_lora_gradients_vs_the_matching_rounded_oracle(mode):
    out, grads, loss = _hip(mode)
    r_out, r_grads, r_loss = _oracle(**ORACLE_FOR[mode])
    p_out, p_grads, p_loss = _oracle()                                   # the plain fp32 oracle: what the arithmetic type costs (reported, bounded loosely)
    report = {"mode": mode, "tokens": GRID[0] * GRID[1] * GRID[2] // 4, "loss_hip": loss, "loss_matching_oracle": r_loss, "loss_fp32_oracle": p_loss}
    fails = []
    for b in range(2):
        rng_ = r_out[b].abs().max().item()
        err = (out[b] - r_out[b]).abs().max().item()
        report[f"out{b}_err_over_range"] = err / rng_
        report[f"out{b}_cos_vs_fp32"] = _cos(out[b], p_out[b])
        if not (err <= 0.03 * rng_ and _cos(out[b], r_out[b]) >= 0.9995):
            fails.append(("out", b, err / rng_, _cos(out[b], r_out[b])))
    if abs(loss - r_loss) > 2e-3 * abs(r_loss):
        fails.append(("loss", loss, r_loss))
    before = _oracle(**dict(ORACLE_FOR[mode], f8_attn="r4"))[1] if mode == "fp8" else None
    per, worst = {}, {"rel": 0.0, "cos": 1.0, "rel_vs_fp32": 0.0}
    for n in sorted(grads):
        for which, i in (("A", 0), ("B", 1)):
            g, r, p = grads[n][i], r_grads[n][i], p_grads[n][i]
            e, c = _rel(g, r), _cos(g, r)
            row = {"rel_vs_matching_oracle": round(e, 5), "cos_vs_matching_oracle": round(c, 6), "rel_vs_fp32": round(_rel(g, p), 5),
                   "matching_oracle_rel_vs_fp32": round(_rel(r, p), 5)}
            if before is not None:
                row["rel_vs_round4_backward_oracle"] = round(_rel(g, before[n][i]), 5)
                row["round4_backward_oracle_rel_vs_fp32"] = round(_rel(before[n][i], p), 5)
            per[f"{n}.lora_{which}"] = row
            worst["rel"], worst["cos"], worst["rel_vs_fp32"] = max(worst["rel"], e), min(worst["cos"], c), max(worst["rel_vs_fp32"], row["rel_vs_fp32"])
            if not (e <= REL and c >= COS):
                fails.append((n, which, "rel", e, "cos", c))
    report.update(worst=worst, per_tensor=per, failed_checks=[str(f) for f in fails])
    os.makedirs(os.path.join(os.path.dirname(HERE), "gpurun_out"), exist_ok=True)
    with open(os.path.join(os.path.dirname(HERE), "gpurun_out", f"wan_parity_{mode}.json"), "w") as f:
        json.dump(report, f, indent=1)
    print(json.dumps({k: v for k, v in report.items() if k != "per_tensor"}))
    assert not fails, fails


def test_wan_full_width_default_path_is_within_the_cfg1_cap_of_fp32_on_every_tensor():
    """the default path is within 12 % of the plain fp32 oracle on EVERY tensor (the cfg1 cap of tests/test_gpu_cfg1.py; measured: 0.7 %), and never further
    from it than the textbook backward by more than noise on the tensors "Precise delta" exists for.  MEASURED here, and worth stating: on this model the two
    backwards agree to 1e-4 -- the q / k adapters of the last block are 0.2 % from fp32 either way, where CogVideoX at the same width was 37-87 % off without
    the completed output.  A plausible reason, not isolated here: Wan rotates q and k by position (RoPE on the training path; the CogVideoX reference trains
    without it), which takes the common component out of sum_j P_ij K_j, the factor the coherent delta error rides on."""
    _, g_def, _ = _hip("bf16")
    _, g_plain, _ = _hip("bf16_plain")
    _, p_grads, _ = _oracle()
    last = CFG["num_layers"] - 1
    gains = {}
    for n in sorted(g_def):
        for i, which in ((0, "A"), (1, "B")):
            e_def, e_plain = _rel(g_def[n][i], p_grads[n][i]), _rel(g_plain[n][i], p_grads[n][i])
            assert e_def <= 0.12, (n, which, e_def)
            if n in (f"blocks.{last}.self_attn.q", f"blocks.{last}.self_attn.k"):
                gains[f"{n}.{which}"] = (round(e_plain, 4), round(e_def, 4))
                assert e_def <= e_plain + 2e-3, (n, which, e_def, e_plain)
    print(json.dumps({"last_block_self_attn_qk_rel_vs_fp32_plain_then_precise": gains}))
