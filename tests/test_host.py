"""Host-side logic of the product (runs on CPU): dataset / collate vs reference goldens, LR schedule, adapter file
format, sharding, scheduler table."""
import json
import os

import torch

from videogpa_amd import dataset as vds
from videogpa_amd.lora import LoraConfig, PeftModel, get_peft_model
from videogpa_amd.optim import cosine_schedule_with_warmup
from videogpa_amd.scheduler import CogVideoXDPMScheduler
from videogpa_amd.transformer import CogVideoXTransformer3DModel


def _materialise(gold, d):
    for name in gold["existing_files"]:
        p = os.path.join(d, name)
        if name.startswith("lat_"):
            gi, vi = name[4:-3].split("_")
            torch.save(torch.full((16, 2, 4, 4), float(int(gi) * 10 + int(vi))).to(torch.bfloat16), p)
        elif name.startswith("cond_"):
            gi, vi = name[5:-3].split("_")
            torch.save({"encoder_hidden_states": torch.full((6, 8), float(int(gi) * 10 + int(vi))).to(torch.bfloat16)}, p)
        else:
            json.dump(gold["meta"], open(p, "w"))


def test_dataset_matches_reference(golden_dir, tmp_path):
    gold = json.load(open(os.path.join(golden_dir, "dataset_pairs.json")))
    _materialise(gold, tmp_path)
    for r in gold["results"]:
        ds = vds.DPODataset(str(tmp_path), os.path.join(tmp_path, "meta_data.json"), metric_name="consistency_score", **r["kwargs"])
        got = [{"group_id": p["group_id"], "winner": p["winner"]["video_path"], "loser": p["loser"]["video_path"], "gap": p["metric_gap"]}
               for p in ds.preference_pairs]
        assert got == r["pairs"]
        if r["batch"] is not None:
            b = vds.collate_fn([ds[0], ds[1]])
            g = r["batch"]
            assert sorted(b.keys()) == g["keys"]
            assert list(b["x_win"].shape) == g["x_win_shape"] and str(b["x_win"].dtype) == g["x_win_dtype"]
            assert [float(b["x_win"][i].flatten()[0]) for i in range(2)] == g["x_win_vals"]
            assert [float(b["x_lose"][i].flatten()[0]) for i in range(2)] == g["x_lose_vals"]
            assert [float(b["prompt_emb"][i].flatten()[0]) for i in range(2)] == g["prompt_emb_vals"]   # winner's condition
            assert b["prompt"] == g["prompt"] and b["m_win"].tolist() == g["m_win"] and b["m_lose"].tolist() == g["m_lose"]
            p = vds.collate_paired([ds[0], ds[1]])
            assert p["x_pair"].shape == (2, 2, 2, 16, 4, 4)
            assert torch.equal(p["x_pair"][:, 0], b["x_win"].permute(0, 2, 1, 3, 4))
            assert torch.equal(p["x_pair"][:, 1], b["x_lose"].permute(0, 2, 1, 3, 4))


def test_missing_groups_key_raises(tmp_path):
    import pytest
    p = tmp_path / "m.json"
    p.write_text("{}")
    with pytest.raises(ValueError, match="missing 'groups'"):
        vds.DPODataset(str(tmp_path), str(p))


def test_lr_schedule_matches_transformers():
    from transformers import get_cosine_schedule_with_warmup
    opt = torch.optim.SGD([torch.nn.Parameter(torch.zeros(1))], lr=5e-6)
    sch = get_cosine_schedule_with_warmup(opt, num_warmup_steps=500, num_training_steps=10000)
    for step in range(0, 10001):
        if step in (0, 1, 250, 499, 500, 501, 5250, 9999, 10000):
            assert abs(opt.param_groups[0]["lr"] - 5e-6 * cosine_schedule_with_warmup(step, 500, 10000)) < 1e-18
        opt.step()
        sch.step()
    assert cosine_schedule_with_warmup(0, 500, 10000) == 0.0
    assert abs(cosine_schedule_with_warmup(250, 500, 10000) - 0.5) < 1e-12
    assert abs(cosine_schedule_with_warmup(5250, 500, 10000) - 0.5) < 1e-12


def test_shard_indices_partition():
    n, w = 23, 4
    parts = [vds.shard_indices(n, r, w, epoch=3) for r in range(w)]
    assert len({len(p) for p in parts}) == 1 and len(parts[0]) == 6
    assert set(sum(parts, [])) == set(range(n))
    assert parts == [vds.shard_indices(n, r, w, epoch=3) for r in range(w)]      # deterministic
    assert parts != [vds.shard_indices(n, r, w, epoch=4) for r in range(w)]
    dl = [vds.shard_indices(n, r, w, drop_last=True) for r in range(w)]
    assert sorted(sum(dl, [])) == sorted(set(sum(dl, []))) and len(sum(dl, [])) == 20


def _tiny():
    return CogVideoXTransformer3DModel(num_attention_heads=2, num_layers=2, time_embed_dim=32, text_embed_dim=48,
                                       use_rotary_positional_embeddings=True)


def test_state_dict_contract():
    m = CogVideoXTransformer3DModel(num_attention_heads=2, num_layers=1, time_embed_dim=32, text_embed_dim=48,
                                    use_rotary_positional_embeddings=True)
    sd = m.state_dict()
    D = 128
    expect = {
        "patch_embed.proj.weight": (D, 16, 2, 2), "patch_embed.text_proj.weight": (D, 48), "time_embedding.linear_1.weight": (32, D),
        "time_embedding.linear_2.weight": (32, 32), "transformer_blocks.0.norm1.linear.weight": (6 * D, 32),
        "transformer_blocks.0.norm1.norm.weight": (D,), "transformer_blocks.0.attn1.to_q.weight": (D, D),
        "transformer_blocks.0.attn1.norm_q.weight": (64,), "transformer_blocks.0.attn1.norm_k.bias": (64,),
        "transformer_blocks.0.attn1.to_out.0.weight": (D, D), "transformer_blocks.0.norm2.linear.bias": (6 * D,),
        "transformer_blocks.0.ff.net.0.proj.weight": (4 * D, D), "transformer_blocks.0.ff.net.2.weight": (D, 4 * D),
        "norm_final.weight": (D,), "norm_out.linear.weight": (2 * D, 32), "norm_out.norm.bias": (D,), "proj_out.weight": (64, D),
    }
    for k, shp in expect.items():
        assert tuple(sd[k].shape) == shp, k
    m.config.use_dynamic_positional_embedding = True   # attribute assignment, as train/CogVideoX1.5-5B/03_train.py:95 does
    assert m.config["use_dynamic_positional_embedding"] is True


def test_adapter_file_format_and_roundtrip(tmp_path):
    from safetensors.torch import load_file
    pm = get_peft_model(_tiny(), LoraConfig(r=4, lora_alpha=8, lora_dropout=0.0, target_modules=["to_q", "to_k", "to_v", "to_out.0"]))
    trainable = [n for n, p in pm.named_parameters() if p.requires_grad]
    assert len(trainable) == 2 * 4 * 2 and all("lora_" in n for n in trainable)
    with torch.no_grad():
        for n, p in pm.named_parameters():
            if "lora_B" in n:
                p.normal_()
    pm.save_pretrained(str(tmp_path / "final_lora"))
    cfg = json.load(open(tmp_path / "final_lora" / "adapter_config.json"))
    ref = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "adapter_config_keys.json")))
    assert set(ref) <= set(cfg), set(ref) - set(cfg)            # every key PEFT wrote for the released adapters
    assert cfg["peft_type"] == "LORA" and cfg["r"] == 4 and cfg["lora_alpha"] == 8 and sorted(cfg["target_modules"]) == ["to_k", "to_out.0", "to_q", "to_v"]
    sd = load_file(str(tmp_path / "final_lora" / "adapter_model.safetensors"))
    assert "base_model.model.transformer_blocks.1.attn1.to_out.0.lora_B.weight" in sd
    assert tuple(sd["base_model.model.transformer_blocks.0.attn1.to_q.lora_A.weight"].shape) == (4, 128)
    assert tuple(sd["base_model.model.transformer_blocks.0.attn1.to_q.lora_B.weight"].shape) == (128, 4)
    pm2 = PeftModel.from_pretrained(_tiny(), str(tmp_path / "final_lora"), adapter_name="other")
    for k, v in sd.items():
        assert torch.equal(pm2.state_dict()[k[:-7] + ".other.weight"], v)
    layer = pm2.lora_layers()[0]
    assert layer.scaling["other"] == 2.0 and layer.r["other"] == 4 and layer.lora_alpha["other"] == 8
    layer.scaling["other"] = 0.2                                    # generate/CogVideoX1.5-5B.py:32-35 override
    # merge: W' = W + scaling * B A
    w0 = layer.base_layer.weight.detach().clone()
    delta = layer.delta_weight("other")
    base = pm2.merge_and_unload()
    assert not any(hasattr(m, "base_layer") for m in base.modules())
    assert torch.allclose(base.transformer_blocks[0].attn1.to_q.weight, w0 + delta, atol=1e-6)


def test_scheduler_table_and_config():
    s = CogVideoXDPMScheduler()
    assert s.config.num_train_timesteps == 1000
    a = s.alphas_cumprod
    assert float(a[-1]) == 0.0 and abs(float(a[0]) - (1 - 0.00085)) < 1e-9 and bool((a[1:] <= a[:-1]).all())


def test_dpm_sampler_step_matches_oracle_restatement():
    """CogVideoXDPMScheduler.set_timesteps / step (host-side float64 math, generate path) vs the oracle's independent
    (alpha, sigma, lambda) restatement, incl. the zero-terminal-SNR first step and the first-order fallback after it."""
    import torch
    from oracle import scheduler as osch
    from videogpa_amd.scheduler import CogVideoXDPMScheduler
    abar = osch.alphas_cumprod()
    for steps in (6, 50):
        s = CogVideoXDPMScheduler()
        s.set_timesteps(steps)
        assert torch.equal(abar, s.alphas_cumprod)
        ts = osch.trailing_timesteps(steps)
        assert ts.tolist() == s.timesteps.tolist()
        g = torch.Generator().manual_seed(steps)
        x = torch.randn(1, 2, 4, 4, 4, generator=g, dtype=torch.float64)
        xo, old, oldo = x.clone(), None, None
        for i, t in enumerate(ts):
            v = torch.randn(x.shape, generator=g, dtype=torch.float64)
            n = torch.randn(2, *x.shape, generator=g, dtype=torch.float64)
            back = ts[i - 1] if i > 0 else None
            x, old = s.step(v, old, t, back, x, noise=n)
            xo, oldo = osch.dpm_step(abar, v, oldo, t, back, xo, steps, n)
            assert torch.allclose(x, xo, rtol=0, atol=1e-12) and torch.allclose(old, oldo, rtol=0, atol=1e-12)


def test_variant_configs_match_the_reference_scripts():
    """Per-script hyper-parameters (train/*/03_train.py DEFAULT_CONFIG + the 1.5 script's hard-coded weight decay)."""
    from videogpa_amd.trainer import variant_config
    t2v, i2v, v15 = variant_config("t2v"), variant_config("i2v"), variant_config("1.5")
    assert (t2v["batch_size"], t2v["accumulate_grad_batches"], t2v["max_steps"], t2v["weight_decay"]) == (1, 2, 10000, 0.01)
    assert (i2v["batch_size"], i2v["accumulate_grad_batches"]) == (2, 1)
    assert (v15["max_steps"], v15["weight_decay"]) == (1500, 1e-3)
    for c in (t2v, i2v, v15):
        assert (c["lora_rank"], c["lora_alpha"], c["learning_rate"], c["warmup_steps"], c["gradient_clip_val"], c["beta"]) == (64, 128, 5e-6, 500, 1.0, 1.0)
    import pytest
    with pytest.raises(ValueError):
        variant_config("2b")


def test_kernel_timer_samples_all_launches_on_the_first_steps_and_the_roofline_kernels_throughout(monkeypatch):
    """ops.KernelTimer(full_steps, always): bench.py's live per-kernel timing -- every launch is timed during the first `full_steps` steps, afterwards
    only the kernels the roofline objects are computed from; per-step figures divide by the steps a kernel was actually timed on."""
    import torch
    from videogpa_amd import ops

    class FakeEvent:
        clock = 0.0

        def __init__(self, enable_timing=True):
            self.t = None

        def record(self):
            FakeEvent.clock += 1.0
            self.t = FakeEvent.clock

        def elapsed_time(self, other):
            return other.t - self.t

    monkeypatch.setattr(torch.cuda, "Event", FakeEvent)
    t = ops.KernelTimer(full_steps=2, always=("attn",))
    ran = []
    for step in range(5):
        for name in ("attn", "ln", "attn", "gelu"):
            t.run(name, 10.0, lambda n=name: ran.append(n), "flop" if name == "attn" else "byte")
        t.next_step()
    assert len(ran) == 20                                   # every launch runs, timed or not
    s = t.summary()
    assert s["attn"]["launches"] == 10 and s["attn"]["steps"] == 5
    assert s["ln"]["launches"] == 2 and s["ln"]["steps"] == 2 and s["gelu"]["launches"] == 2
    assert s["attn"]["avg_ms"] == 1.0 and s["attn"]["unit"] == "flop" and s["ln"]["unit"] == "byte"
    full = ops.KernelTimer()                                # default: everything, always
    for step in range(3):
        full.run("ln", 1.0, lambda: None, "byte")
        full.next_step()
    assert full.summary()["ln"]["launches"] == 3 and full.summary()["ln"]["steps"] == 3


def test_bench_energy_report_arithmetic_and_missing_counter():
    """bench.py::energy_report: joules per step, mean power and TFLOP per joule from two counter readings; None when the counter is not there (no
    librocm_smi64 in this container) or did not advance -- the JSON line then carries "energy": null instead of a made-up number."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.energy_report(None, 5.0, 1.0, 2, 1e12) is None and bench.energy_report(5.0, None, 1.0, 2, 1e12) is None
    assert bench.energy_report(10.0, 10.0, 1.0, 2, 1e12) is None
    r = bench.energy_report(100.0, 6100.0, 4.0, 2, 2.4e15, 3.0e15)
    assert r["joules_per_step"] == 3000.0 and r["mean_power_w"] == 1500.0
    assert abs(r["algorithmic_tflop_per_joule"] - 0.8) < 1e-12 and abs(r["executed_tflop_per_joule"] - 1.0) < 1e-12
    j = bench.read_joules(0)
    assert j is None or j >= 0.0


def test_bench_fences_the_cpu_leg_and_the_secondary_config_children_onto_disjoint_cores(monkeypatch):
    """bench.py::split_host_cpus: the default run times the CPU baseline WHILE child processes build and bench cfg3 / cfg4 / cfg5; the two get
    disjoint hardware threads -- the children the CHILD_CPUS highest-numbered ones this process may use -- and a host too small to split is reported as such"""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod2", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(256)), raising=False)
    leg, child = bench.split_host_cpus()
    assert len(child) == bench.CHILD_CPUS and len(leg) == 256 - bench.CHILD_CPUS and not set(leg) & set(child) and min(child) > max(leg)
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: {3, 5, 7, 9} | set(range(100, 140)), raising=False)
    leg, child = bench.split_host_cpus()
    assert sorted(child) == list(range(124, 140)) and set(leg) == {3, 5, 7, 9} | set(range(100, 124))
    monkeypatch.setattr(os, "sched_getaffinity", lambda pid: set(range(2 * bench.CHILD_CPUS)), raising=False)
    assert bench.split_host_cpus() == (None, None)


def _tiny_trainer(**cfg):
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    m = CogVideoXTransformer3DModel(num_attention_heads=2, num_layers=1, time_embed_dim=32, text_embed_dim=48, use_rotary_positional_embeddings=True).to(torch.bfloat16)
    return CogVideoXDPOTrainer(dict({"lora_rank": 4, "lora_alpha": 8}, **cfg), transformer=m)


def test_lean_activations_setting_is_normalised_and_anything_else_is_an_error():
    """a truthy non-bool (1, "true") used to run FULL activations silently because of an `is True` check."""
    import pytest
    from videogpa_amd.trainer import CogVideoXDPOTrainer
    f = CogVideoXDPOTrainer._lean_setting
    assert [f(v) for v in (True, 1, "true", "True", "1", "on")] == [True] * 6
    assert [f(v) for v in (False, 0, "false", "0", "off")] == [False] * 5
    assert f("auto") == "auto" and f(None) == "auto" and f(" AUTO ") == "auto"
    for bad in ("lean", 2, 0.5, [True]):
        with pytest.raises(ValueError):
            f(bad)
    tr = _tiny_trainer(lean_activations=1)
    assert tr.config["lean_activations"] is True and tr.transformer.get_base_model().lean_activations
    assert _tiny_trainer(lean_activations="false").config["lean_activations"] is False


def test_validation_step_refuses_to_run_over_a_pending_optimizer_step_and_holds_the_hook_off():
    """DPOEngine defers the optimizer step into the NEXT micro-step's `after_reference` hook; a validation pass must neither trigger it (the
    policy pass would see adapters updated in the middle of the batch) nor run while one is pending."""
    import pytest

    class Eng:
        _pending = object()
        fired = 0

        def _from_hook(self):
            self.fired += 1
    eng = Eng()
    tr = _tiny_trainer()
    tr.after_reference = eng._from_hook
    with pytest.raises(RuntimeError, match="flush"):
        tr.validation_step({"x_pair": torch.zeros(1, 2, 1, 16, 4, 4, dtype=torch.bfloat16), "prompt_emb": torch.zeros(1, 2, 48, dtype=torch.bfloat16)})
    assert eng.fired == 0 and tr.after_reference == eng._from_hook          # restored
    eng._pending = None
    with pytest.raises(RuntimeError, match="no CPU fallback|MI355X only"):          # gets as far as the model (no GPU here), with the hook held off
        tr.validation_step({"x_pair": torch.zeros(1, 2, 1, 16, 4, 4, dtype=torch.bfloat16), "prompt_emb": torch.zeros(1, 2, 48, dtype=torch.bfloat16)})
    assert eng.fired == 0 and tr.after_reference == eng._from_hook


def test_attention_forward_policy_switches_on_the_kernels_own_redo_count():
    from videogpa_amd import ops
    p = ops.AttnFwdPolicy()
    assert p.mode == "bound" and p.wants_flags()
    p.observe(0.0); p.calls += 1
    assert p.mode == "bound" and p.wants_flags()
    p.observe(0.4); p.calls += 1
    assert p.mode == "bound" and not p.wants_flags()             # two checks done, next re-check at call RECHECK
    p.calls = ops.AttnFwdPolicy.RECHECK
    assert p.wants_flags()
    p.observe(0.6)
    assert p.mode == "online" and p.switched_at == ops.AttnFwdPolicy.RECHECK and not p.wants_flags()
    q = ops.AttnFwdPolicy(mode="bound", fixed=True)
    q.observe(1.0)
    assert q.mode == "bound" and not q.wants_flags()


def test_f8_attention_policy_decides_once_from_the_score_range():
    """ops.F8AttnPolicy (WanModel.enable_fp8(attention="auto")): the estimate is ERR_PER_BOUND x max|q| max|k| scale log2(e) per (batch, head); unit-gain RMS-normed
    operands stay on the e4m3 forward, gains of 3.5 (a row bound of 200 log2 units) go to bf16; the decision is taken at the first call and kept (reference and policy pass of a step run the same
    forward), fixed=True never looks at the data."""
    from videogpa_amd import ops
    g = torch.Generator().manual_seed(0)
    B, L, H, d = 1, 96, 2, 128

    def rmsn(gain):
        t = torch.randn(B, L, H, d, generator=g)
        return (gain * t * torch.rsqrt(t.pow(2).mean(-1, keepdim=True))).reshape(B, L, H * d).bfloat16()
    q1, k1, q3, k3 = rmsn(1.0), rmsn(1.0), rmsn(3.5), rmsn(3.5)
    scale = d ** -0.5
    e1 = ops.F8AttnPolicy.score_error_estimate(q1, k1, H, scale)
    want = 0.004 * (128 ** 0.5) ** 2 * scale * 1.4426950408889634            # |q| = |k| = sqrt(128) after the RMS norm
    assert abs(e1 - want) < 0.02 * want, (e1, want)
    p = ops.F8AttnPolicy()
    assert p.use_f8(q1, k1, H, scale) and p.mode == "f8" and p.decided
    assert p.use_f8(q3, k3, H, scale) and p.mode == "f8"                     # decided: later data does not move it (loss = ln 2 at B = 0 needs both passes alike)
    p.reset()
    assert not p.use_f8(q3, k3, H, scale) and p.mode == "bf16" and p.estimated_score_error > 0.5
    assert not p.use_f8(q1, k1, H, scale)                                    # ... and it stays on bf16 for good
    f = ops.F8AttnPolicy(fixed=True)
    assert f.use_f8(q3, k3, H, scale) and f.estimated_score_error is None


def test_tuned_gemm_file_is_wired_but_off_without_a_gpu(monkeypatch):
    """ops.use_tuned_gemms: the shipped TunableOp results file names only hipBLASLt solutions of shapes the step issues; without a GPU nothing is switched on, and a
    caller that runs TunableOp its own way is left alone."""
    from videogpa_amd import ops
    assert os.path.isfile(ops.TUNED_GEMM_FILE)
    rows = [ln.strip().split(",") for ln in open(ops.TUNED_GEMM_FILE) if ln.strip()]
    assert {r[1] for r in rows if r[0] == "Validator"} >= {"PT_VERSION", "HIPBLASLT_VERSION", "GCN_ARCH_NAME"}
    entries = [r for r in rows if r[0] != "Validator"]
    assert entries and all(r[0].startswith(("GemmTunableOp_BFloat16", "GemmAndBiasTunableOp_BFloat16", "ScaledGemmTunableOp")) and r[2].startswith("Gemm_Hipblaslt_") for r in entries)
    if not torch.cuda.is_available():
        st = ops.use_tuned_gemms(True)
        assert st["enabled"] is False
    monkeypatch.setenv("PYTORCH_TUNABLEOP_ENABLED", "0")
    st = ops.use_tuned_gemms(True)
    assert "left as is" in st.get("note", "")
