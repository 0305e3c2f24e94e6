"""CPU checks of the Wan2.2 host side and of its oracle (no GPU, no kernels): state-dict names follow the upstream checkpoint, the RoPE / timestep
tables of videogpa_amd/wan_model.py equal the oracle's complex-arithmetic forms, the per-token timestep tensor follows the reference step, and
known-answer properties of oracle/wan.py (the parity-UNPINNED restatement of the un-vendored WanModel): LoRA B = 0 -> no change, zero output
head -> zero output, first-frame tokens see t = 0, text padding positions are attended (upstream passes k_lens = None)."""
import torch

from oracle import wan as ow

CFG = dict(patch_size=(1, 2, 2), text_len=8, in_dim=4, dim=128, ffn_dim=64, freq_dim=16, text_dim=12, out_dim=4, num_heads=1, num_layers=2, cross_attn_norm=True,
           eps=1e-6)


def _model():
    from videogpa_amd.wan_model import WanModel
    torch.manual_seed(0)
    return WanModel(model_type="ti2v", **CFG)


def test_state_dict_names_follow_the_upstream_checkpoint():
    m = _model()
    keys = set(m.state_dict().keys())
    expect = {"patch_embedding.weight", "patch_embedding.bias", "text_embedding.0.weight", "text_embedding.2.bias", "time_embedding.0.weight", "time_embedding.2.weight",
              "time_projection.1.weight", "time_projection.1.bias", "head.head.weight", "head.head.bias", "head.modulation"}
    for i in range(2):
        p = f"blocks.{i}."
        expect |= {p + "modulation", p + "norm3.weight", p + "norm3.bias", p + "ffn.0.weight", p + "ffn.2.bias"}
        for a in ("self_attn", "cross_attn"):
            expect |= {p + f"{a}.{n}.{w}" for n in "qkvo" for w in ("weight", "bias")} | {p + f"{a}.norm_q.weight", p + f"{a}.norm_k.weight"}
    assert expect <= keys, sorted(expect - keys)
    assert not any("norm1" in k or "norm2" in k for k in keys)          # WanLayerNorm without affine parameters
    assert m.blocks[0].modulation.shape == (1, 6, 128) and m.head.modulation.shape == (1, 2, 128)
    assert m.patch_embedding.weight.shape == (128, 4, 1, 2, 2) and (m.head.head.weight == 0).all()       # upstream zero-inits the output layer
    # PEFT targets of the reference config (03_train.py:82): exactly the eight attention projections of every block
    from videogpa_amd.lora import LoraConfig, get_peft_model
    pm = get_peft_model(m, LoraConfig(r=4, lora_alpha=8.0, lora_dropout=0.0, target_modules=["q", "k", "v", "o"]))
    assert len(pm.lora_layers()) == 2 * 8
    assert all(n.endswith(("lora_A.default.weight", "lora_B.default.weight")) for n, p in pm.named_parameters() if p.requires_grad)


def test_rope_and_timestep_tables_equal_the_oracle_forms():
    from videogpa_amd.wan_model import rope_tables, sinusoidal_embedding_1d
    from videogpa_amd.wan import ti2v_timestep_tensor
    d, grid = 128, (3, 4, 5)
    cos, sin = rope_tables(grid, d, "cpu")
    freqs = torch.cat([ow.rope_params(1024, d - 4 * (d // 6)), ow.rope_params(1024, 2 * (d // 6)), ow.rope_params(1024, 2 * (d // 6))], dim=1)
    x = torch.randn(1, 60, 2, d, dtype=torch.float64)
    want = ow.rope_apply(x, grid, freqs)
    a, b = x[..., 0::2], x[..., 1::2]
    c, s = cos.double()[None, :, None], sin.double()[None, :, None]
    got = torch.stack([a * c - b * s, a * s + b * c], dim=-1).flatten(3)
    assert cos.shape == (60, 64) and (got - want).abs().max().item() < 1e-6       # tables are fp32
    t = torch.tensor([0.0, 3.0, 999.0])
    assert torch.equal(sinusoidal_embedding_1d(16, t), ow.sinusoidal_embedding_1d(16, t))
    # 03_train.py:119-125,181-187: zeros on the tokens of latent frame 0 (mask[:, ::2, ::2] of a [C, F, H, W] mask), t elsewhere
    tt = ti2v_timestep_tensor(torch.tensor([417, 20]), (4, 3, 8, 12), 72, (1, 2, 2))
    assert tt.shape == (2, 72) and (tt[:, :24] == 0).all() and (tt[0, 24:] == 417).all() and (tt[1, 24:] == 20).all()


def _oracle_run(state, lora=None, t=None, ctx_len=5, seed=1):
    g = torch.Generator().manual_seed(seed)
    x = [torch.randn(4, 2, 4, 6, generator=g, dtype=torch.float64)]
    L = 2 * 2 * 3
    t = torch.tensor([500.0]) if t is None else t
    ctx = [torch.randn(ctx_len, 12, generator=g, dtype=torch.float64)]
    return ow.forward(ow.Params(state, lora, dtype=torch.float64), dict(CFG), x, t, ctx, L)[0]


def test_oracle_known_answers():
    m = _model()
    state = {k: v.clone() for k, v in m.state_dict().items()}
    assert (_oracle_run(state) == 0).all()                                                   # zero output head (upstream init)
    g = torch.Generator().manual_seed(3)
    state["head.head.weight"] = torch.randn(state["head.head.weight"].shape, generator=g) * 0.1
    base = _oracle_run(state)
    assert base.shape == (4, 2, 4, 6) and base.abs().max() > 0
    A, B0 = torch.randn(4, 128, generator=g, dtype=torch.float64), torch.zeros(128, 4, dtype=torch.float64)
    assert torch.equal(_oracle_run(state, {"blocks.0.self_attn.q": (A, B0, 2.0)}), base)     # PEFT init: B = 0 leaves the model unchanged
    B1 = torch.randn(128, 4, generator=g, dtype=torch.float64) * 0.1
    assert not torch.allclose(_oracle_run(state, {"blocks.1.cross_attn.o": (A, B1, 2.0)}), base)
    # per-token timesteps: changing t on the first-frame tokens only changes the output (the modulation is per token) ...
    t_tok = torch.full((1, 12), 500.0)
    assert torch.allclose(_oracle_run(state, t=t_tok), base)
    t_tok[:, :6] = 0.0
    assert not torch.allclose(_oracle_run(state, t=t_tok), base)
    # ... and the zero-padded text positions ARE attended (k_lens = None upstream): a shorter prompt with explicit zero rows is the same input
    g2 = torch.Generator().manual_seed(1)
    _ = torch.randn(4, 2, 4, 6, generator=g2, dtype=torch.float64)
    c5 = torch.randn(5, 12, generator=g2, dtype=torch.float64)
    x = [torch.randn(4, 2, 4, 6, generator=torch.Generator().manual_seed(1), dtype=torch.float64)]
    P = ow.Params(state, None, dtype=torch.float64)
    o_short = ow.forward(P, dict(CFG), x, torch.tensor([500.0]), [c5], 12)[0]
    o_pad = ow.forward(P, dict(CFG), x, torch.tensor([500.0]), [torch.cat([c5, torch.zeros(3, 12, dtype=torch.float64)])], 12)[0]
    assert torch.equal(o_short, o_pad)


def test_from_pretrained_reads_the_upstream_checkpoint_layout(tmp_path):
    """train/Wan2.2-TI2V-5B/03_train.py:140,166 build policy and reference with `WanModel.from_pretrained(config['model_path'])`: config.json +
    diffusion_pytorch_model.safetensors, or indexed shards.  Round trip of a random model through both layouts: same config, bit-identical parameters,
    strict names; unknown config keys and a missing directory are errors."""
    import json
    import os
    import pytest
    from videogpa_amd.wan_model import WanModel
    m = _model()
    with torch.no_grad():
        m.head.head.weight.normal_(std=0.05)
    m = m.to(torch.bfloat16)
    one, many = str(tmp_path / "one"), str(tmp_path / "many")
    m.save_pretrained(one)
    m.save_pretrained(many, max_shard_size=40_000)
    assert sorted(os.listdir(one)) == ["config.json", "diffusion_pytorch_model.safetensors"]
    shards = [f for f in os.listdir(many) if f.endswith(".safetensors")]
    assert len(shards) > 2 and "diffusion_pytorch_model.safetensors.index.json" in os.listdir(many)
    cfg = json.load(open(os.path.join(one, "config.json")))
    assert cfg["_class_name"] == "WanModel" and cfg["dim"] == 128 and cfg["patch_size"] == [1, 2, 2]
    want = m.state_dict()
    for path in (one, many):
        got = WanModel.from_pretrained(path, torch_dtype="auto")                       # "auto": every tensor keeps its stored dtype
        assert not got.training and dict(got.config) == dict(m.config) and got.config.num_layers == 2
        sd = got.state_dict()
        assert sd.keys() == want.keys()
        assert all(sd[k].dtype == torch.bfloat16 and sd[k].device.type == "cpu" and torch.equal(sd[k], want[k]) for k in want)
    # torch_dtype=None is diffusers' default: float32 parameters whatever the file stores (the reference casts to bf16 right after, 03_train.py:141)
    for f32 in (WanModel.from_pretrained(one), WanModel.from_pretrained(one, torch_dtype=torch.float32)):
        sd = f32.state_dict()
        assert all(v.dtype == torch.float32 for v in sd.values()) and all(torch.equal(sd[k], want[k].float()) for k in want)
    assert all(torch.equal(v, want[k]) for k, v in WanModel.from_pretrained(one).to(torch.bfloat16).state_dict().items())
    # a checkpoint that mixes dtypes loads (upstream does): fp32 by default, as stored with "auto"
    from safetensors.torch import load_file, save_file
    mixed = str(tmp_path / "mixed")
    m.save_pretrained(mixed)
    raw = load_file(os.path.join(mixed, "diffusion_pytorch_model.safetensors"))
    raw["head.modulation"] = raw["head.modulation"].float()
    save_file(raw, os.path.join(mixed, "diffusion_pytorch_model.safetensors"))
    auto = WanModel.from_pretrained(mixed, torch_dtype="auto").state_dict()
    assert auto["head.modulation"].dtype == torch.float32 and auto["head.head.weight"].dtype == torch.bfloat16
    assert all(v.dtype == torch.float32 for v in WanModel.from_pretrained(mixed).state_dict().values())
    cfg["rope_scaling"] = 2
    json.dump(cfg, open(os.path.join(one, "config.json"), "w"))
    with pytest.raises(ValueError):
        WanModel.from_pretrained(one)
    with pytest.raises(FileNotFoundError):
        WanModel.from_pretrained(str(tmp_path / "nowhere"))
