"""PEFT-compatible LoRA for the MI355X transformer (drop-in for the `peft` calls the reference makes:
`LoraConfig` / `get_peft_model` train/CogVideoX-5B/03_train.py:102-106, `save_pretrained` :286-287,
`PeftModel.from_pretrained(...).merge_and_unload()` generate/CogVideoX-5B.py:29-30, `module.scaling[adapter]`
overrides generate/CogVideoX1.5-5B.py:30-36).

On-disk format = PEFT's: `adapter_config.json` (fields as checkpoints/VideoGPA-T2V-lora/adapter_config.json) +
`adapter_model.safetensors` with keys `base_model.model.<module path>.lora_{A,B}.weight`
(A [r, in], B [out, r]).  Semantics follow peft.tuners.lora.layer.Linear: y = base(x) + B(A(x)) * (alpha / r),
A ~ kaiming_uniform(a=sqrt 5), B = 0, adapter weights fp32, cast to the activation dtype at use.
"""
import contextlib
import json
import math
import os
from dataclasses import asdict, dataclass, field
from typing import List, Optional, Union

import torch
import torch.nn as nn
import torch.nn.functional as F


MAX_LORA_RANK = 64


@dataclass
class LoraConfig:
    r: int = 8
    lora_alpha: float = 8
    lora_dropout: float = 0.0
    target_modules: Optional[Union[List[str], str]] = None
    bias: str = "none"
    init_lora_weights: bool = True
    fan_in_fan_out: bool = False
    inference_mode: bool = False
    use_rslora: bool = False
    use_dora: bool = False
    base_model_name_or_path: Optional[str] = None
    task_type: Optional[str] = None
    peft_type: str = "LORA"
    modules_to_save: Optional[List[str]] = None
    rank_pattern: dict = field(default_factory=dict)
    alpha_pattern: dict = field(default_factory=dict)

    def to_dict(self):
        d = asdict(self)
        if isinstance(d["target_modules"], (set, tuple)):
            d["target_modules"] = list(d["target_modules"])
        # remaining keys PEFT writes (kept so real PEFT can load the file)
        d.update({"auto_mapping": {"base_model_class": "CogVideoXTransformer3DModel",
                                   "parent_library": "diffusers.models.transformers.cogvideox_transformer_3d"},
                  "corda_config": None, "eva_config": None, "exclude_modules": None, "layer_replication": None,
                  "layers_pattern": None, "layers_to_transform": None, "loftq_config": {}, "lora_bias": False,
                  "megatron_config": None, "megatron_core": "megatron.core", "revision": None,
                  "trainable_token_indices": None})
        return d

    @classmethod
    def from_dict(cls, d):
        known = {k: v for k, v in d.items() if k in cls.__dataclass_fields__}
        return cls(**known)


class LoraLinear(nn.Module):
    def __init__(self, base_layer: nn.Linear):
        super().__init__()
        self.base_layer = base_layer
        self.in_features, self.out_features = base_layer.in_features, base_layer.out_features
        self.lora_A = nn.ModuleDict()
        self.lora_B = nn.ModuleDict()
        self.r, self.lora_alpha, self.scaling = {}, {}, {}
        self.active_adapters = []
        self.disable_adapters = False
        self.merged_adapters = []

    @property
    def weight(self):
        return self.base_layer.weight

    @property
    def bias(self):
        return self.base_layer.bias

    @property
    def merged(self):
        return bool(self.merged_adapters)

    def update_layer(self, name, r, lora_alpha, init=True, use_rslora=False):
        dev = self.base_layer.weight.device
        self.r[name], self.lora_alpha[name] = r, lora_alpha
        self.scaling[name] = lora_alpha / math.sqrt(r) if use_rslora else lora_alpha / r
        self.lora_A[name] = nn.Linear(self.in_features, r, bias=False, device=dev, dtype=torch.float32)
        self.lora_B[name] = nn.Linear(r, self.out_features, bias=False, device=dev, dtype=torch.float32)
        if init:
            nn.init.kaiming_uniform_(self.lora_A[name].weight, a=math.sqrt(5))
            nn.init.zeros_(self.lora_B[name].weight)
        if name not in self.active_adapters:
            self.active_adapters.append(name)

    def active_lora(self):
        """(A [r,in], B [out,r], scaling) of the active adapter, or None when disabled / merged / absent."""
        if self.disable_adapters or self.merged or not self.active_adapters:
            return None
        if len(self.active_adapters) != 1:
            raise NotImplementedError("one active adapter at a time")
        n = self.active_adapters[0]
        return self.lora_A[n].weight, self.lora_B[n].weight, self.scaling[n]

    def delta_weight(self, name):
        return (self.lora_B[name].weight.float() @ self.lora_A[name].weight.float()) * self.scaling[name]

    def merge(self):
        for n in self.active_adapters:
            if n not in self.merged_adapters:
                w = self.base_layer.weight
                with torch.no_grad():
                    w.add_(self.delta_weight(n).to(w.dtype))   # in place + version bump (invalidates fused-QKV caches)
                self.merged_adapters.append(n)

    def unmerge(self):
        while self.merged_adapters:
            n = self.merged_adapters.pop()
            w = self.base_layer.weight
            with torch.no_grad():
                w.sub_(self.delta_weight(n).to(w.dtype))

    def forward(self, x):
        """peft.tuners.lora.layer.Linear.forward.  On the GPU in bf16 (the training path, e.g. the q/k/v/o linears of a Wan
        model wrapped by get_peft_model) the base product goes to hipBLASLt and the A / B contractions to the MFMA kernels of
        csrc/lora.hip through ops.linear_lora; the torch expression below only serves host-side use (CPU adapter-format and
        merge checks on CPU tensors).  A GPU tensor that is not bf16 raises: nothing on the device runs through torch eager."""
        lora = self.active_lora()
        if x.is_cuda and x.dtype == torch.bfloat16 and self.base_layer.weight.dtype == torch.bfloat16:
            from . import ops
            W, b = self.base_layer.weight, self.base_layer.bias
            if self.merged or len(self.active_adapters) != 1:
                return ops.frozen_linear(x, W, b)
            n = self.active_adapters[0]
            if getattr(self, "_ext", None) is None:
                self._ext = ops.LoraExt()
            # adapter switched off (frozen-reference pass): the SAME extended GEMM with a zero LoRA tail, so policy and reference
            # share the base partial sums bit for bit (ops.LoraExt)
            return ops.linear_lora_ext(x, W, b, self._ext, [(self.lora_A[n].weight, self.lora_B[n].weight, self.scaling[n])],
                                       enabled=not self.disable_adapters)
        if x.is_cuda:
            raise RuntimeError(f"videogpa_amd LoraLinear: the GPU path computes in bf16 (got activations {x.dtype}, base weight "
                               f"{self.base_layer.weight.dtype}); cast the model and its inputs to torch.bfloat16 -- there is no eager GPU fallback")
        y = F.linear(x, self.base_layer.weight, self.base_layer.bias)
        if lora is not None:
            A, B, s = lora
            y = y + F.linear(F.linear(x, A.to(x.dtype)), B.to(x.dtype)) * s
        return y


def _matches(name, targets):
    if isinstance(targets, str):
        import re
        return re.fullmatch(targets, name) is not None
    return any(name == t or name.endswith("." + t) for t in targets)


class LoraModel(nn.Module):
    def __init__(self, model, config: LoraConfig, adapter_name="default", init=True):
        super().__init__()
        self.model = model
        self.peft_config = {adapter_name: config}
        self.inject(config, adapter_name, init)

    def inject(self, config, adapter_name, init=True):
        if not config.target_modules:
            raise ValueError("LoraConfig.target_modules must be given")
        # accepted-but-ignored settings would train something other than what the config says: refuse them up front
        if config.lora_dropout and config.lora_dropout > 0:
            raise NotImplementedError("lora_dropout > 0 is not implemented on this path (every reference config uses 0.0)")
        if config.rank_pattern or config.alpha_pattern:
            raise NotImplementedError("rank_pattern / alpha_pattern are not implemented (the released adapters use neither)")
        if config.use_dora or config.bias != "none":
            raise NotImplementedError("use_dora / bias != 'none' are not implemented")
        if config.r > MAX_LORA_RANK:
            raise NotImplementedError(f"LoRA rank {config.r} > {MAX_LORA_RANK}: the MFMA adapter kernels (csrc/lora.hip) cover padded "
                                      f"ranks up to {MAX_LORA_RANK} (three adapters share one down-projection of width 3 x rank)")
        hit = 0
        for name, mod in list(self.model.named_modules()):
            if not _matches(name, config.target_modules):
                continue
            if isinstance(mod, LoraLinear):
                mod.update_layer(adapter_name, config.r, config.lora_alpha, init and config.init_lora_weights, config.use_rslora)
                hit += 1
                continue
            if not isinstance(mod, nn.Linear):
                continue
            parent_name, _, child = name.rpartition(".")
            parent = self.model.get_submodule(parent_name) if parent_name else self.model
            wrapped = LoraLinear(mod)
            wrapped.update_layer(adapter_name, config.r, config.lora_alpha, init and config.init_lora_weights, config.use_rslora)
            if isinstance(parent, (nn.ModuleList, nn.Sequential)) and child.isdigit():
                parent[int(child)] = wrapped
            else:
                setattr(parent, child, wrapped)
            hit += 1
        if hit == 0:
            raise ValueError(f"Target modules {config.target_modules} not found in the base model.")
        for n, p in self.model.named_parameters():
            p.requires_grad_(".lora_A." in n or ".lora_B." in n)

    def forward(self, *a, **k):
        return self.model(*a, **k)


class PeftModel(nn.Module):
    def __init__(self, model, peft_config: LoraConfig, adapter_name="default", _init=True):
        super().__init__()
        self.base_model = LoraModel(model, peft_config, adapter_name, _init)
        self.peft_config = self.base_model.peft_config
        self.active_adapter = adapter_name

    def __getattr__(self, name):
        try:
            return super().__getattr__(name)
        except AttributeError:
            return getattr(self.base_model.model, name)

    def forward(self, *a, **k):
        return self.base_model.model(*a, **k)

    def get_base_model(self):
        return self.base_model.model

    def lora_layers(self):
        return [m for m in self.base_model.model.modules() if isinstance(m, LoraLinear)]

    @contextlib.contextmanager
    def disable_adapter(self):
        layers = self.lora_layers()
        old = [l.disable_adapters for l in layers]
        for l in layers:
            l.disable_adapters = True
        try:
            yield
        finally:
            for l, o in zip(layers, old):
                l.disable_adapters = o

    def print_trainable_parameters(self):
        tr = sum(p.numel() for p in self.parameters() if p.requires_grad)
        al = sum(p.numel() for p in self.parameters())
        print(f"trainable params: {tr:,d} || all params: {al:,d} || trainable%: {100 * tr / al:.4f}")

    def adapter_state_dict(self, adapter_name=None):
        adapter_name = adapter_name or self.active_adapter
        out = {}
        for k, v in self.state_dict().items():
            tag = f".{adapter_name}.weight"
            if (".lora_A." in k or ".lora_B." in k) and k.endswith(tag):
                out[k[: -len(tag)] + ".weight"] = v.detach()
        return out

    def save_pretrained(self, save_directory, safe_serialization=True, **kw):
        from safetensors.torch import save_file
        os.makedirs(save_directory, exist_ok=True)
        cfg = self.peft_config[self.active_adapter].to_dict()
        cfg["inference_mode"] = True
        with open(os.path.join(save_directory, "adapter_config.json"), "w") as f:
            json.dump(cfg, f, indent=2, sort_keys=True)
        sd = {k: v.cpu().contiguous() for k, v in self.adapter_state_dict().items()}
        save_file(sd, os.path.join(save_directory, "adapter_model.safetensors"), metadata={"format": "pt"})

    @classmethod
    def from_pretrained(cls, model, model_id, adapter_name="default", is_trainable=False, **kw):
        from safetensors.torch import load_file
        with open(os.path.join(model_id, "adapter_config.json")) as f:
            cfg = LoraConfig.from_dict(json.load(f))
        cfg.inference_mode = not is_trainable
        pm = cls(model, cfg, adapter_name, _init=False)
        sd = load_file(os.path.join(model_id, "adapter_model.safetensors"))
        own = pm.state_dict()
        loaded = 0
        for k, v in sd.items():
            kk = k[: -len(".weight")] + f".{adapter_name}.weight"
            if kk not in own:
                raise KeyError(f"unexpected adapter key {k}")
            own[kk].copy_(v.to(own[kk].dtype))
            loaded += 1
        expected = sum(1 for k in own if ".lora_A." in k or ".lora_B." in k)
        if loaded != expected:
            raise RuntimeError(f"adapter file holds {loaded} tensors, model expects {expected}")
        if not is_trainable:
            for p in pm.parameters():
                p.requires_grad_(False)
        return pm

    def merge_and_unload(self):
        """W += scaling * B A for every wrapped linear, then strip the wrappers; returns the base model."""
        model = self.base_model.model
        for name, mod in list(model.named_modules()):
            if isinstance(mod, LoraLinear):
                mod.merge()
                parent_name, _, child = name.rpartition(".")
                parent = model.get_submodule(parent_name) if parent_name else model
                if isinstance(parent, (nn.ModuleList, nn.Sequential)) and child.isdigit():
                    parent[int(child)] = mod.base_layer
                else:
                    setattr(parent, child, mod.base_layer)
        return model


def get_peft_model(model, peft_config: LoraConfig, adapter_name="default"):
    return PeftModel(model, peft_config, adapter_name)
