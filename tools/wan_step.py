"""One Diffusion-DPO pair step of Wan2.2-TI2V-5B at BASELINE.json configs[5] shapes (81 frames x 704 x 1280 -> latent 48 x 21 x 44 x 80,
18480 tokens, 30 layers x dim 3072 x 24 heads of 128, text 512, LoRA r64 on q/k/v/o, block checkpointing), random-init weights, batch 1:
step time + per-kernel table (the KernelTimer of bench.py).   gpurun -- 'PYTHONPATH=. python tools/wan_step.py [--layers N] [--steps K]'"""
import argparse
import json
import time

import torch

from videogpa_amd import ops
from videogpa_amd.wan import WanDPOTrainer
from videogpa_amd.wan_model import WanModel

ap = argparse.ArgumentParser()
ap.add_argument("--layers", type=int, default=30)
ap.add_argument("--steps", type=int, default=2)
ap.add_argument("--frames", type=int, default=21)
ap.add_argument("--out", default="")
ap.add_argument("--fp8", action="store_true")
ap.add_argument("--ckpt-stride", type=int, default=1, help="recompute every k-th block; 0 = keep all activations (no recompute)")
a = ap.parse_args()
torch.manual_seed(0)
dev = "cuda"
with torch.device(dev):
    torch.set_default_dtype(torch.bfloat16)
    m = WanModel(num_layers=a.layers)
    torch.set_default_dtype(torch.float32)
with torch.no_grad():
    torch.nn.init.normal_(m.head.head.weight, std=0.02)
m.enable_gradient_checkpointing(a.ckpt_stride > 0, max(1, a.ckpt_stride))
m.enable_fp8(a.fp8)
tr = WanDPOTrainer({}, m)
opt = tr.configure_optimizers()
g = torch.Generator(device=dev).manual_seed(1)
batch = {"x_win": torch.randn(1, 48, a.frames, 44, 80, device=dev, generator=g).bfloat16(), "x_lose": torch.randn(1, 48, a.frames, 44, 80, device=dev, generator=g).bfloat16(),
         "prompt_emb": torch.randn(1, 300, 4096, device=dev, generator=g).bfloat16(), "image_latent": torch.randn(1, 48, 1, 44, 80, device=dev, generator=g).bfloat16()}


def step():
    loss, logs = tr.training_step(batch)
    loss.backward()
    opt.step()
    opt.zero_grad()
    return loss


loss = step()                       # warmup (hipBLASLt heuristics, LoRA caches)
torch.cuda.synchronize()
t0 = time.time()
for _ in range(a.steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.time() - t0) / a.steps
ops.TIMER = ops.KernelTimer()
step()
torch.cuda.synchronize()
summ = ops.TIMER.summary()
ops.TIMER = None
tot = sum(v["total_ms"] for v in summ.values())
res = {"workload": f"Wan2.2-TI2V-5B pair step, {a.layers} layers, 48x{a.frames}x44x80 latent, batch 1, LoRA r64, " + ("no recompute" if a.ckpt_stride == 0 else f"recompute every {a.ckpt_stride}. block") + (", fp8 feed-forward" if a.fp8 else ""), "s_per_step": dt, "pair_steps_per_s": 1 / dt,
       "loss": float(loss.detach()), "peak_mem_GB": torch.cuda.max_memory_allocated() / 1e9, "timed_kernels_ms": tot, "kernels": {}}
print(f"{res['workload']}: {dt:.3f} s/step, loss {float(loss.detach()):.4f}, peak {res['peak_mem_GB']:.1f} GB; timed kernels {tot:.0f} ms")
for k, v in sorted(summ.items(), key=lambda kv: -kv[1]["total_ms"]):
    rate = v["work_per_launch"] / (v["avg_ms"] * 1e-3)
    r = f"{rate / 1e12:.0f} TFLOP/s" if v["unit"] == "flop" else f"{rate / 1e9:.0f} GB/s"
    res["kernels"][k] = {"launches": v["launches"], "avg_ms": v["avg_ms"], "total_ms": v["total_ms"], "rate": r}
    print(f"  {k:34s} {v['launches']:5d} x {v['avg_ms']:8.3f} ms = {v['total_ms']:8.1f} ms   {r}")
if a.out:
    json.dump(res, open(a.out, "w"), indent=1)
