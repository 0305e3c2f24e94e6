"""Where does the block-1 q/k LoRA-gradient error of the cfg1 parity test come from?  Records the attention-backward and
QK-norm-backward calls of the HIP step and re-computes each one in fp32 torch on the GPU from the SAME bf16 inputs."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import cfg1_common as c1
from videogpa_amd import ops
import test_gpu_cfg1 as tg

rec = []
orig_bwd = ops.attention_bwd_raw


def spy(q, k, v, o, do, lse, dq, dk, dv, **kw):
    orig_bwd(q, k, v, o, do, lse, dq, dk, dv, **kw)
    rec.append(dict(q=q.clone(), k=k.clone(), v=v.clone(), o=o.clone(), do=do.clone(), lse=lse.clone(), dq=dq.clone(), dk=dk.clone(), dv=dv.clone()))


ops.attention_bwd_raw = spy
variant = sys.argv[1] if len(sys.argv) > 1 else "r64"
tg._hip_step(variant)
LOG2E = 1.4426950408889634


def cmp(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return f"cos {float((a * b).sum() / (a.norm() * b.norm())):.6f} |a|/|b| {float(a.norm() / b.norm()):.4f} |b| {float(b.norm()):.3e}"


for li, r in enumerate(rec):          # backward order: last block first
    print(f"--- attention backward call {li} (block {len(rec) - 1 - li})")
    for b in range(r["q"].shape[0]):
        for h in (0, 17, 47):
            q = (r["q"][b, h].float() / (0.125 * LOG2E)).requires_grad_(True)
            k = r["k"][b, h].float().requires_grad_(True)
            v = r["v"][b, h].float().requires_grad_(True)
            do = r["do"][b, h].float()
            s = (q @ k.t()) * 0.125
            p = torch.softmax(s, dim=-1)
            o = p @ v
            o.backward(do)
            lse_ref = torch.logsumexp(s.detach(), dim=-1) * LOG2E
            print(f" b{b} h{h:2d}: o {cmp(r['o'][b, h], o.detach())} | lse maxdiff {float((r['lse'][b, h] - lse_ref).abs().max()):.2e} | pmax {float(p.max()):.3e}")
            print(f"          dq {cmp(r['dq'][b, h], q.grad)}")
            print(f"          dk {cmp(r['dk'][b, h], k.grad)}")
            print(f"          dv {cmp(r['dv'][b, h], v.grad)}")
            # the same backward, but with every operand the kernel rounds to bf16 rounded here too (P, dS): the bf16 floor
            with torch.no_grad():
                pb = p.detach().bfloat16().float()
                dp = do @ v.detach().t()
                delta = (do * r["o"][b, h].float()).sum(-1, keepdim=True)
                ds = (pb * (dp - delta)).bfloat16().float()
                dq_b = (ds @ k.detach()) * 0.125
                dk_b = (ds.t() @ q.detach()) * 0.125
            print(f"          dq(bf16 P,dS model) {cmp(dq_b, q.grad)} ; dk {cmp(dk_b, k.grad)}")

# ---- plain-torch bf16 run of the same step on the GPU (oracle code, bf16 weights / activations): the bf16 noise floor
from oracle import cogvideox as ocv, scheduler as osch
del rec
torch.cuda.empty_cache()
cfg = c1.config()
sd = {k: v.cuda() for k, v in c1.base_state_dict(cfg).items()}
lora, r = c1.lora_state_dict(cfg, variant)
lora = {k: v.cuda().requires_grad_(True) for k, v in lora.items()}
xw, xl, prompt, t, noise = c1.inputs()
out = ocv.dpo_pair_step(sd, cfg, lora, osch.alphas_cumprod().cuda(), xw.cuda(), xl.cuda(), prompt.cuda(), t.cuda(), noise.cuda(), beta=1.0)
out["loss"].backward()
gold = torch.load(os.path.join(ROOT, "tests", "golden", f"cfg1_{variant}.pt"), weights_only=False)
print(f"--- plain torch bf16 (oracle code on the GPU) vs fp32 golden: loss {float(out['loss']):.7f} vs {float(gold['loss']):.7f}")
for k, p in lora.items():
    ref = gold["lora_grads"][k]
    g = p.grad.float().cpu()
    idx = c1.sample_index(g.numel(), k)
    got, rs = g.flatten()[idx].double(), ref["samples"].double()
    print(f"  {k.replace('base_model.model.transformer_blocks.', ''):32s} norm_rel {abs(float(g.double().norm()) / float(ref['norm']) - 1):.4f} "
          f"sample_err/max {float((got - rs).abs().max()) / float(ref['absmax']):.4f} cos {float((got * rs).sum() / (got.norm() * rs.norm())):.6f}")
