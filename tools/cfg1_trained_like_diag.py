import sys, os, json
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch
from videogpa_amd import ops
import test_gpu_cfg1 as t
res = {}
for mode in ("bound", "online"):
    orig = ops.AttnFwdPolicy.__init__
    def init(self, mode_=mode, fixed=True, _o=orig):
        _o(self, mode=mode_, fixed=True)
    ops.AttnFwdPolicy.__init__ = init
    try:
        out, preds, grads = t._hip_step("r64", qk_gain=2.5)
    finally:
        ops.AttnFwdPolicy.__init__ = orig
    res[mode] = (out.loss.item(), grads)
    print(mode, "loss", out.loss.item())
ref_loss, ref_grads = t._oracle_on_gpu("r64", torch.float32, qk_gain=2.5)
print("fp32 loss", ref_loss)
def rel(a, r): return float((a.double() - r.double()).norm() / r.double().norm())
for k in list(ref_grads)[:16]:
    print(k.replace("base_model.model.transformer_blocks.", ""), "bound", round(rel(res["bound"][1][k], ref_grads[k]), 4), "online", round(rel(res["online"][1][k], ref_grads[k]), 4),
          "bound-vs-online", round(rel(res["bound"][1][k], res["online"][1][k]), 4))
