import torch, torch.nn.functional as F, sys, os
sys.path.insert(0, os.getcwd())
from videogpa_amd import ops
M, K, N = 2*17776, 3072, 12288
x = torch.randn(M, K, device="cuda").bfloat16()
W = (0.02*torch.randn(N, K, device="cuda")).bfloat16()
b = (0.02*torch.randn(N, device="cuda")).bfloat16()
def t(f, n=10):
    f(); torch.cuda.synchronize()
    a, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): f()
    e.record(); torch.cuda.synchronize()
    return a.elapsed_time(e)/n
print("linear+bias        %.3f ms" % t(lambda: F.linear(x, W, b)))
print("addmm_act gelu     %.3f ms" % t(lambda: torch._addmm_activation(b, x, W.t(), use_gelu=True)))
u = F.linear(x, W, b)
print("gelu kernel        %.3f ms" % t(lambda: ops.gelu_tanh(u)))
g1 = torch._addmm_activation(b, x, W.t(), use_gelu=True)
g2 = F.gelu(u.float(), approximate="tanh")
g3 = F.gelu((x.float() @ W.float().t() + b.float()), approximate="tanh")
print("fused vs tanh-gelu(bf16 u): max abs %.4g ; vs fp32 chain %.4g ; erf-gelu diff %.4g" % ((g1.float()-g2).abs().max().item(), (g1.float()-g3).abs().max().item(), (g1.float()-F.gelu(u.float())).abs().max().item()))
