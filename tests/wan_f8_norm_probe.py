"""Not a test: the three e4m3 attention models of oracle/wan.py at the full-width configuration of tests/test_gpu_wan_cfg1.py, each against the plain fp32
oracle -- how far from fp32 the self-attention q / k adapter gradients sit with (a) the round-4 backward, (b) the consistent backward (the device), (c) the
output normalised by the sum of the QUANTISED weights (what a further kernel change would buy).   python tests/wan_f8_norm_probe.py"""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import test_gpu_wan_cfg1 as w  # noqa: E402

p = w._oracle()[1]
out = {}
for tag in ("r4", True, "lq"):
    g = w._oracle(round_activations=True, exact_delta=True, fp8_ffn=True, f8_attn=tag)[1]
    out[str(tag)] = {n + "." + "AB"[i]: round(w._rel(g[n][i], p[n][i]), 5) for n in sorted(g) if ".self_attn." in n and n[-1] in "qk" for i in range(2)}
print(json.dumps(out, indent=1))
