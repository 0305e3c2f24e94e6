"""Decode the FWD_DIAG build of attn_fwd_kernel (timestamps written into the LSE buffer): how workgroups pair up on a
SIMD and whether co-resident waves run their tile loops in phase.  Needs var/lib_diag.so copied over libvgpa_hip.so."""
import os
import sys
from collections import defaultdict

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videogpa_amd import ops  # noqa: E402

B, H, S = 2, 48, 17776
g = torch.Generator(device="cuda").manual_seed(0)
qkv = torch.randn(B, S, 3, H, 64, generator=g, device="cuda").to(torch.bfloat16)
q = qkv[:, :, 0].permute(0, 2, 1, 3).contiguous()
k = qkv[:, :, 1].permute(0, 2, 1, 3).contiguous()
v = qkv[:, :, 2].permute(0, 2, 1, 3)
ops.attention_fwd_raw(q, k, v)
torch.cuda.synchronize()
o, lse = ops.attention_fwd_raw(q, k, v)
torch.cuda.synchronize()
n_wg = B * H * ((S + 255) // 256)
raw = lse.view(torch.int32).flatten()[: n_wg * 4 * 32].cpu().numpy().astype(np.uint32).reshape(n_wg, 4, 32)
hwid, xcc = raw[:, :, 0], raw[:, :, 1] & 0xF
start = raw[:, :, 2].astype(np.uint64) | (raw[:, :, 3].astype(np.uint64) << 32)
end = raw[:, :, 4].astype(np.uint64) | (raw[:, :, 5].astype(np.uint64) << 32)
st = raw[:, :, 16:32].astype(np.int64).reshape(n_wg, 4, 4, 4)   # [wg, wave, tile 128..131, stamp 0..3]
tt = st[:, :, :, 0]
t0 = start.min()
print("kernel span (cycles of s_memtime):", int(end.max() - t0))
dur = (end - start).astype(np.int64)
print("WG duration: mean %.0f min %d max %d" % (dur.mean(), dur.min(), dur.max()))
per = np.diff(tt, axis=2) & 0xFFFFFFFF
print("tile period per wave (cycles): mean %.0f  p10 %.0f p50 %.0f p90 %.0f" % (per.mean(), *np.percentile(per, [10, 50, 90])))
seg = np.diff(st, axis=3) & 0xFFFFFFFF          # top->after softmax, ->after PV, ->after barrier
nxt = (st[:, :, 1:, 0] - st[:, :, :-1, 3]) & 0xFFFFFFFF
for i, nm in enumerate(["loads+QK+softmax (stamp0->1)", "PV (1->2)", "tile store+barrier (2->3)"]):
    x = seg[:, :, :, i]
    print("  %-30s mean %.0f  p10 %.0f p50 %.0f p90 %.0f" % (nm, x.mean(), *np.percentile(x, [10, 50, 90])))
print("  %-30s mean %.0f" % ("loop back-edge (3->next 0)", nxt.mean()))
wave_id, simd_id, cu_id, sh_id, se_id = hwid & 0xF, (hwid >> 4) & 3, (hwid >> 8) & 0xF, (hwid >> 12) & 1, (hwid >> 13) & 7
print("wave_id histogram:", np.bincount(wave_id.flatten(), minlength=16))
print("simd_id of waves 0..3 in first WGs:", simd_id[:6].tolist())
print("same wave_id across the 4 waves of a WG: %.3f" % np.mean((wave_id == wave_id[:, :1]).all(axis=1)))
# group by physical SIMD
groups = defaultdict(list)
for w in range(n_wg):
    for j in range(4):
        groups[(int(xcc[w, j]), int(se_id[w, j]), int(sh_id[w, j]), int(cu_id[w, j]), int(simd_id[w, j]))].append((int(start[w, j] - t0), int(end[w, j] - t0), int(wave_id[w, j]), w, j))
print("distinct (xcc,se,sh,cu,simd):", len(groups))
# phase relation of co-resident waves: for each wave find the partner overlapping most in time on the same SIMD
rel = []
shown = 0
for key, lst in groups.items():
    lst.sort()
    for a in range(len(lst)):
        for b in range(a + 1, len(lst)):
            s1, e1, _, w1, j1 = lst[a]
            s2, e2, _, w2, j2 = lst[b]
            if s2 >= e1:
                break
            # both inside tiles 128..135 at overlapping times?
            ta, tb = tt[w1, j1], tt[w2, j2]
            p = float(np.median(np.diff(ta) & 0xFFFFFFFF))
            d = ((tb[0] - ta[0]) & 0xFFFFFFFF)
            if d > 1 << 31:
                d -= 1 << 32
            if abs(d) < 20 * p:
                rel.append((d % p) / p)
    if shown < 3:
        print(key, [(s, e, wid, w) for s, e, wid, w, _ in lst[:6]])
        shown += 1
rel = np.array(rel)
print("pairs analysed:", len(rel))
print("phase offset (fraction of a tile period) histogram, 10 bins:", np.histogram(rel, bins=10, range=(0, 1))[0])
