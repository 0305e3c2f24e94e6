"""Decode the -DFWD_DIAG build of attn_fwd_kernel (s_memtime stamps + HW_ID written into the LSE buffer): how workgroups
pair up on a SIMD, how long the phases of a tile take, and whether co-resident waves run their tile loops in phase.

    tools/build_variant.sh diag -DFWD_V1 -DFWD_DIAG ; cp var/lib_diag.so videogpa_amd/csrc/libvgpa_hip.so ; python tools/fwd_diag.py
    (add -DFWD_DYN_LDS=90000 for one workgroup per CU)
"""
import os
import sys
from collections import defaultdict

import numpy as np

B, H, S = 2, 48, 17776
N_WG = B * H * ((S + 255) // 256)
M32 = 0xFFFFFFFF


def decode(raw):
    """raw: uint32 [n_wg, 4 waves, 32 words] as written by the kernel."""
    n_wg = raw.shape[0]
    hwid, xcc = raw[:, :, 0], raw[:, :, 1] & 0xF
    start = raw[:, :, 2].astype(np.int64) | (raw[:, :, 3].astype(np.int64) << 32)
    end = raw[:, :, 4].astype(np.int64) | (raw[:, :, 5].astype(np.int64) << 32)
    st = raw[:, :, 16:32].astype(np.int64).reshape(n_wg, 4, 4, 4)   # [wg, wave, tile 128..131, stamp 0..3]
    tt = st[:, :, :, 0]
    t0 = int(start[start > 0].min())
    print("kernel span (s_memtime cycles):", int(end.max()) - t0)
    dur = end - start
    print("WG duration: mean %.0f min %d max %d" % (dur.mean(), dur.min(), dur.max()))
    per = np.diff(tt, axis=2) & M32
    print("tile period per wave (cycles): mean %.0f  p10 %.0f p50 %.0f p90 %.0f" % (per.mean(), *np.percentile(per, [10, 50, 90])))
    seg = np.diff(st, axis=3) & M32
    nxt = (st[:, :, 1:, 0] - st[:, :, :-1, 3]) & M32
    for i, nm in enumerate(["prefetch issue + QK^T + softmax (stamp 0->1)", "LDS store + PV (1->2)", "barrier (2->3)"]):
        x = seg[:, :, :, i]
        print("  %-46s mean %.0f  p10 %.0f p50 %.0f p90 %.0f" % (nm, x.mean(), *np.percentile(x, [10, 50, 90])))
    print("  %-46s mean %.0f" % ("loop edge + next prefetch issue (3->next 0)", nxt.mean()))
    wave_id, simd_id, cu_id, sh_id, se_id = hwid & 0xF, (hwid >> 4) & 3, (hwid >> 8) & 0xF, (hwid >> 12) & 1, (hwid >> 13) & 7
    print("wave slot histogram:", np.bincount(wave_id.flatten().astype(np.int64), minlength=4)[:4])
    print("SIMD of waves 0..3 in the first workgroups:", simd_id[:4].tolist())
    groups = defaultdict(list)
    for w in range(n_wg):
        for j in range(4):
            key = (int(xcc[w, j]), int(se_id[w, j]), int(sh_id[w, j]), int(cu_id[w, j]), int(simd_id[w, j]))
            groups[key].append((int(start[w, j]) - t0, int(end[w, j]) - t0, w, j))
    print("distinct (xcc, se, sh, cu, simd):", len(groups))
    # arbitration between the waves that share a SIMD: tile period of the earlier-started ("older") and the later wave
    older, younger, lag = [], [], []
    for lst in groups.values():
        lst.sort()
        for a in range(len(lst) - 1):
            s1, e1, w1, j1 = lst[a]
            s2, e2, w2, j2 = lst[a + 1]
            if s2 >= e1 or s2 - s1 > 0.2 * (e1 - s1):
                continue      # not co-resident for (almost) the whole life of the older one
            ta, tb = tt[w1, j1], tt[w2, j2]
            older.append(float(np.mean(np.diff(ta) & M32)))
            younger.append(float(np.mean(np.diff(tb) & M32)))
            d = int((int(tb[0]) - int(ta[0])) & M32)
            lag.append(d - (1 << 32) if d > 1 << 31 else d)
    print("wave pairs that started together on one SIMD:", len(older))
    if older:
        print("  tile period of the older wave %.0f, of the younger %.0f cycles; the younger reaches tile 128 %.0f cycles later"
              % (np.mean(older), np.mean(younger), np.mean(lag)))
    return len(older)


def selftest():
    rng = np.random.default_rng(0)
    raw = np.zeros((64, 4, 32), dtype=np.uint32)
    for w in range(64):
        for j in range(4):
            cu, slot = (w // 2) % 16, w % 2
            raw[w, j, 0] = slot | (j << 4) | (cu << 8)
            s0 = 1_000_000 + (w // 32) * 1_200_000 + slot * 1500
            raw[w, j, 2], raw[w, j, 4] = s0, s0 + 1_100_000
            base = s0 + 500_000
            for t in range(4):
                for k in range(4):
                    raw[w, j, 16 + 4 * t + k] = base + 4000 * t + 1000 * k + int(rng.integers(0, 50))
    assert decode(raw) > 0


if __name__ == "__main__":
    if "--selftest" in sys.argv:
        selftest()
        sys.exit(0)
    import torch

    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from videogpa_amd import ops  # noqa: E402

    g = torch.Generator(device="cuda").manual_seed(0)
    qkv = torch.randn(B, S, 3, H, 64, generator=g, device="cuda").to(torch.bfloat16)
    q = qkv[:, :, 0].permute(0, 2, 1, 3).contiguous()
    k = qkv[:, :, 1].permute(0, 2, 1, 3).contiguous()
    v = qkv[:, :, 2].permute(0, 2, 1, 3)
    ops.attention_fwd_raw(q, k, v)
    torch.cuda.synchronize()
    o, lse = ops.attention_fwd_raw(q, k, v)
    torch.cuda.synchronize()
    decode(lse.view(torch.int32).flatten()[: N_WG * 4 * 32].cpu().numpy().view(np.uint32).reshape(N_WG, 4, 32))
