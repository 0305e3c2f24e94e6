#!/usr/bin/env python3
"""Generator of the hand-scheduled main loops of the "w1" attention kernels (videogpa_amd/csrc/attention_w1.hip).

    python tools/gen_w1_asm.py            # rewrites videogpa_amd/csrc/w1_*_loop.inc
    python tools/gen_w1_asm.py --check    # exit 1 if a committed .inc differs from what this script generates

Why a generator: with one wave per SIMD nothing but the wave's own instruction order overlaps the matrix pipe with the
VALU / LDS work, and hipcc's register allocator answers a 300+-register software pipeline with hundreds of tuple copies per
iteration (measured on the C++ form of this loop: 94 v_accvgpr_read + 80 v_accvgpr_write + 37 v_mov_b64 per 52 MFMAs).  So
the loop is emitted as ONE inline-asm statement with explicit registers: every MFMA is followed by its share of the VALU
work (about three single-issue instructions per 32-cycle matrix-pipe slot, never more than five), LDS fragment reads are
issued LEAD MFMAs ahead with counted `s_waitcnt lgkmcnt(N)`, LDS-DMA runs two tiles ahead with counted `vmcnt`.  The C++
around it (prologue, epilogue, tail-split bookkeeping) stays compiler-made.

Conventions shared by all loops (see attn_w1.h for the LDS image):
  ring slot s (0..3) at LDS byte s * 16384: operand X tile at +0, operand Y tile at +8192; 32-row block rb at + 4096 rb;
  lane-constant read offsets come in as v[LA..LA+7] = {row[ks = 0..3], tr[db = 0][r3 = 0,1], tr[db = 1][r3 = 0,1]}.
Hazards honoured by construction (LLVM GCNHazardRecognizer rules for gfx940/950): an MFMA result is read by the VALU only
in the NEXT half-step (>= 8 MFMAs later); a VALU-written MFMA operand is consumed >= 8 MFMAs later; a v_exp result is never
used by the very next instruction; the same accumulator is never the destination of two consecutive MFMAs; M0 is written
one instruction slot before the LDS-DMA that reads it.
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MFMA = "v_mfma_f32_32x32x16_bf16"
# Timing-only ablations for tools/w1_variants.sh (results are WRONG with any of them): W1_ABLATE=novalu,nolds,nosync,nomfma
ABLATE = set(x for x in os.environ.get("W1_ABLATE", "").split(",") if x)
KNOB = {k: int(v) for k, v in (kv.split("=") for kv in os.environ.get("W1_KNOBS", "").split(",") if kv)}


def vr(lo, n=1):
    return f"v{lo}" if n == 1 else f"v[{lo}:{lo + n - 1}]"


def ar(lo, n=1):
    return f"a{lo}" if n == 1 else f"a[{lo}:{lo + n - 1}]"


class Emitter:
    """Collects instructions and keeps the LGKM (LDS) queue so waits can be counted exactly."""

    def __init__(self):
        self.lines = []
        self.lgkm = []          # tags of outstanding LDS reads, oldest first

    def raw(self, text):
        if KNOB.get("noexp") and text.startswith("v_exp_f32"):        # timing experiment: a plain VALU op in the place of every exp
            text = text.replace("v_exp_f32", "v_mov_b32")
        if KNOB.get("nomul") and (text.startswith("v_mul_f32") or text.startswith("v_add_f32")):
            return
        if KNOB.get("nocvt") and text.startswith("v_cvt_pk"):
            return
        self.lines.append(text)

    def ds(self, text, tag):
        if "nolds" in ABLATE:
            return
        self.lines.append(text)
        self.lgkm.append(tag)

    def wait_tag(self, tag):
        """all LDS reads carrying `tag` must have returned: lgkmcnt(number of reads issued after the last of them)"""
        idx = [i for i, t in enumerate(self.lgkm) if t == tag]
        if not idx:
            return
        last = idx[-1]
        n_after = len(self.lgkm) - 1 - last
        self.lines.append(f"s_waitcnt lgkmcnt({n_after})")
        self.lgkm = self.lgkm[last + 1:]

    def drain(self):
        self.lgkm = []

    def arrived(self, tag):
        return tag not in self.lgkm

    def retag(self, mapping):
        self.lgkm = [mapping.get(t, t) for t in self.lgkm]

    def text(self):
        return "\n".join('    "%s\\n\\t"' % ln for ln in self.lines)


# ------------------------------------------------------------------------------------------------------ stream scheduling
def schedule(em, mfmas, frag_issue, frag_need, valu, lead, pre_issued=(), post_issue=(), fill_first=(), raw_after=None):
    """Emit one straight-line stretch: `mfmas` = list of (text, frag id or None); `frag_need[f]` = index of the first MFMA that
    reads fragment f; `frag_issue(f)` emits its LDS read(s) (tags them f).  Fragments are requested `lead` MFMAs ahead (those in
    `pre_issued` already are); `post_issue` = callables emitting the next stretch's first requests, placed in the last gaps.
    `valu` = ordered VALU/SALU/VMEM filler instructions, spread so that every MFMA gap carries about the same number of
    instructions; `fill_first` = fillers that must come first, one per gap from gap 0 on (the LDS-DMA issue pairs).
    Waits are merged per two fragments (one s_waitcnt per four MFMAs)."""
    n = len(mfmas)
    gap_ds = [[] for _ in range(n)]
    for f, need in sorted(frag_need.items(), key=lambda kv: kv[1]):
        if f in pre_issued:
            continue
        g = max(need - lead - 1, 0)
        gap_ds[g].append(f)
    for k, fn in enumerate(post_issue):
        gap_ds[n - len(post_issue) + k].append(fn)
    # which MFMAs carry a wait: the first reader of every second fragment (in need order) waits for that fragment AND the next
    order = [f for f, _ in sorted(frag_need.items(), key=lambda kv: kv[1])]
    wait_at = {}
    for k in range(0, len(order), 2):
        wait_at[frag_need[order[k]]] = order[min(k + 1, len(order) - 1)]
    # the merged wait may only name a fragment that has been requested by then; otherwise fall back to single waits
    fixed = list(fill_first)
    weights = [len(gap_ds[g]) * 1 + (1 if (g + 1) in wait_at else 0) + (len(fixed[g]) if g < len(fixed) else 0) for g in range(n)]
    total = sum(weights) + len(valu)
    per_gap = [0] * n
    # water-fill the VALU list over the gaps
    left = len(valu)
    level = max(weights) if weights else 0
    target = -(-total // n)
    for g in range(n):
        room = max(target - weights[g], 0)
        per_gap[g] = room
    # trim / grow to match exactly
    extra = sum(per_gap) - left
    g = n - 1
    while extra > 0:
        if per_gap[g] > 0:
            per_gap[g] -= 1
            extra -= 1
        g = g - 1 if g > 0 else n - 1
    g = 0
    while extra < 0:
        per_gap[g] += 1
        extra += 1
        g = (g + 1) % n
    vk = 0
    requested = set(pre_issued)
    for i, (text, f) in enumerate(mfmas):
        if i in wait_at:
            tgt = wait_at[i]
            if tgt not in requested:
                tgt = f
            em.wait_tag(tgt)
        elif f is not None and frag_need.get(f) == i and not em.arrived(f):
            em.wait_tag(f)
        if "nomfma" not in ABLATE:
            em.raw(text)
        if i < len(fixed):
            for t in fixed[i]:
                em.raw(t)
        for t in (raw_after or {}).get(i, ()):       # a (mostly branch-skipped) block pinned behind MFMA i: not part of the water-fill
            em.raw(t)
        for item in gap_ds[i]:
            if callable(item):
                item()
            else:
                frag_issue(item)
                requested.add(item)
        for _ in range(per_gap[i]):
            em.raw(valu[vk])
            vk += 1
        for _ in range(KNOB.get("pad", 0)):      # timing experiment: idle issue slots in every MFMA gap
            em.raw("s_nop 0")
        if KNOB.get("nop7"):
            em.raw("s_nop 7")
    assert vk == len(valu), (vk, len(valu))


# ---------------------------------------------------------------------------------------------------------------- dQ loop
class DqLoop:
    """dQ:  per 32-key half-tile g (tile = g >> 1 in the K|V ring, key block kb = g & 1), two 32-row q-blocks j per wave:
         A(g): S[g&1][j] = -lse + K_g Q_j^T, DP[g&1][j] = -delta + V_g dO_j^T          16 MFMAs (-lse / -delta enter as srcC)
         B(g): D[g&1][j] = bf16(exp2(S) * DP)                                             80 VALU
         C(g): dq[j] += K_g^T D[g&1][j]                                                   8 MFMAs (transpose-read K)
       half-step(g) issues C(g-1) | A(g+1) | B(g) interleaved (the C products first: their fragments can be requested before
       the tile barrier).  The loop is uniform from g = -1 to 2 nt: the pipeline is filled and drained with zeros (S, DP, D start
       as 0; the slot "before the first tile" is zero-filled by the caller; tiles past the end arrive as zeros because their
       rows are out of the buffer descriptor's range), and rows past S need no masking at all: their K rows are zero, so
       whatever dS they get multiplies 0.

       register map      a[0:63] dq[j][db]   a[64:95] qf[j][ks]   a[96:127] dof[j][ks]   a[128:151] fragment ring (6 x 4)
                         v[0:127] S/DP[p][j] (p-major: s0 s1 dp0 dp1)   v[128:159] D[p][j][cc]
                         v[160:223] srcC of the first k-step: -lse2[q] x16 (j = 0,1), -delta[q] x16 (j = 0,1)
                         v[224:231] lane LDS offsets   v[232:235] LDS-DMA source offsets (K0 K1 V0 V1; advanced per tile)"""

    LA = 224
    VOFF = 232
    CI = 160
    FR = 128       # AGPR
    NFR = 6
    LEAD = KNOB.get("lead", 6)       # fragment reads are issued this many MFMAs ahead of their first use

    def S(self, p, j):
        return p * 64 + j * 16

    def DP(self, p, j):
        return p * 64 + 32 + j * 16

    def D(self, p, j, cc):
        return 128 + p * 16 + j * 8 + cc * 4

    # fragments of one half-step: 0..3 K transposed (cc, db) = (f >> 1, f & 1) of the PREVIOUS tile; 4..7 K rows ks; 8..11 V rows ks.
    # ring position advances continuously (12 per half-step, ring of 6: every half-step starts at position 0)
    def frag_reg(self, f):
        return (160 if KNOB.get("ringv") else self.FR) + 4 * (f % self.NFR)

    def fr(self, f):
        return (vr if KNOB.get("ringv") else ar)(self.frag_reg(f), 4)

    def issue_frag(self, em, f, slotA, kbA, slotC, kbC, tag=None):
        r = self.frag_reg(f)
        ar = vr if KNOB.get("ringv") else globals()["ar"]
        tag = f if tag is None else tag
        if f < 4:
            cc, db = f >> 1, f & 1
            off = slotC * 16384 + kbC * 4096 + cc * 2048
            em.ds(f"ds_read_b64_tr_b16 {ar(r, 2)}, v{self.LA + 4 + 2 * db} offset:{off}", tag)
            em.ds(f"ds_read_b64_tr_b16 {ar(r + 2, 2)}, v{self.LA + 5 + 2 * db} offset:{off}", tag)
        elif f < 8:
            em.ds(f"ds_read_b128 {ar(r, 4)}, v{self.LA + f - 4} offset:{slotA * 16384 + kbA * 4096}", tag)
        else:
            em.ds(f"ds_read_b128 {ar(r, 4)}, v{self.LA + f - 8} offset:{slotA * 16384 + 8192 + kbA * 4096}", tag)

    def mfmas(self, pa, pc):
        out = []
        for i in range(24):
            j = i & 1
            if i < 8:
                c = i >> 1
                cc, db = c >> 1, c & 1
                d = ar(32 * j + 16 * db, 16)
                out.append((f"{MFMA} {d}, {self.fr(c)}, {vr(self.D(pc, j, cc), 4)}, {d}", c))
            elif i < 16:
                ks = (i - 8) >> 1
                d = vr(self.S(pa, j), 16)
                c = (vr(self.CI + 16 * j, 16) if KNOB.get("cinit", 1) in (1, 2) else "0") if ks == 0 else d
                out.append((f"{MFMA} {d}, {self.fr(4 + ks)}, {ar(64 + 16 * j + 4 * ks, 4)}, {c}", 4 + ks))
            else:
                ks = (i - 16) >> 1
                d = vr(self.DP(pa, j), 16)
                c = (vr(self.CI + 32 + 16 * j, 16) if KNOB.get("cinit", 1) in (1, 3) else "0") if ks == 0 else d
                out.append((f"{MFMA} {d}, {self.fr(8 + ks)}, {ar(96 + 16 * j + 4 * ks, 4)}, {c}", 8 + ks))
        return out

    def valu_ops(self, pb):
        """B stage on S[pb], DP[pb] -> D[pb]: 16 units of (exp, exp, mul, mul, cvt), software-pipelined by one unit"""
        if "novalu" in ABLATE:
            return []

        def unit(u):
            j, p = u >> 3, u & 7
            s0, d0 = self.S(pb, j) + 2 * p, self.DP(pb, j) + 2 * p
            w = self.D(pb, j, p >> 2) + (p & 3)
            if KNOB.get("pkmul"):
                return ([f"v_exp_f32 v{s0}, v{s0}", f"v_exp_f32 v{s0 + 1}, v{s0 + 1}"],
                        [f"v_pk_mul_f32 {vr(s0, 2)}, {vr(d0, 2)}, {vr(s0, 2)}", None, f"v_cvt_pk_bf16_f32 v{w}, v{s0}, v{s0 + 1}"])
            return ([f"v_exp_f32 v{s0}, v{s0}", f"v_exp_f32 v{s0 + 1}, v{s0 + 1}"],
                    [f"v_mul_f32 v{s0}, v{d0}, v{s0}", f"v_mul_f32 v{s0 + 1}, v{d0 + 1}, v{s0 + 1}", f"v_cvt_pk_bf16_f32 v{w}, v{s0}, v{s0 + 1}"])
        if KNOB.get("vorder", 1) == 0:
            ops = list(unit(0)[0])
            for u in range(16):
                if u + 1 < 16:
                    ops += unit(u + 1)[0]
                ops += unit(u)[1]
            return ops
        # no instruction directly follows one it depends on: step t = mul(t).a, exp(t+1).a, mul(t).b, exp(t+1).b, cvt(t-1)
        ops = list(unit(0)[0])
        for t in range(17):
            x = unit(t + 1)[0] if t + 1 < 16 else [None, None]
            m = unit(t)[1][:2] if t < 16 else [None, None]
            c = unit(t - 1)[1][2] if 1 <= t else None
            ops += [o for o in (m[0], x[0], m[1], x[1], c) if o is not None]
        return ops

    def half_step(self, em, slotA, kbA, slotC, kbC, pa, nxt, fill_first=()):
        """C reads D[pa] (tile slotC, kbC); A writes S/DP[pa] (tile slotA, kbA); B reads S/DP[pa ^ 1], writes D[pa ^ 1].
        nxt = (slotC, kbC) of the following half-step, whose transposed fragments are requested in the last four gaps (their ring positions are free by then)."""
        need = {f: 2 * f for f in range(12)}
        post = [lambda f=f: self.issue_frag(em, f, 0, 0, nxt[0], nxt[1], tag=("n", f)) for f in range(4)]
        # the prefetched fragments carry the tag ("n", f): rename them to plain f for this stretch
        em.retag({("n", f): f for f in range(4)})
        schedule(em, self.mfmas(pa, pa), lambda f: self.issue_frag(em, f, slotA, kbA, slotC, kbC), need, self.valu_ops(pa ^ 1), self.LEAD,
                 pre_issued=(0, 1, 2, 3), post_issue=post, fill_first=fill_first)

    def generate(self):
        em = Emitter()
        SAVE_M0, CNT = "%0", "%1"
        RK, RV, KSTEP, VSTEP, WBASE, NITER = "%[rk]", "%[rv]", "%[kstep]", "%[vstep]", "%[wbase]", "%[niter]"
        em.raw(f"s_mov_b32 {SAVE_M0}, m0")
        em.raw(f"s_mov_b32 {CNT}, {NITER}")
        for i in range(64):
            em.raw(f"v_accvgpr_write_b32 a{i}, 0")
        for r in list(range(64, 128)) + list(range(128, 144)):      # S/DP[1], D[0]
            em.raw(f"v_mov_b32 v{r}, 0")
        # the first half-step's transposed fragments: tile "-1" = ring slot 3 (zero-filled by the caller), key block 0
        for f in range(4):
            self.issue_frag(em, f, 0, 0, 3, 0, tag=("n", f))
        for _ in range(KNOB.get("pad4", 0)):      # placement experiment: shift the loop body by 4 bytes per unit (MI355X_MICROARCH.md "code-placement sensitivity")
            em.raw("s_nop 0")
        em.raw("L_w1dq_loop_%=:")
        for ph in range(4):
            # tile i (ring slot ph) must have landed for every wave.  Every read of tile i-2's slot has been waited for by its
            # consumer MFMA, so after the barrier LDS-DMA may refill that slot with tile i+2 (issued inside the first gaps).
            em.raw("s_waitcnt vmcnt(4)")
            if "nosync" not in ABLATE:
                em.raw("s_barrier")
            dst = ((ph + 2) & 3) * 16384
            fill = []
            for k, (rs, extra) in enumerate([(RK, 0), (RK, 1024), (RV, 8192), (RV, 9216)]):
                fill.append([f"s_add_u32 m0, {WBASE}, {dst + extra}"])
                fill.append([f"buffer_load_dwordx4 v{self.VOFF + k}, {rs}, 0 offen lds",
                             f"v_add_u32 v{self.VOFF + k}, {KSTEP if k < 2 else VSTEP}, v{self.VOFF + k}"])
            sp = (ph - 1) & 3
            self.half_step(em, ph, 0, sp, 0, 0, nxt=(sp, 1), fill_first=fill)
            self.half_step(em, ph, 1, sp, 1, 1, nxt=(ph, 0))
            em.raw(f"s_sub_u32 {CNT}, {CNT}, 1")
            em.raw(f"s_cmp_eq_u32 {CNT}, 0")
            if ph < 3:
                em.raw("s_cbranch_scc1 L_w1dq_done_%=")
            else:
                em.raw("s_cbranch_scc0 L_w1dq_loop_%=")
        em.raw("L_w1dq_done_%=:")
        em.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        em.raw("s_nop 7")
        em.raw("s_nop 7")            # last MFMA results -> the compiler's v_accvgpr_read
        em.raw(f"s_mov_b32 m0, {SAVE_M0}")
        return em.text() + "\n"


# --------------------------------------------------------------------------------------------------------------- dK/dV loop
class DkvLoop:
    """dK, dV:  a wave owns two 32-key blocks kb (K, V fragments and the dK^T, dV^T accumulators stay in AGPRs for the whole
       sweep) and streams 64-row Q|dO tiles; per 32-row half-tile g (tile = g >> 1, row block qb = g & 1):
         A(g): S[g&1][kb] = -lse[q] + Q_g K_kb^T,  DP[g&1][kb] = -delta[q] + dO_g V_kb^T   16 MFMAs; q runs over the accumulator
               ROWS here, so the srcC of each chain's first k-step is a 16-register tuple read from the tile's statistics
               (fp32 -lse2 / -delta of the 64 rows, brought in by LDS-DMA next to the tile)
         B(g): P = exp2(S) -> PK[g&1][kb] (bf16), dS = P * DP -> DSK[g&1][kb] (bf16)          96 VALU
         C(g): dV^T[kb] += dO_g^T P,  dK^T[kb] += Q_g^T dS                                    16 MFMAs on transpose-read fragments
       half-step(g) issues C(g-1) | A(g+1) | B(g) interleaved; the loop is uniform (pipeline filled and drained with zeros; rows
       past S need no mask: their Q / dO rows are zero, whatever P / dS they get multiplies 0).

       register map   a[0:63] dk[kb][db]  a[64:127] dv[kb][db]  a[128:159] kf[kb][ks]  a[160:191] vf[kb][ks]  a[192:223] fragment ring (8 x 4)
                      v[0:127] S/DP[p][kb] (p-major: s0 s1 dp0 dp1)   v[128:191] PK / DSK[p][kb][cc]   v[192:207] srcC -lse   v[208:223] srcC -delta
                      v[224:231] lane LDS offsets   v[232:236] LDS-DMA source offsets (Q0 Q1 dO0 dO1 stats)   v237 statistics read base
       LDS            ring slot s at 16384 s: Q tile | dO tile;  statistics of slot s at 65536 + 1024 s: wave w's 256-byte piece holds
                      -lse2 of rows 16w..16w+15 at +0 and -delta of the same rows at +64"""

    LA = 224
    VOFF = 232
    SB = 237
    FR = 192
    NFR = 8
    LEAD = KNOB.get("lead", 6)

    def S(self, p, kb):
        return p * 64 + kb * 16

    def DP(self, p, kb):
        return p * 64 + 32 + kb * 16

    def PK(self, p, kb, cc):
        return 128 + p * 32 + kb * 8 + cc * 4

    def DSK(self, p, kb, cc):
        return 128 + p * 32 + 16 + kb * 8 + cc * 4

    def frag_reg(self, f):
        return self.FR + 4 * (f % self.NFR)

    def issue_frag(self, em, f, slotA, qbA, slotC, qbC, tag=None):
        """f 0..7: transposed fragments of the PREVIOUS half (even: dO tile, odd: Q tile; (cc, db) = (f >> 2, (f >> 1) & 1));
        8..11: Q rows ks; 12..15: dO rows ks; "cs" / "cd": the srcC tuples (-lse / -delta of the 32 rows)"""
        tag = f if tag is None else tag
        if f == "cs" or f == "cd":
            base = 192 if f == "cs" else 208
            for g in range(4):
                off = slotA * 1024 + 512 * qbA + 256 * (g >> 1) + 32 * (g & 1) + (64 if f == "cd" else 0)
                em.ds(f"ds_read_b128 {vr(base + 4 * g, 4)}, v{self.SB} offset:{off}", tag)
            return
        r = self.frag_reg(f)
        if f < 8:
            c = f >> 1
            cc, db = c >> 1, c & 1
            off = slotC * 16384 + (8192 if (f & 1) == 0 else 0) + qbC * 4096 + cc * 2048
            em.ds(f"ds_read_b64_tr_b16 {ar(r, 2)}, v{self.LA + 4 + 2 * db} offset:{off}", tag)
            em.ds(f"ds_read_b64_tr_b16 {ar(r + 2, 2)}, v{self.LA + 5 + 2 * db} offset:{off}", tag)
        elif f < 12:
            em.ds(f"ds_read_b128 {ar(r, 4)}, v{self.LA + f - 8} offset:{slotA * 16384 + qbA * 4096}", tag)
        else:
            em.ds(f"ds_read_b128 {ar(r, 4)}, v{self.LA + f - 12} offset:{slotA * 16384 + 8192 + qbA * 4096}", tag)

    def mfmas(self, pa, pc):
        out = []
        for i in range(32):
            kb = i & 1
            if i < 16:
                f = i >> 1
                c = f >> 1
                cc, db = c >> 1, c & 1
                if (f & 1) == 0:
                    d = ar(64 + 32 * kb + 16 * db, 16)
                    b = vr(self.PK(pc, kb, cc), 4)
                else:
                    d = ar(32 * kb + 16 * db, 16)
                    b = vr(self.DSK(pc, kb, cc), 4)
                out.append((f"{MFMA} {d}, {ar(self.frag_reg(f), 4)}, {b}, {d}", f))
            elif i < 24:
                ks = (i - 16) >> 1
                d = vr(self.S(pa, kb), 16)
                c = vr(192, 16) if ks == 0 else d
                out.append((f"{MFMA} {d}, {ar(self.frag_reg(8 + ks), 4)}, {ar(128 + 16 * kb + 4 * ks, 4)}, {c}", 8 + ks))
            else:
                ks = (i - 24) >> 1
                d = vr(self.DP(pa, kb), 16)
                c = vr(208, 16) if ks == 0 else d
                out.append((f"{MFMA} {d}, {ar(self.frag_reg(12 + ks), 4)}, {ar(160 + 16 * kb + 4 * ks, 4)}, {c}", 12 + ks))
        return out

    def valu_ops(self, pb):
        if "novalu" in ABLATE:
            return []

        def unit(u):
            kb, p = u >> 3, u & 7
            s0, d0 = self.S(pb, kb) + 2 * p, self.DP(pb, kb) + 2 * p
            wp, wd = self.PK(pb, kb, p >> 2) + (p & 3), self.DSK(pb, kb, p >> 2) + (p & 3)
            return ([f"v_exp_f32 v{s0}, v{s0}", f"v_exp_f32 v{s0 + 1}, v{s0 + 1}"],
                    [f"v_mul_f32 v{d0}, v{d0}, v{s0}", f"v_mul_f32 v{d0 + 1}, v{d0 + 1}, v{s0 + 1}"],
                    [f"v_cvt_pk_bf16_f32 v{wp}, v{s0}, v{s0 + 1}", f"v_cvt_pk_bf16_f32 v{wd}, v{d0}, v{d0 + 1}"])
        ops = list(unit(0)[0])
        for t in range(17):
            x = unit(t + 1)[0] if t + 1 < 16 else [None, None]
            m = unit(t)[1] if t < 16 else [None, None]
            c = unit(t - 1)[2] if 1 <= t else [None, None]
            ops += [o for o in (m[0], x[0], c[0], m[1], x[1], c[1]) if o is not None]
        return ops

    def half_step(self, em, slotA, qbA, slotC, qbC, pa, nxt, fill_first=()):
        need = {f: 2 * f for f in range(16)}
        need["cs"] = 16
        need["cd"] = 24
        post = [lambda f=f: self.issue_frag(em, f, 0, 0, nxt[0], nxt[1], tag=("n", f)) for f in range(4)]
        em.retag({("n", f): f for f in range(4)})
        schedule(em, self.mfmas(pa, pa), lambda f: self.issue_frag(em, f, slotA, qbA, slotC, qbC), need, self.valu_ops(pa ^ 1), self.LEAD,
                 pre_issued=(0, 1, 2, 3), post_issue=post, fill_first=fill_first)

    def generate(self):
        em = Emitter()
        SAVE_M0, CNT = "%0", "%1"
        RQ, RDO, RST, QSTEP, DSTEP, WBASE, SBASE, NITER = "%[rq]", "%[rdo]", "%[rst]", "%[qstep]", "%[dstep]", "%[wbase]", "%[sbase]", "%[niter]"
        em.raw(f"s_mov_b32 {SAVE_M0}, m0")
        em.raw(f"s_mov_b32 {CNT}, {NITER}")
        for i in range(128):
            em.raw(f"v_accvgpr_write_b32 a{i}, 0")
        for r in list(range(64, 128)) + list(range(128, 160)):      # S/DP[1], PK/DSK[0]
            em.raw(f"v_mov_b32 v{r}, 0")
        for f in range(4):
            self.issue_frag(em, f, 0, 0, 3, 0, tag=("n", f))
        for _ in range(KNOB.get("pad4", 0)):      # placement experiment: shift the loop body by 4 bytes per unit (MI355X_MICROARCH.md "code-placement sensitivity")
            em.raw("s_nop 0")
        em.raw("L_w1dkv_loop_%=:")
        for ph in range(4):
            em.raw("s_waitcnt vmcnt(5)")
            if "nosync" not in ABLATE:
                em.raw("s_barrier")
            dst = ((ph + 2) & 3) * 16384
            fill = []
            for k, (rs, extra) in enumerate([(RQ, 0), (RQ, 1024), (RDO, 8192), (RDO, 9216)]):
                fill.append([f"s_add_u32 m0, {WBASE}, {dst + extra}"])
                fill.append([f"buffer_load_dwordx4 v{self.VOFF + k}, {rs}, 0 offen lds",
                             f"v_add_u32 v{self.VOFF + k}, {QSTEP if k < 2 else DSTEP}, v{self.VOFF + k}"])
            fill.append([f"s_add_u32 m0, {SBASE}, {((ph + 2) & 3) * 1024}"])
            fill.append([f"buffer_load_dword v{self.VOFF + 4}, {RST}, 0 offen lds", f"v_add_u32 v{self.VOFF + 4}, 256, v{self.VOFF + 4}"])
            sp = (ph - 1) & 3
            self.half_step(em, ph, 0, sp, 0, 0, nxt=(sp, 1), fill_first=fill)
            self.half_step(em, ph, 1, sp, 1, 1, nxt=(ph, 0))
            em.raw(f"s_sub_u32 {CNT}, {CNT}, 1")
            em.raw(f"s_cmp_eq_u32 {CNT}, 0")
            if ph < 3:
                em.raw("s_cbranch_scc1 L_w1dkv_done_%=")
            else:
                em.raw("s_cbranch_scc0 L_w1dkv_loop_%=")
        em.raw("L_w1dkv_done_%=:")
        em.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        em.raw("s_nop 7")
        em.raw("s_nop 7")
        em.raw(f"s_mov_b32 m0, {SAVE_M0}")
        return em.text() + "\n"


# --------------------------------------------------------------------------------------------------------------- forward loop
class FwdLoop:
    """forward:  a wave owns two 32-row q-blocks j (Q fragments and O^T accumulators in AGPRs) and streams 64-key K|V tiles; per
       32-key half-tile g:
         A(g): S[g&1][j] = -M[q] + K_g Q_j^T                     8 MFMAs; -M[q] is the srcC of the first k-step (loop-invariant tuple)
         B(g): P = exp2(S), PK[g&1][j] = bf16(P)                      48 VALU (32 exp + 16 packs).  The row sums l[j] += rowsum(P) ride the MATRIX pipe since
               round 6 (mfsum, the default; W1_KNOBS=mfsum=0 restores the 32 v_add_f32 into 4 partial sums per q-block): four v_mfma_f32_16x16x32_bf16 per
               half-step multiply the packed P registers by a SPARSE selector (2 of its 16 rows hold ones: see mfmas()), 64 matrix-pipe cycles for 128 VALU
               cycles, and the selector barely toggles the multiplier array: 6.95 -> 6.58 ms and 9.13 -> 8.63 J per launch in one session
               (profiles/r06_fwd_mfsum_ab.txt).  Round 3 had tried the sums as four FULL 32x32x16 products against an all-ones operand: + 128 pipe cycles at full
               toggling, same time.  l now sums the bf16-rounded P -- exactly the weights the PV product uses, so O = sum(P~ v) / sum(P~) is a true convex
               combination -- and lse2 = M' + log2(l~) differs from the exact one by the mean rounding error of a row's weights (<= 2^-8 relative on a one-hot row: tests/attn_tol.py,
               ~1e-4 typical): the backward's recomputed P = exp2(s - lse2) is scaled per ROW by that factor, which leaves every row's dS summing to zero.
         C(g): O^T[j] += V_g^T P                                  8 MFMAs on transpose-read V fragments
       half-step(g) issues C(g-1) | A(g+1) | B(g).  M[q] = |q| max_k |k| bounds every score of the row from above (the caller
       computes it), so P <= 1 and the loop carries no running maximum, no rescale and no branch.  Keys past the end must not
       count in l: while fewer than 64 keys remain (the ragged last tile and the drain step) a small block in front of each
       half-step rewrites the srcC tuples to -inf for the missing keys -- otherwise the loop is uniform; S starts as -inf (P = 0).

       register map   a[0:63] O[j][db]   a[64:95] qf[j][ks]   a[96:127] fragment ring (8 x 4)
                      v[0:63] S[p][j]   v[64:95] PK[p][j][cc]   v[96:127] srcC tuples (j = 0, 1)   v[128:135] l[j][0..3]   v136 v137 -M[q]
                      v[144:151] lane LDS offsets   v[152:155] LDS-DMA source offsets (K0 K1 V0 V1)   v156 4 * (lane >> 5)   v157 -inf"""

    LA = 144
    VOFF = 152
    FR = 96
    NFR = 8
    LEAD = KNOB.get("lead", 6)

    def S(self, p, j):
        return p * 32 + j * 16

    def PK(self, p, j, cc):
        return 64 + p * 16 + j * 8 + cc * 4

    def frag_reg(self, f):
        return self.FR + 4 * (f % self.NFR)

    def issue_frag(self, em, f, slotA, kbA, slotC, kbC, tag=None):
        """f 0..3: transposed V fragments (cc, db) = (f >> 1, f & 1) of the PREVIOUS half; 4..7: K rows ks"""
        tag = f if tag is None else tag
        r = self.frag_reg(f)
        if f < 4:
            cc, db = f >> 1, f & 1
            off = slotC * 16384 + 8192 + kbC * 4096 + cc * 2048
            em.ds(f"ds_read_b64_tr_b16 {ar(r, 2)}, v{self.LA + 4 + 2 * db} offset:{off}", tag)
            em.ds(f"ds_read_b64_tr_b16 {ar(r + 2, 2)}, v{self.LA + 5 + 2 * db} offset:{off}", tag)
        else:
            em.ds(f"ds_read_b128 {ar(r, 4)}, v{self.LA + f - 4} offset:{slotA * 16384 + kbA * 4096}", tag)

    def mfmas(self, pa, pc):
        out = []
        for i in range(16):
            j = i & 1
            if i < 8:
                c = i >> 1
                cc, db = c >> 1, c & 1
                d = ar(32 * j + 16 * db, 16)
                out.append((f"{MFMA} {d}, {ar(self.frag_reg(c), 4)}, {vr(self.PK(pc, j, cc), 4)}, {d}", c))
                if KNOB.get("mfsum", 1) and db == 1 and j == 1:
                    # W1_KNOBS=mfsum=1: l[q] += sum_k P[k][q] as a 16x16x32 product of a SPARSE selector against the same packed P registers the PV product
                    # reads.  Read as the B operand of v_mfma_f32_16x16x32_bf16, lane L of the 32x32x16 B fragment (column q = L % 32, keys 8 (L / 32) ..+7 of the
                    # 16-key chunk) is column n' = L % 16, k'-block L / 16: blocks 0, 2 hold q = n' (keys 0-7, 8-15), blocks 1, 3 hold q = n' + 16.  With
                    # A'[0][k'] = 1 on blocks 0, 2 and A'[1][k'] = 1 on blocks 1, 3 (a[128:131]: 1.0 pairs in lanes 0, 32 and 17, 49, zero elsewhere)
                    # D'[0][n'] = rowsum(q = n'), D'[1][n'] = rowsum(q = n' + 16): lanes 0-15 of the first two of the four accumulator registers.  64 matrix-pipe
                    # cycles per half-step (4 passes each) against the 32 v_add_f32 they replace; 2 of A's 16 rows are non-zero, so the multiplier array barely toggles.
                    for jj in range(2):
                        dl = vr(128 + 4 * jj, 4)
                        out.append((f"v_mfma_f32_16x16x32_bf16 {dl}, {ar(128, 4)}, {vr(self.PK(pc, jj, cc), 4)}, {dl}", None))
            else:
                ks = (i - 8) >> 1
                d = vr(self.S(pa, j), 16)
                c = vr(96 + 16 * j, 16) if ks == 0 else d
                out.append((f"{MFMA} {d}, {ar(self.frag_reg(4 + ks), 4)}, {ar(64 + 16 * j + 4 * ks, 4)}, {c}", 4 + ks))
        return out

    def valu_ops(self, pb):
        if "novalu" in ABLATE:
            return []

        def unit(u):
            j, p = u >> 3, u & 7
            s0 = self.S(pb, j) + 2 * p
            w = self.PK(pb, j, p >> 2) + (p & 3)
            l0, l1 = 128 + 4 * j + ((2 * p) & 3), 128 + 4 * j + ((2 * p + 1) & 3)
            if KNOB.get("mfsum", 1):      # row sums on the matrix pipe (see mfmas): no adds at all
                return ([f"v_exp_f32 v{s0}, v{s0}", f"v_exp_f32 v{s0 + 1}, v{s0 + 1}"], [None, None], f"v_cvt_pk_bf16_f32 v{w}, v{s0}, v{s0 + 1}")
            if KNOB.get("pksum"):      # the two row-sum adds as ONE packed add (same fp32 additions in the same order: bit-identical results)
                return ([f"v_exp_f32 v{s0}, v{s0}", f"v_exp_f32 v{s0 + 1}, v{s0 + 1}"],
                        [f"v_pk_add_f32 {vr(l0, 2)}, {vr(l0, 2)}, {vr(s0, 2)}", None],
                        f"v_cvt_pk_bf16_f32 v{w}, v{s0}, v{s0 + 1}")
            return ([f"v_exp_f32 v{s0}, v{s0}", f"v_exp_f32 v{s0 + 1}, v{s0 + 1}"],
                    [f"v_add_f32 v{l0}, v{l0}, v{s0}", f"v_add_f32 v{l1}, v{l1}, v{s0 + 1}"],
                    f"v_cvt_pk_bf16_f32 v{w}, v{s0}, v{s0 + 1}")
        ops = list(unit(0)[0])
        for t in range(16):
            x = unit(t + 1)[0] if t + 1 < 16 else [None, None]
            m = unit(t)[1]
            ops += [o for o in (m[0], x[0], m[1], x[1], unit(t)[2]) if o is not None]
        return ops

    def mask_block(self, em, kb, krem, tmp, label):
        """srcC tuples for a half-tile with fewer than 32 valid keys left: row r of this lane is key 32 kb + rowconst(r) + 4 hi"""
        em.raw(f"s_cmp_ge_i32 {krem}, 64")
        em.raw(f"s_cbranch_scc1 {label}")
        for r in range(16):
            rowc = (r & 3) + 8 * (r >> 2) + 32 * kb
            em.raw(f"s_sub_i32 {tmp}, {krem}, {rowc}")
            em.raw(f"v_cmp_lt_i32 vcc, v156, {tmp}")
            for j in range(2):
                em.raw(f"v_cndmask_b32 v{96 + 16 * j + r}, v157, v{136 + j}, vcc")
        em.raw(f"{label}:")

    def half_step(self, em, slotA, kbA, slotC, kbC, pa, nxt, fill_first=()):
        ms = self.mfmas(pa, pa)
        need = {}
        for idx, (_, f) in enumerate(ms):           # first MFMA that reads each fragment (f: 2 f without the mfsum products)
            if f is not None and f not in need:
                need[f] = idx
        post = [lambda f=f: self.issue_frag(em, f, 0, 0, nxt[0], nxt[1], tag=("n", f)) for f in range(4)]
        em.retag({("n", f): f for f in range(4)})
        schedule(em, ms, lambda f: self.issue_frag(em, f, slotA, kbA, slotC, kbC), need, self.valu_ops(pa ^ 1), self.LEAD,
                 pre_issued=(0, 1, 2, 3), post_issue=post, fill_first=fill_first)

    def generate(self):
        em = Emitter()
        SAVE_M0, CNT, KREM, TMP = "%0", "%1", "%2", "%3"
        RK, RV, KSTEP, VSTEP, WBASE, NITER, KREM0 = "%[rk]", "%[rv]", "%[kstep]", "%[vstep]", "%[wbase]", "%[niter]", "%[krem]"
        em.raw(f"s_mov_b32 {SAVE_M0}, m0")
        if KNOB.get("mfsum", 1):
            for i in range(4):                                     # the selector operand A' (v158: 0x3f803f80 in lanes 0, 17, 32, 49, else 0)
                em.raw(f"v_accvgpr_write_b32 a{128 + i}, v158")
        em.raw(f"s_mov_b32 {CNT}, {NITER}")
        em.raw(f"s_mov_b32 {KREM}, {KREM0}")
        for i in range(64):
            em.raw(f"v_accvgpr_write_b32 a{i}, 0")
        em.raw("v_mov_b32 v157, 0xff800000")                      # -inf
        for r in range(32, 64):                                   # S[1] = -inf: the pipeline's first B stage adds nothing
            em.raw(f"v_mov_b32 v{r}, v157")
        for r in list(range(64, 80)) + list(range(128, 136)):    # PK[0], l
            em.raw(f"v_mov_b32 v{r}, 0")
        for j in range(2):                                        # srcC tuples = -M[q]
            for r in range(16):
                em.raw(f"v_mov_b32 v{96 + 16 * j + r}, v{136 + j}")
        for f in range(4):
            self.issue_frag(em, f, 0, 0, 3, 0, tag=("n", f))
        em.raw("s_memtime %[c0]")
        for _ in range(KNOB.get("pad4", 0)):      # placement experiment: shift the loop body by 4 bytes per unit (MI355X_MICROARCH.md "code-placement sensitivity")
            em.raw("s_nop 0")
        em.raw("L_w1fwd_loop_%=:")
        for ph in range(4):
            em.raw("s_waitcnt vmcnt(4)")
            if "nosync" not in ABLATE:
                em.raw("s_barrier")
            dst = ((ph + 2) & 3) * 16384
            fill = []
            for k, (rs, extra) in enumerate([(RK, 0), (RK, 1024), (RV, 8192), (RV, 9216)]):
                fill.append([f"s_add_u32 m0, {WBASE}, {dst + extra}"])
                fill.append([f"buffer_load_dwordx4 v{self.VOFF + k}, {rs}, 0 offen lds",
                             f"v_add_u32 v{self.VOFF + k}, {KSTEP if k < 2 else VSTEP}, v{self.VOFF + k}"])
            sp = (ph - 1) & 3
            self.mask_block(em, 0, KREM, TMP, f"L_w1fwd_m{2 * ph}_%=")
            self.half_step(em, ph, 0, sp, 0, 0, nxt=(sp, 1), fill_first=fill)
            self.mask_block(em, 1, KREM, TMP, f"L_w1fwd_m{2 * ph + 1}_%=")
            self.half_step(em, ph, 1, sp, 1, 1, nxt=(ph, 0))
            em.raw(f"s_sub_i32 {KREM}, {KREM}, 64")
            em.raw(f"s_sub_u32 {CNT}, {CNT}, 1")
            em.raw(f"s_cmp_eq_u32 {CNT}, 0")
            if ph < 3:
                em.raw("s_cbranch_scc1 L_w1fwd_done_%=")
            else:
                em.raw("s_cbranch_scc0 L_w1fwd_loop_%=")
        em.raw("L_w1fwd_done_%=:")
        em.raw("s_memtime %[c1]")
        em.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        em.raw("s_nop 7")
        em.raw("s_nop 7")
        em.raw(f"s_mov_b32 m0, {SAVE_M0}")
        return em.text() + "\n"


# ------------------------------------------------------------------------------------------------------- forward, head_dim 128
class Fwd128Loop:
    """FwdLoop for head_dim 128 (csrc/attention_hd128.hip, the Wan2.2 shapes): same pipeline -- half-step(g) issues C(g-1) | A(g+1) | B(g)
       on 32-key half-tiles, two q-blocks per wave, row-bound shift, fp32 row sums -- with
         A(g): S[g&1][j] = -M[q] + K_g Q_j^T       16 MFMAs (8 k-steps x 2 q-blocks)
         C(g): O^T[j][db] += V_g^T P               16 MFMAs (2 key sub-blocks x 4 d-blocks x 2 q-blocks)
         B(g): 80 VALU, as at head_dim 64: the loop is MFMA-bound here (2.5 VALU per MFMA, was 5).
       LDS ring slot = [K tile 64 x 128 | V tile] = 32 KiB, 4 slots; rows are 256 B = 16 chunks stored at c ^ f(r),
       f(r) = ((r & 3) << 2) | ((r >> 2) & 3): the b128 row reads (16 lanes = 16 rows, one logical chunk) and the transpose reads
       (32 lanes = 4 rows x 4 chunks x 2 halves) are both conflict-free.  ds offsets are 16 bit: lane offsets exist per slot PAIR.
       register map   a[0:127] O[j][db]   a[128:191] qf[j][ks]   a[192:255] fragment ring (16 x 4)
                      v[0:63] S[p][j]   v[64:95] PK[p][j][cc]   v[96:127] srcC tuples   v[128:135] l[j][0..3]   v136 v137 -M[q]
                      v[144:175] lane LDS offsets [slot pair][K rows ks 0..7 | V transposed (db, second read) 0..7]
                      v[176:183] LDS-DMA source offsets (K x4, V x4)   v184 4 * (lane >> 5)   v185 -inf"""

    LA = 144
    VOFF = 176
    HI4 = 184
    NINF = 185
    FR = 192
    LEAD = KNOB.get("lead", 6)

    def S(self, p, j):
        return p * 32 + j * 16

    def PK(self, p, j, cc):
        return 64 + p * 16 + j * 8 + cc * 4

    def frag_reg(self, f):
        return self.FR + 4 * f

    def issue_frag(self, em, f, slotA, kbA, slotC, kbC, tag=None):
        """f 0..7: transposed V fragments (cc, db) = (f >> 2, f & 3) of the PREVIOUS half; 8..15: K rows ks = f - 8"""
        tag = f if tag is None else tag
        r = self.frag_reg(f)
        if f < 8:
            cc, db = f >> 2, f & 3
            base = self.LA + 16 * (slotC >> 1) + 8 + 2 * db
            off = (slotC & 1) * 32768 + 16384 + kbC * 8192 + cc * 4096
            em.ds(f"ds_read_b64_tr_b16 {ar(r, 2)}, v{base} offset:{off}", tag)
            em.ds(f"ds_read_b64_tr_b16 {ar(r + 2, 2)}, v{base + 1} offset:{off}", tag)
        else:
            base = self.LA + 16 * (slotA >> 1) + (f - 8)
            em.ds(f"ds_read_b128 {ar(r, 4)}, v{base} offset:{(slotA & 1) * 32768 + kbA * 8192}", tag)

    def mfmas(self, pa, pc):
        out = []
        for c in range(8):
            cc, db = c >> 2, c & 3
            for j in range(2):
                d = ar(64 * j + 16 * db, 16)
                out.append((f"{MFMA} {d}, {ar(self.frag_reg(c), 4)}, {vr(self.PK(pc, j, cc), 4)}, {d}", c))
        for ks in range(8):
            for j in range(2):
                d = vr(self.S(pa, j), 16)
                c = vr(96 + 16 * j, 16) if ks == 0 else d
                out.append((f"{MFMA} {d}, {ar(self.frag_reg(8 + ks), 4)}, {ar(128 + 32 * j + 4 * ks, 4)}, {c}", 8 + ks))
        return out

    def valu_ops(self, pb):
        """as FwdLoop.valu_ops plus the softmax scale: S holds q.k - M / c (q arrives UNscaled here: the backward kernels of
        attention_hd128.hip read the same q), so one v_mul by c = scale * log2(e) (an SGPR) precedes each exp"""
        if "novalu" in ABLATE:
            return []

        def unit(u):
            j, p = u >> 3, u & 7
            s0 = self.S(pb, j) + 2 * p
            w = self.PK(pb, j, p >> 2) + (p & 3)
            l0, l1 = 128 + 4 * j + ((2 * p) & 3), 128 + 4 * j + ((2 * p + 1) & 3)
            return ([f"v_mul_f32 v{s0}, %[cs], v{s0}", f"v_mul_f32 v{s0 + 1}, %[cs], v{s0 + 1}", f"v_exp_f32 v{s0}, v{s0}", f"v_exp_f32 v{s0 + 1}, v{s0 + 1}"],
                    [f"v_add_f32 v{l0}, v{l0}, v{s0}", f"v_add_f32 v{l1}, v{l1}, v{s0 + 1}"],
                    f"v_cvt_pk_bf16_f32 v{w}, v{s0}, v{s0 + 1}")
        ops = list(unit(0)[0])
        for t in range(16):
            x = unit(t + 1)[0] if t + 1 < 16 else [None] * 4
            m = unit(t)[1]
            ops += [o for o in (x[0], m[0], x[1], x[2], m[1], x[3], unit(t)[2]) if o is not None]
        return ops

    def mask_block(self, em, kb, krem, tmp, label):
        em.raw(f"s_cmp_ge_i32 {krem}, 64")
        em.raw(f"s_cbranch_scc1 {label}")
        for r in range(16):
            rowc = (r & 3) + 8 * (r >> 2) + 32 * kb
            em.raw(f"s_sub_i32 {tmp}, {krem}, {rowc}")
            em.raw(f"v_cmp_lt_i32 vcc, v{self.HI4}, {tmp}")
            for j in range(2):
                em.raw(f"v_cndmask_b32 v{96 + 16 * j + r}, v{self.NINF}, v{136 + j}, vcc")
        em.raw(f"{label}:")

    def half_step(self, em, slotA, kbA, slotC, kbC, pa, nxt, fill_first=()):
        need = {f: 2 * f for f in range(16)}
        post = [lambda f=f: self.issue_frag(em, f, 0, 0, nxt[0], nxt[1], tag=("n", f)) for f in range(4)]
        em.retag({("n", f): f for f in range(4)})
        schedule(em, self.mfmas(pa, pa), lambda f: self.issue_frag(em, f, slotA, kbA, slotC, kbC), need, self.valu_ops(pa ^ 1), self.LEAD,
                 pre_issued=(0, 1, 2, 3), post_issue=post, fill_first=fill_first)

    def generate(self):
        em = Emitter()
        SAVE_M0, CNT, KREM, TMP = "%0", "%1", "%2", "%3"
        RK, RV, KSTEP, VSTEP, WBASE, NITER, KREM0 = "%[rk]", "%[rv]", "%[kstep]", "%[vstep]", "%[wbase]", "%[niter]", "%[krem]"
        em.raw(f"s_mov_b32 {SAVE_M0}, m0")
        em.raw(f"s_mov_b32 {CNT}, {NITER}")
        em.raw(f"s_mov_b32 {KREM}, {KREM0}")
        for i in range(128):
            em.raw(f"v_accvgpr_write_b32 a{i}, 0")
        em.raw(f"v_mov_b32 v{self.NINF}, 0xff800000")
        for r in range(32, 64):
            em.raw(f"v_mov_b32 v{r}, v{self.NINF}")
        for r in list(range(64, 80)) + list(range(128, 136)):
            em.raw(f"v_mov_b32 v{r}, 0")
        for j in range(2):
            for r in range(16):
                em.raw(f"v_mov_b32 v{96 + 16 * j + r}, v{136 + j}")
        for f in range(4):
            self.issue_frag(em, f, 0, 0, 3, 0, tag=("n", f))
        for _ in range(KNOB.get("pad4", 0)):      # placement experiment: shift the loop body by 4 bytes per unit (MI355X_MICROARCH.md "code-placement sensitivity")
            em.raw("s_nop 0")
        em.raw("L_w1f128_loop_%=:")
        for ph in range(4):
            em.raw("s_waitcnt vmcnt(8)")
            if "nosync" not in ABLATE:
                em.raw("s_barrier")
            dst = ((ph + 2) & 3) * 32768
            fill = []
            for k in range(8):
                isV = k >= 4
                fill.append([f"s_add_u32 m0, {WBASE}, {dst + (16384 if isV else 0) + (k & 3) * 1024}"])
                fill.append([f"buffer_load_dwordx4 v{self.VOFF + k}, {RV if isV else RK}, 0 offen lds",
                             f"v_add_u32 v{self.VOFF + k}, {VSTEP if isV else KSTEP}, v{self.VOFF + k}"])
            sp = (ph - 1) & 3
            self.mask_block(em, 0, KREM, TMP, f"L_w1f128_m{2 * ph}_%=")
            self.half_step(em, ph, 0, sp, 0, 0, nxt=(sp, 1), fill_first=fill)
            self.mask_block(em, 1, KREM, TMP, f"L_w1f128_m{2 * ph + 1}_%=")
            self.half_step(em, ph, 1, sp, 1, 1, nxt=(ph, 0))
            em.raw(f"s_sub_i32 {KREM}, {KREM}, 64")
            em.raw(f"s_sub_u32 {CNT}, {CNT}, 1")
            em.raw(f"s_cmp_eq_u32 {CNT}, 0")
            if ph < 3:
                em.raw("s_cbranch_scc1 L_w1f128_done_%=")
            else:
                em.raw("s_cbranch_scc0 L_w1f128_loop_%=")
        em.raw("L_w1f128_done_%=:")
        em.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        em.raw("s_nop 7")
        em.raw("s_nop 7")
        em.raw(f"s_mov_b32 m0, {SAVE_M0}")
        return em.text() + "\n"


# ---------------------------------------------------------------------------------------------------- forward, head_dim 128, e4m3 operands
MFMA8 = "v_mfma_scale_f32_32x32x64_f8f6f4"


class Fwd128F8Loop:
    """The head_dim-128 forward on OCP-e4m3 operands (csrc/attention_f8.hip; BASELINE configs[4] "fp8 MFMA path"): both products of a 64-key tile
       run as v_mfma_scale_f32_32x32x64_f8f6f4 (K = 64 per instruction, 64 cycles: twice the bf16 matrix rate).  Per tile T, one straight-line block:
         C(T-2): O^T[j][db] += V8^T P8            8 MFMAs (4 d-blocks x 2 q-blocks; contraction = the tile's 64 keys)
         A(T)  : S[T&1][j][b] = -M[q] + K8 Q8^T   8 MFMAs (2 key blocks x 2 k-steps of 64 x 2 q-blocks)
         B(T-1): p = exp2(S), tile sums, P8 = e4m3(p / 2^x)   ~180 VALU
       Scales ride the instruction's E8M0 operands: 2^eq, 2^ek, 2^ev (one power of two per (batch, head) and tensor, made by the prep kernels) and, for P,
       a power of two PER TILE AND QUERY ROW: x = exponent of the row's sum over the tile's 64 keys (both lane halves, one v_permlane32_swap) minus 8, so
       every p / 2^x < 2^8 <= 448 whatever the distance between the row bound M and the true maximum -- no running maximum, and e4m3's 17 binades sit
       right below each tile's own largest weight.  Row sums stay fp32 (of the unquantised p).
       Operand layout of the instruction (measured, tools/f8_probe.hip): lane (i = lane % 32, h = lane / 32) holds row i of A (column i of B), bytes
       0..15 = k in [16 h, 16 h + 16), bytes 16..31 = k in [32 + 16 h, 48 + 16 h); its scale operand covers k in [32 h, 32 h + 32) of row i.
       Key order: row i of key block b is key 32 b + 16 ((i >> 2) & 1) + (i & 3) + 4 (i >> 3), so that score register r of lane half h' is key
       32 b + 16 h' + r and the packed P8 bytes (16 b + r) are the B operand of the PV product against V8^T rows = d, columns = the tile's keys in order.
       LDS ring: 4 slots of [K8 tile 64 x 128 B | V8^T tile 128 x 64 B] = 16 KiB; 16-B chunks of K row r at c ^ ((r >> 1) & 7), of V^T row d at
       c ^ ((d >> 2) & 3) (both ds_read_b128 patterns conflict-free).  LDS-DMA one tile ahead (V8^T of tile T-2 is still being read when K8 of T is).
       register map   a[0:127] O[j][db]   a[128:159] Q8 fragments [j][ks]   a[160:223] fragment ring: V^T[db] (4 x 8), K[b][ks] (4 x 8)
                      v[0:127] S[p][j][b]   v[128:159] P8[p][j]   v[160:191] srcC tuples -M[j]   v[192:197] lane LDS offsets K[ks][u], V[u]
                      v[200:203] LDS-DMA source offsets   v204 v205 l[j]   v[206:213] partial sums   v[214:217] E8M0 of x [p][j]   v218 v219 2^x
                      v220 v221 v222 E8M0 of eq, ek, ev   v223 -inf   v224 16 h   v225 v226 -M[j]   v227 v228 temporaries"""

    T0, LAK, LAV, VOFF, L, TP, XS, FS, EQ, EK, EV, NINF, HI16, NEGM, U, E = 160, 192, 196, 200, 204, 206, 214, 218, 220, 221, 222, 223, 224, 225, 227, 228
    QF, FR = 128, 160
    LEAD = KNOB.get("lead8", 6)

    def S(self, p, j, b):
        return 64 * p + 32 * j + 16 * b

    def P8(self, p, j):
        return 128 + 16 * p + 8 * j

    def frag_reg(self, f):
        return self.FR + 8 * f

    def issue_frag(self, em, f, slotK, slotV, tag=None):
        """f 0..3: V8^T d-block f of ring slot slotV; f 4..7: K8 (b, ks) = ((f - 4) >> 1, (f - 4) & 1) of ring slot slotK.  Two 16-byte reads each."""
        tag = f if tag is None else tag
        r = self.frag_reg(f)
        for u in range(2):
            if f < 4:
                em.ds(f"ds_read_b128 {ar(r + 4 * u, 4)}, v{self.LAV + u} offset:{slotV * 16384 + 8192 + f * 2048}", tag)
            else:
                b, ks = (f - 4) >> 1, (f - 4) & 1
                em.ds(f"ds_read_b128 {ar(r + 4 * u, 4)}, v{self.LAK + 2 * ks + u} offset:{slotK * 16384 + b * 4096}", tag)

    def mfmas(self, pa, pc):
        out = []
        for db in range(4):
            for j in range(2):
                d = ar(64 * j + 16 * db, 16)
                out.append((f"{MFMA8} {d}, {ar(self.frag_reg(db), 8)}, {vr(self.P8(pc, j), 8)}, {d}, v{self.EV}, v{self.XS + 2 * pc + j} op_sel_hi:[0,0,0]", db))
        for b in range(2):
            for ks in range(2):
                for j in range(2):
                    d = vr(self.S(pa, j, b), 16)
                    c = vr(self.T0 + 16 * j, 16) if ks == 0 else d
                    out.append((f"{MFMA8} {d}, {ar(self.frag_reg(4 + 2 * b + ks), 8)}, {ar(self.QF + 16 * j + 8 * ks, 8)}, {c}, v{self.EK}, v{self.EQ} op_sel_hi:[0,0,0]",
                                4 + 2 * b + ks))
        return out

    def valu_ops(self, pb):
        if "novalu" in ABLATE:
            return []
        ops = []
        for j in range(2):
            s = [self.S(pb, j, b) + r for b in range(2) for r in range(16)]          # byte order of the packed P8: 16 b + r
            tp = [self.TP + 4 * j + k for k in range(4)]
            # four partial sums; an add consumes values whose v_exp was issued at least four instructions earlier (gfx940 forwards a transcendental's
            # result to the NEXT instruction only through a hazard wait state)
            for i in range(0, 32, 4):
                for k in range(4):
                    ops.append(f"v_exp_f32 v{s[i + k]}, v{s[i + k]}")
                    if i == 8:
                        ops.append(f"v_add_f32 v{tp[k]}, v{s[k]}, v{s[4 + k]}")
                    elif i > 8:
                        ops.append(f"v_add_f32 v{tp[k]}, v{tp[k]}, v{s[i - 4 + k]}")
            for k in range(4):
                ops.append(f"v_add_f32 v{tp[k]}, v{tp[k]}, v{s[28 + k]}")
            ops += [f"v_add_f32 v{tp[0]}, v{tp[0]}, v{tp[1]}", f"v_add_f32 v{tp[2]}, v{tp[2]}, v{tp[3]}", f"v_add_f32 v{tp[0]}, v{tp[0]}, v{tp[2]}",
                    f"v_mov_b32 v{self.U}, v{tp[0]}", "s_nop 1", f"v_permlane32_swap_b32 v{self.U}, v{tp[0]}", "s_nop 0", f"v_add_f32 v{tp[0]}, v{tp[0]}, v{self.U}",
                    f"v_add_f32 v{self.L + j}, v{self.L + j}, v{tp[0]}",                                    # l[j] += the row's sum over all 64 keys (both halves)
                    f"v_frexp_exp_i32_f32 v{self.E}, v{tp[0]}",                                              # sum in [2^(e-1), 2^e)
                    f"v_add_u32 v{self.E}, 119, v{self.E}",                                                  # E8M0 of 2^(e - 8)
                    f"v_max_i32 v{self.XS + 2 * pb + j}, 1, v{self.E}",                                        # e + 119 <= 247 for any finite sum
                    f"v_lshlrev_b32 v{self.FS + j}, 23, v{self.XS + 2 * pb + j}"]                           # the same power of two as an fp32 number
            for w in range(8):
                d = self.P8(pb, j) + w
                ops.append(f"v_cvt_scalef32_pk_fp8_f32 v{d}, v{s[4 * w]}, v{s[4 * w + 1]}, v{self.FS + j}")
                ops.append(f"v_cvt_scalef32_pk_fp8_f32 v{d}, v{s[4 * w + 2]}, v{s[4 * w + 3]}, v{self.FS + j} op_sel:[0,0,0,1]")
        return ops

    def mask_lines(self, b, krem, tmp, label):
        """keys >= krem of key block b get -inf in the srcC tuples (score register r of lane half h = key 32 b + 16 h + r); skipped for full tiles"""
        out = [f"s_cmp_ge_i32 {krem}, 64", f"s_cbranch_scc1 {label}"]
        for r in range(16):
            out += [f"s_sub_i32 {tmp}, {krem}, {32 * b + r}", f"v_cmp_lt_i32 vcc, v{self.HI16}, {tmp}"]
            out += [f"v_cndmask_b32 v{self.T0 + 16 * j + r}, v{self.NINF}, v{self.NEGM + j}, vcc" for j in range(2)]
        out += ["s_nop 1", f"{label}:"]
        return out

    def generate(self):
        em = Emitter()
        SAVE_M0, CNT, KREM, TMP = "%0", "%1", "%2", "%3"
        RK, RV, KSTEP, VSTEP, WBASE, NITER, KREM0 = "%[rk]", "%[rv]", "%[kstep]", "%[vstep]", "%[wbase]", "%[niter]", "%[krem]"
        em.raw(f"s_mov_b32 {SAVE_M0}, m0")
        em.raw(f"s_mov_b32 {CNT}, {NITER}")
        em.raw(f"s_mov_b32 {KREM}, {KREM0}")
        for i in range(128):
            em.raw(f"v_accvgpr_write_b32 a{i}, 0")
        em.raw(f"v_mov_b32 v{self.NINF}, 0xff800000")
        for r in range(64, 128):                       # S[1]: the first B stage (tile -1) must produce p = 0
            em.raw(f"v_mov_b32 v{r}, v{self.NINF}")
        for r in list(range(128, 160)) + [self.L, self.L + 1]:
            em.raw(f"v_mov_b32 v{r}, 0")
        for r in range(self.XS, self.XS + 4):
            em.raw(f"v_mov_b32 v{r}, 127")
        for j in range(2):
            for r in range(16):
                em.raw(f"v_mov_b32 v{self.T0 + 16 * j + r}, v{self.NEGM + j}")
        for f in range(4):
            self.issue_frag(em, f, 0, 2, tag=("n", f))
        for _ in range(KNOB.get("pad4", 0)):      # placement experiment: shift the loop body by 4 bytes per unit (MI355X_MICROARCH.md "code-placement sensitivity")
            em.raw("s_nop 0")
        em.raw("L_w1f8_loop_%=:")
        for ph in range(4):
            em.raw("s_waitcnt vmcnt(0)")
            if "nosync" not in ABLATE:
                em.raw("s_barrier")
            dst = ((ph + 1) & 3) * 16384
            fill = []
            for k in range(4):
                isV = k >= 2
                fill.append([f"s_add_u32 m0, {WBASE}, {dst + (8192 if isV else 0) + (k & 1) * 1024}"])
                fill.append([f"buffer_load_dwordx4 v{self.VOFF + k}, {RV if isV else RK}, 0 offen lds",
                             f"v_add_u32 v{self.VOFF + k}, {VSTEP if isV else KSTEP}, v{self.VOFF + k}"])
            pa = ph & 1
            for ln in self.mask_lines(0, KREM, TMP, f"L_w1f8_m{2 * ph}_%="):
                em.raw(ln)
            need = {0: 0, 1: 2, 2: 4, 3: 6, 4: 8, 5: 10, 6: 12, 7: 14}
            slotV_next = (ph - 1) & 3                    # C of the NEXT iteration reads V8^T of tile T - 1
            post = [lambda f=f: self.issue_frag(em, f, 0, slotV_next, tag=("n", f)) for f in range(4)]
            em.retag({("n", f): f for f in range(4)})
            schedule(em, self.mfmas(pa, pa), lambda f: self.issue_frag(em, f, ph, (ph - 2) & 3), need, self.valu_ops(pa ^ 1), self.LEAD,
                     pre_issued=(0, 1, 2, 3), post_issue=post, fill_first=fill, raw_after={11: self.mask_lines(1, KREM, TMP, f"L_w1f8_m{2 * ph + 1}_%=")})
            em.raw(f"s_sub_i32 {KREM}, {KREM}, 64")
            em.raw(f"s_sub_u32 {CNT}, {CNT}, 1")
            em.raw(f"s_cmp_eq_u32 {CNT}, 0")
            if ph < 3:
                em.raw("s_cbranch_scc1 L_w1f8_done_%=")
            else:
                em.raw("s_cbranch_scc0 L_w1f8_loop_%=")
        em.raw("L_w1f8_done_%=:")
        em.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        em.raw("s_nop 7")
        em.raw("s_nop 7")
        em.raw(f"s_mov_b32 m0, {SAVE_M0}")
        return em.text() + "\n"


# ---------------------------------------------------------------------------------------------------------- dK, dV, head_dim 128
class Dkv128Loop:
    """DkvLoop for head_dim 128 (csrc/attention_hd128.hip): ONE 32-key block per wave -- dK^T, dV^T (4 d-blocks each) and the K, V
       fragments (8 k-steps each) already take 192 AGPRs -- streaming 64-row Q|dO tiles; per 32-row half-tile g:
         A(g): S[g&1] = -lse2[q] / c + Q_g K^T,  DP[g&1] = -delta[q] + dO_g V^T        16 MFMAs (8 k-steps each); q unscaled, so
         B(g): P = exp2(c S) -> PK (bf16), dS = P * DP -> DSK (bf16)                    64 VALU (16 of them the v_mul by c)
         C(g): dV^T[db] += dO_g^T P,  dK^T[db] += Q_g^T dS                              16 MFMAs on transpose-read fragments
       Every streamed fragment feeds one MFMA here (two at head_dim 64): the loop needs one 1 KiB LDS read per MFMA and is bound by
       LDS bandwidth, not by the matrix pipe.
       LDS: ring slot s at 32768 s = [Q tile 64 x 128 | dO tile], rows swizzled as in Fwd128Loop; statistics of slot s at
       131072 + 1024 s (layout as DkvLoop).
       register map   a[0:63] dk[db]  a[64:127] dv[db]  a[128:159] kf[ks]  a[160:191] vf[ks]  a[192:255] fragment ring (16 x 4)
                      v[0:63] S / DP [p] (s dp)   v[64:95] PK / DSK [p][cc]   v[96:111] srcC -lse2/c   v[112:127] srcC -delta
                      v[128:159] lane LDS offsets [slot pair][rows ks 0..7 | transposed (db, second read) 0..7]
                      v[160:167] LDS-DMA source offsets (Q x4, dO x4)   v168 statistics source offset   v169 statistics read base"""

    LA = 128
    VOFF = 160
    SB = 169
    FR = 192
    LEAD = KNOB.get("lead", 6)

    def S(self, p):
        return p * 32

    def DP(self, p):
        return p * 32 + 16

    def PK(self, p, cc):
        return 64 + p * 16 + cc * 4

    def DSK(self, p, cc):
        return 64 + p * 16 + 8 + cc * 4

    def frag_reg(self, f):
        return self.FR + 4 * (f % 16)

    def issue_frag(self, em, f, slotA, qbA, slotC, qbC, tag=None):
        """f 0..15: transposed fragments of the PREVIOUS half (even: dO tile, odd: Q tile; (cc, db) = (f >> 3, (f >> 1) & 3));
        16..23: Q rows ks; 24..31: dO rows ks; "cs" / "cd": the srcC tuples"""
        tag = f if tag is None else tag
        if f == "cs" or f == "cd":
            base = 96 if f == "cs" else 112
            for g in range(4):
                off = slotA * 1024 + 512 * qbA + 256 * (g >> 1) + 32 * (g & 1) + (64 if f == "cd" else 0)
                em.ds(f"ds_read_b128 {vr(base + 4 * g, 4)}, v{self.SB} offset:{off}", tag)
            return
        r = self.frag_reg(f)
        if f < 16:
            c = f >> 1
            cc, db = c >> 2, c & 3
            base = self.LA + 16 * (slotC >> 1) + 8 + 2 * db
            off = (slotC & 1) * 32768 + (16384 if (f & 1) == 0 else 0) + qbC * 8192 + cc * 4096
            em.ds(f"ds_read_b64_tr_b16 {ar(r, 2)}, v{base} offset:{off}", tag)
            em.ds(f"ds_read_b64_tr_b16 {ar(r + 2, 2)}, v{base + 1} offset:{off}", tag)
        else:
            isdo = f >= 24
            ks = (f - 16) & 7
            base = self.LA + 16 * (slotA >> 1) + ks
            em.ds(f"ds_read_b128 {ar(r, 4)}, v{base} offset:{(slotA & 1) * 32768 + (16384 if isdo else 0) + qbA * 8192}", tag)

    def mfmas(self, pa, pc):
        out = []
        for f in range(16):
            c = f >> 1
            cc, db = c >> 2, c & 3
            if (f & 1) == 0:
                d, b = ar(64 + 16 * db, 16), vr(self.PK(pc, cc), 4)
            else:
                d, b = ar(16 * db, 16), vr(self.DSK(pc, cc), 4)
            out.append((f"{MFMA} {d}, {ar(self.frag_reg(f), 4)}, {b}, {d}", f))
        for ks in range(8):
            d = vr(self.S(pa), 16)
            c = vr(96, 16) if ks == 0 else d
            out.append((f"{MFMA} {d}, {ar(self.frag_reg(16 + ks), 4)}, {ar(128 + 4 * ks, 4)}, {c}", 16 + ks))
        for ks in range(8):
            d = vr(self.DP(pa), 16)
            c = vr(112, 16) if ks == 0 else d
            out.append((f"{MFMA} {d}, {ar(self.frag_reg(24 + ks), 4)}, {ar(160 + 4 * ks, 4)}, {c}", 24 + ks))
        return out

    def valu_ops(self, pb):
        if "novalu" in ABLATE:
            return []

        def unit(p):
            s0, d0 = self.S(pb) + 2 * p, self.DP(pb) + 2 * p
            wp, wd = self.PK(pb, p >> 2) + (p & 3), self.DSK(pb, p >> 2) + (p & 3)
            return ([f"v_mul_f32 v{s0}, %[cs], v{s0}", f"v_mul_f32 v{s0 + 1}, %[cs], v{s0 + 1}", f"v_exp_f32 v{s0}, v{s0}", f"v_exp_f32 v{s0 + 1}, v{s0 + 1}"],
                    [f"v_mul_f32 v{d0}, v{d0}, v{s0}", f"v_mul_f32 v{d0 + 1}, v{d0 + 1}, v{s0 + 1}"],
                    [f"v_cvt_pk_bf16_f32 v{wp}, v{s0}, v{s0 + 1}", f"v_cvt_pk_bf16_f32 v{wd}, v{d0}, v{d0 + 1}"])
        ops = list(unit(0)[0])
        for t in range(9):
            x = unit(t + 1)[0] if t + 1 < 8 else [None] * 4
            m = unit(t)[1] if t < 8 else [None, None]
            c = unit(t - 1)[2] if 1 <= t else [None, None]
            ops += [o for o in (x[0], m[0], x[1], c[0], x[2], m[1], x[3], c[1]) if o is not None]
        return ops

    def half_step(self, em, slotA, qbA, slotC, qbC, pa, nxt, fill_first=()):
        need = {f: f for f in range(32)}
        need["cs"] = 16
        need["cd"] = 24
        post = [lambda f=f: self.issue_frag(em, f, 0, 0, nxt[0], nxt[1], tag=("n", f)) for f in range(4)]
        em.retag({("n", f): f for f in range(4)})
        schedule(em, self.mfmas(pa, pa), lambda f: self.issue_frag(em, f, slotA, qbA, slotC, qbC), need, self.valu_ops(pa ^ 1), self.LEAD,
                 pre_issued=(0, 1, 2, 3), post_issue=post, fill_first=fill_first)

    def generate(self):
        em = Emitter()
        SAVE_M0, CNT = "%0", "%1"
        RQ, RDO, RST, QSTEP, DSTEP, WBASE, SBASE, NITER = "%[rq]", "%[rdo]", "%[rst]", "%[qstep]", "%[dstep]", "%[wbase]", "%[sbase]", "%[niter]"
        em.raw(f"s_mov_b32 {SAVE_M0}, m0")
        em.raw(f"s_mov_b32 {CNT}, {NITER}")
        for i in range(128):
            em.raw(f"v_accvgpr_write_b32 a{i}, 0")
        for r in list(range(32, 64)) + list(range(64, 80)):      # S / DP[1], PK / DSK[0]
            em.raw(f"v_mov_b32 v{r}, 0")
        for f in range(4):
            self.issue_frag(em, f, 0, 0, 3, 0, tag=("n", f))
        for _ in range(KNOB.get("pad4", 0)):      # placement experiment: shift the loop body by 4 bytes per unit (MI355X_MICROARCH.md "code-placement sensitivity")
            em.raw("s_nop 0")
        em.raw("L_w1dkv128_loop_%=:")
        for ph in range(4):
            em.raw("s_waitcnt vmcnt(9)")
            if "nosync" not in ABLATE:
                em.raw("s_barrier")
            dst = ((ph + 2) & 3) * 32768
            fill = []
            for k in range(8):
                isdo = k >= 4
                fill.append([f"s_add_u32 m0, {WBASE}, {dst + (16384 if isdo else 0) + (k & 3) * 1024}"])
                fill.append([f"buffer_load_dwordx4 v{self.VOFF + k}, {RDO if isdo else RQ}, 0 offen lds",
                             f"v_add_u32 v{self.VOFF + k}, {DSTEP if isdo else QSTEP}, v{self.VOFF + k}"])
            fill.append([f"s_add_u32 m0, {SBASE}, {((ph + 2) & 3) * 1024}"])
            fill.append([f"buffer_load_dword v{self.VOFF + 8}, {RST}, 0 offen lds", f"v_add_u32 v{self.VOFF + 8}, 256, v{self.VOFF + 8}"])
            sp = (ph - 1) & 3
            self.half_step(em, ph, 0, sp, 0, 0, nxt=(sp, 1), fill_first=fill)
            self.half_step(em, ph, 1, sp, 1, 1, nxt=(ph, 0))
            em.raw(f"s_sub_u32 {CNT}, {CNT}, 1")
            em.raw(f"s_cmp_eq_u32 {CNT}, 0")
            if ph < 3:
                em.raw("s_cbranch_scc1 L_w1dkv128_done_%=")
            else:
                em.raw("s_cbranch_scc0 L_w1dkv128_loop_%=")
        em.raw("L_w1dkv128_done_%=:")
        em.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        em.raw("s_nop 7")
        em.raw("s_nop 7")
        em.raw(f"s_mov_b32 m0, {SAVE_M0}")
        return em.text() + "\n"


# --------------------------------------------------------------------------------------------------------------- dQ, head_dim 128
class Dq128Loop:
    """DqLoop for head_dim 128: ONE 32-row q-block per wave (dQ^T 64 + Q, dO fragments 64 AGPRs; two blocks would leave no room for the
       fragment ring), streaming 64-key K|V tiles (LDS image as Fwd128Loop); per 32-key half-tile g:
         A(g): S[g&1] = -lse2[q] / c + K_g Q^T,  DP[g&1] = -delta[q] + V_g dO^T         16 MFMAs; the constants are loop-invariant srcC tuples
         B(g): D[g&1] = bf16(exp2(c S) * DP)                                             56 VALU
         C(g): dQ^T[db] += K_g^T D                                                       8 MFMAs on transpose-read K fragments
       24 fragment reads per 24 MFMAs.  Keys past the end need no mask (their K rows are zero).
       register map   a[0:63] dq[db]   a[64:95] qf[ks]   a[96:127] dof[ks]   a[128:223] fragment ring (24 x 4)
                      v[0:63] S / DP [p]   v[64:79] D[p][cc]   v[80:95] srcC -lse2/c   v[96:111] srcC -delta
                      v[112:143] lane LDS offsets [slot pair][16]   v[144:151] LDS-DMA source offsets (K x4, V x4)"""

    LA = 112
    VOFF = 144
    FR = 128
    LEAD = KNOB.get("lead", 6)

    def S(self, p):
        return p * 32

    def DP(self, p):
        return p * 32 + 16

    def D(self, p, cc):
        return 64 + p * 8 + cc * 4

    def frag_reg(self, f):
        return self.FR + 4 * f

    def issue_frag(self, em, f, slotA, kbA, slotC, kbC, tag=None):
        """f 0..7: transposed K fragments (cc, db) = (f >> 2, f & 3) of the PREVIOUS half; 8..15: K rows ks; 16..23: V rows ks"""
        tag = f if tag is None else tag
        r = self.frag_reg(f)
        if f < 8:
            cc, db = f >> 2, f & 3
            base = self.LA + 16 * (slotC >> 1) + 8 + 2 * db
            off = (slotC & 1) * 32768 + kbC * 8192 + cc * 4096
            em.ds(f"ds_read_b64_tr_b16 {ar(r, 2)}, v{base} offset:{off}", tag)
            em.ds(f"ds_read_b64_tr_b16 {ar(r + 2, 2)}, v{base + 1} offset:{off}", tag)
        else:
            isv = f >= 16
            base = self.LA + 16 * (slotA >> 1) + ((f - 8) & 7)
            em.ds(f"ds_read_b128 {ar(r, 4)}, v{base} offset:{(slotA & 1) * 32768 + (16384 if isv else 0) + kbA * 8192}", tag)

    def mfmas(self, pa, pc):
        out = []
        for c in range(8):
            cc, db = c >> 2, c & 3
            d = ar(16 * db, 16)
            out.append((f"{MFMA} {d}, {ar(self.frag_reg(c), 4)}, {vr(self.D(pc, cc), 4)}, {d}", c))
        for ks in range(8):
            d = vr(self.S(pa), 16)
            c = vr(80, 16) if ks == 0 else d
            out.append((f"{MFMA} {d}, {ar(self.frag_reg(8 + ks), 4)}, {ar(64 + 4 * ks, 4)}, {c}", 8 + ks))
        for ks in range(8):
            d = vr(self.DP(pa), 16)
            c = vr(96, 16) if ks == 0 else d
            out.append((f"{MFMA} {d}, {ar(self.frag_reg(16 + ks), 4)}, {ar(96 + 4 * ks, 4)}, {c}", 16 + ks))
        return out

    def valu_ops(self, pb):
        if "novalu" in ABLATE:
            return []

        def unit(p):
            s0, d0 = self.S(pb) + 2 * p, self.DP(pb) + 2 * p
            w = self.D(pb, p >> 2) + (p & 3)
            return ([f"v_mul_f32 v{s0}, %[cs], v{s0}", f"v_mul_f32 v{s0 + 1}, %[cs], v{s0 + 1}", f"v_exp_f32 v{s0}, v{s0}", f"v_exp_f32 v{s0 + 1}, v{s0 + 1}"],
                    [f"v_mul_f32 v{s0}, v{d0}, v{s0}", f"v_mul_f32 v{s0 + 1}, v{d0 + 1}, v{s0 + 1}"],
                    f"v_cvt_pk_bf16_f32 v{w}, v{s0}, v{s0 + 1}")
        ops = list(unit(0)[0])
        for t in range(9):
            x = unit(t + 1)[0] if t + 1 < 8 else [None] * 4
            m = unit(t)[1] if t < 8 else [None, None]
            c = unit(t - 1)[2] if 1 <= t else None
            ops += [o for o in (x[0], m[0], x[1], x[2], m[1], x[3], c) if o is not None]
        return ops

    def half_step(self, em, slotA, kbA, slotC, kbC, pa, nxt, fill_first=()):
        need = {f: f for f in range(24)}
        post = [lambda f=f: self.issue_frag(em, f, 0, 0, nxt[0], nxt[1], tag=("n", f)) for f in range(4)]
        em.retag({("n", f): f for f in range(4)})
        schedule(em, self.mfmas(pa, pa), lambda f: self.issue_frag(em, f, slotA, kbA, slotC, kbC), need, self.valu_ops(pa ^ 1), self.LEAD,
                 pre_issued=(0, 1, 2, 3), post_issue=post, fill_first=fill_first)

    def generate(self):
        em = Emitter()
        SAVE_M0, CNT = "%0", "%1"
        RK, RV, KSTEP, VSTEP, WBASE, NITER = "%[rk]", "%[rv]", "%[kstep]", "%[vstep]", "%[wbase]", "%[niter]"
        em.raw(f"s_mov_b32 {SAVE_M0}, m0")
        em.raw(f"s_mov_b32 {CNT}, {NITER}")
        for i in range(64):
            em.raw(f"v_accvgpr_write_b32 a{i}, 0")
        for r in list(range(32, 64)) + list(range(64, 72)):      # S / DP[1], D[0]
            em.raw(f"v_mov_b32 v{r}, 0")
        for f in range(4):
            self.issue_frag(em, f, 0, 0, 3, 0, tag=("n", f))
        for _ in range(KNOB.get("pad4", 0)):      # placement experiment: shift the loop body by 4 bytes per unit (MI355X_MICROARCH.md "code-placement sensitivity")
            em.raw("s_nop 0")
        em.raw("L_w1dq128_loop_%=:")
        for ph in range(4):
            em.raw("s_waitcnt vmcnt(8)")
            if "nosync" not in ABLATE:
                em.raw("s_barrier")
            dst = ((ph + 2) & 3) * 32768
            fill = []
            for k in range(8):
                isv = k >= 4
                fill.append([f"s_add_u32 m0, {WBASE}, {dst + (16384 if isv else 0) + (k & 3) * 1024}"])
                fill.append([f"buffer_load_dwordx4 v{self.VOFF + k}, {RV if isv else RK}, 0 offen lds",
                             f"v_add_u32 v{self.VOFF + k}, {VSTEP if isv else KSTEP}, v{self.VOFF + k}"])
            sp = (ph - 1) & 3
            self.half_step(em, ph, 0, sp, 0, 0, nxt=(sp, 1), fill_first=fill)
            self.half_step(em, ph, 1, sp, 1, 1, nxt=(ph, 0))
            em.raw(f"s_sub_u32 {CNT}, {CNT}, 1")
            em.raw(f"s_cmp_eq_u32 {CNT}, 0")
            if ph < 3:
                em.raw("s_cbranch_scc1 L_w1dq128_done_%=")
            else:
                em.raw("s_cbranch_scc0 L_w1dq128_loop_%=")
        em.raw("L_w1dq128_done_%=:")
        em.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        em.raw("s_nop 7")
        em.raw("s_nop 7")
        em.raw(f"s_mov_b32 m0, {SAVE_M0}")
        return em.text() + "\n"


# ------------------------------------------------------------------------------------- dQ, head_dim 128, two q-blocks per wave
class Dq128x2Loop:
    """Dq128Loop with TWO 32-row q-blocks j per wave, so that every streamed K / V fragment feeds two MFMAs (the matrix pipe's energy floor is 0.74 J per
       TFLOP and every 1 KiB fragment read per MFMA adds 0.24: tools/mfma_energy_probe.hip, DESIGN section 4.2 -- one read per MFMA was the price of
       Dq128Loop).  dQ^T (128) + the Q and dO fragments of both blocks (64 + 64) fill the accumulator half of the register file, so the fragment ring
       lives in VGPRs and the -lse2 / -delta srcC tuples (64 registers for two blocks) are gone: the score chain starts from the inline constant 0 and
         B(g): p = exp2(fma(c, S, -lse2[q])),  dS = (DP - delta[q]) * p,  D[g&1][j] = bf16(dS)        144 VALU per half-step (9 per element pair)
         A(g): S[g&1][j] = K_g Q_j^T,  DP[g&1][j] = V_g dO_j^T                                         32 MFMAs
         C(g): dQ^T[j][db] += K_g^T D[g&1][j]                                                          16 MFMAs on transpose-read K fragments
       24 fragment reads per 48 MFMAs.  Pipeline fill / drain and keys past the end as in Dq128Loop: whatever dS such a key gets multiplies a zero K row.
       register map   a[0:127] dq[j][db]   a[128:191] qf[j][ks]   a[192:255] dof[j][ks]
                      v[0:127] S / DP [p][j] (s dp)   v[128:159] D[p][j][cc]   v[160:191] fragment ring (8 x 4)
                      v[192:223] lane LDS offsets [slot pair][16]   v[224:231] LDS-DMA source offsets (K x4, V x4)   v232 v233 -lse2[q_j]   v234 v235 -delta[q_j]"""

    LA = 192
    VOFF = 224
    NL = 232
    ND = 234
    FR = 160
    NFR = 8
    LEAD = KNOB.get("lead", 6)

    def S(self, p, j):
        return 64 * p + 32 * j

    def DP(self, p, j):
        return 64 * p + 32 * j + 16

    def D(self, p, j, cc):
        return 128 + 16 * p + 8 * j + 4 * cc

    def frag_reg(self, f):
        return self.FR + 4 * (f % self.NFR)

    def issue_frag(self, em, f, slotA, kbA, slotC, kbC, tag=None):
        """f 0..7: transposed K fragments (cc, db) = (f >> 2, f & 3) of the PREVIOUS half; 8..15: K rows ks; 16..23: V rows ks"""
        tag = f if tag is None else tag
        r = self.frag_reg(f)
        if f < 8:
            cc, db = f >> 2, f & 3
            base = self.LA + 16 * (slotC >> 1) + 8 + 2 * db
            off = (slotC & 1) * 32768 + kbC * 8192 + cc * 4096
            em.ds(f"ds_read_b64_tr_b16 {vr(r, 2)}, v{base} offset:{off}", tag)
            em.ds(f"ds_read_b64_tr_b16 {vr(r + 2, 2)}, v{base + 1} offset:{off}", tag)
        else:
            isv = f >= 16
            base = self.LA + 16 * (slotA >> 1) + ((f - 8) & 7)
            em.ds(f"ds_read_b128 {vr(r, 4)}, v{base} offset:{(slotA & 1) * 32768 + (16384 if isv else 0) + kbA * 8192}", tag)

    def mfmas(self, pa, pc):
        out = []
        for c in range(8):
            cc, db = c >> 2, c & 3
            for j in range(2):
                d = ar(64 * j + 16 * db, 16)
                out.append((f"{MFMA} {d}, {vr(self.frag_reg(c), 4)}, {vr(self.D(pc, j, cc), 4)}, {d}", c))
        for ks in range(8):
            for j in range(2):
                d = vr(self.S(pa, j), 16)
                out.append((f"{MFMA} {d}, {vr(self.frag_reg(8 + ks), 4)}, {ar(128 + 32 * j + 4 * ks, 4)}, {'0' if ks == 0 else d}", 8 + ks))
        for ks in range(8):
            for j in range(2):
                d = vr(self.DP(pa, j), 16)
                out.append((f"{MFMA} {d}, {vr(self.frag_reg(16 + ks), 4)}, {ar(192 + 32 * j + 4 * ks, 4)}, {'0' if ks == 0 else d}", 16 + ks))
        return out

    def valu_ops(self, pb):
        if "novalu" in ABLATE:
            return []

        def unit(u):
            j, p = u >> 3, u & 7
            s0, d0 = self.S(pb, j) + 2 * p, self.DP(pb, j) + 2 * p
            w = self.D(pb, j, p >> 2) + (p & 3)
            return ([f"v_fma_f32 v{s0}, %[cs], v{s0}, v{self.NL + j}", f"v_fma_f32 v{s0 + 1}, %[cs], v{s0 + 1}, v{self.NL + j}",
                     f"v_exp_f32 v{s0}, v{s0}", f"v_exp_f32 v{s0 + 1}, v{s0 + 1}"],
                    [f"v_add_f32 v{d0}, v{self.ND + j}, v{d0}", f"v_add_f32 v{d0 + 1}, v{self.ND + j}, v{d0 + 1}"],
                    [f"v_mul_f32 v{s0}, v{d0}, v{s0}", f"v_mul_f32 v{s0 + 1}, v{d0 + 1}, v{s0 + 1}"],
                    f"v_cvt_pk_bf16_f32 v{w}, v{s0}, v{s0 + 1}")
        ops = list(unit(0)[0])
        for t in range(17):
            x = unit(t + 1)[0] if t + 1 < 16 else [None] * 4
            a = unit(t)[1] if t < 16 else [None, None]
            m = unit(t)[2] if t < 16 else [None, None]
            c = unit(t - 1)[3] if 1 <= t else None
            # no instruction directly follows one it depends on; a v_exp result is first used two steps later
            ops += [o for o in (a[0], x[0], a[1], x[1], m[0], x[2], m[1], x[3], c) if o is not None]
        return ops

    def half_step(self, em, slotA, kbA, slotC, kbC, pa, nxt, fill_first=()):
        need = {f: 2 * f for f in range(24)}
        post = [lambda f=f: self.issue_frag(em, f, 0, 0, nxt[0], nxt[1], tag=("n", f)) for f in range(4)]
        em.retag({("n", f): f for f in range(4)})
        schedule(em, self.mfmas(pa, pa), lambda f: self.issue_frag(em, f, slotA, kbA, slotC, kbC), need, self.valu_ops(pa ^ 1), self.LEAD,
                 pre_issued=(0, 1, 2, 3), post_issue=post, fill_first=fill_first)

    def generate(self):
        em = Emitter()
        SAVE_M0, CNT = "%0", "%1"
        RK, RV, KSTEP, VSTEP, WBASE, NITER = "%[rk]", "%[rv]", "%[kstep]", "%[vstep]", "%[wbase]", "%[niter]"
        em.raw(f"s_mov_b32 {SAVE_M0}, m0")
        em.raw(f"s_mov_b32 {CNT}, {NITER}")
        for i in range(128):
            em.raw(f"v_accvgpr_write_b32 a{i}, 0")
        for r in list(range(64, 128)) + list(range(128, 144)):      # S / DP[1], D[0]
            em.raw(f"v_mov_b32 v{r}, 0")
        for f in range(4):
            self.issue_frag(em, f, 0, 0, 3, 0, tag=("n", f))
        for _ in range(KNOB.get("pad4", 0)):      # placement experiment: shift the loop body by 4 bytes per unit (MI355X_MICROARCH.md "code-placement sensitivity")
            em.raw("s_nop 0")
        em.raw("L_w1dq128x2_loop_%=:")
        for ph in range(4):
            em.raw("s_waitcnt vmcnt(8)")
            if "nosync" not in ABLATE:
                em.raw("s_barrier")
            dst = ((ph + 2) & 3) * 32768
            fill = []
            for k in range(8):
                isv = k >= 4
                fill.append([f"s_add_u32 m0, {WBASE}, {dst + (16384 if isv else 0) + (k & 3) * 1024}"])
                fill.append([f"buffer_load_dwordx4 v{self.VOFF + k}, {RV if isv else RK}, 0 offen lds",
                             f"v_add_u32 v{self.VOFF + k}, {VSTEP if isv else KSTEP}, v{self.VOFF + k}"])
            sp = (ph - 1) & 3
            self.half_step(em, ph, 0, sp, 0, 0, nxt=(sp, 1), fill_first=fill)
            self.half_step(em, ph, 1, sp, 1, 1, nxt=(ph, 0))
            em.raw(f"s_sub_u32 {CNT}, {CNT}, 1")
            em.raw(f"s_cmp_eq_u32 {CNT}, 0")
            if ph < 3:
                em.raw("s_cbranch_scc1 L_w1dq128x2_done_%=")
            else:
                em.raw("s_cbranch_scc0 L_w1dq128x2_loop_%=")
        em.raw("L_w1dq128x2_done_%=:")
        em.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        em.raw("s_nop 7")
        em.raw("s_nop 7")
        em.raw(f"s_mov_b32 m0, {SAVE_M0}")
        return em.text() + "\n"


# --------------------------------------------------------------------------------------------------------------------- GEMM loop
class GemmLoop:
    """C^T tile = W A^T for a 256 (M) x 128 (N) output tile, operands both K-contiguous (x [M, K], W [N, K]: y = x W^T), BK = 64 per
       stage, 4 waves as 2 (M) x 2 (N): a wave owns 128 x 64 of the tile = 2 (n) x 4 (m) accumulator blocks, D[n][m] (column m on the
       lanes, so a lane ends up with runs of 4 consecutive n = 8-byte pieces of a C row).
       LDS stage (48 KiB, ring of 3): A panel = 4 sub-tiles of [64 rows][64 k] at +0, W panel = 2 sub-tiles at +32768, every sub-tile
       in the attn_w1.h image; each wave brings 8 pieces of A sub-tile `wave` and 4 pieces of W sub-tile wave/2 per stage.
       register map   a[0:127] acc[ni][mi]   a[128:175] fragment ring: 2 sets x (W n0 n1, A m0..m3)
                      v[0:11] LDS-DMA source offsets (8 A pieces, 4 W pieces; + 128 bytes per stage)
                      v[12:35] lane read offsets [stage][A ks0..3, W ks0..3] (wave's panel offset and the stage base folded in)"""

    def frag(self, setno, i):          # i: 0,1 = W n-block; 2..5 = A m-block
        return 128 + 24 * setno + 4 * i

    def issue_set(self, em, stage, ks, setno, tag):
        for i in range(6):
            isW = i < 2
            blk = i if isW else i - 2                     # 32-row block inside the wave's panel
            base = 12 + 8 * stage + (4 if isW else 0) + ks
            em.ds(f"ds_read_b128 {ar(self.frag(setno, i), 4)}, v{base} offset:{4096 * blk}", tag)

    def generate(self):
        em = Emitter()
        SAVE_M0, CNT = "%0", "%1"
        RA, RW, WBA, WBW, NITER = "%[ra]", "%[rw]", "%[wba]", "%[wbw]", "%[niter]"
        em.raw(f"s_mov_b32 {SAVE_M0}, m0")
        em.raw(f"s_mov_b32 {CNT}, {NITER}")
        for i in range(128):
            em.raw(f"v_accvgpr_write_b32 a{i}, 0")
        for _ in range(KNOB.get("pad4", 0)):      # placement experiment: shift the loop body by 4 bytes per unit (MI355X_MICROARCH.md "code-placement sensitivity")
            em.raw("s_nop 0")
        em.raw("L_w1gemm_loop_%=:")
        for st in range(3):
            # stage `st` landed for every wave (this wave's 12 pieces of the NEXT stage may still fly); everyone is past stage st-1
            em.raw("s_waitcnt vmcnt(12)")
            em.raw("s_barrier")
            nxt = (st + 2) % 3
            dma = []
            for k in range(12):
                isW = k >= 8
                dst = nxt * 49152 + (32768 if isW else 0) + (k - 8 if isW else k) * 1024
                dma.append([f"s_add_u32 m0, {WBW if isW else WBA}, {dst}"])
                dma.append([f"buffer_load_dwordx4 v{k}, {RW if isW else RA}, 0 offen lds", f"v_add_u32 v{k}, 128, v{k}"])
            self.issue_set(em, st, 0, 0, ("s", 0))
            slot = 0
            for ks in range(4):
                setno = ks & 1
                if ks + 1 < 4:
                    pass
                em.wait_tag(("s", ks))
                mf = []
                for ni in range(2):
                    for mi in range(4):
                        d = ar((ni * 4 + mi) * 16, 16)
                        mf.append(f"{MFMA} {d}, {ar(self.frag(setno, ni), 4)}, {ar(self.frag(setno, 2 + mi), 4)}, {d}")
                for i, text in enumerate(mf):
                    em.raw(text)
                    if i == 1 and ks + 1 < 4:      # the next k-step's fragments go to the other register set (its readers are done)
                        self.issue_set(em, st, ks + 1, setno ^ 1, ("s", ks + 1))
                    if slot < len(dma):
                        for t in dma[slot]:
                            em.raw(t)
                        slot += 1
            assert slot == len(dma)
            em.raw(f"s_sub_u32 {CNT}, {CNT}, 1")
            em.raw(f"s_cmp_eq_u32 {CNT}, 0")
            if st < 2:
                em.raw("s_cbranch_scc1 L_w1gemm_done_%=")
            else:
                em.raw("s_cbranch_scc0 L_w1gemm_loop_%=")
        em.raw("L_w1gemm_done_%=:")
        em.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
        em.raw("s_nop 7")
        em.raw("s_nop 7")
        em.raw(f"s_mov_b32 m0, {SAVE_M0}")
        return em.text() + "\n"


def clobbers(ranges, aranges=()):
    regs = []
    for lo, hi in ranges:
        regs += [f'"v{i}"' for i in range(lo, hi + 1)]
    for lo, hi in aranges:
        regs += [f'"a{i}"' for i in range(lo, hi + 1)]
    lines = [", ".join(regs[i:i + 16]) for i in range(0, len(regs), 16)]
    return ",\n".join("    " + ln for ln in lines) + "\n"


TARGETS = {"w1_dq_loop.inc": lambda: DqLoop().generate(),
           "w1_dq_clobbers.inc": lambda: clobbers([(0, 159)], [(128, 151)]),
           "w1_dkv_loop.inc": lambda: DkvLoop().generate(),
           "w1_dkv_clobbers.inc": lambda: clobbers([(0, 223)], [(192, 223)]),
           "w1_fwd128_loop.inc": lambda: Fwd128Loop().generate(),
           "w1_fwd128_clobbers.inc": lambda: clobbers([(0, 127), (185, 185)], [(192, 255)]),
           "w1_fwd128f8_loop.inc": lambda: Fwd128F8Loop().generate(),
           "w1_fwd128f8_clobbers.inc": lambda: clobbers([(0, 191), (206, 219), (223, 223), (227, 228)], [(160, 223)]),
           "w1_dkv128_loop.inc": lambda: Dkv128Loop().generate(),
           "w1_dkv128_clobbers.inc": lambda: clobbers([(0, 127)], [(192, 255)]),
           "w1_dq128_loop.inc": lambda: Dq128Loop().generate(),
           "w1_dq128_clobbers.inc": lambda: clobbers([(0, 79)], [(128, 223)]),
           "w1_dq128x2_loop.inc": lambda: Dq128x2Loop().generate(),
           "w1_dq128x2_clobbers.inc": lambda: clobbers([(0, 191)], []),
           "w1_gemm_loop.inc": lambda: GemmLoop().generate(),
           "w1_gemm_clobbers.inc": lambda: clobbers([], [(128, 175)]),
           "w1_fwd_loop.inc": lambda: FwdLoop().generate(),
           "w1_fwd_clobbers.inc": lambda: clobbers([(0, 127), (157, 157)], [(96, 127)] + ([(128, 131)] if KNOB.get("mfsum", 1) else [])),
           # what the C++ around the forward loop must know about the knobs the loop was generated with (attention_w1.hip: selector operand, epilogue)
           "w1_fwd_knobs.inc": lambda: f"#define W1_FWD_MFSUM {1 if KNOB.get('mfsum', 1) else 0}\n"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    bad = 0
    for name, gen in TARGETS.items():
        # the GEMM probe's loop belongs to a variant build only (tools/variants/gemm_w1.hip); everything else is product source
        path = os.path.join(ROOT, "tools", "variants", name) if name.startswith("w1_gemm_") else os.path.join(ROOT, "videogpa_amd", "csrc", name)
        text = f"// GENERATED by tools/gen_w1_asm.py -- do not edit; regenerate with `python tools/gen_w1_asm.py`\n" + gen()
        if a.check:
            if not os.path.exists(path) or open(path).read() != text:
                print(f"{name}: out of date")
                bad = 1
        else:
            open(path, "w").write(text)
            print(f"wrote {path} ({text.count(chr(10))} lines)")
    sys.exit(bad)


if __name__ == "__main__":
    main()
