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

    def text(self):
        return "\n".join('    "%s\\n\\t"' % ln for ln in self.lines)


# ---------------------------------------------------------------------------------------------------------------- dQ loop
class DqLoop:
    """dQ:  per 32-key half-tile g (tile = g >> 1 in the K|V ring, key block kb = g & 1), two 32-row q-blocks j per wave:
         A(g): S[g&1][j] = -lse + K_g Q_j^T, DP[g&1][j] = -delta + V_g dO_j^T          20 MFMAs (4 folds + 8 + 8)
         B(g): D[g&1][j] = bf16(exp2(S) * DP)                                             80 VALU
         C(g): dq[j] += K_g^T D[g&1][j]                                                   8 MFMAs (transpose-read K)
       half-step(g) issues A(g+1) | B(g) | C(g-1) interleaved.  The loop is uniform from g = -1 to 2 nt: the pipeline is
       filled and drained with zeros (S, DP, D start as 0; the slot "before the first tile" is zero-filled by the caller;
       tiles past the end arrive as zeros because their rows are out of the buffer descriptor's range), and rows past S
       need no masking at all: their K rows are zero, so whatever dS they get multiplies 0.

       register map      a[0:63] dq[j][db]   a[64:95] qf[j][ks]   a[96:127] dof[j][ks]   a[128:143] qx0 qx1 dx0 dx1   a[144:147] kx
                         v[0:127] S/DP[p][j] (p-major: s0 s1 dp0 dp1)   v[128:159] D[p][j][cc]   v[160:183] fragment ring (6)
                         v[192:199] lane LDS offsets   v[200:203] LDS-DMA source offsets (K0 K1 V0 V1; advanced per tile)"""

    LA = 192
    VOFF = 200
    FR = 160
    NFR = 6
    LEAD = KNOB.get("lead", 6)       # fragment reads are issued this many MFMAs ahead of their first use

    def S(self, p, j):
        return p * 64 + j * 16

    def DP(self, p, j):
        return p * 64 + 32 + j * 16

    def D(self, p, j, cc):
        return 128 + p * 16 + j * 8 + cc * 4

    def frag_reg(self, f):
        return self.FR + 4 * (f % self.NFR)

    def issue_frag(self, em, f, slotA, kbA, slotC, kbC):
        r = self.frag_reg(f)
        if f < 4:      # K rows, ks = f
            em.ds(f"ds_read_b128 {vr(r, 4)}, v{self.LA + f} offset:{slotA * 16384 + kbA * 4096}", f)
        elif f < 8:    # V rows
            em.ds(f"ds_read_b128 {vr(r, 4)}, v{self.LA + f - 4} offset:{slotA * 16384 + 8192 + kbA * 4096}", f)
        else:          # K transposed, (cc, db) = ((f-8) >> 1, (f-8) & 1)
            c = f - 8
            cc, db = c >> 1, c & 1
            off = slotC * 16384 + kbC * 4096 + cc * 2048
            em.ds(f"ds_read_b64_tr_b16 {vr(r, 2)}, v{self.LA + 4 + 2 * db} offset:{off}", f)
            em.ds(f"ds_read_b64_tr_b16 {vr(r + 2, 2)}, v{self.LA + 5 + 2 * db} offset:{off}", f)

    def mfma(self, i, pa, pc):
        """text of MFMA slot i (0..27) and the fragment it needs (or None)"""
        j = i & 1
        if i < 2:
            return f"{MFMA} {vr(self.S(pa, j), 16)}, {ar(144, 4)}, {ar(128 + 4 * j, 4)}, 0", None
        if i < 4:
            return f"{MFMA} {vr(self.DP(pa, j), 16)}, {ar(144, 4)}, {ar(136 + 4 * j, 4)}, 0", None
        if i < 12:
            ks = (i - 4) >> 1
            d = vr(self.S(pa, j), 16)
            return f"{MFMA} {d}, {vr(self.frag_reg(ks), 4)}, {ar(64 + 16 * j + 4 * ks, 4)}, {d}", ks
        if i < 20:
            ks = (i - 12) >> 1
            d = vr(self.DP(pa, j), 16)
            return f"{MFMA} {d}, {vr(self.frag_reg(4 + ks), 4)}, {ar(96 + 16 * j + 4 * ks, 4)}, {d}", 4 + ks
        c = (i - 20) >> 1
        cc, db = c >> 1, c & 1
        d = ar(32 * j + 16 * db, 16)
        return f"{MFMA} {d}, {vr(self.frag_reg(8 + c), 4)}, {vr(self.D(pc, j, cc), 4)}, {d}", 8 + c

    def valu_ops(self, pb):
        """B stage on S[pb], DP[pb] -> D[pb]: 16 units of (exp, exp, mul, mul, cvt), software-pipelined by one unit"""
        def unit(u):
            j, p = u >> 3, u & 7
            s0, d0 = self.S(pb, j) + 2 * p, self.DP(pb, j) + 2 * p
            w = self.D(pb, j, p >> 2) + (p & 3)
            return ([f"v_exp_f32 v{s0}, v{s0}", f"v_exp_f32 v{s0 + 1}, v{s0 + 1}"],
                    [f"v_mul_f32 v{s0}, v{d0}, v{s0}", f"v_mul_f32 v{s0 + 1}, v{d0 + 1}, v{s0 + 1}", f"v_cvt_pk_bf16_f32 v{w}, v{s0}, v{s0 + 1}"])
        ops = list(unit(0)[0])
        for u in range(16):
            if u + 1 < 16:
                ops += unit(u + 1)[0]
            ops += unit(u)[1]
        return ops

    def half_step(self, em, slotA, kbA, slotC, kbC, pa, extra_valu=()):
        """A writes S/DP[pa]; B reads S/DP[pa ^ 1], writes D[pa ^ 1]; C reads D[pa]"""
        pb, pc = pa ^ 1, pa
        valu = list(extra_valu) + ([] if "novalu" in ABLATE else self.valu_ops(pb))
        nv = len(valu)
        need = {f: 4 + 2 * f for f in range(12)}
        issued = set()
        for f in range(12):                      # fragments wanted within the first LEAD MFMAs
            if need[f] - self.LEAD <= 0:
                self.issue_frag(em, f, slotA, kbA, slotC, kbC)
                issued.add(f)
        vk = 0
        for i in range(28):
            text, f = self.mfma(i, pa, pc)
            if f is not None and (i & 1) == 0:
                em.wait_tag(f)
            if "nomfma" not in ABLATE:
                em.raw(text)
            for g in range(12):
                if g not in issued and need[g] - self.LEAD <= i + 1:
                    self.issue_frag(em, g, slotA, kbA, slotC, kbC)
                    issued.add(g)
            vend = nv * (i + 1) // 28
            while vk < vend:
                em.raw(valu[vk])
                vk += 1
        assert vk == nv and len(issued) == 12

    def generate(self):
        em = Emitter()
        # operands: %0-%3 scratch SGPRs (=&s), %4.. inputs -- see attention_w1.hip
        SAVE_M0, CNT, T0 = "%0", "%1", "%2"
        RK, RV, KSTEP, VSTEP, WBASE, NITER = "%[rk]", "%[rv]", "%[kstep]", "%[vstep]", "%[wbase]", "%[niter]"
        em.raw(f"s_mov_b32 {SAVE_M0}, m0")
        em.raw(f"s_mov_b32 {CNT}, {NITER}")
        for i in range(64):
            em.raw(f"v_accvgpr_write_b32 a{i}, 0")
        for r in list(range(64, 128)) + list(range(128, 144)):      # S/DP[1], D[0]
            em.raw(f"v_mov_b32 v{r}, 0")
        em.raw("L_w1dq_loop_%=:")
        for ph in range(4):
            # tile i (ring slot ph) must have landed for every wave; every read of tile i-2's slot is done -> DMA tile i+2 into it
            em.raw("s_waitcnt vmcnt(4) lgkmcnt(0)")
            em.drain()
            if "nosync" not in ABLATE:
                em.raw("s_barrier")
            dst = ((ph + 2) & 3) * 16384
            adds = []
            for k, (rs, extra) in enumerate([(RK, 0), (RK, 1024), (RV, 8192), (RV, 9216)]):
                em.raw(f"s_add_u32 m0, {WBASE}, {dst + extra}")
                em.raw("s_nop 0")
                em.raw(f"buffer_load_dwordx4 v{self.VOFF + k}, {rs}, 0 offen lds")
                adds.append(f"v_add_u32 v{self.VOFF + k}, {KSTEP if k < 2 else VSTEP}, v{self.VOFF + k}")
            sp = (ph - 1) & 3
            # half-step(2i-1): A(2i) -> S[0] (tile i, kb 0) | B(2i-1) on S[1] -> D[1] | C(2i-2): D[0], tile i-1, kb 0
            self.half_step(em, ph, 0, sp, 0, 0, extra_valu=adds)
            # half-step(2i):   A(2i+1) -> S[1] (tile i, kb 1) | B(2i) on S[0] -> D[0] | C(2i-1): D[1], tile i-1, kb 1
            self.half_step(em, ph, 1, sp, 1, 1)
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


def clobbers(ranges):
    regs = []
    for lo, hi in ranges:
        regs += [f'"v{i}"' for i in range(lo, hi + 1)]
    lines = [", ".join(regs[i:i + 16]) for i in range(0, len(regs), 16)]
    return ",\n".join("    " + ln for ln in lines) + "\n"


TARGETS = {"w1_dq_loop.inc": lambda: DqLoop().generate(),
           "w1_dq_clobbers.inc": lambda: clobbers([(0, 183)])}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    a = ap.parse_args()
    bad = 0
    for name, gen in TARGETS.items():
        path = os.path.join(ROOT, "videogpa_amd", "csrc", name)
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
