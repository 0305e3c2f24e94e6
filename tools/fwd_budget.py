"""Instruction budget of the generated forward loop (videogpa_amd/csrc/w1_fwd_loop.inc, tools/gen_w1_asm.py::FwdLoop) by CLASS, per half-step (32 keys x 64 query rows of
one wave: 16 MFMAs), next to what the PMC pass measured over the launch.  Static part: parse the checked-in loop body; the loop's four phases
hold 8 half-steps.  Issue-cycle weights are the measured per-instruction costs of profiles/r01p_coissue_table.txt / profiles/HISTORY.md (v_fma_f32 = 4 cycles).
    python tools/fwd_budget.py [--pmc gpurun_out/pmc_fwd_budget.json] [--md profiles/r06_fwd_budget.md]"""
import argparse
import json
import os
import re
from collections import OrderedDict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CLASSES = OrderedDict([
    ("mfma 32x32x16 bf16", (r"^v_mfma_f32_32x32x16", 32.0, "matrix pipe: 8 passes x 4 cycles")),
    ("mfma 16x16x32 bf16 (row sums: sparse selector x packed P)", (r"^v_mfma_f32_16x16x32", 16.0, "matrix pipe: 4 passes; replaces the 32 row-sum adds since round 6 (mfsum)")),
    ("exp2 (v_exp_f32)", (r"^v_exp_f32", 8.0, "one per score; transcendental unit, measured 1.6-2.7 x v_fma_f32; gfx950 has no packed / bf16 form (llvm-mc rejects v_exp_bf16, v_pk_exp_f16)")),
    ("row-sum add (v_add_f32)", (r"^v_add_f32", 4.0, "one per score into 4 partial sums per q-block; v_pk_add_f32 halves the count but holds the matrix pipe ~13 cycles each (r01p_coissue_table)")),
    ("bf16 pack (v_cvt_pk_bf16_f32)", (r"^v_cvt_pk_bf16_f32", 5.8, "one per two scores: the PV product's B operand")),
    ("address (v_add_u32: LDS-DMA source offsets)", (r"^v_add_u32", 4.0, "4 per 64-key tile")),
    ("other VALU (mask block: cmp / cndmask, branch-skipped except on the ragged tile)", (r"^v_(cmp|cndmask|mov)", 4.0, "executed on the last tile only")),
    ("LDS fragment reads (ds_read_b128 / ds_read_b64_tr_b16)", (r"^ds_read", 0.0, "issue only; 8 KiB per half-step")),
    ("LDS-DMA (buffer_load ... lds)", (r"^buffer_load", 0.0, "4 per 64-key tile, 2 tiles ahead")),
    ("SALU / waits / barrier", (r"^s_", 0.0, "scalar pipe")),
])


def static_budget():
    src = open(os.path.join(ROOT, "videogpa_amd", "csrc", "w1_fwd_loop.inc")).read()
    lines = [m.group(1) for m in re.finditer(r'^\s*"(.*?)\\n\\t"', src, flags=re.M)]
    a = next(i for i, ln in enumerate(lines) if ln.startswith("L_w1fwd_loop"))
    b = next(i for i, ln in enumerate(lines) if ln.startswith("L_w1fwd_done"))
    body = lines[a + 1:b]
    # the mask blocks are skipped by their branch on every tile but the last: count them apart
    out = OrderedDict((k, 0) for k in CLASSES)
    skipped = 0
    in_mask = False
    for ln in body:
        if ln.startswith("s_cbranch_scc1 L_w1fwd_m"):
            in_mask = True
            out["SALU / waits / barrier"] += 1
            continue
        if re.match(r"^L_w1fwd_m\d+_", ln):
            in_mask = False
            continue
        if ln.endswith(":"):
            continue
        if in_mask:
            skipped += 1
            continue
        for k, (pat, _, _) in CLASSES.items():
            if re.match(pat, ln):
                out[k] += 1
                break
        else:
            raise SystemExit(f"unclassified instruction: {ln}")
    return out, skipped, len(body)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--pmc", default=os.path.join(ROOT, "gpurun_out", "pmc_fwd_budget.json"))
    ap.add_argument("--md", default=None)
    a = ap.parse_args()
    counts, skipped, n_body = static_budget()
    HS = 8.0
    rows = []
    for k, n in counts.items():
        _, cyc, note = CLASSES[k]
        rows.append((k, n / HS, cyc, n / HS * cyc, note))
    valu = sum(r[3] for r in rows if r[0].startswith(("exp2", "row-sum", "bf16 pack", "address")))
    mfma = sum(r[3] for r in rows if r[0].startswith("mfma"))
    txt = ["| class | instructions per half-step | issue cycles each | cycles per half-step | note |", "|---|---|---|---|---|"]
    for k, n, cyc, tot, note in rows:
        txt.append(f"| {k} | {n:.2f} | {cyc if cyc else '-'} | {tot:.0f} | {note} |")
    txt.append(f"| **VALU total / matrix pipe** | | | **{valu:.0f} / {mfma:.0f}** | one wave per SIMD: both streams come from the same wave; perfect overlap would need "
               f"max({valu:.0f}, {mfma:.0f}) = {max(valu, mfma):.0f} cycles, i.e. mfma_busy <= {mfma / max(valu, mfma):.2f} |")
    txt.append(f"\n{n_body} instructions in the loop body (4 phases = 8 half-steps), {skipped} of them inside the eight branch-skipped mask blocks.")
    if os.path.isfile(a.pmc):
        p = json.load(open(a.pmc))
        txt.append("\nPMC over one launch at the headline shape (2 x 48 heads x 17 776 x 64; `tools/pmc_fwd_budget.sh`, mean over the dispatches of `attn_fwd_w1_kernel<false>`):\n")
        txt.append("| counter | value | per half-step and wave | static count |")
        txt.append("|---|---|---|---|")
        half_steps = p.get("half_steps_per_launch")
        for c, v in p["counters"].items():
            per = v / half_steps if half_steps else float("nan")
            st = {"SQ_INSTS_VALU": valu_count(counts), "SQ_INSTS_MFMA": sum(v for k, v in counts.items() if k.startswith("mfma")) / HS, "SQ_INSTS_VALU_TRANS_F32": counts["exp2 (v_exp_f32)"] / HS,
                  "SQ_INSTS_LDS": counts["LDS fragment reads (ds_read_b128 / ds_read_b64_tr_b16)"] / HS}.get(c)
            txt.append(f"| {c} | {v:.4g} | {per:.2f} | {'' if st is None else f'{st:.2f}'} |")
        for k, v in p.get("derived", {}).items():
            txt.append(f"\n{k}: {v}")
    out = "\n".join(txt)
    print(out)
    if a.md:
        with open(a.md, "w") as f:
            f.write(out + "\n")


def valu_count(counts):
    return sum(v for k, v in counts.items() if k.startswith(("exp2", "row-sum", "bf16 pack", "address"))) / 8.0


if __name__ == "__main__":
    main()
