"""Per-kernel HBM traffic, matrix-pipe occupancy and clock from the PMC passes of tools/profile_round.sh:
    python tools/pmc_traffic.py gpurun_out TAG   ->  gpurun_out/pmc_traffic_TAG.json  (copy to profiles/pmc_traffic.json for bench.py)
HBM bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) KB: on gfx950 FETCH_SIZE counts 128-byte requests as 64 bytes (MI355X_MICROARCH.md, HBM),
WRITE_SIZE is taken as reported.  GRBM_GUI_ACTIVE comes back summed over the 8 XCDs: active cycles = GRBM_GUI_ACTIVE / 8;
mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (active cycles x 1024 SIMDs); clock = active cycles / kernel duration."""
import collections
import csv
import glob
import json
import sys

root, tag = sys.argv[1], sys.argv[2]


def kname(full):
    """kernel name with its template arguments, without the parameter list"""
    depth, out = 0, []
    for ch in full.replace("void ", ""):
        if ch == "<":
            depth += 1
        if ch == "(" and depth == 0:
            break
        out.append(ch)
        if ch == ">":
            depth -= 1
    return "".join(out).strip()


acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{root}/pmc_{tag}_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[kname(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
dur = {}
for stats in (f"{root}/{tag}_attn_kernel_stats.csv", f"{root}/{tag}_attn128_kernel_stats.csv"):
    try:
        for r in csv.DictReader(open(stats)):
            dur[kname(r["Name"])] = float(r["AverageNs"])
    except OSError as e:
        print("no attention kernel stats:", e)
# the full-round launches carry the bench names; tail-split (<true>) launches, merges and the redo pass keep their own
alias = {"attn_fwd_pipe_kernel<2, 4, false>": "attn_fwd_kernel (online-softmax form)", "attn_fwd_w1_kernel<false>": "attn_fwd_kernel",
         "attn_bwd_dkv_w1_kernel<false>": "attn_bwd_dkv_kernel", "attn_bwd_dq_w1_kernel<false>": "attn_bwd_dq_kernel",
         "attn_bwd_dkv_kernel<false>": "attn_bwd_dkv_kernel (2 waves per SIMD)", "attn_bwd_dq_kernel<2, false>": "attn_bwd_dq_kernel (2 waves per SIMD)",
         "w1_bwd_prep_kernel": "attn_delta_kernel", "attn128_fwd_w1_kernel": "attn128_fwd"}
out = {}
for k, cs in acc.items():
    if k.startswith("at::") or "Cijk" in k or "FETCH_SIZE" not in cs or "WRITE_SIZE" not in cs:
        continue
    mean = lambda name: sum(cs[name]) / len(cs[name]) if name in cs else None
    fetch, write = mean("FETCH_SIZE"), mean("WRITE_SIZE")
    e = {"fetch_size_kb_reported": fetch, "write_size_kb_reported": write, "hbm_bytes_per_launch": (2.0 * fetch + write) * 1024.0,
         "note": "2 x FETCH_SIZE (gfx950 wide-load correction) + WRITE_SIZE, KB -> bytes; mean over the dispatches of this kernel"}
    gui, busy, nm = mean("GRBM_GUI_ACTIVE"), mean("SQ_VALU_MFMA_BUSY_CYCLES"), mean("SQ_INSTS_MFMA")
    gui = gui / 8.0 if gui else gui
    if gui and busy:
        e["mfma_busy"] = busy / (gui * 1024.0)
        e["mfma_instructions"] = nm
    if gui and k in dur:
        e["clock_mhz"] = gui / dur[k] * 1e3
        e["avg_ns_under_rocprof"] = dur[k]
    if mean("SQ_LDS_IDX_ACTIVE"):
        e["lds_bank_conflict_frac"] = mean("SQ_LDS_BANK_CONFLICT") / mean("SQ_LDS_IDX_ACTIVE")
    out[alias.get(k, k)] = e
# bench.py times the head_dim-128 backward as ONE entry (delta + dK/dV + dQ launches of vgpa_attn128_bwd): bytes add up, occupancy and clock are
# duration-weighted means of the two matrix kernels
dq_name = "attn128_dq_w1x2_kernel" if "attn128_dq_w1x2_kernel" in out else "attn128_dq_w1_kernel"      # two q-blocks per wave since round 4
parts = [out[k] for k in ("attn128_dkv_w1_kernel", dq_name) if k in out]
if len(parts) == 2:
    e = {"hbm_bytes_per_launch": sum(p["hbm_bytes_per_launch"] for p in parts) + out.get("attn128_delta_kernel", {}).get("hbm_bytes_per_launch", 0.0),
         "note": f"attn128_dkv_w1_kernel + {dq_name} (+ attn128_delta_kernel) of one vgpa_attn128_bwd call"}
    if all("avg_ns_under_rocprof" in p for p in parts):
        w = [p["avg_ns_under_rocprof"] for p in parts]
        for key in ("mfma_busy", "clock_mhz"):
            if all(key in p for p in parts):
                e[key] = sum(p[key] * wi for p, wi in zip(parts, w)) / sum(w)
        e["avg_ns_under_rocprof"] = sum(w)
    out["attn128_bwd"] = e
out["__source__"] = {"profile_round": tag, "made_by": "tools/profile_round.sh -> tools/pmc_traffic.py", "files": f"profiles/{tag}_pmc_traffic.json"}
json.dump(out, open(f"{root}/pmc_traffic_{tag}.json", "w"), indent=1)
for k, v in out.items():
    if "hbm_bytes_per_launch" not in v:
        continue
    print(f"{k:48s} hbm {v['hbm_bytes_per_launch'] / 1e6:9.1f} MB  mfma_busy {v.get('mfma_busy', float('nan')):.3f}  clock {v.get('clock_mhz', float('nan')):7.0f} MHz")
