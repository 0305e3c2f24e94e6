#!/bin/bash
# PMC passes over the PRODUCT forward kernel at the headline shape for tools/fwd_budget.py (run on the GPU box):  tools/pmc_fwd_budget.sh
# Separate passes per counter set, --pmc only (no trace domains next to it).  -> gpurun_out/pmc_fwd_budget.json
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY" \
           "SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_VALU_MFMA_COEXEC_CYCLES SQ_WAVES" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_fwdbudget_$i --output-format csv -- python $R/tools/attn_bench.py --which fwd --iters 2 > $R/gpurun_out/pmc_fwdbudget_$i.log 2>&1
done
python - "$R" <<'PYEOF'
import csv, glob, json, sys
from collections import defaultdict
R = sys.argv[1]
acc = defaultdict(list)
for f in glob.glob(f"{R}/gpurun_out/pmc_fwdbudget_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "attn_fwd_w1_kernel<false>" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
c = {k: sum(v) / len(v) for k, v in sorted(acc.items())}
B, H, S = 2, 48, 17776
tiles = (S + 63) // 64
waves = B * H * ((S + 255) // 256) * 4                  # main launch + tail chunks cover every (strip, wave) once
half_steps = waves * (tiles + 1) * 2                    # + the drain step
d = {}
if "GRBM_GUI_ACTIVE" in c and "SQ_VALU_MFMA_BUSY_CYCLES" in c:
    cyc = c["GRBM_GUI_ACTIVE"] / 8.0
    d["mfma_busy (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs))"] = round(c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), 4)
if "SQ_ACTIVE_INST_VALU" in c and "SQ_WAVE_CYCLES" in c:
    d["VALU-active fraction of wave cycles (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES; both are per-SIMD-quad sampled: x4 each)"] = round(c["SQ_ACTIVE_INST_VALU"] / c["SQ_WAVE_CYCLES"], 4)
if "SQ_INST_CYCLES_VMEM" in c and "SQ_WAVE_CYCLES" in c:
    d["VMEM issue cycles / wave cycles"] = round(c["SQ_INST_CYCLES_VMEM"] / c["SQ_WAVE_CYCLES"], 5)
json.dump({"kernel": "attn_fwd_w1_kernel<false>", "shape": [B, H, S, 64], "half_steps_per_launch": half_steps, "counters": c, "derived": d},
          open(f"{R}/gpurun_out/pmc_fwd_budget.json", "w"), indent=1)
print(json.dumps(d, indent=1))
PYEOF
python $R/tools/fwd_budget.py --md $R/gpurun_out/fwd_budget.md
