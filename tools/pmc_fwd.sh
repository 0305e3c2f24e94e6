#!/bin/bash
# PMC passes over tools/attn_bench.py for one library variant:  tools/pmc_fwd.sh VARIANT WHICH   (run on the GPU box)
v=$1; which=${2:-fwd}
R=${GRAFT_REPO_ROOT:-/root/repo}
cp $R/var/lib_$v.so $R/videogpa_amd/csrc/libvgpa_hip.so
cd /tmp && export TMPDIR=/tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_VMEM_RD" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" \
           "GRBM_GUI_ACTIVE SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVES SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_ANY SQ_INSTS_BRANCH"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set -d $R/gpurun_out/pmc_${v}_$i --output-format csv -- python $R/tools/attn_bench.py --which $which --iters 2 > $R/gpurun_out/pmc_${v}_$i.log 2>&1
done
python $R/tools/pmc_summary.py --all $(find $R/gpurun_out/pmc_${v}_* -name "*counter_collection.csv") | grep -A40 "attn_"
