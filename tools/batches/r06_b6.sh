#!/bin/bash
# round-6 GPU batch 6: full output of the batch-5 failures (forward lse under mfsum, policy timing), the hipBLASLt GELU-epilogue probes
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_attn_policy.py "tests/test_gpu_kernels.py::test_attention_extreme_outliers_strip_redo" \
    tests/test_gpu_wan_kernels.py::test_attention128_e4m3_flagged_strips_are_redone_on_the_dequantised_operands -m gpu -q -s 2>&1 | cut -c1-1500 > $O/r06_b6_fail.log
timeout 300 python tools/gelu_probe.py > $O/gelu_probe.txt 2>&1
hipcc -O2 tools/hipblaslt_epilogue_time.cpp -lhipblaslt -o /tmp/hipblaslt_epilogue_time > $O/hipblaslt_epilogue_time.txt 2>&1
timeout 600 /tmp/hipblaslt_epilogue_time >> $O/hipblaslt_epilogue_time.txt 2>&1
tail -n 30 $O/r06_b6_fail.log; cat $O/gelu_probe.txt $O/hipblaslt_epilogue_time.txt
