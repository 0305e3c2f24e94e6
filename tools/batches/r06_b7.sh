#!/bin/bash
# round-6 GPU batch 7: the batch-5 failures after the lse-contract tolerance (tests/attn_tol.py); where the trained_like bench arm loses its loss to NaN
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_attn_policy.py tests/test_gpu_kernels.py tests/test_gpu_wan_kernels.py -m gpu -q 2>&1 | cut -c1-600 | tail -60 > $O/r06_b7_tests.log
timeout 900 python tools/trained_like_diag.py --qk-gain 2.5 --json $O/trained_like_diag_g2.5.json > $O/trained_like_diag_g2.5.txt 2>&1
timeout 900 python tools/trained_like_diag.py --qk-gain 1.0 --steps 1 --json $O/trained_like_diag_g1.json > $O/trained_like_diag_g1.txt 2>&1
tail -n 25 $O/r06_b7_tests.log; cat $O/trained_like_diag_g2.5.txt $O/trained_like_diag_g1.txt | cut -c1-400
