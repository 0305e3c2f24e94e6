#!/bin/bash
# round-6 GPU batch 8: the overflow-window fix of the forward (W1_L_MAX + accumulator check): regression tests, the trained_like diag and bench arm again
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_attn_policy.py tests/test_gpu_kernels.py tests/test_gpu_wan_kernels.py -m gpu -q 2>&1 | cut -c1-600 | tail -60 > $O/r06_b8_tests.log
timeout 900 python tools/trained_like_diag.py --qk-gain 2.5 --json $O/trained_like_diag_g2.5.json > $O/trained_like_diag_g2.5.txt 2>&1
timeout 900 python tools/trained_like_diag.py --qk-gain 3.5 --steps 1 --json $O/trained_like_diag_g3.5.json > $O/trained_like_diag_g3.5.txt 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-scorer --weights trained_like --qk-gain 2.5 > $O/r06_bench_trained_like.json 2> $O/r06_bench_trained_like.err
timeout 600 python tools/attn_bench.py --which fwd --iters 20 > $O/attn_bench_fwd_after_fix.txt 2>&1
tail -n 25 $O/r06_b8_tests.log; cat $O/trained_like_diag_g2.5.txt $O/trained_like_diag_g3.5.txt $O/attn_bench_fwd_after_fix.txt | cut -c1-400
