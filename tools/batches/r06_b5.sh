#!/bin/bash
# round-6 GPU batch 5: the whole GPU suite on the mfsum forward + sampled shift (hd64 and hd128) + tuned GEMMs, smoke, fp8 GEMM table, the round's profile evidence, default bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | tail -40 > $O/r06_gputest.log
timeout 300 python __graft_entry__.py smoke > $O/r06_smoke.log 2>&1
timeout 600 python tools/attn_robust.py > $O/attn_robust.log 2>&1
timeout 600 python tools/gemm_tune.py table --file tools/batches/tunableop_scaled_fp8.csv --json $O/gemm_tune_table_fp8.json > $O/gemm_tune_table_fp8.txt 2>&1
timeout 1500 bash tools/pmc_fwd_budget.sh > $O/pmc_fwd_budget.log 2>&1
timeout 3000 bash tools/profile_round.sh r06a > $O/profile_round_r06a.log 2>&1
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-scorer --weights trained_like --qk-gain 2.5 > $O/r06_bench_trained_like.json 2> $O/r06_bench_trained_like.err
timeout 1200 python bench.py > $O/r06a_bench.json 2> $O/r06a_bench.err
tail -n 6 $O/r06_gputest.log $O/r06_smoke.log $O/profile_round_r06a.log | cut -c1-300
