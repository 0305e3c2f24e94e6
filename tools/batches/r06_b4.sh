#!/bin/bash
# round-6 GPU batch 4: sampled-max shift of the forward (policy test, robustness table, kernel regression), loss curve, Wan depth with the noise-floor-relative bounds,
# row sums on the matrix pipe (W1_KNOBS=mfsum=1) A/B with joules, GEMM cross-entry table + cfg5 solution search, scorer probe / bench, forward PMC budget, step A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_attn_policy.py -q -s 2>&1 | tail -25 > $O/attn_policy.log
timeout 900 python tools/attn_robust.py > $O/attn_robust.log 2>&1
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -x -k "attention or attn" 2>&1 | tail -6 > $O/kernels_regress.log
timeout 900 python -m pytest tests/test_gpu_cfg1.py -q -x 2>&1 | tail -6 >> $O/kernels_regress.log
timeout 900 python -m pytest tests/test_gpu_loss_curve.py -x -q 2>&1 | tail -15 > $O/loss_curve.log
timeout 1500 python -m pytest tests/test_gpu_depth_wan.py -q 2>&1 | tail -25 > $O/depthwan_L30.log
timeout 900 bash tools/fwd_knob_ab.sh mfsum=1 > $O/fwd_knob_mfsum.log 2>&1
timeout 600 python tools/gemm_tune.py table --file tools/batches/tunableop_exp.csv --json $O/gemm_tune_table_exp.json > $O/gemm_tune_table_exp.txt 2>&1
timeout 2400 python tools/gemm_tune.py tune --configs cfg5,cfg5bf16,cfg4 --fresh --out $O/tunableop_cfg5.csv > $O/gemm_tune_cfg5.log 2>&1
timeout 900 python tools/gemm_tune.py table --file $O/tunableop_cfg5.csv --json $O/gemm_tune_table_cfg5.json > $O/gemm_tune_table_cfg5.txt 2>&1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/zbuf_atomic_probe.hip -o /tmp/zbuf_atomic_probe > $O/zbuf_probe.log 2>&1 && timeout 300 /tmp/zbuf_atomic_probe >> $O/zbuf_probe.log 2>&1
timeout 600 python tools/scorer_bench.py > $O/scorer_bench.log 2>&1
timeout 1500 bash tools/pmc_fwd_budget.sh > $O/pmc_fwd_budget.log 2>&1
for i in 1 2; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --no-scorer --no-tuned-gemms > $O/ab2_gemm_default_$i.json 2> $O/ab2_gemm_default_$i.err
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --no-scorer > $O/ab2_gemm_tuned_$i.json 2> $O/ab2_gemm_tuned_$i.err
done
for f in attn_policy kernels_regress loss_curve depthwan_L30 fwd_knob_mfsum zbuf_probe scorer_bench; do echo "== $f"; tail -n 4 $O/$f.log | cut -c1-300; done
