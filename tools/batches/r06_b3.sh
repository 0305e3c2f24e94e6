#!/bin/bash
# round-6 GPU batch 3 (run through gpurun from the repo root): new parity tests, attention policy, kernel regression after the shift change, GEMM solution search + A/B
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
export VGPA_DEPTH_LAYERS=2
timeout 600 python -m pytest tests/test_gpu_depth_wan.py -x -q 2>&1 | tail -8 > $O/depthwan_L2.log
unset VGPA_DEPTH_LAYERS
timeout 900 python -m pytest tests/test_gpu_attn_policy.py -q -s 2>&1 | tail -25 > $O/attn_policy.log
timeout 900 python tools/attn_robust.py > $O/attn_robust.log 2>&1
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_wan_kernels.py tests/test_gpu_model.py tests/test_gpu_cfg1.py -q -x 2>&1 | tail -15 > $O/kernels_regress.log
timeout 900 python -m pytest tests/test_gpu_loss_curve.py -x -q 2>&1 | tail -15 > $O/loss_curve.log
timeout 1500 python -m pytest tests/test_gpu_depth_wan.py -q 2>&1 | tail -25 > $O/depthwan_L30.log
timeout 900 python -m pytest tests/test_gpu_depth.py -x -q -k order_one 2>&1 | tail -12 > $O/depth_rich.log
# vendor-GEMM solution search at the cfg2 shapes, then per-shape table and the step-level A/B (default heuristic vs tuned, interleaved)
timeout 2400 python tools/gemm_tune.py tune --configs cfg2 --fresh > $O/gemm_tune.log 2>&1
timeout 900 python tools/gemm_tune.py table > $O/gemm_tune_table.txt 2>&1
for i in 1 2; do
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --no-scorer --no-tuned-gemms > $O/ab_gemm_default_$i.json 2> $O/ab_gemm_default_$i.err
  timeout 600 python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-other-configs --no-scorer > $O/ab_gemm_tuned_$i.json 2> $O/ab_gemm_tuned_$i.err
done
cp videogpa_amd/tuned/tunableop_gfx950.csv $O/ 2>/dev/null
tail -3 $O/depthwan_L2.log $O/attn_policy.log $O/kernels_regress.log $O/loss_curve.log $O/depthwan_L30.log $O/depth_rich.log $O/gemm_tune.log
