#!/bin/bash
# round-6 GPU batch 11: whole GPU suite on the tree with the e4m3 sampled shift + exact q_deq, attn_robust incl. head_dim 128 / e4m3, cfg5 profile, default bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | cut -c1-400 | tail -30 > $O/r06_gputest3.log
timeout 900 python tools/attn_robust.py --json $O/attn_trained_like_d.json > $O/attn_robust_d.log 2>&1
timeout 600 python bench.py --config cfg5 --steps 3 --warmup 1 --weights trained_like --qk-gain 2.5 > $O/r06_bench_cfg5_trained_like_after.json 2> $O/r06_bench_cfg5_trained_like_after.err
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_r06b_cfg5 --output-format csv -- python $R/bench.py --config cfg5 --steps 2 --warmup 1 > $R/gpurun_out/prof_r06b_bench_cfg5.json 2> $R/gpurun_out/prof_r06b_cfg5.log
f5=$(find $R/gpurun_out/prof_r06b_cfg5 -name "*kernel_stats.csv" | head -1)
[ -n "$f5" ] && cp $f5 $R/gpurun_out/r06b_bench_cfg5_kernel_stats.csv && python $R/profiles/summarize.py $f5 40 > $R/gpurun_out/r06b_bench_cfg5_kernel_stats_summary.txt
find $R/gpurun_out/prof_r06b_cfg5 -name "*kernel_trace.csv" -delete
cd $R
timeout 1500 python bench.py > $O/r06b_bench.json 2> $O/r06b_bench.err
tail -n 8 $O/r06_gputest3.log; tail -n 9 $O/attn_robust_d.log | cut -c1-220
