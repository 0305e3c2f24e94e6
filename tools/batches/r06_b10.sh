#!/bin/bash
# round-6 GPU batch 10: sampled shift in the e4m3 forward -- kernel tests, Wan parity tests, cfg5 on trained-like gains and on the bench init
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout 900 python -m pytest tests/test_gpu_wan_kernels.py tests/test_gpu_wan_cfg1.py tests/test_gpu_wan.py -m gpu -q -s 2>&1 | cut -c1-500 | tail -50 > $O/r06_b10_tests.log
timeout 600 python bench.py --config cfg5 --steps 2 --warmup 1 --weights trained_like --qk-gain 2.5 > $O/r06_bench_cfg5_trained_like_after.json 2> $O/r06_bench_cfg5_trained_like_after.err
timeout 600 python bench.py --config cfg5 --steps 3 --warmup 1 > $O/r06_bench_cfg5_after.json 2> $O/r06_bench_cfg5_after.err
for f in $O/r06_bench_cfg5_trained_like_after.json $O/r06_bench_cfg5_after.json; do python - $f <<'PYEOF'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernels"]
    print(sys.argv[1], "ms", round(d["ms_per_step"], 1), "loss", d["loss"], {n: round(v["avg_ms"], 3) for n, v in k.items() if "attn128" in n})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PYEOF
done
tail -n 30 $O/r06_b10_tests.log
