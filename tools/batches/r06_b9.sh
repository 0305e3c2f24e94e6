#!/bin/bash
# round-6 GPU batch 9: cfg5 on trained-like QK-norm gains (e4m3 and bf16 attention), then the whole GPU suite on the fixed tree
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout 600 python bench.py --config cfg5 --steps 2 --warmup 1 --weights trained_like --qk-gain 2.5 > $O/r06_bench_cfg5_trained_like.json 2> $O/r06_bench_cfg5_trained_like.err
timeout 600 python bench.py --config cfg5 --steps 2 --warmup 1 --weights trained_like --qk-gain 2.5 --no-fp8-attn > $O/r06_bench_cfg5_trained_like_bf16attn.json 2> $O/r06_bench_cfg5_trained_like_bf16attn.err
timeout 600 python bench.py --config cfg5 --steps 2 --warmup 1 --weights trained_like --qk-gain 4.0 > $O/r06_bench_cfg5_trained_like_g4.json 2> $O/r06_bench_cfg5_trained_like_g4.err
for f in $O/r06_bench_cfg5_trained_like*.json; do python - $f <<'PYEOF'
import json, sys
try:
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    k = d["kernels"]
    print(sys.argv[1], "ms", round(d["ms_per_step"], 1), "loss", d["loss"], {n: round(v["avg_ms"], 3) for n, v in k.items() if "attn128" in n})
except Exception as e:
    print(sys.argv[1], "FAILED", e)
PYEOF
done
timeout 1800 python -m pytest tests -m gpu -q 2>&1 | cut -c1-400 | tail -40 > $O/r06_gputest2.log
tail -n 12 $O/r06_gputest2.log
