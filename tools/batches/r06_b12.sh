#!/bin/bash
# round-6 GPU batch 12: the final tree -- default GPU suite (with durations), smoke, default bench
cd ${GRAFT_REPO_ROOT:-/root/repo}
export GRAFT_REPO_ROOT=${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out
timeout 1800 python -m pytest tests -m gpu -q --durations=8 2>&1 | cut -c1-300 | tail -25 > $O/r06_gputest4.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/r06_smoke_final.log 2>&1
timeout 1500 python bench.py > $O/r06c_bench.json 2> $O/r06c_bench.err
tail -n 16 $O/r06_gputest4.log; tail -n 3 $O/r06_smoke_final.log
