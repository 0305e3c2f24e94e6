python -m pytest tests/test_gpu_wan_model.py tests/test_gpu_vggt.py tests/test_gpu_preprocess.py -q 2>&1 | tail -8
for i in 1 2; do for pd in 0 1; do
VGPA_PRECISE_DELTA=$pd python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('precise_delta=$pd', round(d['ms_per_step'],1), round(d['max_memory_gb'],1), {k:(round(v['avg_ms'],3)) for k,v in d['kernels'].items() if 'attn' in k})"
done; done
