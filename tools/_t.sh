timeout -k 10 1000 python -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests.txt 2>&1; tail -3 gpurun_out/gpu_tests.txt
timeout -k 10 200 python bench.py --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], [round(x) for x in d['rep_values']], d['kernel_ms'])"
for i in 1 2; do
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['value'], [round(x) for x in d['rep_values']])"
done
