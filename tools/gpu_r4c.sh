#!/bin/bash
# round 4, session 2 first pass: the GPU suite, the bench line, kernel stats of the bench command, configs 4 / 5 timing.  Usage: tools/gpu_r4c.sh
export TMPDIR=/tmp
O=gpurun_out/r4c; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest.txt
timeout -k 10 500 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; cat $O/bench.json
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench20 rc=$?"; cat $O/bench_steps20.json
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-secondary --reps 2 > $O/bench_prof.json 2> $O/bench_prof.err; echo "prof rc=$?"
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; head -12 $O/bench_kernel_stats.csv
