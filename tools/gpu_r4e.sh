#!/bin/bash
# final pass of round 4, part 1: the bench lines and the kernel trace of the bench command (profiles/r04_bench*.json, r04_bench_kernel_stats.csv)
export TMPDIR=/tmp
O=gpurun_out/r04; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench20 rc=$?"
rm -rf $O/prof
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-secondary --reps 2 > $O/bench_prof.json 2> $O/bench_prof.err; echo "prof rc=$?"
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv
python tools/e2e_bayesnmf.py > $O/e2e_bayesNMF.json 2> $O/e2e.err; echo "e2e rc=$?"
python tools/runoverhead.py > $O/run_overhead.txt 2>&1
tail -c 600 $O/bench_steps20.json
