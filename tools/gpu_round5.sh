#!/bin/bash
# One GPU-box pass of round 5: the evidence committed under profiles/r05_*.  Usage: tools/gpu_round5.sh (from the repository root)
export TMPDIR=/tmp
O=gpurun_out/r05; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_steps20.json 2> $O/bench_steps20.err; echo "bench20 rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 bench.py --no-cpu-baseline --no-secondary --reps 2 > $O/bench_prof.json 2> $O/bench_prof.err; echo "prof rc=$?"
cp $(ls $O/prof/*/*kernel_stats.csv | head -1) $O/bench_kernel_stats.csv; rm -rf $O/prof
bash tools/pmc_r5.sh $O/pmc > $O/pmc.txt 2>&1; echo "pmc rc=$?"; tail -1 $O/pmc.txt; rm -rf $O/pmc/*/
for c in 3 3c 4; do CFG=$c ITERS=30 bash tools/prof_cfg.sh $c r05/cfg$c 14 > $O/cfg$c.txt 2>&1; cp $(ls $O/cfg$c/*/*kernel_stats.csv | head -1) $O/cfg${c}_kernel_stats.csv; rm -rf $O/cfg$c; done
G5=50000 ITERS=8 WINDOW=2 CFG=5 bash tools/prof_cfg.sh 5 r05/cfg5 10 > $O/cfg5.txt 2>&1; cp $(ls $O/cfg5/*/*kernel_stats.csv | head -1) $O/cfg5_kernel_stats.csv; rm -rf $O/cfg5
CFG=m ITERS=800 WINDOW=1000 bash tools/prof_cfg.sh m r05/cfgm 18 > $O/timeline_steady_state.txt 2>&1; rm -rf $O/cfgm
CFG=4 ITERS=40 bash tools/pmc_cfg.sh $O/pmc4 > $O/pmc4.txt 2>&1; echo "pmc4 rc=$?"; rm -rf $O/pmc4/*/
G5=50000 ITERS=6 WINDOW=2 CFG=5 bash tools/pmc_cfg.sh $O/pmc5 > $O/pmc5.txt 2>&1; echo "pmc5 rc=$?"; rm -rf $O/pmc5/*/
python tools/e2e_bayesnmf.py > $O/e2e_bayesNMF.json 2> $O/e2e.err; echo "e2e rc=$?"
BNMF_RANKDBG=1 python tools/rankdbg.py > $O/rank_stamps.txt 2>&1
python tools/runoverhead.py > $O/run_overhead.txt 2>&1
ls $O | head -40
