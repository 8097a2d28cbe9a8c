#!/bin/bash
# round 4: section shares of k_zalloc_step (ZPPROF build) and FETCH / WRITE counters of configs 4 and 5.  Usage: tools/gpu_r4d.sh
export TMPDIR=/tmp
O=gpurun_out/r4d; mkdir -p $O
CFG=4 timeout -k 10 200 python tools/zpprof.py > $O/zpprof_cfg4.txt 2>&1; cat $O/zpprof_cfg4.txt
CFG=5 G5=12800 timeout -k 10 300 python tools/zpprof.py > $O/zpprof_cfg5.txt 2>&1; cat $O/zpprof_cfg5.txt
CFG=4 ITERS=40 timeout -k 10 400 bash tools/pmc_cfg.sh $O/pmc4 > $O/pmc4.txt 2>&1; tail -12 $O/pmc4.txt
CFG=5 G5=50000 ITERS=6 WINDOW=2 timeout -k 10 600 bash tools/pmc_cfg.sh $O/pmc5 > $O/pmc5.txt 2>&1; tail -12 $O/pmc5.txt
