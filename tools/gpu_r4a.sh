#!/bin/bash
# round 4, first pass: the allocation-kernel tests, the whole GPU suite, config 5 at full size.  Usage: tools/gpu_r4a.sh
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest_a.txt 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_a.txt
CFG=5 G5=50000 ITERS=3 ORDER=1,0 timeout -k 10 600 python tools/zstep_time.py > $O/cfg5_full.txt 2>&1; echo "cfg5 rc=$?"; cat $O/cfg5_full.txt
