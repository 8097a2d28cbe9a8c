#!/bin/bash
# round 4: rank-sweep tests + config 4 timing + stamps.  Usage: tools/gpu_r4b.sh
export TMPDIR=/tmp
O=gpurun_out/r4; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "rank or config4 or chains or bic or learned" > $O/pytest_b.txt 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_b.txt
ORDER=1,1 timeout -k 10 300 python tools/zstep_time.py > $O/cfg4_b.txt 2>&1; cat $O/cfg4_b.txt
BNMF_RANKDBG=1 timeout -k 10 300 python tools/rankdbg.py > $O/rankdbg_b.txt 2>&1; tail -8 $O/rankdbg_b.txt
