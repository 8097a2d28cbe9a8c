#!/bin/bash
# Two PROCESSES on one GPU at once, each running tools/concurrent_check.py with full-size chains (a rank-learning chain in both): every chain
# the bits of its solo run, no time-out.  usage: tools/two_process_check.sh [repetitions per process, default 3]   (BNMF_DEVLOCK=0: without the lock file)
R=${1:-3}
CONC_BIG=1 python tools/concurrent_check.py $R > /tmp/tp_a.log 2>&1 &
A=$!
CONC_BIG=1 python tools/concurrent_check.py $R > /tmp/tp_b.log 2>&1 &
B=$!
wait $A; ra=$?
wait $B; rb=$?
tail -2 /tmp/tp_a.log; tail -2 /tmp/tp_b.log
echo "exit codes $ra $rb"
[ $ra -eq 0 ] && [ $rb -eq 0 ]
