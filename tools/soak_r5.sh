#!/bin/bash
# Round-5 soak on the final kernels: random parity cases against the oracle, chains beside each other, mixed entry points, two processes.
export TMPDIR=/tmp
O=gpurun_out/r05soak; mkdir -p $O
{
echo "soak of round 5 (one MI355X box): tools/fuzz_parity.py seeds 101-104 (300 cases each, G up to 500), seeds 105-106 rank learning only, tools/concurrent_check.py 40, CONC_BIG=1 tools/concurrent_check.py 8, tools/api_mix_check.py 20, tools/two_process_check.sh 3, tools/soak.py"
for s in 101 102 103 104; do FUZZ_SEED=$s FUZZ_N=300 FUZZ_GMAX=500 timeout -k 5 240 python tools/fuzz_parity.py 2>&1 | tail -1; done
for s in 105 106; do FUZZ_LR_ONLY=1 FUZZ_SEED=$s FUZZ_N=200 FUZZ_GMAX=300 timeout -k 5 240 python tools/fuzz_parity.py 2>&1 | tail -1; done
timeout -k 5 300 python tools/concurrent_check.py 40 2>&1 | tail -2
CONC_BIG=1 timeout -k 5 300 python tools/concurrent_check.py 8 2>&1 | tail -2
timeout -k 5 300 python tools/api_mix_check.py 20 2>&1 | tail -2
timeout -k 5 400 bash tools/two_process_check.sh 3 2>&1 | tail -5
timeout -k 5 200 python tools/soak.py 2>&1 | tail -4
} > $O/soak.txt 2>&1
cat $O/soak.txt
