#!/bin/bash
# One experiment pass on the GPU box: parity subset with an experiment build, then the A/B timing against the product library.
# usage: tools/exp.sh <experiment .so> [pytest -k expression]
L=${1:-tools/bin/libbnmf_fast.so}; K=${2:-"full_size_chain or merged_draw"}
export TMPDIR=/tmp
BNMF_TEST_LIB=$L timeout -k 5 300 python -m pytest tests/test_gpu_parity.py -x -q -k "$K" 2>&1 | tail -15
timeout -k 5 200 python tools/ab20.py bayesnmf_amd/libbnmf.so $L 2>&1 | tail -8
