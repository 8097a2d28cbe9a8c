#!/bin/bash
# A/B timing of several builds on one box (tools/ab20.py): usage tools/ab3.sh lib1 lib2 ...
export TMPDIR=/tmp
timeout -k 5 400 python tools/ab20.py "$@" 2>&1 | tail -12
