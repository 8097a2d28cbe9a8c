#!/bin/bash
# One GPU-box pass: GPU tests, bench line, rocprofv3 kernel stats of the bench command.  Usage: tools/gpu_round.sh TAG
TAG=${1:-r02a}
export TMPDIR=/tmp
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q > gpurun_out/${TAG}_pytest.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/${TAG}_pytest.log
tail -3 gpurun_out/${TAG}_pytest.log
python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc=$?"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_prof -- python3 bench.py --no-cpu-baseline --no-secondary --reps 2 > gpurun_out/${TAG}_bench_prof.json 2> gpurun_out/${TAG}_bench_prof.err; echo "prof rc=$?"
head -c 1500 gpurun_out/${TAG}_bench.json
