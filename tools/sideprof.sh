#!/bin/bash
# usage: tools/sideprof.sh <outdir> ; serialised per-kernel times + VALU counters of the hyper sweep (k_side) and the allocation kernel
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
python3 - > $OUT/serial.log 2>&1 <<PY
import sys; sys.path.insert(0, ".")
from bayesnmf_amd import Engine
from bayesnmf_amd.setup import synth_counts, apply_hyperprior_params
M, _, _ = synth_counts(96, 10000, 8, 20250218)
e = Engine(M, 20, prior="gamma", seed=1); apply_hyperprior_params(e, "gamma", M, 20); e.init(); e.run(50, metrics=False)
for r in range(3):
    p = e.profile(40)
    print({k: round(v * 1e3, 1) for k, v in p.items() if v > 0})
PY
cat $OUT/serial.log
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_SALU"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/prof.py > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(p + "*/*counter_collection.csv"):
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][:40] + "/g" + r.get("Grid_Size", "")
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for c in sorted(acc): print(p.split("/")[-2], c[0], c[1], "%.4g" % (acc[c] / cnt[c]), cnt[c])
PY
