#!/bin/bash
# usage: tools/pmc4.sh <outdir> [env assignments...] ; latency / FIFO counters of k_zalloc* (separate --pmc passes)
export TMPDIR=/tmp
OUT=$1; shift; mkdir -p $OUT
for a in "$@"; do export "$a"; done
i=0
for set in "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_LDS_CMD_FIFO_FULL SQ_LDS_DATA_FIFO_FULL" "SQ_INSTS_LDS_ATOMIC SQ_INSTS_LDS_LOAD SQ_INSTS_LDS_STORE SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_CVT SQ_INSTS_VALU_TRANS_F64"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/prof.py > $OUT/p$i.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for p in sorted(glob.glob("$OUT/p*/")):
    for f in glob.glob(p + "*/*counter_collection.csv"):
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "zalloc" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        for c in acc: print(p.split("/")[-2], c, "%.4g" % (acc[c] / cnt[c]))
PY
