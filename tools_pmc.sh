#!/bin/bash
# usage: tools_pmc.sh <outdir> ; collects SQ counters for the kernels of tools_prof.py (separate --pmc passes)
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -- python3 tools_prof.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/p2 -- python3 tools_prof.py > $OUT/p2.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % p):
        acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0][-24:]
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
        for k, d in acc.items():
            print(p, k, {c: "%.3g" % v for c, v in d.items()})
PY
