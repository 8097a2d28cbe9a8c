#!/bin/bash
# usage: tools/pmc3.sh <outdir> [env assignments...] ; SQ counters of k_zalloc* only (separate --pmc passes)
export TMPDIR=/tmp
OUT=$1; shift; mkdir -p $OUT
for a in "$@"; do export "$a"; done
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS --output-format csv -d $OUT/p1 -- python3 tools/prof.py > $OUT/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $OUT/p2 -- python3 tools/prof.py > $OUT/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR --output-format csv -d $OUT/p3 -- python3 tools/prof.py > $OUT/p3.log 2>&1
python3 - <<PY
import csv, glob, collections
for p in ("p1","p2","p3"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % p):
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            if "zalloc" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
        for c in acc: print(p, c, "%.4g" % (acc[c] / cnt[c]))
PY
