#!/bin/bash
# usage: CFG=4|5 [G5=12800 ITERS=6 WINDOW=2] tools/pmc_sq_cfg.sh <outdir> : SQ counters per launch of every kernel of one BASELINE.json
# configuration (tools/prof_cfg.py), one rocprofv3 --pmc pass per counter set -> <outdir>/pmc_sq.json
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/s$i -- python3 tools/prof_cfg.py > $OUT/s$i.log 2>&1; echo "pass $i rc=$?"
done
python3 - <<PY
import csv, glob, collections, json, os
res = collections.OrderedDict()
for f in sorted(glob.glob("$OUT/s*/*/*counter_collection.csv")):
    per = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void bnmf::", "").replace("bnmf::", "")
        per[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
    for (k, c), v in sorted(per.items()):
        v2 = v[len(v) // 4:]
        res.setdefault(k, {})[c] = sum(v2) / len(v2)
json.dump({"note": "rocprofv3 --pmc SQ counters per launch (first quarter of the launches dropped). CFG=%s G5=%s" % (os.environ.get("CFG"), os.environ.get("G5", "-")), "kernels": res}, open("$OUT/pmc_sq.json", "w"), indent=1)
for k, d in res.items():
    if "zalloc" in k or "rank" in k:
        print(k[:60])
        for c, v in d.items(): print("   %-24s %.5g" % (c, v))
        if "SQ_WAVE_CYCLES" in d and "SQ_WAVES" in d and d["SQ_WAVES"]:
            simd_cyc = d["SQ_WAVE_CYCLES"] / d["SQ_WAVES"] * 1024.0
            print("   VALU busy frac (ACTIVE_INST_VALU / (WAVE_CYCLES / WAVES x 1024 SIMDs)) %.3f" % (d.get("SQ_ACTIVE_INST_VALU", 0) / simd_cyc))
        if "SQ_LDS_IDX_ACTIVE" in d: print("   LDS bank conflict share of LDS-active %.3f" % (d["SQ_LDS_BANK_CONFLICT"] / max(d["SQ_LDS_IDX_ACTIVE"], 1)))
PY
