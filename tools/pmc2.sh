#!/bin/bash
# HBM traffic of the kernels (separate --pmc passes, as MI355X_MICROARCH.md prescribes): FETCH_SIZE and WRITE_SIZE
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
ITERS=30 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/f -- python3 tools/prof.py > $OUT/f.log 2>&1
ITERS=30 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/w -- python3 tools/prof.py > $OUT/w.log 2>&1
ITERS=30 SAVE_Z=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fz -- python3 tools/prof.py > $OUT/fz.log 2>&1
ITERS=30 SAVE_Z=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/wz -- python3 tools/prof.py > $OUT/wz.log 2>&1
ITERS=30 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/s -- python3 tools/prof.py > $OUT/s.log 2>&1
ITERS=30 SAVE_Z=1 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/sz -- python3 tools/prof.py > $OUT/sz.log 2>&1
python3 - <<PY
import csv, glob, collections, json
res = {}
for tag in ("f", "w", "fz", "wz"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % tag):
        acc = collections.defaultdict(float); cnt = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void bnmf::", "").replace("bnmf::", "")
            acc[(k, r["Counter_Name"])] += float(r["Counter_Value"]); cnt[(k, r["Counter_Name"])] += 1
        for (k, c), v in acc.items():
            res.setdefault(tag, {})[k + ":" + c] = {"sum": v, "launches": cnt[(k, c)], "per_launch": v / cnt[(k, c)]}
json.dump(res, open("$OUT/pmc_traffic.json", "w"), indent=1)
for tag, d in res.items():
    for k, v in d.items():
        if "zalloc" in k or "side" in k or "edraw" in k: print(tag, k, "per launch %.1f" % v["per_launch"])
PY
for t in s sz; do echo "== kernel stats ($t)"; cat $(ls $OUT/$t/*/*kernel_stats.csv | head -1) | cut -c1-160 | head -8; done
