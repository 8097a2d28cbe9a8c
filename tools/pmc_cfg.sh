#!/bin/bash
# usage: CFG=4|5 [G5=50000 ITERS=6 WINDOW=2] tools/pmc_cfg.sh <outdir> : FETCH_SIZE / WRITE_SIZE per launch of every kernel of one
# BASELINE.json configuration (tools/prof_cfg.py), one rocprofv3 --pmc pass each (serial-safe mode of the library) -> <outdir>/pmc_traffic.json
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/$c -- python3 tools/prof_cfg.py > $OUT/$c.log 2>&1; echo "$c rc=$?"
done
python3 - <<PY
import csv, glob, collections, json, os
res = collections.defaultdict(dict)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = collections.defaultdict(list)
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv" % c):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != c: continue
            k = r["Kernel_Name"].split("(")[0].replace("void bnmf::", "").replace("bnmf::", "")
            per[k].append(float(r["Counter_Value"]))
    for k, v in per.items():
        v2 = v[len(v) // 4:]
        res[k][c + "_KiB"] = sum(v2) / len(v2); res[k]["launches"] = len(v)
out = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, per launch (first quarter of the launches dropped), KiB; "
               "hbm_bytes_per_launch = (2 * FETCH + WRITE) * 1024 (gfx950: FETCH_SIZE reports half of a coalesced read stream). CFG=%s G5=%s" % (os.environ.get("CFG"), os.environ.get("G5", "-"))}
for k, v in sorted(res.items()):
    if "FETCH_SIZE_KiB" in v and "WRITE_SIZE_KiB" in v and not k.startswith("__"):
        v["hbm_bytes_per_launch"] = (2 * v["FETCH_SIZE_KiB"] + v["WRITE_SIZE_KiB"]) * 1024
        out[k] = v
json.dump(out, open("$OUT/pmc_traffic.json", "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict): print(k[:50].ljust(50), v["launches"], round(v["hbm_bytes_per_launch"] / 1e6, 2), "MB")
PY
