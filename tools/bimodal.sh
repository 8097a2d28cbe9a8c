#!/bin/bash
# several processes of the same build under rocprofv3 --kernel-trace --stats: per-kernel average durations, to see which kernel differs between
# the "fast" and the "slow" processes (steady iteration 80.6 vs 83.3 us: tools/ablong.py)
export TMPDIR=/tmp ITERS=1500 WINDOW=${WINDOW:-1000} CFG=m
for i in 1 2 3 4 5 6 7 8; do
  rm -rf gpurun_out/bim; rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bim -- python3 tools/prof_cfg.py > /dev/null 2>&1
  python3 - <<PY
import csv, glob
f = glob.glob("gpurun_out/bim/*/*kernel_stats.csv")[0]
d = {r["Name"].split("(")[0].replace("void bnmf::","").replace("bnmf::","")[:14]: float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(f))}
rows = list(csv.DictReader(open(glob.glob("gpurun_out/bim/*/*kernel_trace.csv")[0])))
zs = sorted([r for r in rows if "k_draw" in r["Kernel_Name"]], key=lambda r: int(r["Start_Timestamp"]))
per = (int(zs[-1]["Start_Timestamp"]) - int(zs[len(zs)//2]["Start_Timestamp"])) / (len(zs) - 1 - len(zs)//2) / 1e3
def nm(r): return r["Kernel_Name"].split("(")[0].replace("void bnmf::","").replace("bnmf::","")[:13]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows)//2:]
import collections
off = collections.defaultdict(list)
last = {}
for r in rows:
    k, s0, e0 = nm(r), int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if k.startswith("k_zalloc_sort"):
        if "k_draw_end" in last: off["draw_end->zalloc_start"].append(s0 - last["k_draw_end"])
        last["z_start"], last["z_end"] = s0, e0
    elif k == "k_draw":
        if "z_end" in last: off["zalloc_end->draw_start"].append(s0 - last["z_end"])
        last["k_draw_end"] = e0
    elif k in ("k_side", "k_side_lp") and "k_draw_end" in last:
        off[k + "_start-draw_end"].append(s0 - last["k_draw_end"])
        if "z_end" in last: pass
print("run $i: iteration %.2f us |" % per, " ".join("%s %.2f" % (k, v) for k, v in sorted(d.items()) if k.startswith(("k_draw", "k_zalloc_sort", "k_side", "k_reduce"))),
      "|", " ".join("%s %.1f" % (k, sum(v) / len(v) / 1e3) for k, v in sorted(off.items())))
PY
done
rm -rf gpurun_out/bim
