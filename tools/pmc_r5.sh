#!/bin/bash
# usage: tools/pmc_r5.sh <outdir> : counters of the steady-state kernels, one rocprofv3 --pmc pass per counter set (the library's
# probe finds the dispatches serialised and runs in its serial-safe mode: stream waits instead of in-kernel polling, same kernels)
export TMPDIR=/tmp
OUT=$1; mkdir -p $OUT
i=0
for set in "FETCH_SIZE" "WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_SMEM" \
           "SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1))
  ITERS=40 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -- python3 tools/prof.py > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? ($set)"; tail -2 $OUT/p$i.log
done
# full mode: records only (what the library does), and with Z expanded every iteration (BNMF_ZEAGER=1)
ITERS=40 SAVE_Z=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fr -- python3 tools/prof.py > $OUT/fr.log 2>&1; echo "fr rc=$?"
ITERS=40 SAVE_Z=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/wr -- python3 tools/prof.py > $OUT/wr.log 2>&1; echo "wr rc=$?"
export BNMF_ZEAGER=1
ITERS=40 SAVE_Z=1 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/fz -- python3 tools/prof.py > $OUT/fz.log 2>&1; echo "fz rc=$?"
ITERS=40 SAVE_Z=1 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/wz -- python3 tools/prof.py > $OUT/wz.log 2>&1; echo "wz rc=$?"
unset BNMF_ZEAGER
python3 - <<PY
import csv, glob, collections, json
res = collections.OrderedDict()
for p in sorted(glob.glob("$OUT/*/")):
    tag = p.rstrip("/").split("/")[-1]
    for f in glob.glob(p + "*/*counter_collection.csv"):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void bnmf::", "").replace("bnmf::", "")
            per[(k, r["Counter_Name"])].append(float(r["Counter_Value"]))
        for (k, c), v in sorted(per.items()):
            v2 = v[len(v) // 4:]                      # steady state: drop the first quarter of the launches
            res.setdefault(tag, {})[k + ":" + c] = {"launches": len(v), "mean": sum(v2) / len(v2), "min": min(v2), "max": max(v2)}
json.dump(res, open("$OUT/pmc_counters.json", "w"), indent=1)
# HBM traffic per launch in the form bench.py reads (gfx950: FETCH_SIZE reports half of a coalesced read stream, MI355X_MICROARCH.md)
def kib(tag, kern, ctr):
    for k, v in res.get(tag, {}).items():
        if k.startswith(kern) and k.endswith(":" + ctr): return v["mean"]
    return None
tr = {"note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes (tools/pmc_r5.sh -> tools/prof.py, record window 1000), K=96 G=10000 N=20, "
              "per launch, steady state (first quarter of the launches dropped), KiB; hbm_bytes_per_launch = (2 * FETCH + WRITE) * 1024.  The "
              "library ran in its serial-safe mode (its probe found the dispatches serialised under counter collection): same kernels, stream waits instead of polling.  "
              "Round 5: k_zalloc_sort also writes Mhat (8KG bytes) for the per-column metric terms, which k_side_lp sums (its traffic includes them)."}
for name, kern, ft, wt in (("k_zalloc_stats", "k_zalloc_sort", "p1", "p2"), ("k_draw", "k_draw", "p1", "p2"),
                           ("k_side", "k_side:", "p1", "p2"), ("k_side_lp", "k_side_lp", "p1", "p2")):
    f, w = kib(ft, kern.rstrip(":") if kern != "k_side:" else "k_side", "FETCH_SIZE"), kib(wt, kern.rstrip(":") if kern != "k_side:" else "k_side", "WRITE_SIZE")
    if f is not None and w is not None: tr[name] = {"FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w, "hbm_bytes_per_launch": (2 * f + w) * 1024}
# full mode (save_Z): the sorted-schedule kernel writes item records (round 5: that is the sample; fr / wr passes); with BNMF_ZEAGER=1
# k_zexpand turns them into Z every iteration: the two launches together (fz / wz passes)
fr, wr = kib("fr", "k_zalloc_sort", "FETCH_SIZE"), kib("wr", "k_zalloc_sort", "WRITE_SIZE")
if fr is not None and wr is not None: tr["k_zalloc_records"] = {"FETCH_SIZE_KiB": fr, "WRITE_SIZE_KiB": wr, "hbm_bytes_per_launch": (2 * fr + wr) * 1024}
fs, ws = [kib("fz", k, "FETCH_SIZE") for k in ("k_zalloc_sort", "k_zexpand")], [kib("wz", k, "WRITE_SIZE") for k in ("k_zalloc_sort", "k_zexpand")]
if None not in fs and None not in ws:
    tr["k_zalloc_full"] = {"FETCH_SIZE_KiB": sum(fs), "WRITE_SIZE_KiB": sum(ws), "hbm_bytes_per_launch": (2 * sum(fs) + sum(ws)) * 1024,
                           "parts": {"k_zalloc_sort": {"FETCH_SIZE_KiB": fs[0], "WRITE_SIZE_KiB": ws[0]}, "k_zexpand": {"FETCH_SIZE_KiB": fs[1], "WRITE_SIZE_KiB": ws[1]}}}
json.dump(tr, open("$OUT/pmc_traffic.json", "w"), indent=1)
for tag, d in res.items():
    for k, v in d.items():
        if any(x in k for x in ("zalloc", "zexpand", "k_draw", "k_side")): print(tag, k, "mean %.5g  min %.5g  max %.5g  n %d" % (v["mean"], v["min"], v["max"], v["launches"]))
PY
grep -l -i "traceback\|error" $OUT/*.log || echo "no traceback / error in any log"
