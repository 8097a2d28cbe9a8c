#!/bin/bash
# usage: tools/prof_cfg.sh CFG TAG  -> rocprofv3 kernel stats + a steady-state timeline excerpt
export TMPDIR=/tmp
CFG=$1 TAG=$2
export CFG
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG} -- python3 tools/prof_cfg.py > gpurun_out/${TAG}.log 2>&1
cut -c1-170 gpurun_out/${TAG}/*/*kernel_stats.csv | head -14
python3 - <<PY
import csv,glob
rows=list(csv.DictReader(open(glob.glob("gpurun_out/${TAG}/*/*kernel_trace.csv")[0])))
rows.sort(key=lambda r:int(r['Start_Timestamp']))
n=len(rows); i0=int(n*0.7)
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i0+int("${3:-24}")]:
    nm=r['Kernel_Name'].split('(')[0].replace('void bnmf::','').replace('bnmf::','')[:26]
    print(f"{nm:28s} q={r['Queue_Id']:>2} start={(int(r['Start_Timestamp'])-t0)/1e3:9.1f} dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:8.1f} grid={r['Grid_Size_X']:>8} vgpr={r['VGPR_Count']} lds={r['LDS_Block_Size']} scr={r['Scratch_Size']}")
PY
