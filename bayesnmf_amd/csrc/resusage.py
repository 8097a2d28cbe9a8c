#!/usr/bin/env python3
"""Print VGPR/SGPR/scratch/occupancy per kernel (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, subprocess, sys
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
       "-fno-fast-math", "-Wno-unused-value", "-Wno-unused-result", "-I../../include",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/libbnmf_res.so", "api.hip"]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for l in out.splitlines():
    m = re.search(r"remark: (.*?) \[-Rpass", l)
    if not m:
        continue
    s = m.group(1)
    if s.startswith("Function Name:"):
        cur = s.split(":")[1].strip(); rows[cur] = {}
    elif cur and ":" in s:
        k, v = s.rsplit(":", 1); rows[cur][k.strip()] = v.strip()
for f, r in rows.items():
    name = subprocess.run(["c++filt", f], capture_output=True, text=True).stdout.strip().split("(")[0]
    print(f"{name[:44]:44s} VGPR {r.get('VGPRs'):>4s} SGPR {r.get('TotalSGPRs'):>4s} scratch {r.get('ScratchSize [bytes/lane]'):>4s} "
          f"occ {r.get('Occupancy [waves/SIMD]')} LDS {r.get('LDS Size [bytes/block]')}")
