#!/usr/bin/env python3
"""VGPR / LDS / occupancy of every kernel in libmghip (hipcc -Rpass-analysis=kernel-resource-usage), one line each.

    python3 tools/kernel_resources.py [substring-of-mangled-name]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "mixed_precision_multigrid_solvers_for_pdes_amd", "csrc", "mghip.hip")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
       "-Rpass-analysis=kernel-resource-usage", "-o", "/tmp/_mg_res.so", SRC] + sys.argv[2:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = t.split(":", 1)[1].strip()
        rows[cur] = {}
    elif cur and ":" in t:
        k, v = t.rsplit(":", 1)
        rows[cur][k.strip()] = v.strip()
flt = sys.argv[1] if len(sys.argv) > 1 else ""
for name, r in sorted(rows.items()):
    if flt in name:
        short = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        short = re.sub(r"\(.*", "", short)[:110]
        print(f"{short:110s} vgpr {r.get('VGPRs'):>4s} agpr {r.get('AGPRs'):>3s} spill {r.get('VGPRs Spill'):>3s} "
              f"lds {r.get('LDS Size [bytes/block]'):>6s} occ {r.get('Occupancy [waves/SIMD]')}")
