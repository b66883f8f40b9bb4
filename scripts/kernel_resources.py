#!/usr/bin/env python3
"""kernel_resources.py file.hip [pattern]: VGPRs / spills / scratch / occupancy of every kernel of one csrc source file
(hipcc -Rpass-analysis=kernel-resource-usage; runs in the build container, no GPU needed)."""
import os
import re
import subprocess
import sys

csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "vector-indexer_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
       "-fno-slp-vectorize", "-c", sys.argv[1], "-o", "/tmp/kr_%d.o" % os.getpid(), "-Rpass-analysis=kernel-resource-usage"]
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
pat = sys.argv[2] if len(sys.argv) > 2 else ""
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = m.group(1)
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
for k, v in rows.items():
    if pat in k:
        print("%-80s vgpr %3s agpr %3s spill %3s scratch %4s occ %s" % (k[-80:], v.get("VGPRs"), v.get("AGPRs"), v.get("VGPRs Spill"),
                                                                        v.get("ScratchSize"), v.get("Occupancy")))
try:
    os.remove("/tmp/kr_%d.o" % os.getpid())
except OSError:
    pass
