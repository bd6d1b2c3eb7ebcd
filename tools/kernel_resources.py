#!/usr/bin/env python3
"""Compile one csrc/*.hip for gfx950 with -Rpass-analysis=kernel-resource-usage and print one
line per kernel (VGPRs, AGPRs, spills, scratch, occupancy, LDS)."""
import re
import subprocess
import sys

src = sys.argv[1]
extra = sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-fno-slp-vectorize", "--offload-arch=gfx950", "-std=c++17", "-c", src,
       "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + extra
out = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True).stdout
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r"remark: [^:]+:\d+:\d+:\s+(.*?) \[-Rpass", line)
    if not m:
        m = re.search(r":\d+:\d+: remark:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        name = t.split(":", 1)[1].strip()
        dem = subprocess.run(["c++filt", name], stdout=subprocess.PIPE, text=True).stdout.strip()
        cur = {"name": re.sub(r"\(.*", "", dem).replace("void arp::", "")}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
for r in rows:
    print("%-70s vgpr %-4s agpr %-4s spill %-4s scratch %-5s occ %-2s lds %s" % (
        r["name"][:70], r.get("VGPRs"), r.get("AGPRs"), r.get("VGPRs Spill"), r.get("ScratchSize [bytes/lane]"),
        r.get("Occupancy [waves/SIMD]"), r.get("LDS Size [bytes/block]")))
