#!/bin/bash
# FETCH_SIZE of every ess_kernel dispatch of tools/ess_bench.py (white noise: exactly one pass over the trace -- the
# calibration of the counter for THIS access pattern, dword loads coalesced to 256 bytes per wave; AR(1); the sampler's
# own trace), one --pmc pass, nothing else traced.  ARP_DEBUG=1 ARP_ESS_TILE=1 in the environment: the one-pass tile kernel.
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/ess_traffic"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/f" -- python3 "$R/tools/ess_bench.py" > "$OUT/f.log" 2>&1
grep -v amdgpu.ids "$OUT/f.log" | tail -4
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/f/*/*counter_collection.csv"):
    rows = [r for r in csv.DictReader(open(f)) if ("ess_kernel" in r["Kernel_Name"] or "ess_tile_kernel" in r["Kernel_Name"]) and r["Counter_Name"] == "FETCH_SIZE"]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    alg = 4.0 * 1000 * 65536 * 71
    print(sorted(set(r["Kernel_Name"].split("(")[0] for r in rows)))
    print("ess kernel dispatches in order (6 per trace: white noise, AR(1), sampler): FETCH_SIZE raw bytes / algorithmic bytes")
    print(" ".join("%.3f" % (float(r["Counter_Value"]) * 1024 / alg) for r in rows))
PY
