#!/usr/bin/env python3
"""Turn gpurun_out/prof (written by tools/profile_bench.sh on the GPU box) into the
tracked evidence under profiles/: the rocprofv3 --stats kernel summary, the
bench line of the same command and the HBM traffic of the dominant kernel.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and
WRITE_SIZE are collected in separate --pmc passes, are in KiB, and on gfx950
FETCH_SIZE under-reports wide coalesced streaming reads by 2x; this kernel's reads
are narrow strided dwords (uncalibrated pattern), so both the raw and the
doubled read figure are recorded and the conservative (doubled) one is used."""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern):
    f = glob.glob(os.path.join(src, pattern))
    if not f:
        raise SystemExit("missing " + pattern)
    return f[0]


stats = list(csv.DictReader(open(one("trace/*/*_kernel_stats.csv"))))
with open(os.path.join(dst, "%s_kernel_stats.csv" % tag), "w") as f:
    w = csv.DictWriter(f, fieldnames=list(stats[0].keys()))
    w.writeheader()
    w.writerows(stats)
ours = [r for r in stats if "arp::" in r["Name"]]
dom = max(ours, key=lambda r: float(r["TotalDurationNs"]))
kname = dom["Name"]
short = kname.split("(")[0].replace("void ", "")


def pmc(pattern, counter):
    vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(one(pattern)))
            if r["Counter_Name"] == counter and short in r["Kernel_Name"]]
    return vals


fetch = pmc("pmc_fetch/*/*_counter_collection.csv", "FETCH_SIZE")
write = pmc("pmc_write/*/*_counter_collection.csv", "WRITE_SIZE")
vg = [r for r in csv.DictReader(open(one("trace/*/*_kernel_trace.csv"))) if short in r["Kernel_Name"]][0]
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
fk, wk = sum(fetch) / len(fetch), sum(write) / len(write)
summary = {
    "kernel": kname, "calls": int(dom["Calls"]), "avg_ns": float(dom["AverageNs"]),
    "percentage_of_gpu_time": float(dom["Percentage"]),
    "vgpr": int(vg["VGPR_Count"]), "agpr": int(vg["Accum_VGPR_Count"]), "sgpr": int(vg["SGPR_Count"]),
    "lds_bytes": int(vg["LDS_Block_Size"]), "scratch_bytes": int(vg["Scratch_Size"]),
    "grid": int(vg["Grid_Size_X"]), "workgroup": int(vg["Workgroup_Size_X"]),
    "FETCH_SIZE_KiB_per_launch": fk, "WRITE_SIZE_KiB_per_launch": wk,
    "hbm_read_bytes_raw": fk * 1024, "hbm_read_bytes_x2_corrected": 2 * fk * 1024, "hbm_write_bytes": wk * 1024,
    "bytes_per_launch": 2 * fk * 1024 + wk * 1024,
    "algorithmic_bytes_per_launch": bench["roofline"]["algorithmic_bytes_per_launch"],
    "bench_kernel_ms": bench["roofline"]["kernel_ms"],
    "note": "rocprofv3 --kernel-trace --stats and two --pmc passes of `bench.py --steps 10 --warmup 2`; "
            "bench line from an un-profiled run of the same command",
}
json.dump(summary, open(os.path.join(dst, "%s_summary.json" % tag), "w"), indent=1)
json.dump({"bytes_per_launch": summary["bytes_per_launch"], "source": "%s_summary.json" % tag},
          open(os.path.join(dst, "hbm_traffic.json"), "w"))
open(os.path.join(dst, "%s_bench.json" % tag), "w").write(json.dumps(bench) + "\n")
print(json.dumps(summary, indent=1))
