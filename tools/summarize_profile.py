#!/usr/bin/env python3
"""Turn gpurun_out/prof (written by tools/profile_bench.sh on the GPU box) into the tracked evidence under
profiles/: the rocprofv3 --kernel-trace --stats summaries, the bench line of the same command, and for the dominant
(headline) kernel the HBM traffic and SQ counters that bench.py quotes in its `roofline` block.

HBM bytes follow /opt/skills/guides/MI355X_MICROARCH.md "HBM": FETCH_SIZE and WRITE_SIZE come from separate --pmc
passes, are in KiB, and on gfx950 FETCH_SIZE under-reports wide coalesced streaming reads by 2x; this kernel's reads
are narrow strided dwords (uncalibrated pattern), so the raw and the doubled read figure are both recorded and the
conservative (doubled) one is used.  SQ_*_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles (x 4 = cycles).

usage: tools/summarize_profile.py <tag>   (e.g. r02)"""
import csv
import glob
import json
import os
import subprocess
import sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)


def one(pattern, required=True):
    f = glob.glob(os.path.join(src, pattern))
    if not f:
        if required:
            raise SystemExit("missing " + pattern)
        return None
    return max(f, key=os.path.getmtime)     # gpurun merges into gpurun_out/: older runs' files may still be there


def short(name):
    return name.split("(")[0].replace("void ", "")


def copy_stats(pattern, out_name):
    rows = list(csv.DictReader(open(one(pattern))))
    with open(os.path.join(dst, out_name), "w") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        w.writerows(rows)
    return rows


def counters(pattern, match=None):
    """{kernel short name: {counter: mean value over its dispatches}}"""
    p = one(pattern, required=False)
    acc = defaultdict(lambda: defaultdict(list))
    if p:
        for r in csv.DictReader(open(p)):
            k = short(r["Kernel_Name"])
            if "arp::" not in k or (match and match not in k):
                continue
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in acc.items()}


stats = copy_stats("trace/*/*_kernel_stats.csv", "%s_kernel_stats.csv" % tag)
copy_stats("trace_full/*/*_kernel_stats.csv", "%s_kernel_stats_full_bench.csv" % tag)
ours = [r for r in stats if "arp::" in r["Name"]]
dom = max(ours, key=lambda r: float(r["TotalDurationNs"]))
kname = dom["Name"]
ks = short(kname)
bench = json.loads(open(os.path.join(src, "bench.json")).read().strip().splitlines()[-1])
open(os.path.join(dst, "%s_bench.json" % tag), "w").write(json.dumps(bench) + "\n")

vg = [r for r in csv.DictReader(open(one("trace/*/*_kernel_trace.csv"))) if ks in r["Kernel_Name"]][0]
fetch = counters("pmc_fetch/*/*_counter_collection.csv", ks).get(ks, {}).get("FETCH_SIZE")
write = counters("pmc_write/*/*_counter_collection.csv", ks).get(ks, {}).get("WRITE_SIZE")
sq = {}
for d in ("pmc_sqa", "pmc_sqb"):
    sq.update(counters("%s/*/*_counter_collection.csv" % d, ks).get(ks, {}))
cfg = bench["config"]
avg_ns = float(dom["AverageNs"])
n_simd = 256 * 4
derived = {}
if "GRBM_GUI_ACTIVE" in sq:
    cyc = sq["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
    derived["kernel_cycles_profiled_pass"] = cyc
    if "SQ_ACTIVE_INST_VALU" in sq:
        derived["valu_active_frac_of_simd_cycles"] = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (n_simd * cyc)
    if "SQ_INSTS_VALU" in sq:
        derived["simd_cycles_per_valu_inst"] = n_simd * cyc / sq["SQ_INSTS_VALU"]
        derived["valu_insts_per_wave_per_sampler_step"] = sq["SQ_INSTS_VALU"] / sq.get("SQ_WAVES", 1) / cfg["transitions_per_step"]
    derived["clock_ghz_estimate"] = cyc / avg_ns
head = subprocess.run(["git", "rev-parse", "--short", "HEAD"], cwd=ROOT, stdout=subprocess.PIPE, text=True).stdout.strip()
# the rocprofv3 average over the TIMED launches only (the last `steps` of the trace; the first launches of a process run
# at a lower clock) next to the HIP-event figure bench.py printed in that same profiled process
ktr = [r for r in csv.DictReader(open(one("trace/*/*_kernel_trace.csv"))) if ks in r["Kernel_Name"]]
ktr.sort(key=lambda r: int(r["Start_Timestamp"]))
durs = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in ktr]
n_timed = 20
try:
    line = [l for l in open(os.path.join(src, "trace.log")) if l.startswith("{")][-1]
    prof_bench = json.loads(line)
    n_timed = prof_bench["steps"]
    events_ms_same_process = prof_bench["roofline"]["kernel_ms"]
except Exception:
    events_ms_same_process = None
avg_timed_ns = sum(durs[-n_timed:]) / max(1, len(durs[-n_timed:]))
summary = {
    "kernel": kname, "calls": int(dom["Calls"]), "avg_ns": avg_ns,
    "avg_ns_timed_launches": avg_timed_ns, "hip_events_ms_same_profiled_process": events_ms_same_process,
    "percentage_of_gpu_time": float(dom["Percentage"]),
    # rocprofv3's VGPR_Count is in half-granules on gfx950 (124 for a kernel the compiler reports at 247 architected
    # VGPRs, allocated as 248 = 31 granules of 8): the register file figure is twice the column
    "vgpr_rocprofv3_column": int(vg["VGPR_Count"]), "vgpr": 2 * int(vg["VGPR_Count"]),
    "vgpr_note": "vgpr = 2 x rocprofv3 VGPR_Count (allocated registers, granule 8); the compiler's own count is in "
                 "tools/kernel_resources.py's output (-Rpass-analysis=kernel-resource-usage)",
    "agpr": int(vg["Accum_VGPR_Count"]), "sgpr": int(vg["SGPR_Count"]),
    "lds_bytes": int(vg["LDS_Block_Size"]), "scratch_bytes": int(vg["Scratch_Size"]),
    "grid": int(vg["Grid_Size_X"]), "workgroup": int(vg["Workgroup_Size_X"]),
    "config": {"chains": cfg["chains_per_gpu"], "transitions": cfg["transitions_per_step"], "thin": cfg["trace_thin"],
               "method": "i" if "interleaved" in cfg["workload"] else "CP", "leapfrog": cfg["num_leapfrog_steps"],
               "dataset": "PA" if cfg["D"] == 71 else "?", "lanes": cfg["lanes_per_chain"]},
    "FETCH_SIZE_KiB_per_launch": fetch, "WRITE_SIZE_KiB_per_launch": write,
    "hbm_read_bytes_raw": fetch * 1024 if fetch is not None else None,
    "hbm_read_bytes_x2_corrected": 2 * fetch * 1024 if fetch is not None else None,
    "hbm_write_bytes": write * 1024 if write is not None else None,
    "hbm_bytes_per_launch": (2 * fetch + write) * 1024 if fetch is not None and write is not None else None,
    "sq_counters_per_launch": sq, "derived": derived,
    "bench_kernel_ms_unprofiled": bench["roofline"]["kernel_ms"],
    "bench_frac_unprofiled": bench["roofline"]["frac"],
    "head": head,
    "note": "rocprofv3 --kernel-trace --stats and separate --pmc passes of `bench.py --steps 20 --warmup 5 "
            "--headline-only`; bench line from an un-profiled run of the same configuration (tools/profile_bench.sh)",
}
json.dump(summary, open(os.path.join(dst, "%s_headline.json" % tag), "w"), indent=1)

# every other kernel of the full bench: duration + SQ counters
full = {}
for d in ("pmc_full_sqa", "pmc_full_sqb"):
    for k, v in counters("%s/*/*_counter_collection.csv" % d).items():
        full.setdefault(k, {}).update(v)
with open(os.path.join(dst, "%s_sq_counters_all_kernels.txt" % tag), "w") as f:
    f.write("# rocprofv3 --pmc (two passes) of the full bench.py run: mean per launch; SQ_*_CYCLES / SQ_WAIT_* / "
            "SQ_ACTIVE_INST_* in quad-cycles\n")
    for k in sorted(full):
        f.write(k + "\n")
        for c in sorted(full[k]):
            f.write("    %-32s %.6g\n" % (c, full[k][c]))
print(json.dumps(summary, indent=1))
