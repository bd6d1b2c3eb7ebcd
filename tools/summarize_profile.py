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
if len(sys.argv) != 2:
    sys.exit(__doc__)
tag = sys.argv[1]
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
if os.path.exists(os.path.join(src, "bench_extras.json")):      # the long secondary figures of the same run
    import shutil
    shutil.copy(os.path.join(src, "bench_extras.json"), os.path.join(dst, "%s_bench_extras.json" % tag))

vg = [r for r in csv.DictReader(open(one("trace/*/*_kernel_trace.csv"))) if ks in r["Kernel_Name"]][0]
fetch = counters("pmc_fetch/*/*_counter_collection.csv", ks).get(ks, {}).get("FETCH_SIZE")
write = counters("pmc_write/*/*_counter_collection.csv", ks).get(ks, {}).get("WRITE_SIZE")
sq = {}
for d in ("pmc_sqa", "pmc_sqb", "pmc_sqc"):
    sq.update(counters("%s/*/*_counter_collection.csv" % d, ks).get(ks, {}))
cfg = bench["config"]
avg_ns = float(dom["AverageNs"])
n_simd = 256 * 4
derived = {}
if "GRBM_GUI_ACTIVE" in sq:
    cyc = sq["GRBM_GUI_ACTIVE"] / 8.0          # the counter is summed over the 8 XCDs
    derived["kernel_cycles_profiled_pass"] = cyc
    if "SQ_ACTIVE_INST_VALU" in sq:
        # NOT a bounded utilisation: the execution of a SIMD's two waves overlaps and is counted twice (1.06 / 1.10 for
        # round 3's electric / radon_stddvs kernels).  Kept under a name that says so; the bounded figure is
        # `issue_bound_frac` below (the ISA ledger's instruction mix priced at its best issue rates / measured cycles).
        derived["sum_over_waves_valu_active_over_simd_cycles_unbounded"] = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (n_simd * cyc)
    if "SQ_WAVE_CYCLES" in sq and "SQ_ACTIVE_INST_VALU" in sq:
        # fraction of a wave's own lifetime it spends executing vector instructions (<= 1)
        derived["valu_active_frac_of_wave_lifetime"] = sq["SQ_ACTIVE_INST_VALU"] / sq["SQ_WAVE_CYCLES"]
    if "SQ_INSTS_VALU" in sq:
        derived["simd_cycles_per_valu_inst"] = n_simd * cyc / sq["SQ_INSTS_VALU"]
        # relay segments: the launch's steps are cut into segments, a wave per (segment, 64 lanes), so a
        # wave runs transitions_per_step / segments of them -- the number of segments is the power of two by which the
        # launch's waves exceed one wave per 64 lanes at 4 lanes per chain (the headline's split)
        waves_whole = cfg["chains_per_gpu"] * 4 / 64.0
        segs = max(1, int(round(sq.get("SQ_WAVES", waves_whole) / waves_whole)))
        derived["relay_segments"] = segs
        steps_per_wave = cfg["transitions_per_step"] / float(segs)
        derived["valu_insts_per_wave_per_sampler_step"] = sq["SQ_INSTS_VALU"] / sq.get("SQ_WAVES", 1) / steps_per_wave
    derived["clock_ghz_estimate"] = cyc / avg_ns
    try:
        import importlib.util
        spec_ = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
        bm = importlib.util.module_from_spec(spec_)
        spec_.loader.exec_module(bm)
        mix, mix_src = bm.headline_mix()                 # the `weighted` row of the newest profiles/rNN_headline_ledger.txt
        derived["issue_mix_source"] = mix_src
        priced = sum(bm.HEADLINE_COST[k] * v for k, v in mix.items())     # SIMD cycles per wave and step
        waves_per_simd_total = sq.get("SQ_WAVES", 0) / n_simd
        derived["issue_bound_cycles_at_2.4GHz_per_wave_step"] = priced
        # time based (the costs are times quoted at 2.4 GHz): priced time / this profiled pass's measured time
        spw = cfg["transitions_per_step"] / float(derived.get("relay_segments", 1))     # steps a wave runs
        derived["issue_bound_frac"] = (priced / 2.4e9) * spw * waves_per_simd_total / (avg_ns * 1e-9)
        derived["issue_bound_frac_if_costs_were_true_cycles"] = priced * spw * waves_per_simd_total / cyc
    except Exception as e:
        derived["issue_bound_error"] = repr(e)
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

# HBM traffic per launch of every kernel of the full bench (FETCH_SIZE / WRITE_SIZE in KiB; reads doubled per the gfx950
# note for wide coalesced streams -- arp_ess's loads are 256-byte coalesced dword streams, so the doubled figure applies)
traffic = {}
for cname, d in (("FETCH_SIZE", "pmc_full_fetch"), ("WRITE_SIZE", "pmc_full_write")):
    p = one("%s/*/*_counter_collection.csv" % d, required=False)
    if not p:
        continue
    per = defaultdict(list)
    for r in csv.DictReader(open(p)):
        if r["Counter_Name"] == cname and "arp::" in r["Kernel_Name"]:
            per[(short(r["Kernel_Name"]), int(r["Grid_Size"]))].append(float(r["Counter_Value"]) * 1024.0)
    for (k, g), v in per.items():
        traffic.setdefault(k, {}).setdefault(str(g), {})[cname + "_bytes"] = v
ess_key = [k for k in traffic if "ess_kernel" in k]
if ess_key:
    ek = traffic[ess_key[0]]
    g = max(ek, key=lambda x: int(x))                       # the headline-size trace: 65 536 x 71 series
    f, w = ek[g].get("FETCH_SIZE_bytes", []), ek[g].get("WRITE_SIZE_bytes", [])
    ess_sum = {"kernel": ess_key[0], "grid": int(g), "dispatches": len(f),
               "fetch_bytes_raw_per_dispatch": f, "write_bytes_per_dispatch": w,
               "hbm_bytes_per_dispatch_x2_reads": [2 * a + (w[i] if i < len(w) else 0.0) for i, a in enumerate(f)],
               "algorithmic_bytes": 4.0 * 1000 * int(g) if True else None,
               "note": "per dispatch, in launch order: bench.py's own 1 000-sample trace (6 calls), then the reference flow's "
                       "two --method=i candidates (num_ls 4 and 8; the second is the kept one); grid = series rounded up to 256",
               "head": head}
    json.dump(ess_sum, open(os.path.join(dst, "%s_ess_kernel.json" % tag), "w"), indent=1)
json.dump(traffic, open(os.path.join(dst, "%s_hbm_traffic_all_kernels.json" % tag), "w"), indent=1)

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
