#!/bin/bash
# SQ counters of the headline kernel (own pass, no tracing flags besides --pmc)
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/pmc_sq"; rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d "$OUT/a" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/a.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS --output-format csv -d "$OUT/b" -- python3 "$R/bench.py" --steps 5 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-} > "$OUT/b.log" 2>&1
python3 - <<PY
import csv,glob,collections
for d in ("a","b"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv"%d):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if ("interleaved_kernel" in r["Kernel_Name"] or "hmc_kernel" in r["Kernel_Name"]) and int(r["Grid_Size"]) >= 262144:
                acc[(r["Kernel_Name"][:60],r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(acc.items()): print(k[0][10:52],k[1],"%.4g"%(sum(v)/len(v)),"n=%d"%len(v))
PY
tail -3 "$OUT/a.log" "$OUT/b.log" | grep -i -E "error|fail" 
