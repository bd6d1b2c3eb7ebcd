#!/bin/bash
# SQ counters of the kernels whose name contains $KERNEL, for an arbitrary python script:
#   KERNEL=hmc_kernel bash tools/pmc_cmd.sh tools/german_probe.py one
# (two --pmc passes, no tracing flags; rocprofv3 is given python3 directly)
R="${GRAFT_REPO_ROOT:-$(pwd)}"; OUT="$R/gpurun_out/pmc_cmd"; rm -rf "$OUT"; mkdir -p "$OUT"
S="$R/$1"; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d "$OUT/a" -- python3 "$S" "$@" > "$OUT/a.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d "$OUT/b" -- python3 "$S" "$@" > "$OUT/b.log" 2>&1
KERNEL="${KERNEL:-arp::}" python3 - <<PY
import csv,glob,collections,os
for d in ("a","b"):
    for f in glob.glob("$OUT/%s/*/*counter_collection.csv"%d):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if os.environ["KERNEL"] in r["Kernel_Name"]:
                acc[(r["Kernel_Name"].split("(")[0][-64:],r["Grid_Size"],r["Counter_Name"])].append(float(r["Counter_Value"]))
        for k,v in sorted(acc.items()): print(k[0],"grid",k[1],k[2],"%.4g"%(sum(v)/len(v)),"n=%d"%len(v))
PY
tail -3 "$OUT/a.log" "$OUT/b.log" | grep -i -E "error|fail"
