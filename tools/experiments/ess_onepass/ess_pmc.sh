cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/ess_pmc; mkdir -p $O
export ARP_DEBUG=1 ARP_ESS_ONEPASS=1 ESS_WHITE_ONLY=1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d $O/a -- python3 $R/tools/experiments/ess_onepass/ess_pair_time.py > $O/a.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_SALU SQ_IFETCH --output-format csv -d $O/b -- python3 $R/tools/experiments/ess_onepass/ess_pair_time.py > $O/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_VMEM --output-format csv -d $O/c -- python3 $R/tools/experiments/ess_onepass/ess_pair_time.py > $O/c.log 2>&1
python3 - <<'PY'
import csv, glob, os, collections
O=os.environ.get("GRAFT_REPO_ROOT")+"/gpurun_out/ess_pmc"
for d in "abc":
    for f in glob.glob(O+"/%s/**/*counter_collection.csv"%d, recursive=True):
        acc=collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "ess_pair" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items(): print(d, k, "n=%d"%len(v), "mean %.4g"%(sum(v)/len(v)))
PY
tail -3 $O/a.log
