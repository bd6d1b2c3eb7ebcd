#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel trace + PMC passes (each in its own run, the
# program itself after `--`).  Outputs under gpurun_out/prof/ ; tools/summarize_profile.py turns them into profiles/*.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/prof"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
HL="--steps 20 --warmup 5 --no-cpu-baseline --headline-only ${BENCH_ARGS:-}"
python3 "$R/bench.py" ${BENCH_ARGS:-} --extras "$OUT/bench_extras.json" > "$OUT/bench.json" 2> "$OUT/bench.err"
# headline kernel alone: every launch in the trace is a warm-up or timed step
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" $HL > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" $HL > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" $HL > "$OUT/pmc_write.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d "$OUT/pmc_sqa" -- python3 "$R/bench.py" $HL > "$OUT/pmc_sqa.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d "$OUT/pmc_sqb" -- python3 "$R/bench.py" $HL > "$OUT/pmc_sqb.log" 2>&1
# instruction classes as the hardware counts them (cross-check of tools/asm_ledger.py's ISA ledger)
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT SQ_BUSY_CU_CYCLES --output-format csv -d "$OUT/pmc_sqc" -- python3 "$R/bench.py" $HL > "$OUT/pmc_sqc.log" 2>&1
# every kernel of the full bench (german, election, plain HMC, VI, ESS ...): kernel trace + the same SQ passes
FULL="--steps 20 --warmup 5 --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_full" -- python3 "$R/bench.py" $FULL > "$OUT/trace_full.log" 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_SALU --output-format csv -d "$OUT/pmc_full_sqa" -- python3 "$R/bench.py" $FULL --no-ess > "$OUT/pmc_full_sqa.log" 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA --output-format csv -d "$OUT/pmc_full_sqb" -- python3 "$R/bench.py" $FULL --no-ess > "$OUT/pmc_full_sqb.log" 2>&1
# HBM traffic of every kernel of the full bench, arp_ess included (separate FETCH_SIZE / WRITE_SIZE passes, MI355X_MICROARCH.md "HBM")
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_full_fetch" -- python3 "$R/bench.py" $FULL > "$OUT/pmc_full_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_full_write" -- python3 "$R/bench.py" $FULL > "$OUT/pmc_full_write.log" 2>&1
find "$OUT" -name '*.csv' | head -60
cat "$OUT/bench.json"; wc -c "$OUT/bench.json" "$OUT/bench.err" "$OUT/bench_extras.json"
