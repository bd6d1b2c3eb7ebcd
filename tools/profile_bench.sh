#!/bin/bash
# Run on the GPU box (via gpurun): bench line + rocprofv3 kernel trace + HBM PMC passes.
# Outputs under gpurun_out/prof/ ; tools/summarize_profile.py turns them into profiles/*.
set -u
R="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$R/gpurun_out/prof"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ARGS="--steps 10 --warmup 2 --no-cpu-baseline --headline-only ${BENCH_ARGS:-}"
python3 "$R/bench.py" ${BENCH_ARGS:-} > "$OUT/bench.json" 2> "$OUT/bench.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 "$R/bench.py" $ARGS > "$OUT/trace.log" 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 "$R/bench.py" $ARGS > "$OUT/pmc_fetch.log" 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- python3 "$R/bench.py" $ARGS > "$OUT/pmc_write.log" 2>&1
find "$OUT" -name '*.csv' | head -50
cat "$OUT/bench.json"
