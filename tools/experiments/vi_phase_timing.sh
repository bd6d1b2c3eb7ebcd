#!/bin/bash
# where a VI step's cycles go: the -DARP_VI_TIMING variant (tools/build_variants.sh vit=-DARP_VI_TIMING) prints shader
# cycles per phase of thread 0 of the first workgroup.  phases: 0 parameters -> LDS, 1 draws (+ skipped words),
# 2 z, 3 gradient, 4 accumulation, 5 wave reduction, 6 workgroup barrier, 7 publish, 8 owners' gather, 9 totals' gather,
# 10 Adam, 11 loop
export ARP_DEBUG=1 ARP_LIB_PATH=$PWD/autoreparam_amd/libautoreparam_hip_vit.so VI_BENCH_KINDS=${VI_BENCH_KINDS:-1}
for m in "$@"; do python tools/vi_bench.py $m 2>&1 | grep -v "DEBUG SWITCH" | sort | uniq -c | sort -rn | head -6; done
