#!/bin/bash
# where the bf16 x 3 tile loop's time goes: variants that skip the operand reads after the first tile / the hand-over
# barrier and LDS-DMA / the backward matrix-core instructions (wrong results; timing only)
export ARP_DEBUG=1
for v in "" _noread _nobar _nomfma; do L=""; [ -n "$v" ] && L="ARP_LIB_PATH=$PWD/autoreparam_amd/libautoreparam_hip$v.so"
  echo "variant '${v:-default}': $(env $L python tools/german_math_ab.py 16384 256 2>&1 | grep "NCP  bf16x3" | sed 's/.*hmc/hmc/')"; done
