#!/bin/bash
# A/B of tagged library variants on ONE box, secondary bench entries: tools/ab_other.sh <name-substring> "" _tag ...
PAT=$1; shift
for rep in 1 2; do for v in "$@"; do
  if [ -z "$v" ]; then L=""; else L="ARP_DEBUG=1 ARP_LIB_PATH=$PWD/autoreparam_amd/libautoreparam_hip$v.so"; fi
  env $L python bench.py --no-cpu-baseline --no-ess --steps 10 --warmup 3 2>/dev/null | python -c "
import sys, json
b=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
for k,v in b['other_models'].items():
    if '$PAT' in k: print('variant \"$v\"', k, '%.3f ms %.4g' % (v['kernel_ms'], v['leapfrog_steps_per_s']), 'short launches %.4g' % v['at_round3_launch_length']['leapfrog_steps_per_s'])
"; done; done
