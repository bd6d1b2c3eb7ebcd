#!/bin/bash
# the headline launch at the same chains x steps product, split differently: 1 024 workgroups are exactly two rounds on the
# 512 resident slots; more workgroups are handed out dynamically (profiles/r05_chain_count_sweep.txt)
for cfg in "65536 1024" "131072 512" "262144 256" "98304 680" "81920 820" "73728 910"; do
  set -- $cfg
  python bench.py --headline-only --no-cpu-baseline --no-ess --steps 10 --warmup 3 --chains $1 --transitions $2 2>/dev/null | python -c "
import sys, json
b = json.loads(sys.stdin.read()); r = b['roofline']
print('chains %7d steps %5d: %.3f ms  %.4f ns per chain and step  %.4e leapfrog-steps/s' % ($1, $2, r['kernel_ms'], 1e6 * r['kernel_ms'] / ($1 * $2), b['value']))"
done
