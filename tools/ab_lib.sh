#!/bin/bash
# A/B of tagged library variants on ONE box: tools/ab_lib.sh "" _tab ...   (headline launch, 20 timed launches, two rounds)
for rep in 1 2 3; do
for v in "$@"; do
  if [ -z "$v" ]; then L=""; else L="ARP_DEBUG=1 ARP_LIB_PATH=$PWD/autoreparam_amd/libautoreparam_hip$v.so"; fi
  R=$(env $L python bench.py --headline-only --no-cpu-baseline --steps 20 --warmup 5 ${BENCH_ARGS:-} 2>/dev/null | python -c "
import sys, json
b = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = b['roofline']
print('%.3f ms frac %.4f clock %.3f GHz' % (r['kernel_ms'], r['frac'], r.get('clock_ghz_live') or 0))")
  echo "variant '${v:-default}' rep $rep: $R"
done; done
