export ARP_DEBUG=1
run() { L=""; [ -n "$1" ] && L="ARP_LIB_PATH=$PWD/autoreparam_amd/libautoreparam_hip$1.so"; env $L python bench.py --headline-only --no-cpu-baseline --no-ess --steps 20 --warmup 5 2>/dev/null | python -c "
import sys, json
b = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r = b['roofline']
print('%.3f ms (min %.3f max %.3f) frac %.4f clock %s' % (r['kernel_ms'], r['kernel_ms_min'], r['kernel_ms_max'], r['frac'], r.get('clock_ghz_live')))"; }
for rep in 1 2 3; do echo "old rep $rep: $(run _old)"; echo "new rep $rep: $(run '')"; done
