R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q 2>&1 | tail -3
run() { ARP_LIB_PATH=$1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --headline-only $2 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$3', round(d['roofline']['kernel_ms'],4), 'ms/step wall', round(d['ms_per_step'],4), '%.3e'%d['value'], d['accept_rate'])"; }
run "" "" deferred
run $R/autoreparam_amd/libautoreparam_hip_nodefer.so "" nodefer
run "" "--no-trace" notrace
run "" "" deferred
run $R/autoreparam_amd/libautoreparam_hip_nodefer.so "" nodefer
