R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -x -q -k "election or density or vi_" 2>&1 | tail -3
for lib in "" $R/autoreparam_amd/libautoreparam_hip_el3.so; do
ARP_LIB_PATH=$lib python tools/model_sweep.py election 2>&1 | tail -12
done
