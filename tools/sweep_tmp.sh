R=$GRAFT_REPO_ROOT
python tools/debug_lanes.py s0
ARP_LIB_PATH=$R/autoreparam_amd/libautoreparam_hip_gs1.so python tools/debug_lanes.py s1
python - <<PY
import numpy as np
for n in (1,2,3,5):
    q0,q1=np.load("/tmp/q_s0_%d.npy"%n),np.load("/tmp/q_s1_%d.npy"%n)
    g0,g1=np.load("/tmp/g_s0_%d.npy"%n),np.load("/tmp/g_s1_%d.npy"%n)
    acc=np.load("/tmp/a_s0_%d.npy"%n)
    dq=np.argwhere(q0!=q1); dg=np.argwhere(g0!=g1)
    print("after %d steps: q globals differ (chain, elem):"%n, dq[:8].tolist(), "| grad:", dg[:8].tolist(), "| accepted so far of first differing:", [int(acc[c]) for c,_ in dq[:4]])
PY
