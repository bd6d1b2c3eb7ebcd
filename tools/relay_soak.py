"""Soak of the relay segments: random chain counts (whole and ragged rounds of workgroups), step counts and
workloads, each run with the library's segments and with one workgroup per chain block (ARP_DEBUG=1 ARP_SEGMENTS=1) from the same
state -- states, gradients, generator states and counters must be equal bit for bit every time (a stale read across a hand-off
would show here).  usage: relay_soak.py [seed] [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ARP_DEBUG"] = "1"
import numpy as np, torch
import helpers
from autoreparam_amd import engine, _lib
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
WORK = [("radon_PA", "i"), ("radon_MN", "CP"), ("election", "NCP"), ("election", "i"), ("electric", "NCP"), ("radon_sd_MN", "CP")]
engines = {}
t0 = time.time(); n = 0; steps = 0
while time.time() - t0 < budget:
    mname, mode = WORK[rs.randint(len(WORK))]
    sp = helpers.spec(mname)
    if mname not in engines: engines[mname] = engine.Engine(sp, "cuda:0")
    eng = engines[mname]
    C = int(rs.choice([8192, 8200, 16384, 32768, 40000, 49152, 65536, 65537, 98304, 131072]))
    T = int(rs.randint(256, 1200))
    L = int(rs.randint(1, 4))
    if mode == "i": eng.set_param(0, "CP"); eng.set_param(1, "NCP")
    else: eng.set_param(0, mode)
    q0 = helpers.states(sp, C, seed=int(rs.randint(1000)), scale=0.05)
    e = np.full(sp.D, 2e-3, np.float32)
    seed = int(rs.randint(1 << 30))
    outs = []
    for segs in (None, "1"):
        if segs is None: os.environ.pop("ARP_SEGMENTS", None)
        else: os.environ["ARP_SEGMENTS"] = segs
        st = engine.ChainState(torch.as_tensor(q0, device="cuda:0"))
        for _ in range(2):
            if mode == "i":
                eng.interleaved_run(st, e, e, L, L, T, seed=seed, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=T)
            else:
                eng.hmc_run(st, e, L, T, seed=seed, adapt_kind=_lib.ADAPT_DUAL, n_adapt=T)
        torch.cuda.synchronize()
        outs.append([t.clone() for t in (st.q, st.grad, st.logp, st.adapt, st.rng, st.accept_count)])
    for k, (x, y) in enumerate(zip(*outs)):
        assert torch.equal(x, y) or (torch.isnan(x) == torch.isnan(y)).all() and torch.equal(torch.nan_to_num(x), torch.nan_to_num(y)), (mname, mode, C, T, L, seed, k)
    n += 1; steps += 4 * T
print("relay soak ok: %d comparisons (segmented = unsegmented, bit for bit), %d launched steps in %.0f s" % (n, steps, time.time() - t0))
