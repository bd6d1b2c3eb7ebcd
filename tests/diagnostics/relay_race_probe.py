"""Hunt for a rare mismatch between a relay-segmented launch and the unsegmented one (seen once in
tests/test_gpu_hmc.py::test_relay_segments_equal_the_unsegmented_launch[8192-0]: accept_count1 differed, everything before it equal).
Repeats the segmented run many times against one reference and prints which arrays differ, for which chains.
usage: relay_race_probe.py [chains] [T] [segs] [repeats] [with_stats]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
os.environ["ARP_DEBUG"] = "1"
import numpy as np, torch
import helpers
from autoreparam_amd import engine, _lib
chains = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
segs = sys.argv[3] if len(sys.argv) > 3 else "8"
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
with_stats = (sys.argv[5] != "0") if len(sys.argv) > 5 else True
gpu = torch.device("cuda:0")
sp = helpers.spec("radon_PA")
eng = engine.Engine(sp, gpu)
eng.set_param(0, "CP"); eng.set_param(1, "NCP")
q0 = helpers.states(sp, chains, seed=2, scale=0.1)
e = np.full(sp.D, 0.06, np.float32); e[2] = 0.015
names = ["q", "grad", "logp", "adapt", "adapt1", "accept_count", "accept_count1", "rng", "trace", "acc0", "acc1", "stats"]

def run(s):
    os.environ["ARP_SEGMENTS"] = s
    st = engine.ChainState(torch.as_tensor(q0, device=gpu))
    n_burn = 101
    S = 2 * ((2 * T - n_burn) // 2 // 2 + 1)
    tr = torch.zeros(S, chains, sp.D, device=gpu)
    a0 = torch.zeros(S, chains, dtype=torch.uint8, device=gpu); a1 = torch.zeros_like(a0)
    extra = dict(stats=torch.zeros(6, chains, sp.D, device=gpu), stats_batch=3, n_samples=S) if with_stats else {}
    for _ in range(2):
        eng.interleaved_run(st, e, e, 4, 4, T, seed=9, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=150, n_burnin=n_burn, thin=2,
                            trace=tr, trace_accept0=a0, trace_accept1=a1, trace_centered=False, lanes=0, **extra)
    torch.cuda.synchronize()
    eng.check()
    out = [st.q, st.grad, st.logp, st.adapt, st.adapt1, st.accept_count, st.accept_count1, st.rng, tr, a0, a1]
    if with_stats:
        out.append(extra["stats"])
    return [t.cpu().numpy() for t in out]

ref = run("1")
bad = 0
for r in range(reps):
    got = run(segs)
    diffs = []
    for k in range(11):
        if not np.array_equal(ref[k], got[k], equal_nan=True):
            x, y = ref[k], got[k]
            if x.ndim >= 2 and x.shape[0] != chains:       # [S, C, ...]: chain axis 1
                ch = np.unique(np.nonzero((x != y).reshape(x.shape[0], chains, -1).any(axis=(0, 2)))[0])
            else:
                ch = np.unique(np.nonzero((x != y).reshape(chains, -1).any(axis=1))[0])
            diffs.append((names[k], len(ch), ch[:8].tolist(), ch[-3:].tolist()))
    if diffs:
        bad += 1
        print("rep %d MISMATCH:" % r, diffs, flush=True)
print("chains %d T %d segs %s stats %s: %d of %d repetitions differ from the unsegmented run" % (chains, T, segs, with_stats, bad, reps), flush=True)
