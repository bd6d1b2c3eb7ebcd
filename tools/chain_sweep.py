"""Throughput against the number of chains on one GPU (C in {4 096, 16 384, 65 536, 262 144, 1 048 576}): the headline
sampler (radon PA, interleaved CP / NCP, 4 + 4 leapfrogs, a trace row every second step), radon MN plain HMC (BASELINE config 2's
kernel), election and German credit plain HMC -- the library's own lanes-per-chain choice at every size."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from autoreparam_amd import engine, _lib
def timeit(f, n=3):
    f(); f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
def lanes_used(st):
    return int((st.rng.view(st.rng.shape[0], -1, 4)[0, :, :2].abs().sum(dim=1) != 0).sum().item())
print("%-28s %9s %6s %6s %10s %14s" % ("workload", "chains", "lanes", "T", "ms", "leapfrogs/s"))
for C in (4096, 16384, 65536, 262144, 1048576):
    T = 256 if C <= 262144 else 64
    sp = helpers.spec("radon_PA"); eng = engine.Engine(sp, "cuda:0"); eng.set_param(0, "CP"); eng.set_param(1, "NCP")
    st = engine.ChainState(torch.as_tensor(helpers.states(sp, C, seed=1, scale=0.1), device="cuda:0"))
    e = np.full(sp.D, 0.08, np.float32); e[2] = 0.02
    tr = torch.empty(T // 2, C, sp.D, device="cuda:0")
    ms = timeit(lambda: eng.interleaved_run(st, e, e, 4, 4, T, seed=3, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10**9, n_burnin=st.step, thin=2, trace=tr, trace_centered=False))
    print("%-28s %9d %6d %6d %10.3f %14.4e" % ("radon PA interleaved 4+4", C, lanes_used(st), T, ms, C * T * 8 / (ms * 1e-3)), flush=True)
    del tr, st
    for mname, L, eps in (("radon_MN", 4, 0.05), ("election", 4, 0.02), ("german", 4, 0.005)):
        if mname == "german" and C > 262144: continue
        Tm = T if mname != "german" else 16
        sp = helpers.spec(mname); eng = engine.Engine(sp, "cuda:0"); eng.set_param(0, "NCP" if mname != "radon_MN" else "CP")
        st = engine.ChainState(torch.as_tensor(helpers.states(sp, C, seed=1, scale=0.05), device="cuda:0"))
        ee = np.full(sp.D, eps, np.float32)
        ms = timeit(lambda: eng.hmc_run(st, ee, L, Tm, seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10**9))
        print("%-28s %9d %6d %6d %10.3f %14.4e" % (mname + " plain HMC L=4", C, lanes_used(st), Tm, ms, C * Tm * L / (ms * 1e-3)), flush=True)
        del st
