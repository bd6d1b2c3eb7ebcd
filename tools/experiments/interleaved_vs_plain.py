"""Every model under --method=i (interleaved CP / NCP, 2 L + 2 gradients per step: the reference re-bootstraps after each change of
coordinates) against plain HMC in both parameterisations: nanoseconds per gradient and chain."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from autoreparam_amd import engine, _lib
def timeit(f, n=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
L, T = 4, 256
for mname, C in (("radon_PA", 65536), ("radon_MN", 65536), ("election", 131072), ("electric", 65536), ("radon_sd_MN", 65536),
                 ("time_series", 65536), ("8schools", 262144)):
    sp = helpers.spec(mname)
    eng = engine.Engine(sp, "cuda:0"); eng.set_param(0, "CP"); eng.set_param(1, "NCP")
    eps = np.full(sp.D, 1e-3, np.float32)
    q0 = helpers.states(sp, C, seed=1, scale=0.05)
    st = engine.ChainState(torch.as_tensor(q0, device="cuda:0"))
    ms_i = timeit(lambda: eng.interleaved_run(st, eps, eps, L, L, T, seed=3, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10**9))
    st2 = engine.ChainState(torch.as_tensor(q0, device="cuda:0"))
    ms_cp = timeit(lambda: eng.hmc_run(st2, eps, L, T, which=0, seed=3, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10**9))
    ms_ncp = timeit(lambda: eng.hmc_run(st2, eps, L, T, which=1, seed=3, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10**9))
    per = lambda ms, n: 1e6 * ms / T / n / C
    print("%-12s C=%-7d interleaved %8.2f ms: %.3f ns per gradient and chain at 2L+2, %.3f counting 2L;  plain CP %.3f  NCP %.3f"
          % (mname, C, ms_i, per(ms_i, 2 * L + 2), per(ms_i, 2 * L), per(ms_cp, L), per(ms_ncp, L)), flush=True)
