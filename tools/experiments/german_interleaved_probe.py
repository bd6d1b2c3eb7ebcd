"""German credit, 16 384 chains: the interleaved CP / NCP sampler (generic interleaved_kernel, re-bootstraps after each change of
coordinates) against two plain HMC runs of the same leapfrog counts -- microseconds per gradient evaluation."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from autoreparam_amd import engine, _lib
sp = helpers.spec("german")
C, L, T = 16384, 4, 64
def timeit(f, n=3):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n
for math in ("bf16x3", "f32"):
    eng = engine.Engine(sp, "cuda:0"); eng.set_option("german_math", math)
    eng.set_param(0, "CP"); eng.set_param(1, "NCP")
    eps = np.full(sp.D, 0.005, np.float32)
    st = engine.ChainState(torch.as_tensor(helpers.states(sp, C, seed=1, scale=0.1), device="cuda:0"))
    ms_i = timeit(lambda: eng.interleaved_run(st, eps, eps, L, L, T, seed=3, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10**9, lanes=4))
    st2 = engine.ChainState(torch.as_tensor(helpers.states(sp, C, seed=1, scale=0.1), device="cuda:0"))
    ms_cp = timeit(lambda: eng.hmc_run(st2, eps, L, T, which=0, seed=3, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10**9, lanes=4))
    ms_ncp = timeit(lambda: eng.hmc_run(st2, eps, L, T, which=1, seed=3, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10**9, lanes=4))
    print("%-7s interleaved %.2f ms per %d steps = %.1f us per gradient (2L + 2 = %d per step); plain CP %.2f ms, NCP %.2f ms = %.1f / %.1f us per gradient"
          % (math, ms_i, T, 1e3 * ms_i / T / (2 * L + 2), 2 * L + 2, ms_cp, ms_ncp, 1e3 * ms_cp / T / L, 1e3 * ms_ncp / T / L), flush=True)
