"""German credit, 16 384 chains, 4 lanes per chain: microseconds per transition at L = 1, 2, 4, 8 leapfrog steps for both
matrix-core forms.  The slope is the cost of an interior gradient, the intercept that of the closing (log-density) gradient plus
the draw / Metropolis test / adaptation of a transition (profiles/r05_german_math.txt)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers
from autoreparam_amd import engine, _lib
sp = helpers.spec("german")
for math in ("bf16x3", "f32"):
    eng = engine.Engine(sp, "cuda:0"); eng.set_option("german_math", math); eng.set_param(0, "NCP")
    for L in (1, 2, 4, 8):
        st = engine.ChainState(torch.as_tensor(helpers.states(sp, 16384, seed=1, scale=0.1), device="cuda:0"))
        eps = np.full(sp.D, 0.005, np.float32)
        kw = dict(seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10**9, lanes=4)
        for _ in range(2): eng.hmc_run(st, eps, L, 128, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3): eng.hmc_run(st, eps, L, 128, **kw)
        e1.record(); torch.cuda.synchronize()
        print(math, "L", L, "us per transition %.1f" % (1e3 * e0.elapsed_time(e1) / 3 / 128), flush=True)
