import os, sys
import numpy as np, torch
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import helpers
from autoreparam_amd import engine, _lib
sp = helpers.spec("german")
for math in sys.argv[1:] or ["bf16x3"]:
    eng = engine.Engine(sp, "cuda:0"); eng.set_option("german_math", math); eng.set_param(0, "NCP")
    st = engine.ChainState(torch.as_tensor(helpers.states(sp, 16384, seed=1, scale=0.1), device="cuda:0"))
    eps = np.full(sp.D, 0.005, np.float32)
    for _ in range(2):
        eng.hmc_run(st, eps, 4, 256, seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10**9, lanes=4)
    torch.cuda.synchronize()
