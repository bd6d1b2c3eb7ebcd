import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from autoreparam_amd import models, engine, _lib
spec = models._spec_radon("MN")
for C in (16384, 65536):
    for lanes in (4, 8, 16):
        eng = engine.Engine(spec, "cuda:0"); eng.set_param(0, "CP"); eng.set_param(1, "NCP")
        rs = np.random.RandomState(0)
        st = engine.ChainState(torch.as_tensor((0.1 * rs.randn(C, spec.D)).astype(np.float32), device="cuda:0"))
        e = np.full(spec.D, 0.05, np.float32)
        for inter in (0, 1):
            def run():
                if inter: eng.interleaved_run(st, e, e, 4, 4, 64, seed=1, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10**9, lanes=lanes)
                else: eng.hmc_run(st, e, 4, 64, seed=1, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10**9, lanes=lanes)
            run(); torch.cuda.synchronize()
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(3): run()
            b.record(); torch.cuda.synchronize()
            ms = a.elapsed_time(b) / 3
            LL = 8 if inter else 4
            print("MN C=%d lanes=%d %s: %.3f ms  %.3e leapfrog/s" % (C, lanes, "interleaved" if inter else "plain", ms, C * 64 * LL / (ms * 1e-3)), flush=True)
