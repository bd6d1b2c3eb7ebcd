"""Headline kernel (radon PA, interleaved CP/NCP, L = 8) per-GPU throughput of a strong-scaling shard for every
lanes-per-chain split: chains x lanes -> ms per 256-step launch, leapfrog-steps/s."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers
from autoreparam_amd import engine, _lib

sp = helpers.spec("radon_PA")
eng = engine.Engine(sp, "cuda:0")
eng.set_param(0, "CP"); eng.set_param(1, "NCP")
T, L = 256, 8
for chains in (8192, 16384, 32768):
    for lanes in (4, 8, 16):
        q0 = torch.as_tensor(helpers.states(sp, chains, seed=1, scale=0.1), device="cuda:0")
        st = engine.ChainState(q0)
        e = np.full(sp.D, 0.04, np.float32); e[2] = 0.01
        tr = torch.empty(T // 2, chains, sp.D, device="cuda:0")
        def run():
            eng.interleaved_run(st, e, e, L, L, T, seed=3, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10 ** 9, n_burnin=st.step,
                                thin=2, trace=tr, trace_centered=False, lanes=lanes)
        run(); run(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): run()
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print("chains %6d lanes %2d  %.3f ms  %.3e leapfrog-steps/s" % (chains, lanes, ms, chains * T * 2 * L / (ms * 1e-3)), flush=True)
