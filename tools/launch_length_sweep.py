#!/usr/bin/env python3
"""Throughput of the fused HMC kernel against the launch length (transitions per launch) for the models of
bench.py: other_models -- short launches pay the state load / store and start inside the clock ramp of an idle GPU."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from autoreparam_amd import models, engine, _lib

def run(tag, spec, rep, C, L, step, flop):
    for T in (16, 64, 256, 1024):
        eng = engine.Engine(spec, "cuda:0"); eng.set_param(0, rep)
        rs = np.random.RandomState(3)
        st = engine.ChainState(torch.as_tensor((0.1 * rs.randn(C, spec.D)).astype(np.float32), device="cuda:0"))
        e = np.full(spec.D, step, np.float32)
        kw = dict(seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10 ** 9)
        for _ in range(2): eng.hmc_run(st, e, L, T, **kw)
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(3): eng.hmc_run(st, e, L, T, **kw)
        b.record(); torch.cuda.synchronize()
        ms = a.elapsed_time(b) / 3; rate = C * T * L / (ms * 1e-3)
        print("%-26s C=%6d L=%d T=%4d: %8.3f ms  %.3e leapfrog/s  %.3f of FP32  acc %.2f" % (
            tag, C, L, T, ms, rate, rate * flop / 157.3e12, st.accept_count.float().mean().item() / st.step), flush=True)

mn = models._spec_radon("MN"); Jm = mn.D - 3
run("radon MN CP", mn, "CP", 65536, 4, 0.05, 30.0 * Jm + 20 + 4.0 * mn.D)
run("radon MN CP (config 2)", mn, "CP", 4096, 4, 0.05, 30.0 * Jm + 20 + 4.0 * mn.D)
sd = models._spec_radon_stddvs("MN")
run("radon_stddvs MN NCP", sd, "NCP", 65536, 8, 0.01, 45.0 * Jm + 20.0 + 4.0 * sd.D)
el = models._spec_electric()
run("electric NCP", el, "NCP", 65536, 8, 0.01, 97 * 60.0 + 12 * 10.0 + 4.0 * el.D)
ts = models._spec_time_series()
run("time_series NCP", ts, "NCP", 65536, 8, 0.05, 60 * 80.0 + 4.0 * ts.D)
es = models._spec_election()
run("election NCP", es, "NCP", 131072, 4, 0.02, 4500.0 + 4.0 * es.D)
