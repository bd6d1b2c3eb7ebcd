#!/usr/bin/env python3
"""German credit at 4 lanes per chain: the f32 matrix-core likelihood against the bf16 x 3 one (arp_model_set_option
"german_math") -- accuracy of log density and gradient against the float64 oracle on the same states, and the fused HMC
kernel's time at 16 384 chains, L = 4, 256 transitions per launch.

    python tests/diagnostics/german_math_ab.py [chains=16384] [transitions=256]
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import oracle  # noqa: E402  (the checker)
from autoreparam_amd import engine, _lib  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
T = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
sp = helpers.spec("german")
orc = oracle.OracleModel(sp)
for kind in ("NCP", "VIP"):
    a, b = helpers.params(sp, kind)
    x = helpers.states(sp, 512, seed=3, scale=0.3)
    lp64, g64 = orc.logp_grad(x, a, b, dtype=np.float64)
    for math in ("f32", "bf16x3"):
        eng = engine.Engine(sp, dev)
        eng.set_option("german_math", math)
        eng.set_param(0, (a, b))
        lp, g = eng.logp_grad(x, lanes=4)
        lp, g = lp.cpu().numpy().astype(np.float64), g.cpu().numpy().astype(np.float64)
        e_lp = np.abs(lp - lp64).max() / np.abs(lp64).max()
        e_g = np.abs(g - g64).max() / np.abs(g64).max()
        e_g_med = np.median(np.abs(g - g64)) / np.abs(g64).max()
        q0 = torch.as_tensor(helpers.states(sp, C, seed=1, scale=0.1), device=dev)
        st = engine.ChainState(q0)
        eps = np.full(sp.D, 0.005, np.float32)
        kw = dict(seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10 ** 9, lanes=4)
        for _ in range(2):
            eng.hmc_run(st, eps, 4, T, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            eng.hmc_run(st, eps, 4, T, **kw)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 3
        acc = float(st.accept_count.float().mean().item()) / st.step
        print("%-4s %-7s logp err %.2e  grad err max %.2e median %.2e (relative to the largest entry)   hmc %8.3f ms per %d transitions = %.3e leapfrog-steps/s  accept %.3f finite %s" % (
            kind, math, e_lp, e_g, e_g_med, ms, T, C * T * 4 / (ms * 1e-3), acc, bool(torch.isfinite(st.q).all())), flush=True)
