"""First-contact GPU script: parity of logp/grad + HMC vs the C oracle, then a
throughput sweep over lanes-per-chain and chain counts for radon."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import oracle
from autoreparam_amd import models, engine, _lib
from autoreparam_amd.models import _spec_radon

import __graft_entry__ as ge
ge.smoke()

def sweep(ds, L, Cs, lanes_list, n_steps=64):
    spec = _spec_radon(ds)
    eng = engine.Engine(spec, "cuda:0")
    eng.set_param(0, "CP")
    D = spec.D
    BT = 4 * (5 * D + 8) + 1
    for C in Cs:
        rs = np.random.RandomState(0)
        q0 = torch.as_tensor((0.1 * rs.randn(C, D)).astype(np.float32), device="cuda:0")
        eps0 = np.full(D, 0.05 / (L / 4.0) ** 2, np.float32)
        for lanes in lanes_list:
            st = engine.ChainState(q0)
            try:
                eng.hmc_run(st, eps0, L, 8, seed=1, lanes=lanes)
            except RuntimeError as e:
                print("skip", ds, C, lanes, e); continue
            torch.cuda.synchronize()
            e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
            e0.record()
            eng.hmc_run(st, eps0, L, n_steps, seed=1, lanes=lanes)
            e1.record(); torch.cuda.synchronize()
            ms = e0.elapsed_time(e1)
            lf = C * n_steps * L / (ms * 1e-3)
            acc = st.accept_count.float().mean().item() / st.step
            print("radon-%s C=%6d L=%d lanes=%2d  %8.3f ms/%d steps  %.3e leapfrog/s  alg %.2f TB/s (%.1f%% of 8TB/s) acc=%.2f"
                  % (ds, C, L, lanes, ms, n_steps, lf, C * n_steps * BT / (ms * 1e-3) / 1e12,
                     100 * C * n_steps * BT / (ms * 1e-3) / 8e12, acc), flush=True)

sweep("PA", 8, [8192, 65536, 262144], [4, 8, 16])
sweep("MN", 4, [4096, 65536], [4, 8, 16])
# launch-per-transition mode
sweep("PA", 8, [65536], [4, 8], n_steps=1)
