"""Randomised density / gradient / converter parity of the HIP lanes against the float64 oracle:
every model x lanes-per-chain x several seeds of (a, b) and states at two scales."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers, oracle
from autoreparam_amd import engine
LANES = {"8schools": [1, 2, 4, 8], "radon_MN": [4, 8, 16], "radon_PA": [4, 8, 16], "election": [4, 8, 16],
         "german": [4, 8, 16], "radon_sd_MN": [8, 16], "funnel": [1], "electric": [8, 16], "time_series": [4, 8, 16]}
worst = {}
for mname, lanes_list in LANES.items():
    sp = helpers.spec(mname); orc = oracle.OracleModel(sp); eng = engine.Engine(sp, "cuda:0")
    for seed in range(4):
        for kind in ("CP", "NCP", "VIP"):
            a, b = helpers.params(sp, kind, seed=seed)
            eng.set_param(0, (a, b))
            for scale in (0.1, 1.0):
                x = helpers.states(sp, 37, seed=100 + seed, scale=scale)
                lo, go = orc.logp_grad(x.astype(np.float64), a, b)
                xo = orc.transform(x.astype(np.float64), a, b, True)
                for lanes in lanes_list:
                    lp, g = eng.logp_grad(torch.as_tensor(x, device="cuda:0"), lanes=lanes)
                    xc = eng.transform(torch.as_tensor(x, device="cuda:0"), 0, True) if hasattr(eng, "transform") else None
                    el = np.abs(lp.cpu().numpy() - lo) / (np.abs(lo) + 1)
                    eg = np.abs(g.cpu().numpy() - go).max(axis=1) / (np.abs(go).max(axis=1) + 1)
                    key = (mname, lanes)
                    w = worst.setdefault(key, [0.0, 0.0, 0.0])
                    ok = np.isfinite(lo) & np.isfinite(go).all(axis=1)
                    w[0] = max(w[0], float(el[ok].max(initial=0))); w[1] = max(w[1], float(eg[ok].max(initial=0)))
                    if xc is not None:
                        ex = np.abs(xc.cpu().numpy() - xo).max(axis=1) / (np.abs(xo).max(axis=1) + 1)
                        w[2] = max(w[2], float(ex[ok].max(initial=0)))
for k, w in worst.items():
    print("%-12s lanes %2d  max rel err: logp %.2e  grad %.2e  centred %.2e" % (k[0], k[1], w[0], w[1], w[2]))
