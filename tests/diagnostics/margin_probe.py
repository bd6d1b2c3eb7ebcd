#!/usr/bin/env python3
"""GPU box: how close to its threshold does a Metropolis test have to sit for the HIP path and the float32 oracle to
decide it differently?  For every model x parameterisation x lanes-per-chain of tests/test_gpu_hmc.py's trajectory
test, prints the chains that branched, the margin |log u - log alpha| of the branching test in float32 ulps of the
energies it compared, and the largest state error of the chains that never branched.  The constants
tests/helpers.py: MARGIN_ABS / MARGIN_ULPS come from this table (profiles/r03_margin_probe.txt)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
import oracle  # noqa: E402
import parity  # noqa: E402
import test_gpu_hmc as T  # noqa: E402
from autoreparam_amd import engine, _lib  # noqa: E402

oracle.build()
gpu = torch.device("cuda:0")
helpers.MARGIN_ABS = 1e30          # classify only: nothing is asserted here
worst = 0.0
print("model kind lanes adapt | branched/of | margin of the branching test: |m|, energies, |m| / (ulp * energies) | max state err of clean chains")
for mname in ["8schools", "radon_MN", "radon_PA", "election", "german", "radon_sd_MN", "funnel", "electric", "time_series"]:
    sp = helpers.spec(mname)
    eng = engine.Engine(sp, gpu)
    orc = oracle.OracleModel(sp)
    for kind in ("CP", "NCP", "VIP", "B1"):
        a, b = helpers.params(sp, kind)
        eng.set_param(0, (a, b))
        for lanes in T.LANES[mname]:
            for adapt, frac, L, n, n_adapt in ((0, 0.05, 4, 12, 0), (1, 0.02, 3, 14, 10), (2, 0.02, 3, 14, 10)):
                if adapt and kind not in ("CP", "NCP"):
                    continue
                if adapt and mname in ("election", "german"):
                    frac = 0.002
                Cn = 96
                q0 = helpers.states(sp, Cn, seed=2, scale=0.1)
                eps0 = T._eps0(oracle, sp, a, b, q0, frac)
                r = parity.hmc_every_step(oracle, eng, orc, (a, b), q0, eps0, L, n, 1e30, "probe", seed=9, chain_offset=1000,
                                          adapt_kind=adapt, n_adapt=n_adapt, lanes=lanes)
                clean, first = r["clean"], r["first"]
                x, xo = r["x"].cpu().numpy(), r["xo"]
                err = np.abs(x - xo).max(axis=2) / r["scale"]
                cerr = err[:, clean].max() if clean.any() else 0.0
                rows = []
                for c in np.where(~clean)[0]:
                    s = int(first[c])
                    m, e = abs(float(r["margin"][s, c])), float(r["escale"][s, c])
                    ratio = m / (helpers.EPS32 * max(e, 1e-30))
                    pre = err[:s, c].max() if s > 0 else 0.0
                    rows.append("c%d@%d |m|=%.2e E=%.2e ulps=%.1f pre=%.1e" % (c, s, m, e, ratio, pre))
                    if adapt != 1:
                        worst = max(worst, ratio if m > 1e-3 else 0.0)
                print("%s %s %d %d | %d/%d | %s | %.2e" % (mname, kind, lanes, adapt, int((~clean).sum()), Cn, "; ".join(rows), cerr), flush=True)
print("largest margin (in ulps of the energies, fixed step / simple adaptation, margins above 1e-3 only):", worst)
