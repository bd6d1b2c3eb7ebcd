#!/usr/bin/env python3
"""Generate tests/golden/density_golden.npz from the float64 autograd restatement
of the reference's model programs (oracle/ed2_ref.py).  The reference itself
cannot run in the build container, so these vectors pin the
analytic kernels against an independent, reference-shaped formulation
(one-hot matmuls + autodiff), not against TFP output."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from oracle import ed2_ref  # noqa: E402

out = {}
for mname in helpers.MODEL_SPECS:
    sp = helpers.spec(mname)
    x = helpers.states(sp, 4, seed=7).astype(np.float64)
    out[mname + "/x"] = x
    for kind in ("CP", "NCP", "VIP"):
        a, b = helpers.params(sp, kind)
        ab = kind if kind != "VIP" else ed2_ref.ab_dict(sp, a, b)
        lp = np.zeros(4); g = np.zeros_like(x); xc = np.zeros_like(x)
        for i in range(4):
            lp[i], g[i] = ed2_ref.log_joint(sp, ab, x[i])
            xc[i] = ed2_ref.convert(sp, ab, x[i], True)
        out["%s/%s/logp" % (mname, kind)] = lp
        out["%s/%s/grad" % (mname, kind)] = g
        out["%s/%s/centred" % (mname, kind)] = xc
        out["%s/%s/a" % (mname, kind)] = a
        out["%s/%s/b" % (mname, kind)] = b
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "density_golden.npz"), **out)
print("wrote", len(out), "arrays")
