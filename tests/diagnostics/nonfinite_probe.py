"""Past float32 overflow: which gradient entries are finite / +-inf / NaN in the HIP path (arp_logp_grad) and in the float32
oracle -- one scale parameter at e^100 (profiles/r05_nonfinite_classes.txt)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, helpers
import oracle; oracle.build()
from autoreparam_amd import engine
def cls(x):
    return np.where(np.isnan(x), 2, np.where(np.isinf(x), 1, 0))
for mname, kind, idx, val in (("election","NCP",1,100.0),("election","CP",1,100.0),("8schools","CP",1,100.0),("8schools","NCP",1,100.0),("german","NCP",0,95.0),("radon_MN","NCP",0,3e38)):
    sp = helpers.spec(mname); eng = engine.Engine(sp, "cuda:0"); orc = oracle.OracleModel(sp)
    a, b = helpers.params(sp, kind); eng.set_param(0, (a, b))
    rs = np.random.RandomState(0)
    x = (0.13 * rs.randn(64, sp.D)).astype(np.float32); x[:, idx] += val
    for lanes in (0,):
        lp, g = eng.logp_grad(x, lanes=lanes); lp = lp.cpu().numpy(); g = g.cpu().numpy()
        lpo, go = orc.logp_grad(x, a, b, dtype=np.float32)
        d = cls(g) != cls(go)
        print(mname, kind, "logp class equal", (cls(lp) == cls(lpo)).all(), "grad class mismatches", int(d.sum()), "of", d.size,
              "columns", np.nonzero(d.any(0))[0][:12].tolist())
        if d.any():
            r, c = np.argwhere(d)[0]
            print("   e.g. chain", r, "col", c, "hip", g[r, c], "oracle", go[r, c], " hip row classes", np.bincount(cls(g[r]), minlength=3), "oracle", np.bincount(cls(go[r]), minlength=3))
