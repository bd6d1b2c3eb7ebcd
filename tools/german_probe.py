"""German credit HMC throughput vs chain count / leapfrog count (lanes = 8)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import model_sweep as ms
from autoreparam_amd import models
sp = models._spec_german()
if len(sys.argv) > 1 and sys.argv[1] == "rows":
    # time vs number of observations (whole tiles of 64 rows): separates the per-tile cost of the
    # likelihood from everything else in a transition
    import copy
    lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    for n in (64, 128, 512, 1000):
        sub = copy.copy(sp)
        sub.raw = dict(sp.raw); sub.raw["X"] = sp.raw["X"][:n]; sub.raw["y"] = sp.raw["y"][:n]
        sub.observed = {"y": sub.raw["y"][None]}
        print("N =", n, end="  ")
        ms.run("german", sub, 16384, 4, lanes, "NCP", T=4, eps=0.005)
    sys.exit(0)
if len(sys.argv) > 1 and sys.argv[1] == "one":
    ms.run("german", sp, 16384, 4, int(sys.argv[2]) if len(sys.argv) > 2 else 8, "NCP", T=4, eps=0.005)
    sys.exit(0)
for C in (8192, 16384, 32768, 65536):
    ms.run("german", sp, C, 4, 8, "NCP", T=4, eps=0.005)
ms.run("german", sp, 16384, 16, 8, "NCP", T=2, eps=0.002)
