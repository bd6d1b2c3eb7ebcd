"""Stability soak: random chain counts (ragged workgroups included), lanes per chain, step counts and
parameterisations through hmc_run / interleaved_run of every model; checks finiteness and that nothing hangs."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import helpers
from autoreparam_amd import engine, _lib
LANES = {"8schools": [1, 2, 4, 8], "radon_MN": [4, 8, 16], "radon_PA": [4, 8, 16], "election": [4, 8, 16],
         "german": [4, 8, 16], "radon_sd_MN": [8, 16], "funnel": [1], "electric": [8, 16], "time_series": [4, 8, 16]}
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60.0
t0 = time.time(); n = 0
engines = {m: engine.Engine(helpers.spec(m), "cuda:0") for m in LANES}
while time.time() - t0 < budget:
    m = list(LANES)[rs.randint(len(LANES))]
    sp = helpers.spec(m); eng = engines[m]
    C = int(rs.choice([1, 3, 17, 63, 64, 65, 257, 1000, 4097, 20000]))
    if m == "german": C = min(C, 4097)
    lanes = int(rs.choice(LANES[m]))
    kind = ["CP", "NCP", "VIP", "B1"][rs.randint(4)]
    eng.set_param(0, helpers.params(sp, kind, seed=rs.randint(100)))
    eng.set_param(1, helpers.params(sp, "NCP"))
    q0 = helpers.states(sp, C, seed=rs.randint(1000), scale=0.05)
    st = engine.ChainState(torch.as_tensor(q0, device="cuda:0"))
    eps = np.full(sp.D, 1e-4 if m in ("time_series", "electric") else 1e-3, np.float32)
    S = int(rs.randint(1, 4))
    tc = int(rs.choice([0, 1, min(C, 5), C]))                   # chains a trace row holds (0 = all)
    tr = torch.zeros(S, tc if 0 < tc < C else C, sp.D, device="cuda:0")
    L, T = int(rs.randint(1, 5)), int(rs.randint(1, 9))
    extra = {}
    if rs.rand() < 0.5:                                         # in-kernel streaming statistics
        extra = dict(stats=torch.zeros(6, C, sp.D, device="cuda:0"), stats_batch=int(rs.randint(1, 4)), n_samples=S,
                     trace_chains=tc)
    elif 0 < tc < C:
        extra = dict(trace_chains=tc)
    if rs.rand() < 0.5:
        eng.hmc_run(st, eps, L, T, seed=int(rs.randint(1 << 30)), adapt_kind=int(rs.randint(3)), n_adapt=3, n_burnin=0,
                    thin=max(1, T // S), trace=tr, lanes=lanes, rec_accept=torch.zeros(C, dtype=torch.int32, device="cuda:0"),
                    **extra)
    else:
        eng.interleaved_run(st, eps, eps, L, L, T, seed=int(rs.randint(1 << 30)), adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=3,
                            n_burnin=0, thin=max(1, T // S), trace=tr, lanes=lanes, **extra)
    torch.cuda.synchronize()
    assert torch.isfinite(st.q).all() and torch.isfinite(tr).all(), (m, C, lanes, kind)
    if "stats" in extra:
        assert torch.isfinite(extra["stats"]).all(), (m, C, lanes, kind)
    n += 1
print("soak ok: %d launches in %.0f s" % (n, time.time() - t0))
