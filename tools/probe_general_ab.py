import sys, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/bench.py") else os.getcwd())
import numpy as np, torch
from autoreparam_amd import models, engine, _lib
def run(tag, spec, rep, C, L, T, step, lanes=0):
    eng = engine.Engine(spec, "cuda:0"); eng.set_param(0, rep)
    rs = np.random.RandomState(3)
    st = engine.ChainState(torch.as_tensor((0.1*rs.randn(C, spec.D)).astype(np.float32), device="cuda:0"))
    e = np.full(spec.D, step, np.float32)
    kw = dict(seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10**9, lanes=lanes)
    for _ in range(2): eng.hmc_run(st, e, L, T, **kw)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(3): eng.hmc_run(st, e, L, T, **kw)
    b.record(); torch.cuda.synchronize()
    ms = a.elapsed_time(b)/3
    print("%-40s lanes %2d %8.3f ms  %.3e leapfrog/s acc %.2f" % (tag, lanes, ms, C*T*L/(ms*1e-3), st.accept_count.float().mean().item()/st.step), flush=True)
rs = np.random.RandomState(7)
es = models._spec_election()
ab = (rs.uniform(0.2,0.8,es.D).astype(np.float32), rs.uniform(0.2,0.8,es.D).astype(np.float32))
for lanes in (0, 4, 8, 16):
    run("election general (a,b)", es, ab, 131072, 4, 128, 0.02, lanes)
run("election NCP (packed)", es, "NCP", 131072, 4, 128, 0.02)
b1 = (ab[0], np.ones(es.D, np.float32))
run("election b=1 (packed)", es, b1, 131072, 4, 128, 0.02)
pa = models._spec_radon("PA")
abp = (rs.uniform(0.2,0.8,pa.D).astype(np.float32), rs.uniform(0.2,0.8,pa.D).astype(np.float32))
for lanes in (0, 4, 8, 16):
    run("radon PA general (a,b)", pa, abp, 65536, 8, 256, 0.02, lanes)
run("radon PA CP (packed)", pa, "CP", 65536, 8, 256, 0.02)
