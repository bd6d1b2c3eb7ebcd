"""Where the wall clock of a sampling run goes at the headline size (sampling / arp_ess / host copies)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from autoreparam_amd import models, engine, _lib, util

dev = torch.device("cuda:0")
spec = models._spec_radon("PA")
eng = engine.Engine(spec, dev); eng.set_param(0, "CP"); eng.set_param(1, "NCP")
C, S, B, L = 65536, 1000, 1000, int(sys.argv[1]) if len(sys.argv) > 1 else 8
D = spec.D
q0 = (0.1 * torch.randn(C, D)).to(dev)
e = np.full(D, 0.08 / (L / 4.0) ** 2, np.float32); e[2] = 0.02 / (L / 4.0) ** 2
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    trace = torch.empty(S, C, D, dtype=torch.float32, device=dev)
    a0 = torch.empty(S, C, dtype=torch.uint8, device=dev); a1 = torch.empty(S, C, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    st = engine.ChainState(q0)
    total = 1 + B + 2 * (S - 1); done = 0
    while done < total:
        n = min(4096, total - done)
        eng.interleaved_run(st, e, e, L, L, n, seed=1, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=600, n_burnin=B, thin=2,
                            trace=trace, trace_accept0=a0, trace_accept1=a1, trace_centered=False)
        done += n
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ess = util.effective_sample_size(trace)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    h0 = a0.cpu().numpy().astype(bool); h1 = a1.cpu().numpy().astype(bool); he = ess.cpu().numpy()
    t4 = time.perf_counter()
    print("rep %d: alloc %.1f ms, sampling %.1f ms (%d steps), arp_ess %.1f ms, D2H %.1f ms; mean min-ESS %.1f" % (
        rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), total, 1e3 * (t3 - t2), 1e3 * (t4 - t3), float(np.nanmin(he, axis=1).mean())))
    del trace, a0, a1
