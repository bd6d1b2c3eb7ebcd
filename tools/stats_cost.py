#!/usr/bin/env python3
"""GPU box: what the in-kernel statistics cost.  Headline radon(PA) interleaved launch and the election plain-HMC launch,
each with no recording / statistics (thin 2, representative batch) -- run once per value of ARP_STATS_LDS (the library
reads it at load): 1 = accumulators in LDS (packed kernels), 0 = the plane-per-sample route."""
import os as _os, sys as _sys
if _os.environ.get('ARP_STATS_LDS') and _os.environ.get("ARP_DEBUG") != "1":
    # the library honours its experiment switches under ARP_DEBUG=1 only: without it this run would silently measure the default
    _sys.exit("tools/stats_cost.py: ARP_STATS_LDS is set but ARP_DEBUG=1 is not -- the library would ignore the switch; set ARP_DEBUG=1")
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoreparam_amd import models, engine, _lib  # noqa: E402

dev = torch.device("cuda:0")
batch = int(os.environ.get("STATS_BATCH", "64"))


def timeit(fn, n=5, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
    for a, b in ev:
        a.record(); fn(); b.record()
    torch.cuda.synchronize()
    return float(np.median([a.elapsed_time(b) for a, b in ev]))


def radon(stats, C=65536, T=256, lanes=0):
    sp = models._spec_radon("PA")
    eng = engine.Engine(sp, dev); eng.set_param(0, "CP"); eng.set_param(1, "NCP")
    g = torch.Generator(device="cpu").manual_seed(1)
    st = engine.ChainState((0.1 * torch.randn(C, sp.D, generator=g)).to(dev))
    e = np.full(sp.D, 0.08, np.float32); e[2] = 0.02
    stt = torch.zeros(6, C, sp.D, device=dev) if stats else None
    kw = dict(stats=stt, stats_batch=batch, n_samples=1 << 30) if stats else {}
    return timeit(lambda: eng.interleaved_run(st, e, e, 4, 4, T, seed=7, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10 ** 9,
                                              n_burnin=0, thin=2, trace_centered=False, lanes=lanes, **kw))


def election(stats, C=131072, T=128, kind="NCP"):
    sp = models._spec_election()
    eng = engine.Engine(sp, dev)
    if kind == "B1":
        eng.set_param(0, (np.full(sp.D, 0.5, np.float32), np.ones(sp.D, np.float32)))
    else:
        eng.set_param(0, kind)
    rs = np.random.RandomState(2)
    st = engine.ChainState(torch.as_tensor((0.05 * rs.randn(C, sp.D)).astype(np.float32), device=dev))
    e = np.full(sp.D, 0.02, np.float32)
    stt = torch.zeros(6, C, sp.D, device=dev) if stats else None
    kw = dict(stats=stt, stats_batch=batch, n_samples=1 << 30) if stats else {}
    return timeit(lambda: eng.hmc_run(st, e, 4, T, seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10 ** 9, n_burnin=0, thin=2,
                                      trace_centered=True, **kw), n=3, warm=1)


print("ARP_STATS_LDS=%s batch=%d" % (os.environ.get("ARP_STATS_LDS", "1"), batch))
for name, fn in (("radon_PA interleaved 65536 x 256", radon), ("radon_PA interleaved 8192 x 256", lambda s: radon(s, C=8192)),
                 ("election NCP 131072 x 128", election), ("election b=1 131072 x 128", lambda s: election(s, kind="B1"))):
    a, b = fn(False), fn(True)
    print("%-36s none %.3f ms   stats %.3f ms   overhead %.1f %%" % (name, a, b, 100 * (b / a - 1)), flush=True)
