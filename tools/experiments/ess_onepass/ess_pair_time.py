#!/usr/bin/env python3
"""GPU box: arp_ess (forced one-pass / two-sweep under ARP_DEBUG=1) on a white-noise and an AR(1) trace of the headline shape."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.getcwd())   # run from the repository root
from autoreparam_amd import util, _lib
dev = torch.device("cuda:0")
C, S, D = int(os.environ.get("ESS_C", "65536")), int(os.environ.get("ESS_S", "1000")), 71
def timeit(x, n=5):
    util.effective_sample_size(x); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); e = util.effective_sample_size(x); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts)), e
print("lib:", _lib.LIB_PATH, "ONEPASS =", os.environ.get("ARP_ESS_ONEPASS"))
x = torch.empty(S, C, D, device=dev).normal_()
ms, _ = timeit(x); print("white noise: %.3f ms" % ms, flush=True)
if os.environ.get("ESS_WHITE_ONLY"):
    sys.exit(0)
rho = torch.full((D,), 0.3, device=dev); rho[:3] = 0.75
prev = torch.randn(C, D, device=dev)
for t in range(S):
    prev = rho * prev + torch.sqrt(1 - rho * rho) * torch.randn(C, D, device=dev); x[t] = prev
ms, e = timeit(x); print("AR(1) 0.75 x3 / 0.3: %.3f ms; mean min-ESS %.1f" % (ms, float(e.min(dim=1).values.mean())), flush=True)
