#!/usr/bin/env python3
"""Time the mean-field VI kernel: 5 learning rates x 3 000 steps x 256 draws (the reference's defaults,
main.py:88-100) per model, HIP events around arp_vi_run, with the launch geometry the library chose.

    python tools/vi_bench.py [model ...]        (ARP_DEBUG=1 ARP_VI_G=.. / ARP_VI_R=.. to force a geometry;
                                                 VI_LAUNCH=plain|cooperative: arp_model_set_option "vi_launch")
"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers  # noqa: E402
from autoreparam_amd import engine  # noqa: E402

names = sys.argv[1:] or ["german", "election", "radon_PA", "radon_MN", "8schools", "electric", "time_series", "radon_sd_MN", "funnel"]
dev = torch.device("cuda", 0)
lrs = [0.02, 0.05, 0.1, 0.2, 0.4][:int(os.environ.get("VI_BENCH_LRS", "5"))]
n_steps, n_mc = 3000, 256
for name in names:
    sp = helpers.spec(name)
    eng = engine.Engine(sp, dev)
    if os.environ.get("VI_LAUNCH"):
        eng.set_option("vi_launch", os.environ["VI_LAUNCH"])       # plain | cooperative | auto
    for kind, learn in (("NCP", False), ("cVIP", True))[:int(os.environ.get("VI_BENCH_KINDS", "2"))]:
        if learn:
            eng.set_param(0, (np.full(sp.D, 0.5, np.float32), np.ones(sp.D, np.float32)))
        else:
            eng.set_param(0, kind)
        best = None
        for rep in range(3):
            rs = np.random.RandomState(0)
            loc = torch.as_tensor((1e-2 * rs.randn(len(lrs), sp.D)).astype(np.float32), device=dev)
            rho = torch.full((len(lrs), sp.D), -2.0, device=dev)
            w = torch.zeros(len(lrs), sp.D, device=dev) if learn else None
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a.record()
            elbo = eng.vi_run(lrs, loc, rho, n_steps, n_mc, w=w, seed=1)
            b.record()
            torch.cuda.synchronize()
            ms = a.elapsed_time(b)
            best = ms if best is None else min(best, ms)
        g = eng.vi_geometry()
        e = elbo[:, -32:].mean(1).cpu().numpy()
        print("%-12s %-5s %8.2f ms  %6.2f us/step  B=%d G=%d R=%d wgs=%d occ=%d  best elbo %.3f finite=%s" % (
            name, kind, best, 1e3 * best / n_steps, g["threads_per_workgroup"], g["sample_groups"], g["row_parts"],
            g["workgroups_resident"], g["workgroups_per_cu"], float(np.nanmax(e)), bool(np.isfinite(elbo.cpu().numpy()).all())),
            flush=True)
