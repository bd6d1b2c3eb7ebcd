#!/usr/bin/env python3
"""Where a tile of ess_tile_kernel spends its time: s_memtime stamps of workgroup 0's thread 0 between the kernel's phases
(a library built with -DARP_ESS_PROF: tools/build_variants.sh prof=-DARP_ESS_PROF; ARP_DEBUG=1 ARP_LIB_PATH=...).  Ticks of
the 100 MHz clock, summed over the workgroup's tiles."""
import ctypes as C, os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoreparam_amd import _lib, util
L = _lib.lib()
S, Cn, D = 1000, int(os.environ.get("ESS_C", "65536")), 71
x = torch.randn(S, Cn, D, device="cuda:0")
if os.environ.get("ESS_AR"):
    rho = float(os.environ["ESS_AR"])
    for t in range(1, S):
        x[t] = rho * x[t - 1] + (1 - rho * rho) ** 0.5 * x[t]
util.effective_sample_size(x); torch.cuda.synchronize()
out = (C.c_ulonglong * 16)()
L.arp_debug_ess_prof(out, 1)
util.effective_sample_size(x); torch.cuda.synchronize()
L.arp_debug_ess_prof(out, 0)
names = ["barrier C (tile free)", "tile write", "prefetch 0 + barrier A", "r + ring + block 0", "prefetch 1", "block 1", "prefetch 2", "sums over the slots", "prefetch 3", "centre + cut", "further lags"]
tiles = (Cn * D + 31) // 32 // 256
tot = sum(out[:11])
for k, nm in enumerate(names):
    print("%-22s %8d ticks  %6.2f us/tile  %5.1f %%" % (nm, out[k], out[k] / 100.0 / tiles, 100.0 * out[k] / max(tot, 1)))
print("total %.2f us/tile over %d tiles = %.2f ms" % (tot / 100.0 / tiles, tiles, tot / 1e5))
