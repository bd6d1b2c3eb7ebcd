import os, sys, torch
sys.path.insert(0, os.getcwd())
from autoreparam_amd import util
dev = torch.device("cuda:0"); C, S, D = 65536, 1000, 71
x = torch.empty(S, C, D, device=dev).normal_()
for i in range(4):
    e = util.effective_sample_size(x).reshape(-1); torch.cuda.synchronize()
    n = C * D; w = ((n + 511) // 512 // 2) * 512
    cyc, ticks = float(e[w]), float(e[w + 1])
    print("sweep of one wave: %.0f shader cycles in %.0f ticks of 100 MHz -> %.3f GHz; %.1f cycles per row" % (cyc, ticks, cyc / ticks * 0.1, cyc / S))
