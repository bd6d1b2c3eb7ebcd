"""German credit (BASELINE configs[2] shape: 16 384 chains, dVIP-like NCP) through inference.hmc with the whole
[S, C, D] trace and with the in-kernel streaming statistics: wall time and agreement of the summaries."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from autoreparam_amd import flags as flags_mod, graphs, inference, models, util
S = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
cfg = models.get_model_by_name("german_credit_lognormalcentered")
sp = cfg.model
f = flags_mod.FlagValues()
f.num_chains, f.num_samples, f.num_leapfrog_steps = 16384, S, 4
f.num_burnin_steps, f.num_adaptation_steps = (10000, 6000) if S >= 50000 else (500, 400)   # reference defaults at full size
f.num_chains_to_save = 4
target, *_ = graphs.make_ncp_graph(cfg, flags=f)
rs = np.random.RandomState(0)
init = [0.1 * rs.randn(f.num_chains, *s).astype(np.float32) for s in sp.part_shapes]
step = [0.02, np.full(62, 0.02), np.full(62, 0.02)]
modes = ("stats",) if len(sys.argv) > 2 and sys.argv[2] == "stats" else ("trace", "stats")
for mode in modes:
    ff = f.copy()
    if mode == "stats":
        ff.trace_chunk_rows = max(64, S // 4)
    torch.cuda.synchronize(); t = time.time()
    _, kr, st, ess = inference.hmc(target, cfg, step, init, "NCP", flags=ff)
    torch.cuda.synchronize(); dt = time.time() - t
    acc = 100.0 * np.sum(kr.inner_results.is_accepted) / (S * f.num_chains)
    print("%-5s S=%d: %.2f s  estimator %s  min-ESS %.1f  accept %.1f %%  kept trace %s" % (
        mode, S, dt, kr.ess_info.estimator, util.get_min_ess(ess)[0], acc, st[0].shape), flush=True)
