#!/usr/bin/env python3
"""Which ESS is right for BASELINE config 3 (german credit, dVIP, L = 4)?  Round 3's full-size CLI run printed 8.32 ESS per
1 000 gradients from the short HMCtuning run (S = 100, tfp's autocorrelation estimator on the whole trace) and 0.81 from
the full schedule (S = 50 000, batch means).  This runs the full schedule at a chain count whose WHOLE trace fits
(1 024 chains x 50 000 samples x 125 = 25.6 GB) in streaming mode with --ess_chains >= C, so one run yields both
estimators on the same samples, and then takes the autocorrelation estimator on PREFIXES of the same trace:
the estimator cannot see autocorrelation times longer than the series it is given."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from autoreparam_amd import flags as flags_mod, main as cli, inference, util

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
d = tempfile.mkdtemp(prefix="arp_rec_")
base = ["--model=german_credit_lognormalcentered", "--results_dir=" + d, "--num_chains=%d" % C]
for m in ("cVIP", "dVIP"):
    cli.main(base + ["--inference=VI", "--method=" + m], flags=flags_mod.FlagValues())
L = 4
t0 = time.time()
cli.main(base + ["--inference=HMC", "--method=dVIP", "--num_leapfrog_steps=%d" % L, "--num_samples=%d" % S,
                 "--trace_chunk_rows=8192", "--ess_chains=%d" % C], flags=flags_mod.FlagValues(), out=(sink := {}))
print("full schedule: %.1f s" % (time.time() - t0))
r = json.load(open(os.path.join(d, "dVIP_eig_tied.json")))
print({k: r[k][-1] for k in ("ess_min", "sem_min", "ess_estimator", "ess_chains", "ess_min_batch_means", "sem_min_batch_means",
                             "batch_means_batch", "acceptance_rate")})
info = sink["kernel_results"].ess_info
# the run's kept trace is gone with the call; run the sampler once more at the API level to hold the trace here
from autoreparam_amd import graphs, models
cfg = models.get_model_by_name("german_credit_lognormalcentered", dataset="")
f = flags_mod.FlagValues(); f.parse(base + ["--inference=HMC", "--method=dVIP", "--num_leapfrog_steps=%d" % L, "--num_samples=%d" % S])
target = cli.create_target_graph(cfg, d, f)[0]
init = list(util.variational_inits_from_params(r["learned_variational_params"], param_names=list(cfg.model.part_names),
                                               num_inits=C, seed=f.seed).values())
_, kr, st, ess = inference.hmc(target, cfg, r["initial_step_size"], init, None, flags=f)
assert kr.ess_info.estimator == "autocorrelation"
trace = torch.cat([p._t.reshape(S, C, -1) for p in st], dim=2)       # [S, C, D] on the device
norm = lambda e, s: 1000.0 * e / (s * L)
print("%-10s %-28s %-14s %-12s %s" % ("prefix S'", "ESS/1000 grads (mean min)", "mean min ESS", "arp_ess ms", "frac of series positive at lag 48"))
for sp in (100, 300, 1000, 3000, 10000, 30000, S):
    if sp > S:
        continue
    x = trace[:sp]
    e = util.effective_sample_size(x); torch.cuda.synchronize()
    ev = util.effective_sample_size.last_events
    ms = ev[0].elapsed_time(ev[1])
    mn = torch.nan_to_num(e).min(dim=1).values
    print("%-10d %-28.4f %-14.2f %-12.2f" % (sp, norm(mn.mean().item(), sp), mn.mean().item(), ms), flush=True)
# batch means on the same whole trace, several batch lengths: ESS = S var / (batch var(batch means)), float64
def batch_means_ess(x, batch):
    out = torch.empty(x.shape[1], x.shape[2], dtype=torch.float64, device=x.device)
    nb = x.shape[0] // batch
    for c0 in range(0, x.shape[1], 64):
        y = x[: nb * batch, c0:c0 + 64].double()
        var = y.var(dim=0, unbiased=True)
        vb = y.reshape(nb, batch, *y.shape[1:]).mean(dim=1).var(dim=0, unbiased=True)
        out[c0:c0 + 64] = torch.minimum(nb * batch * var / (batch * vb), torch.full_like(var, float(nb * batch)))
    return out

e_ac = util.effective_sample_size(trace).double()
stuck = torch.isnan(e_ac).any(dim=1)            # a chain that never moved after burn-in: constant series, ESS 0/0 (as in TFP)
print("chains with a constant series (ESS = nan, counted as 0 by get_min_ess): %d of %d -> ids %s" % (
    int(stuck.sum().item()), C, torch.nonzero(stuck).flatten().tolist()))
print("   (they never accept after burn-in: the frozen averaged step is too long where they stand -- the algorithm, in float64 too:"
      " profiles/r05_stuck_chains.txt; ess_min above counts them as 0, the tau / batch-means lines below leave them out)")
e_ac = e_ac[~stuck]
trace = trace[:, ~stuck]
tau_ac = (S / e_ac).mean(dim=0)
print("tau = S / ESS per element (mean over chains), autocorrelation estimator: median %.1f, max %.1f samples" % (
    tau_ac.median().item(), tau_ac.max().item()))
for batch in (64, 256, 1024, 4096):
    e_bm = batch_means_ess(trace, batch)
    mn = e_bm.min(dim=1).values
    ratio = ((S / e_bm).mean(dim=0) / tau_ac).cpu().numpy()
    print("batch means, batch %-5d (%4d batches): ESS/1000 grads %.4f (mean min ESS %.2f); tau_bm / tau_ac per element: min %.3f median %.3f max %.3f"
          % (batch, S // batch, norm(mn.mean().item(), S), mn.mean().item(), ratio.min(), np.median(ratio), ratio.max()), flush=True)
