#!/usr/bin/env python3
"""GPU box: arp_ess on the trace the reference flow produces at the headline size (VI step sizes, tuned leapfrog count),
timed repeatedly, with the distribution of the lag at which series are cut (float64 oracle on a sample of chains)."""
import json, os, sys, tempfile, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from autoreparam_amd import main as cli, inference, util, models, graphs
from autoreparam_amd.flags import FLAGS
from oracle import ess_ref

tmp = tempfile.mkdtemp(prefix="arp_essdbg_")
base = ["--model=radon", "--dataset=PA", "--results_dir=%s" % tmp, "--seed=1", "--device=cuda:0"]
for m in ("CP", "NCP"):
    cli.main(base + ["--inference=VI", "--method=%s" % m], flags=FLAGS.copy())
cp = json.load(open(os.path.join(tmp, "CP_tied.json"))); ncp = json.load(open(os.path.join(tmp, "NCP_tied.json")))
fl = FLAGS.copy(); fl.parse(base + ["--inference=HMC", "--method=i", "--num_chains=65536", "--num_samples=1000",
                                    "--num_burnin_steps=1000", "--num_adaptation_steps=600"])
mc = models.get_model_by_name("radon", dataset="PA")
tcp, _, _, _, _ = graphs.make_cp_graph(mc, flags=fl)[:5] if hasattr(graphs, "make_cp_graph") else (None,) * 5
names = list(mc.model.part_names)
init = list(util.variational_inits_from_params(cp["learned_variational_params"], param_names=names, num_inits=65536, seed=1).values())
for num_ls in (4, 8):
    fl.num_leapfrog_steps = 2 * num_ls
    target, model, elbo, vp, lp, ar = cli.create_target_graph(mc, tmp, fl)
    states, kr, ess = inference.hmc_interleaved(mc, target[0], target[1], num_ls, num_ls, cp["initial_step_size"],
                                                ncp["initial_step_size"], init, flags=fl)
    tr = torch.cat([s._t.reshape(s._t.shape[0], s._t.shape[1], -1) for s in states], dim=2).contiguous()
    ts = []
    for _ in range(4):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); e = util.effective_sample_size(tr); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    x = tr[:, :64].cpu().numpy().astype(np.float64)
    xc = x - x.mean(axis=0)
    S = x.shape[0]
    cut = np.zeros(x.shape[1:], int)
    for c in range(x.shape[1]):
        for d in range(x.shape[2]):
            y = xc[:, c, d]; c0 = (y * y).sum() / S
            k = 1
            while k < S and (y[:S - k] * y[k:]).sum() / (S - k) / c0 >= 0: k += 1
            cut[c, d] = k
    print("num_ls %d: arp_ess %s ms; mean min-ESS %.1f; cut lag: median %d, 90%% %d, 99%% %d, max %d; series cut past 16: %.3f, past 80: %.3f; per-element max %s" % (
        num_ls, ["%.2f" % t for t in ts], float(e.min(dim=1).values.mean()), np.median(cut), np.percentile(cut, 90),
        np.percentile(cut, 99), cut.max(), (cut > 16).mean(), (cut > 80).mean(), cut.max(axis=0)[:8]), flush=True)
    del states, tr
