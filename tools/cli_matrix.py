#!/usr/bin/env python3
"""Every model x every method of the reference's CLI (CP, NCP, cVIP, dVIP, interleaved), small sizes, end to end through
autoreparam_amd.main on one GPU: VI -> HMCtuning -> HMC (cVIP's VI before dVIP, CP and NCP before `i`, as main.py's
sequencing demands).  One line per run: ESS per 1000 gradients, acceptance rate, wall time."""
import json, os, sys, tempfile, time, io, contextlib
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoreparam_amd import flags as flags_mod
from autoreparam_amd import main as cli

base = tempfile.mkdtemp(prefix="arp_matrix_")
MODELS = [("8schools", None), ("radon", "MN"), ("radon", "PA"), ("german_credit_lognormalcentered", None), ("election", None),
          ("radon_stddvs", "MN"), ("neals_funnel", None), ("electric", None), ("time_series", None)]
only = sys.argv[1:]
hm = ["--num_samples=200", "--num_burnin_steps=300", "--num_adaptation_steps=250", "--num_leapfrog_steps=4"]
bad = 0
for model, ds in MODELS:
    if only and model not in only:
        continue
    d = os.path.join(base, model + (ds or ""))
    c = ["--model=" + model, "--results_dir=" + d, "--num_chains=256", "--seed=1", "--num_optimization_steps=400"]
    if ds:
        c.append("--dataset=" + ds)
    for method in ("CP", "NCP", "cVIP", "dVIP", "i"):
        m = ["--method=" + method]
        try:
            t = time.time()
            with contextlib.redirect_stdout(io.StringIO()):
                if method != "i":
                    cli.main(c + m + ["--inference=VI"], flags=flags_mod.FlagValues())
                    cli.main(c + m + ["--inference=HMCtuning"] + hm, flags=flags_mod.FlagValues())
                cli.main(c + m + ["--inference=HMC"] + hm, flags=flags_mod.FlagValues())
            dt = time.time() - t
            fn = [f for f in os.listdir(d) if f.startswith(method) and f.endswith(".json")][0]
            r = json.load(open(os.path.join(d, fn)))
            acc = r.get("acceptance_rate", r.get("acceptance_rate_cp"))
            ok = all(map(lambda v: v == v and abs(v) < 1e30, [r["ess_min"][-1], acc[-1]]))
            bad += 0 if ok else 1
            print("%-34s %-5s ess_min/1000 grad %9.3f  acceptance %5.1f %%  %5.1f s  %s" % (
                model + (" " + ds if ds else ""), method, r["ess_min"][-1], acc[-1], dt, "" if ok else "NON-FINITE"), flush=True)
        except Exception as e:      # a failing cell is reported, the matrix goes on
            bad += 1
            print("%-34s %-5s FAILED: %r" % (model + (" " + ds if ds else ""), method, e), flush=True)
print("cells failed: %d" % bad)
sys.exit(1 if bad else 0)
