#!/usr/bin/env python3
"""BASELINE configurations 3, 4 and 5 at the reference's full schedule (num_samples = 50 000, burn-in 10 000,
6 000 adaptation steps: 110 000 transitions per chain) through the CLI on one GPU.  The [S, C, D] traces would be
410 TB / 930 TB / 1.4 PB; the runs stream (automatic when the trace does not fit): moments and batch means of every chain inside the kernels, the
reference's autocorrelation ESS on the --ess_chains (1 024) chains that keep their whole trace."""
import json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoreparam_amd import flags as flags_mod, main as cli

base = tempfile.mkdtemp(prefix="arp_full_")

def stage(tag, args):
    t = time.time(); cli.main(args, flags=flags_mod.FlagValues()); dt = time.time() - t
    print("%-40s %7.1f s" % (tag, dt), flush=True)
    return dt

def summary(d, fname, keys):
    r = json.load(open(os.path.join(d, fname)))
    print("    ", {k: (r[k][-1] if isinstance(r[k], list) else r[k]) for k in keys if k in r}, flush=True)

small = ["--num_samples=100", "--num_burnin_steps=300", "--num_adaptation_steps=250"]
d = os.path.join(base, "c3"); c = ["--model=german_credit_lognormalcentered", "--results_dir=" + d, "--num_chains=16384"]
stage("config3 VI cVIP", c + ["--inference=VI", "--method=cVIP"]); stage("config3 VI dVIP", c + ["--inference=VI", "--method=dVIP"])
stage("config3 HMCtuning L=4 (short)", c + ["--inference=HMCtuning", "--method=dVIP", "--num_leapfrog_steps=4"] + small)
dt = stage("config3 HMC dVIP, full schedule", c + ["--inference=HMC", "--method=dVIP"])
print("     %.3g leapfrog-steps/s end to end" % (16384 * 110000 * 4 / dt))
summary(d, "dVIP_eig_tied.json", ["ess_min", "sem_min", "ess_estimator", "ess_chains", "ess_min_batch_means", "batch_means_batch", "acceptance_rate", "mcmc_time_sec"])
d = os.path.join(base, "c4"); c = ["--model=radon", "--dataset=PA", "--results_dir=" + d, "--num_chains=65536"]
for m in ("CP", "NCP"):
    stage("config4 VI " + m, c + ["--inference=VI", "--method=" + m])
    stage("config4 HMCtuning %s L=4 (short)" % m, c + ["--inference=HMCtuning", "--method=" + m, "--num_leapfrog_steps=4"] + small)
dt = stage("config4 HMC i, full schedule", c + ["--inference=HMC", "--method=i"])
print("     %.3g leapfrog-steps/s end to end" % (65536 * 110000 * 8 / dt))
summary(d, "i_tied.json", ["num_leapfrog_steps", "ess_min", "sem_min", "ess_estimator", "ess_chains", "ess_min_batch_means", "batch_means_batch", "acceptance_rate_cp", "acceptance_rate_ncp", "mcmc_time_sec"])
d = os.path.join(base, "c5"); c = ["--model=election", "--method=cVIP", "--results_dir=" + d, "--num_chains=131072"]
stage("config5 VI cVIP", c + ["--inference=VI"])
stage("config5 HMCtuning L=4 (short)", c + ["--inference=HMCtuning", "--num_leapfrog_steps=4"] + small)
dt = stage("config5 HMC cVIP, full schedule", c + ["--inference=HMC"])
print("     %.3g leapfrog-steps/s end to end" % (131072 * 110000 * 4 / dt))
summary(d, "cVIP_eig_tied.json", ["ess_min", "sem_min", "ess_estimator", "ess_chains", "ess_min_batch_means", "batch_means_batch", "acceptance_rate", "mcmc_time_sec"])
