#!/usr/bin/env python3
"""Run the five BASELINE.json configurations end to end through the reference-shaped CLI
(autoreparam_amd.main) on one GPU, with the sample counts scaled down so the whole script
takes a few minutes; prints one summary line per stage.  Chain counts are the BASELINE ones."""
import json, os, shutil, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from autoreparam_amd import flags as flags_mod
from autoreparam_amd import main as cli

base = tempfile.mkdtemp(prefix="arp_cfg_")

def run(args):
    f = flags_mod.FlagValues()
    t = time.time()
    out = cli.main(args, flags=f)
    return time.time() - t, out

def stage(tag, args):
    dt, _ = run(args)
    print("%-46s %6.1f s   %s" % (tag, dt, " ".join(a for a in args if not a.startswith("--results_dir"))), flush=True)

def summary(d, fname, keys):
    r = json.load(open(os.path.join(d, fname)))
    print("    ", {k: (r[k][-1] if isinstance(r[k], list) and k != "tuning_runs" else r[k]) for k in keys if k in r}, flush=True)

hm = ["--num_samples=500", "--num_burnin_steps=500", "--num_adaptation_steps=400"]
# config 1: 8schools CP, 4 chains, 4 leapfrog steps, 1000 samples
d = os.path.join(base, "c1"); c = ["--model=8schools", "--method=CP", "--results_dir=" + d, "--num_chains=4"]
stage("config1 VI", c + ["--inference=VI"])
stage("config1 HMCtuning L=4", c + ["--inference=HMCtuning", "--num_leapfrog_steps=4", "--num_samples=1000", "--num_burnin_steps=1000", "--num_adaptation_steps=600"])
stage("config1 HMC", c + ["--inference=HMC", "--num_samples=1000", "--num_burnin_steps=1000", "--num_adaptation_steps=600"])
summary(d, "CP_tied.json", ["elbo", "ess_min", "acceptance_rate", "mcmc_time_sec"])
# config 2: radon MN CP, 4096 chains, 4 leapfrog steps
d = os.path.join(base, "c2"); c = ["--model=radon", "--dataset=MN", "--method=CP", "--results_dir=" + d, "--num_chains=4096"]
stage("config2 VI", c + ["--inference=VI"])
stage("config2 HMCtuning L=4", c + ["--inference=HMCtuning", "--num_leapfrog_steps=4"] + hm)
stage("config2 HMC", c + ["--inference=HMC"] + hm)
summary(d, "CP_tied.json", ["elbo", "ess_min", "acceptance_rate", "mcmc_time_sec"])
# config 3: german credit dVIP: VI 3000 steps (cVIP then dVIP), HMC 16384 chains
d = os.path.join(base, "c3"); c = ["--model=german_credit_lognormalcentered", "--results_dir=" + d, "--num_chains=16384"]
stage("config3 VI cVIP (3000 steps x 5 lr)", c + ["--inference=VI", "--method=cVIP"])
stage("config3 VI dVIP (3000 steps x 5 lr)", c + ["--inference=VI", "--method=dVIP"])
stage("config3 HMCtuning dVIP L=4", c + ["--inference=HMCtuning", "--method=dVIP", "--num_leapfrog_steps=4", "--num_samples=150", "--num_burnin_steps=300", "--num_adaptation_steps=250"])
stage("config3 HMC dVIP", c + ["--inference=HMC", "--method=dVIP", "--num_samples=150", "--num_burnin_steps=300", "--num_adaptation_steps=250"])
summary(d, "dVIP_eig_tied.json", ["elbo", "ess_min", "acceptance_rate", "mcmc_time_sec"])
# config 4: radon PA interleaved, 65536 chains
d = os.path.join(base, "c4"); c = ["--model=radon", "--dataset=PA", "--results_dir=" + d, "--num_chains=65536"]
for m in ("CP", "NCP"):
    stage("config4 VI " + m, c + ["--inference=VI", "--method=" + m])
    stage("config4 HMCtuning %s L=4" % m, c + ["--inference=HMCtuning", "--method=" + m, "--num_leapfrog_steps=4", "--num_samples=200", "--num_burnin_steps=300", "--num_adaptation_steps=250"])
stage("config4 HMC i (4+4 leapfrog steps)", c + ["--inference=HMC", "--method=i", "--num_samples=200", "--num_burnin_steps=300", "--num_adaptation_steps=250"])
summary(d, "i_tied.json", ["num_leapfrog_steps", "ess_min", "acceptance_rate_cp", "acceptance_rate_ncp", "mcmc_time_sec"])
# config 5: election cVIP, 131072 chains, tuned L
d = os.path.join(base, "c5"); c = ["--model=election", "--method=cVIP", "--results_dir=" + d, "--num_chains=131072"]
stage("config5 VI cVIP", c + ["--inference=VI"])
for L in (4, 8):
    stage("config5 HMCtuning cVIP L=%d" % L, c + ["--inference=HMCtuning", "--num_leapfrog_steps=%d" % L, "--num_samples=100", "--num_burnin_steps=300", "--num_adaptation_steps=250"])
stage("config5 HMC cVIP (tuned L)", c + ["--inference=HMC", "--num_samples=100", "--num_burnin_steps=300", "--num_adaptation_steps=250"])
summary(d, "cVIP_eig_tied.json", ["elbo", "ess_min", "acceptance_rate", "mcmc_time_sec"])
shutil.rmtree(base, ignore_errors=True)
