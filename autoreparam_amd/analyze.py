"""Pretty-printer for the result files main.py writes (reference analyze.py:30-185):

    python -m autoreparam_amd.analyze --results_dir=. --model=radon_MN --elbos --ess --reparams

The reference's method list is stale (`cVIP_exp_tied`, analyze.py:19-26) and it expects a
`num_leapfrog_steps` key that main.py only writes for interleaved runs; here every
`<method>.json` found in the model's directory is reported and the leapfrog count of a
plain HMC run is recovered from its best tuning run (main.py:292-294).
"""
import argparse
import glob
import json
import os

import numpy as np


def load(results_dir, model_name):
    out = {}
    for path in sorted(glob.glob(os.path.join(results_dir, model_name, "*.json"))):
        with open(path) as f:
            out[os.path.basename(path)[:-5]] = json.load(f)
    return out


def leapfrog_steps(res):
    if "num_leapfrog_steps" in res:
        return res["num_leapfrog_steps"][-1]
    runs = res.get("tuning_runs")
    return max(runs, key=lambda d: d["ess_min"])["num_leapfrog_steps"] if runs else None


def report_elbos(results):
    lines = []
    for m, r in results.items():
        if "elbo" in r:
            lines.append("{0:.4f} +/- {1:.2f}   : {2}".format(r["elbo"], r["estimated_elbo_std"], m))
    return lines


def report_reparams(results):
    lines = []
    for m, r in results.items():
        if m.startswith("cVIP") and r.get("learned_reparam"):
            lines.append("   {}".format(m))
            for k, v in r["learned_reparam"].items():
                lines.append("{:>10}: {}".format(k, np.array(v, np.float32)))
    return lines


def report_ess(results, normalize_times=False, num_samples=10000):
    lines = []
    vi_times = {m: r.get("variational_fit_time_secs") for m, r in results.items()}
    base = next((m for m in results if m.startswith("CP")), None)
    for m, r in results.items():
        if "ess_min" not in r:
            continue
        L = leapfrog_steps(r)
        ess, sem, t = r["ess_min"][-1], r["sem_min"][-1], r["mcmc_time_sec"][-1]
        if not normalize_times:
            lines.append("{} +/- {} : {} ({} leapfrog steps)".format(r["ess_min"], r["sem_min"], m, L))
            continue
        per_sample = 2 * L if m.startswith("i") else L
        if m.startswith("i"):
            vi = sum(v for k, v in vi_times.items() if k.startswith(("CP", "NCP")) and v)
        else:
            vi = vi_times.get(m) or 0.0
        grads = per_sample * float(num_samples)
        line = "{} +/- {} in {}s ({}s VI + {}s MCMC): {} ({} leapfrog steps".format(
            ess * grads / 1000.0, sem * grads / 1000.0, vi + t, vi, t, m, L)
        if base and "mcmc_time_sec" in results[base] and vi_times.get(base):
            Lb = leapfrog_steps(results[base])
            rel_vi = (vi / 3000.0) / (vi_times[base] / 3000.0) if vi else float("nan")
            rel_mc = (t / (num_samples * per_sample)) / (results[base]["mcmc_time_sec"][-1] / (num_samples * Lb))
            line += ", {:.2f}x/{:.2f}x CP time per VI/MCMC step".format(rel_vi, rel_mc)
        lines.append(line + ")")
    return lines


def main(argv=None):
    ap = argparse.ArgumentParser()
    for f in ("elbos", "ess", "reparams", "normalize_times"):
        ap.add_argument("--" + f, action="store_true")
    ap.add_argument("--model", default="all")
    ap.add_argument("--results_dir", default="")
    args = ap.parse_args(argv)
    root = args.results_dir or "."
    names = sorted(d for d in os.listdir(root) if os.path.isdir(os.path.join(root, d))) if args.model == "all" \
        else [args.model]
    for name in names:
        results = load(root, name)
        if not results:
            continue
        print(" ******  {}  ****** ".format(name))
        if args.elbos:
            print("\n".join(report_elbos(results)) + "\n")
        if args.reparams:
            print("\n".join(report_reparams(results)) + "\n")
        if args.ess:
            print("\n".join(report_ess(results, args.normalize_times)) + "\n")


if __name__ == "__main__":
    main()
