"""Command line of the reference (main.py:37-589) on the HIP engine:

    python -m autoreparam_amd.main --model=radon --dataset=MN --inference=VI --method=CP
    python -m autoreparam_amd.main --model=radon --dataset=MN --inference=HMCtuning --method=CP --num_leapfrog_steps=4
    python -m autoreparam_amd.main --model=radon --dataset=MN --inference=HMC --method=CP

Same flags, result directory, file names, JSON keys and run sequencing (VI before
HMC, cVIP before dVIP, tuning runs before an untuned HMC).  Two defects of the
reference's interleaved entry point are not reproduced: the
NameError at main.py:515 and the CP.json / CP_tied.json file-name mismatch (both
spellings are looked up).
"""
import collections
import json
import os
import sys
import time
from collections import OrderedDict

import numpy as np

from . import graphs, inference, models, parallel, util
from .flags import FLAGS


def _vip_suffix(flags):
    return "{}{}{}{}".format(flags.learnable_parameterisation_type,
                             "_tied" if flags.tied_pparams else "",
                             "_reparam_variational" if flags.reparameterise_variational else "",
                             "_discrete_prior" if flags.discrete_prior else "")


TargetBundle = collections.namedtuple("TargetBundle", ["target", "model", "elbo", "variational_parameters",
                                                         "learnable_parameters", "reparam"])


def _stored_cvip_reparam(results_dir, flags, required_for):
    """`learned_reparam` of the cVIP fit this results directory holds (the dVIP / cVIP-HMC steps start from it)."""
    path = os.path.join(results_dir, "cVIP_{}.json".format(_vip_suffix(flags)))
    if not os.path.exists(path):
        raise Exception("Run cVIP first to find reparameterisation" if required_for == "dVIP" else
                        "no cVIP fit at {}: run --inference=VI --method=cVIP first".format(path))
    with open(path, "r") as f:
        return json.load(f)["learned_reparam"]


def _bundle_fixed(kind):
    """CP / NCP: a fixed parameterisation."""
    def build(model_config, results_dir, flags):
        make = graphs.make_cp_graph if kind == "CP" else graphs.make_ncp_graph
        return TargetBundle(*make(model_config, flags=flags), reparam=kind)
    return build


def _bundle_interleaved(model_config, results_dir, flags):
    """`i`: the pair (centred, non-centred) for the interleaved sampler; there is nothing to fit."""
    if flags.inference == "VI":
        raise Exception("Cannot run interleaved VI. Use `i` method with HMC only.")
    cp = graphs.make_cp_graph(model_config, flags=flags)
    ncp = graphs.make_ncp_graph(model_config, flags=flags)
    return TargetBundle((cp[0], ncp[0]), (cp[1], ncp[1]), None, None, None, None)


def _bundle_cvip(model_config, results_dir, flags):
    """cVIP: the learnable parameterisation while fitting; the fitted (continuous) one when sampling."""
    ptype = flags.learnable_parameterisation_type
    if flags.inference == "VI":
        return TargetBundle(*graphs.make_cvip_graph(model_config, parameterisation_type=ptype,
                                                    tied_pparams=flags.tied_pparams, flags=flags), reparam=None)
    fitted = _stored_cvip_reparam(results_dir, flags, "cVIP")
    return TargetBundle(*graphs.make_dvip_graph(model_config, fitted, parameterisation_type=ptype, flags=flags),
                        reparam=fitted)


def _bundle_dvip(model_config, results_dir, flags):
    """dVIP: the cVIP fit rounded to {0, 1} element by element (reference main.py:170-172)."""
    fitted = _stored_cvip_reparam(results_dir, flags, "dVIP")
    rounded = collections.OrderedDict((name, (np.array(value) >= 0.5).astype(np.float32)) for name, value in fitted.items())
    util.print_("dVIP parameterisation (cVIP fit thresholded at 0.5): {}".format(rounded))
    return TargetBundle(*graphs.make_dvip_graph(model_config, rounded,
                                                parameterisation_type=flags.learnable_parameterisation_type, flags=flags),
                        reparam=rounded)


_BUNDLES = {"CP": _bundle_fixed("CP"), "NCP": _bundle_fixed("NCP"), "i": _bundle_interleaved, "cVIP": _bundle_cvip,
            "dVIP": _bundle_dvip}


def create_target_graph(model_config, results_dir, flags=FLAGS):
    """The target (log joint under the method's parameterisation), ELBO and variational parameters of a run -- the
    reference's create_target_graph (main.py:117-187), same 6-tuple; one builder per --method."""
    try:
        build = _BUNDLES[flags.method]
    except KeyError:
        raise Exception("unknown method {}".format(flags.method))
    return tuple(build(model_config, results_dir, flags))


def _clean_dict(d):
    if d is None:
        return None
    return OrderedDict([(k, np.asarray(d[k]).item() if np.ndim(d[k]) == 0 else np.asarray(d[k]).tolist())
                        for k in d.keys()])


def run_vi(model_config, results_dir, file_path, flags=FLAGS):
    """reference main.py:234-290"""
    target, model, elbo, vp, lp, actual_reparam = create_target_graph(model_config, results_dir, flags)
    if os.path.exists(file_path):
        util.print_("Already ran experiment {}-{} on model {} with dataset {}. Skipping".format(
            flags.inference, flags.method, flags.model, flags.dataset))
        return
    prior = None
    if flags.discrete_prior:
        # a mixture of Laplace (not Beta or Kumaraswamy): finite at 0 and 1 (reference main.py:244-253)
        prior = inference.DiscretePrior()
    start_time = time.time()
    (elbo_final, elbo_timeline, learning_rate, initial_step_size, learned_variational_params,
     learned_reparam) = inference.find_best_learning_rate(
         elbo, vp, learnable_parameters_prior=prior, learnable_parameters=lp, flags=flags)
    end_time = time.time()
    if learned_reparam is None and isinstance(actual_reparam, dict):
        learned_reparam = actual_reparam
    results = {
        "elbo": float(elbo_final),
        "variational_fit_time_secs": end_time - start_time,
        "actual_num_variational_steps": len(elbo_timeline),
        "estimated_elbo_std": float(np.std(elbo_timeline[-32:])),
        "learning_rate": learning_rate,
        "initial_step_size": [np.asarray(i).item() if np.ndim(i) == 0 else np.asarray(i).tolist()
                              for i in initial_step_size],
        "learned_reparam": _clean_dict(learned_reparam),
        "learned_variational_params": _clean_dict(learned_variational_params),
    }
    _write_json_atomic(file_path, results)
    return results


def get_best_num_leapfrog_steps_from_tuning_runs(tuning_runs):
    best_run = max(tuning_runs, key=lambda d: d["ess_min"])
    return best_run["num_leapfrog_steps"]


def _param_names(model_config):
    return list(model_config.model.part_names)


def _ess_chain_count(info, model_config, flags):
    """Chains of the whole job the ESS parts cover: all of them, or None for the chain subset of a streaming run
    (inference.EssInfo; the per-rank block lengths are then gathered with the values)."""
    if info is not None and info.batch_means is not None and info.estimator == "autocorrelation":
        return None
    return flags.num_chains


def _ess_report(info, model_config, flags, dev, ess_parts=None):
    """Build-specific keys next to the reference's `ess_min`: which estimator it is and on how many chains, how many of
    them never moved after burn-in, and -- for a streaming run -- the batch-means figure of ALL chains from the in-kernel
    accumulators.  (A collective when ws > 1: every rank calls it.)"""
    if info is None:
        return {}
    n_const = None
    if ess_parts is not None:
        # A chain that accepts nothing after burn-in has a constant recorded series: tfp's estimator gives 0 / 0 = nan and
        # util.get_min_ess counts the chain as 0.  With a step size frozen at the end of adaptation that happens to a few
        # chains per thousand in funnel-shaped posteriors (profiles/r05_stuck_chains.txt: the algorithm, not the
        # arithmetic) -- the summary says so instead of hiding them in the mean.
        parts = [np.asarray(e) for e in ess_parts]
        n = parts[0].shape[0] if parts else 0
        stuck = np.zeros(n, bool)
        for p in parts:                                        # [C, *event] each: one vector pass per part, not one per chain
            if n:                                              # (a rank can hold no ESS chains in a streaming run)
                stuck |= np.isnan(p.reshape(n, int(np.prod(p.shape[1:], dtype=np.int64)))).any(axis=1)
        local = int(stuck.sum())
        n_const = int(parallel.all_reduce_sum(float(local), dev).item())
        if n_const:
            util.print_("    {} chain(s) never moved after burn-in (constant series: ESS nan, counted as 0 by get_min_ess)".format(n_const))
    # every per-run key is a list with one entry per run (the reference's contract): whole-trace runs append None here
    out = {"ess_estimator": info.estimator, "ess_min_batch_means": None, "sem_min_batch_means": None,
           "batch_means_batch": None, "ess_constant_chains": n_const}
    n_local = int(info.chains)
    out["ess_chains"] = int(parallel.all_reduce_sum(float(n_local), dev).item())
    if info.batch_means is not None:
        norm = 1000.0 / (flags.num_samples * flags.num_leapfrog_steps)
        bm = np.nan_to_num(info.batch_means.cpu().numpy()) * norm          # [C_local, D]
        mins = parallel.all_gather_chains(bm.min(axis=1).astype(np.float32), flags.num_chains, dev).cpu().numpy()
        m, sem = parallel.mean_sem(mins)
        out.update(ess_min_batch_means=m, sem_min_batch_means=sem, batch_means_batch=int(info.batch))
        util.print_("    batch-means ESS of all {} chains (batches of {}): {} +/- {}".format(len(mins), info.batch, m, sem))
    return out


def _read_vi_fit(file_path):
    """The JSON a VI run of the same method left behind (step sizes, variational parameters, tuning runs so far)."""
    if not os.path.exists(file_path):
        raise Exception("Run VI first to find initial step sizes")
    with open(file_path, "r") as f:
        return json.load(f)


def _initial_population(fit, model_config, flags):
    """num_chains draws from the fitted mean-field Normal (reference util.py:394-410), one array per latent part."""
    names = _param_names(model_config)
    return list(util.variational_inits_from_params(fit["learned_variational_params"], param_names=names,
                                                   num_inits=flags.num_chains, seed=flags.seed).values())


def _settle_leapfrog_count(fit, tuning, flags):
    """--num_leapfrog_steps as given; a tuning run insists on it (and is skipped when the file already has that count),
    a sampling run without it takes the best tuning run's.  Returns False when there is nothing to do."""
    if tuning:
        if not flags.num_leapfrog_steps:
            raise ValueError("You must specify the number of leapfrog steps for a tuning run.")
        done = [t for t in fit.get("tuning_runs", []) if t["num_leapfrog_steps"] == flags.num_leapfrog_steps]
        if done:
            util.print_("tuning run with {} leapfrog steps is already recorded ({}): skipped".format(
                flags.num_leapfrog_steps, done[0]))
            return False
    elif not flags.num_leapfrog_steps:
        flags.num_leapfrog_steps = get_best_num_leapfrog_steps_from_tuning_runs(fit["tuning_runs"])
    util.print_("\nsampling with {} leapfrog steps per transition\n".format(flags.num_leapfrog_steps))
    if flags.count_in_leapfrog_steps:
        # schedule lengths given in gradient evaluations (reference main.py:331-336)
        for name in ("num_samples", "num_burnin_steps", "num_adaptation_steps"):
            setattr(flags, name, int(getattr(flags, name) / float(flags.num_leapfrog_steps)))
    return True


def run_hmc(model_config, results_dir, file_path, tuning=False, flags=FLAGS, out=None):
    """One HMC run (or one HMCtuning run) of a fitted method: the reference's run_hmc (main.py:296-398) -- same inputs,
    files and keys; chains sharded over the ranks of the job, statistics combined at the end."""
    fit = _read_vi_fit(file_path)
    if not _settle_leapfrog_count(fit, tuning, flags):
        return
    initial_states = _initial_population(fit, model_config, flags)
    target, _, _, _, _, reparam = create_target_graph(model_config, results_dir, flags)
    # one process per GPU: every rank draws the same initial population and keeps its block of chains
    rank, ws = parallel.world()
    initial_states, chain_offset = parallel.shard_states(initial_states, rank, ws)
    clock = time.time()
    _, kernel_results, samples, ess_final = inference.hmc(
        target, model_config, fit["initial_step_size"], initial_states=initial_states, reparam=reparam, flags=flags,
        chain_offset=chain_offset)
    mcmc_time = time.time() - clock
    if out is not None:
        out["kernel_results"] = kernel_results      # .ess_info, .moments: what the run knows beyond the reference's tuple
    per_1000_gradients = 1000.0 / (flags.num_samples * flags.num_leapfrog_steps)
    normalized_ess_final = [e * per_1000_gradients for e in ess_final]
    info = kernel_results.ess_info
    dev = flags.device if ws > 1 else None
    n_ess = _ess_chain_count(info, model_config, flags)          # None: a chain subset (streaming run)
    ess_min, sem_min, acceptance_rate, _ = parallel.summarize(
        normalized_ess_final, kernel_results.inner_results.is_accepted, flags.num_samples, flags.num_chains, device=dev,
        ess_chains_total=n_ess)
    util.print_("ESS per 1000 gradients: {} +/- {}".format(ess_min, sem_min))
    extra = _ess_report(info, model_config, flags, dev, normalized_ess_final)
    if ws > 1 and not tuning:
        # _ess.npz / _ess.txt hold every chain's per-element ESS: collect the other ranks' blocks (a collective: all ranks)
        normalized_ess_final = parallel.gather_parts(normalized_ess_final, n_ess, flags.device)
        if flags.num_chains_to_save > 0:
            # _traces.npz holds the JOB's first chains, whichever ranks own them (rank 0's block may be shorter)
            samples = parallel.gather_leading_chains(samples, flags.num_chains_to_save, chain_offset, flags.device)
    summary = (ess_min, sem_min, acceptance_rate, mcmc_time)
    if rank != 0:
        return summary
    if tuning:
        # the reference's keys of a tuning_runs entry, no more
        save_hmc_results(file_path=file_path, tuning_runs=dict(
            num_leapfrog_steps=flags.num_leapfrog_steps, ess_min=float(ess_min), sem_min=float(sem_min),
            acceptance_rate=float(acceptance_rate), mcmc_time=mcmc_time, num_samples=flags.num_samples,
            num_burnin_steps=flags.num_burnin_steps))
        return summary
    save_hmc_results(file_path=file_path, ess_min=float(ess_min), sem_min=float(sem_min),
                     acceptance_rate=float(acceptance_rate), mcmc_time_sec=mcmc_time, **extra)
    save_ess(file_path_base=file_path[:-5], samples=samples, param_names=_param_names(model_config),
             normalized_ess_final=normalized_ess_final, num_chains_to_save=flags.num_chains_to_save)
    return summary


def run_interleaved_hmc_with_leapfrog_steps(model_config, results_dir, num_leapfrog_steps_cp,
                                            num_leapfrog_steps_ncp, initial_step_size_cp, initial_step_size_ncp,
                                            initial_states_cp, flags=FLAGS):
    """reference main.py:401-449"""
    target, model, elbo, vp, lp, actual_reparam = create_target_graph(model_config, results_dir, flags)
    target_cp, target_ncp = target
    rank, ws = parallel.world()
    initial_states_cp, chain_offset = parallel.shard_states(list(initial_states_cp), rank, ws)
    start_time = time.time()
    states, kernel_results, ess_final = inference.hmc_interleaved(
        model_config, target_cp, target_ncp, num_leapfrog_steps_cp=num_leapfrog_steps_cp,
        num_leapfrog_steps_ncp=num_leapfrog_steps_ncp, step_size_cp=initial_step_size_cp,
        step_size_ncp=initial_step_size_ncp, initial_states_cp=initial_states_cp, flags=flags,
        chain_offset=chain_offset)
    mcmc_time = time.time() - start_time
    is_accepted_cp = kernel_results.cp_results.inner_results.is_accepted
    is_accepted_ncp = kernel_results.ncp_results.inner_results.is_accepted
    normalized_ess_final = [1000 * e / (flags.num_samples * flags.num_leapfrog_steps) for e in ess_final]
    dev = flags.device if ws > 1 else None
    info = kernel_results.ess_info
    n_ess = _ess_chain_count(info, model_config, flags)
    ess_min, sem_min, acc_cp, _ = parallel.summarize(normalized_ess_final, is_accepted_cp, flags.num_samples,
                                                     flags.num_chains, device=dev, ess_chains_total=n_ess)
    extra = _ess_report(info, model_config, flags, dev, normalized_ess_final)
    acc_ncp = float(parallel.all_reduce_sum(float(np.sum(is_accepted_ncp)), dev).item()) * 100.0 / float(
        flags.num_samples * flags.num_chains)
    util.print_("ESS: {} +/- {}".format(ess_min, sem_min))
    if ws > 1:
        normalized_ess_final = parallel.gather_parts(normalized_ess_final, n_ess, flags.device)
    # Only `[:, :num_chains_to_save]` of the samples is ever read again (save_ess).  Taking that slice to the host now
    # releases the [S, C, D] device trace before the next candidate leapfrog count allocates its own (two 18.6 GB traces
    # alive at once at the headline size, and a fresh device allocation of that size can cost half a second).
    k = max(0, int(flags.num_chains_to_save))
    if k > 0:
        states = parallel.gather_leading_chains(states, k, chain_offset, flags.device if ws > 1 else None)
    else:
        states = [np.zeros((flags.num_samples, 0), np.float32) for _ in states]
    del kernel_results, is_accepted_cp, is_accepted_ncp
    return (ess_min, sem_min, acc_cp, acc_ncp, mcmc_time, states, normalized_ess_final, extra)


def _first_existing(results_dir, names):
    for n in names:
        p = os.path.join(results_dir, n)
        if os.path.exists(p):
            return p
    return None


def _tuned_fit(results_dir, names, what):
    """(step sizes, best tuned leapfrog count, variational parameters) of a CP / NCP fit this directory holds; the
    reference writes `CP_tied.json` and looks for `CP.json`: both spellings are accepted."""
    path = _first_existing(results_dir, names)
    if path is None:
        raise Exception("Run VI first to find initial step sizes, and HMC first to find num_leapfrog_steps.")
    with open(path, "r") as f:
        fit = json.load(f)
    if not fit.get("tuning_runs"):
        raise Exception("no HMCtuning run recorded for {}: run --inference=HMCtuning --method={} first".format(path, what))
    return fit["initial_step_size"], get_best_num_leapfrog_steps_from_tuning_runs(fit["tuning_runs"]), \
        fit["learned_variational_params"]


def run_interleaved_hmc(model_config, results_dir, file_path, flags=FLAGS):
    """--method=i: the interleaved CP / NCP sampler started from the centred fit, once per tuned leapfrog count of the two
    fits, keeping the run with the larger ESS (reference main.py:452-528, intended behaviour: module docstring)."""
    tied = "_tied" if flags.tied_pparams else ""
    step_cp, ls_cp, vparams_cp = _tuned_fit(results_dir, ["CP.json", "CP%s.json" % tied], "CP")
    step_ncp, ls_ncp, _ = _tuned_fit(results_dir, ["NCP.json", "NCP%s.json" % tied], "NCP")
    initial_states_cp = _initial_population({"learned_variational_params": vparams_cp}, model_config, flags)
    kept, kept_ls = None, None
    for num_ls in sorted({ls_ncp, ls_cp}):
        flags.num_leapfrog_steps = 2 * num_ls          # what the ESS normalisation counts (reference main.py:493)
        util.print_("\ninterleaved sampling with {0} + {0} leapfrog steps per step\n".format(num_ls))
        run = run_interleaved_hmc_with_leapfrog_steps(
            model_config=model_config, results_dir=results_dir, num_leapfrog_steps_cp=num_ls,
            num_leapfrog_steps_ncp=num_ls, initial_step_size_cp=step_cp, initial_step_size_ncp=step_ncp,
            initial_states_cp=initial_states_cp, flags=flags)
        if kept is None or float(run[0]) > float(kept[0]):
            kept, kept_ls = run, num_ls
    ess_min, sem_min, acceptance_rate_cp, acceptance_rate_ncp, mcmc_time, samples, normalized_ess_final, extra = kept
    flags.num_leapfrog_steps = 2 * kept_ls
    if parallel.world()[0] != 0:
        return kept
    save_hmc_results(file_path=file_path, initial_step_size_ncp=step_ncp, initial_step_size_cp=step_cp,
                     num_leapfrog_steps=kept_ls, ess_min=float(ess_min), sem_min=float(sem_min),
                     acceptance_rate_cp=float(acceptance_rate_cp), acceptance_rate_ncp=float(acceptance_rate_ncp),
                     mcmc_time_sec=mcmc_time, **extra)
    save_ess(file_path_base=file_path[:-5], samples=samples, param_names=_param_names(model_config),
             normalized_ess_final=normalized_ess_final, num_chains_to_save=flags.num_chains_to_save)
    return kept


def save_hmc_results(file_path, **record):
    """Append one run to the result file: every key of the JSON object is a list with one entry per run (the file contract
    of reference main.py:531-550 -- analyze.py and a later cVIP -> dVIP step read these lists)."""
    history = {}
    if os.path.exists(file_path):
        with open(file_path) as f:
            history = json.load(f)
    for key, value in record.items():
        history.setdefault(key, []).append(value)
    _write_json_atomic(file_path, history)


def _write_json_atomic(file_path, obj):
    """A reader (another rank at the top of its next phase, a later run) sees the old file or the new one, never a
    half-written one: write next to it, then rename over it."""
    tmp = "%s.tmp.%d" % (file_path, os.getpid())
    with open(tmp, "w") as f:
        json.dump(obj, f)
        f.flush()
        os.fsync(f.fileno())
    os.replace(tmp, file_path)


def save_ess(file_path_base, samples, normalized_ess_final, param_names, num_chains_to_save=0):
    """The side files of a sampling run (file contract of reference main.py:552-585): `<base>_ess.npz` with one [C, *event]
    array of normalised ESS per latent part, `<base>_ess.txt` with the same arrays printed, then their mean and standard
    deviation over chains, and -- with --num_chains_to_save -- `<base>_traces.npz` with the first chains' samples."""
    ess_by_part = {name: np.asarray(e) for name, e in zip(param_names, normalized_ess_final)}
    np.savez(file_path_base + "_ess.npz", **ess_by_part)
    lines = ["{}: {}\n\n".format(name, e) for name, e in ess_by_part.items()]
    lines.append("\n\n")
    for name, e in ess_by_part.items():
        lines.append("{} mean: {}\n".format(name, e.mean(axis=0)))
        lines.append("{} stddev: {}\n\n".format(name, e.std(axis=0)))
    with open(file_path_base + "_ess.txt", "w") as f:
        f.writelines(lines)
    if num_chains_to_save > 0:
        np.savez(file_path_base + "_traces.npz",
                 **{name: x[:, :num_chains_to_save] for name, x in zip(param_names, samples)})


def main(argv=None, flags=FLAGS, out=None):
    """reference main.py:190-231.  `out`: an optional dict a plain HMC run leaves its kernel results in (tests, tools)."""
    if argv is not None:
        flags.parse(list(argv))
    if flags.reparameterise_variational:
        # util.make_variational_model_special (reference util.py:334-391) is outside the hot path this build covers;
        # accepting the flag silently would write a *_reparam_variational*.json that holds ordinary mean-field results
        raise NotImplementedError("--reparameterise_variational is not supported by this build (SURVEY.md section 2)")
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws > 1:
        # launched by torch.distributed.run: one rank per GPU, RCCL for the statistics exchange
        import torch
        import torch.distributed as dist
        local = int(os.environ.get("LOCAL_RANK", "0"))
        if util.debug_switch("ARP_SHARE_GPU") and torch.cuda.device_count():
            local %= torch.cuda.device_count()     # tests only: several ranks on one GPU (with the gloo backend below)
        flags.device = "cuda:%d" % local
        if torch.cuda.is_available():          # (the engine itself fails loudly without a GPU)
            torch.cuda.set_device(local)
        if flags.inference == "VI":
            # VI is a one-workgroup-per-learning-rate job with nothing to exchange: rank 0 runs it and writes the
            # JSON, the other ranks leave, and NO process group is created -- a communicator whose peers have exited
            # (or a barrier sitting in the RCCL watchdog for as long as the fit takes) is exactly what to avoid.
            if (dist.get_rank() if dist.is_initialized() else int(os.environ.get("RANK", "0"))) != 0:
                return None
        elif not dist.is_initialized():
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            backend = util.debug_switch("ARP_DIST_BACKEND") or "nccl"   # "nccl" is RCCL; "gloo": tests that share one GPU
            if backend != "nccl":
                dist.init_process_group(backend)
            else:
                try:
                    dist.init_process_group("nccl", device_id=torch.device(flags.device))
                except TypeError:                                     # older torch: no device_id keyword
                    dist.init_process_group("nccl")
    util.print_("Loading model {} with dataset {}.".format(flags.model, flags.dataset))
    model_config = models.get_model_by_name(flags.model, dataset=flags.dataset)
    results_dir = flags.results_dir if flags.results_dir != "" else flags.model + "_" + flags.dataset
    if not os.path.exists(results_dir):
        os.makedirs(results_dir)
    filename = "{}{}{}{}{}.json".format(
        flags.method,
        ("_" + flags.learnable_parameterisation_type if "VIP" in flags.method else ""),
        ("_tied" if flags.tied_pparams else ""),
        ("_reparam_variational" if "VIP" in flags.method and flags.reparameterise_variational else ""),
        ("_discrete_prior" if "VIP" in flags.method and flags.discrete_prior else ""))
    file_path = os.path.join(results_dir, filename)
    if flags.inference == "VI":
        return run_vi(model_config, results_dir, file_path, flags)
    if flags.inference not in ("HMC", "HMCtuning"):
        raise Exception("unknown inference {}".format(flags.inference))
    try:
        if flags.inference == "HMC":
            if flags.method == "i":
                return run_interleaved_hmc(model_config, results_dir, file_path, flags)
            return run_hmc(model_config, results_dir, file_path, tuning=False, flags=flags, out=out)
        return run_hmc(model_config, results_dir, file_path, tuning=True, flags=flags)
    finally:
        # Every rank reads the result file at the top of a sampling phase (step sizes, tuning_runs -> the leapfrog count)
        # and rank 0 rewrites it at the end of one: no rank may enter the next phase before rank 0's write has landed,
        # or the ranks could sample with different leapfrog counts.  (A rank that raised never reaches its peers'
        # barrier; the launcher then tears the job down.)
        if sys.exc_info()[0] is None:
            parallel.barrier()


if __name__ == "__main__":
    main(sys.argv[1:])
