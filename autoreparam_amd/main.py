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


def create_target_graph(model_config, results_dir, flags=FLAGS):
    """reference main.py:117-187"""
    cVIP_path = os.path.join(results_dir, "cVIP_{}.json".format(_vip_suffix(flags)))
    actual_reparam = None
    if flags.method == "CP":
        target, model, elbo, vp, lp = graphs.make_cp_graph(model_config, flags=flags)
        actual_reparam = "CP"
    elif flags.method == "NCP":
        target, model, elbo, vp, lp = graphs.make_ncp_graph(model_config, flags=flags)
        actual_reparam = "NCP"
    elif flags.method == "i":
        if flags.inference == "VI":
            raise Exception("Cannot run interleaved VI. Use `i` method with HMC only.")
        target_cp, model_cp, _, _, _ = graphs.make_cp_graph(model_config, flags=flags)
        target_ncp, model_ncp, _, _, _ = graphs.make_ncp_graph(model_config, flags=flags)
        target, model = (target_cp, target_ncp), (model_cp, model_ncp)
        elbo, vp, lp = None, None, None
    elif flags.method == "cVIP":
        if flags.inference == "VI":
            target, model, elbo, vp, lp = graphs.make_cvip_graph(
                model_config, parameterisation_type=flags.learnable_parameterisation_type,
                tied_pparams=flags.tied_pparams, flags=flags)
        else:
            with open(cVIP_path, "r") as f:
                actual_reparam = json.load(f)["learned_reparam"]
            target, model, elbo, vp, lp = graphs.make_dvip_graph(
                model_config, actual_reparam, parameterisation_type=flags.learnable_parameterisation_type,
                flags=flags)
    elif flags.method == "dVIP":
        if os.path.exists(cVIP_path):
            with open(cVIP_path, "r") as f:
                reparam = json.load(f)["learned_reparam"]
        else:
            raise Exception("Run cVIP first to find reparameterisation")
        discrete = collections.OrderedDict(
            [(key, (np.array(reparam[key]) >= 0.5).astype(np.float32)) for key in reparam.keys()])
        print("discrete parameterisation is", discrete)
        target, model, elbo, vp, lp = graphs.make_dvip_graph(
            model_config, discrete, parameterisation_type=flags.learnable_parameterisation_type, flags=flags)
        actual_reparam = discrete
    else:
        raise Exception("unknown method {}".format(flags.method))
    return target, model, elbo, vp, lp, actual_reparam


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


def _ess_report(info, model_config, flags, dev):
    """Build-specific keys next to the reference's `ess_min`: which estimator it is and on how many chains, and -- for a
    streaming run -- the batch-means figure of ALL chains from the in-kernel accumulators.  (A collective when ws > 1:
    every rank calls it.)"""
    if info is None:
        return {}
    out = {"ess_estimator": info.estimator}
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


def run_hmc(model_config, results_dir, file_path, tuning=False, flags=FLAGS):
    """reference main.py:296-398"""
    if os.path.exists(file_path):
        with open(file_path, "r") as f:
            prev_results = json.load(f)
    else:
        raise Exception("Run VI first to find initial step sizes")
    param_names = _param_names(model_config)
    initial_step_size = prev_results["initial_step_size"]
    initial_states = list(util.variational_inits_from_params(
        prev_results["learned_variational_params"], param_names=param_names, num_inits=flags.num_chains,
        seed=flags.seed).values())
    if tuning:
        if not flags.num_leapfrog_steps:
            raise ValueError("You must specify the number of leapfrog steps for a tuning run.")
        for existing_run in prev_results.get("tuning_runs", []):
            if existing_run["num_leapfrog_steps"] == flags.num_leapfrog_steps:
                print("A tuning run already exists for HMC with {} leapfrog steps, skipping. ({})".format(
                    flags.num_leapfrog_steps, existing_run))
                return
    if not flags.num_leapfrog_steps:
        flags.num_leapfrog_steps = get_best_num_leapfrog_steps_from_tuning_runs(prev_results["tuning_runs"])
    util.print_("\nNumber of leaprog steps is set to {}.\n".format(flags.num_leapfrog_steps))
    if flags.count_in_leapfrog_steps:
        flags.num_samples = int(flags.num_samples / float(flags.num_leapfrog_steps))
        flags.num_burnin_steps = int(flags.num_burnin_steps / float(flags.num_leapfrog_steps))
        flags.num_adaptation_steps = int(flags.num_adaptation_steps / float(flags.num_leapfrog_steps))
    target, _, elbo, vp, lp, actual_reparam = create_target_graph(model_config, results_dir, flags)
    # one process per GPU: every rank draws the same initial population and keeps its block of chains
    rank, ws = parallel.world()
    initial_states, chain_offset = parallel.shard_states(initial_states, rank, ws)
    start_time = time.time()
    states_orig, kernel_results, samples, ess_final = inference.hmc(
        target, model_config, initial_step_size, initial_states=initial_states, reparam=actual_reparam, flags=flags,
        chain_offset=chain_offset)
    is_accepted = kernel_results.inner_results.is_accepted
    mcmc_time = time.time() - start_time
    normalized_ess_final = [1000 * e / (flags.num_samples * flags.num_leapfrog_steps) for e in ess_final]
    info = getattr(inference.hmc, "last_ess_info", None)
    dev = flags.device if ws > 1 else None
    n_ess = _ess_chain_count(info, model_config, flags)          # None: a chain subset (streaming run)
    ess_min, sem_min, acceptance_rate, _ = parallel.summarize(
        normalized_ess_final, is_accepted, flags.num_samples, flags.num_chains, device=dev, ess_chains_total=n_ess)
    util.print_("ESS per 1000 gradients: {} +/- {}".format(ess_min, sem_min))
    extra = _ess_report(info, model_config, flags, dev)
    if ws > 1 and not tuning:
        # _ess.npz / _ess.txt hold every chain's per-element ESS: collect the other ranks' blocks (a collective: all ranks)
        normalized_ess_final = parallel.gather_parts(normalized_ess_final, n_ess, flags.device)
    if rank != 0:
        return ess_min, sem_min, acceptance_rate, mcmc_time
    if tuning:
        save_hmc_results(file_path=file_path,
                         tuning_runs={"num_leapfrog_steps": flags.num_leapfrog_steps, "ess_min": float(ess_min),
                                      "sem_min": float(sem_min), "acceptance_rate": float(acceptance_rate),
                                      "mcmc_time": mcmc_time, "num_samples": flags.num_samples,
                                      "num_burnin_steps": flags.num_burnin_steps})   # the reference's keys, no more
    else:
        save_hmc_results(file_path=file_path, ess_min=float(ess_min), sem_min=float(sem_min),
                         acceptance_rate=float(acceptance_rate), mcmc_time_sec=mcmc_time, **extra)
        save_ess(file_path_base=file_path[:-5], samples=samples, param_names=param_names,
                 normalized_ess_final=normalized_ess_final, num_chains_to_save=flags.num_chains_to_save)
    return ess_min, sem_min, acceptance_rate, mcmc_time


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
    info = getattr(inference.hmc_interleaved, "last_ess_info", None)
    n_ess = _ess_chain_count(info, model_config, flags)
    ess_min, sem_min, acc_cp, _ = parallel.summarize(normalized_ess_final, is_accepted_cp, flags.num_samples,
                                                     flags.num_chains, device=dev, ess_chains_total=n_ess)
    extra = _ess_report(info, model_config, flags, dev)
    acc_ncp = float(parallel.all_reduce_sum(float(np.sum(is_accepted_ncp)), dev).item()) * 100.0 / float(
        flags.num_samples * flags.num_chains)
    util.print_("ESS: {} +/- {}".format(ess_min, sem_min))
    if ws > 1:
        normalized_ess_final = parallel.gather_parts(normalized_ess_final, n_ess, flags.device)
    # Only `[:, :num_chains_to_save]` of the samples is ever read again (save_ess).  Taking that slice to the host now
    # releases the [S, C, D] device trace before the next candidate leapfrog count allocates its own (two 18.6 GB traces
    # alive at once at the headline size, and a fresh device allocation of that size can cost half a second).
    k = max(0, int(flags.num_chains_to_save))
    states = [np.asarray(s[:, :k]) for s in states] if k > 0 else [np.zeros((flags.num_samples, 0), np.float32) for _ in states]
    del kernel_results, is_accepted_cp, is_accepted_ncp
    return (ess_min, sem_min, acc_cp, acc_ncp, mcmc_time, states, normalized_ess_final, extra)


def _first_existing(results_dir, names):
    for n in names:
        p = os.path.join(results_dir, n)
        if os.path.exists(p):
            return p
    return None


def run_interleaved_hmc(model_config, results_dir, file_path, flags=FLAGS):
    """reference main.py:452-528 (intended behaviour, see module docstring)"""
    tied = "_tied" if flags.tied_pparams else ""
    file_path_cp = _first_existing(results_dir, ["CP.json", "CP%s.json" % tied])
    file_path_ncp = _first_existing(results_dir, ["NCP.json", "NCP%s.json" % tied])
    param_names = _param_names(model_config)
    if file_path_cp and file_path_ncp:
        with open(file_path_cp, "r") as f:
            prev = json.load(f)
            initial_step_size_cp = prev["initial_step_size"]
            num_leapfrog_steps_cp = get_best_num_leapfrog_steps_from_tuning_runs(prev["tuning_runs"])
            learned_variational_params_cp = prev["learned_variational_params"]
        with open(file_path_ncp, "r") as f:
            prev = json.load(f)
            initial_step_size_ncp = prev["initial_step_size"]
            num_leapfrog_steps_ncp = get_best_num_leapfrog_steps_from_tuning_runs(prev["tuning_runs"])
    else:
        raise Exception("Run VI first to find initial step sizes, and HMC first to find num_leapfrog_steps.")
    initial_states_cp = list(util.variational_inits_from_params(
        learned_variational_params_cp, param_names=param_names, num_inits=flags.num_chains,
        seed=flags.seed).values())
    best_ess_min, best_num_ls, results = 0, None, ()
    for num_ls in sorted(set([num_leapfrog_steps_ncp, num_leapfrog_steps_cp])):
        flags.num_leapfrog_steps = num_ls + num_ls
        util.print_("\nNumber of leaprog steps is set to {}.\n".format(flags.num_leapfrog_steps))
        res = run_interleaved_hmc_with_leapfrog_steps(
            model_config=model_config, results_dir=results_dir, num_leapfrog_steps_cp=num_ls,
            num_leapfrog_steps_ncp=num_ls, initial_step_size_cp=initial_step_size_cp,
            initial_step_size_ncp=initial_step_size_ncp, initial_states_cp=initial_states_cp, flags=flags)
        if float(res[0]) > best_ess_min or best_num_ls is None:
            best_ess_min, best_num_ls, results = float(res[0]), num_ls, res
    ess_min, sem_min, acceptance_rate_cp, acceptance_rate_ncp, mcmc_time, samples, normalized_ess_final, extra = results
    flags.num_leapfrog_steps = best_num_ls + best_num_ls
    if parallel.world()[0] != 0:
        return results
    save_hmc_results(file_path=file_path, initial_step_size_ncp=initial_step_size_ncp,
                     initial_step_size_cp=initial_step_size_cp, num_leapfrog_steps=best_num_ls,
                     ess_min=float(ess_min), sem_min=float(sem_min), acceptance_rate_cp=float(acceptance_rate_cp),
                     acceptance_rate_ncp=float(acceptance_rate_ncp), mcmc_time_sec=mcmc_time, **extra)
    save_ess(file_path_base=file_path[:-5], samples=samples, param_names=param_names,
             normalized_ess_final=normalized_ess_final, num_chains_to_save=flags.num_chains_to_save)
    return results


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


def main(argv=None, flags=FLAGS):
    """reference main.py:190-231"""
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
            return run_hmc(model_config, results_dir, file_path, tuning=False, flags=flags)
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
