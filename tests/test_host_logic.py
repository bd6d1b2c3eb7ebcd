"""CPU: host-side logic around the engine -- flags, ESS, result-file bookkeeping."""
import json
import os

import numpy as np
import pytest
import torch

from autoreparam_amd import flags as flags_mod
from autoreparam_amd import util
import helpers


def test_flag_defaults_match_reference():
    f = flags_mod.FlagValues()
    # reference main.py:37-113
    assert (f.model, f.inference, f.method) == ("8schools", "VI", "CP")
    assert f.learning_rates == [0.02, 0.05, 0.1, 0.2, 0.4]
    assert (f.num_optimization_steps, f.num_mc_samples) == (3000, 256)
    assert (f.num_samples, f.num_chains, f.num_burnin_steps, f.num_adaptation_steps) == (50000, 100, 10000, 6000)
    assert f.tied_pparams is True and f.num_leapfrog_steps is None and f.learnable_parameterisation_type == "eig"


def test_flag_parsing():
    f = flags_mod.FlagValues()
    rest = f.parse(["--model=radon", "--dataset", "PA", "--notied_pparams", "--learning_rates=0.1,0.2",
                    "--num_chains=7", "--count_in_leapfrog_steps", "pos"])
    assert rest == ["pos"]
    assert (f.model, f.dataset, f.tied_pparams, f.num_chains, f.count_in_leapfrog_steps) == \
        ("radon", "PA", False, 7, True)
    assert [float(v) for v in f.learning_rates] == [0.1, 0.2]
    try:
        f.parse(["--nonsense=1"])
        assert False
    except ValueError:
        pass


def test_ess_of_ar1_and_white_noise():
    """tfp.mcmc.effective_sample_size semantics: ESS/S -> (1-rho)/(1+rho) for AR(1)."""
    torch.manual_seed(0)
    S, rho = 4000, 0.7
    x = torch.randn(S, 200, 2)
    for t in range(1, S):
        x[t] = rho * x[t - 1] + (1 - rho ** 2) ** 0.5 * x[t]
    e = (util.effective_sample_size(x) / S).numpy()
    assert abs(e.mean() - (1 - rho) / (1 + rho)) < 0.01
    w = (util.effective_sample_size(torch.randn(S, 50, 3)) / S).numpy()
    assert abs(w.mean() - 1.0) < 0.05
    # direct O(S^2) restatement on one series
    xs = x[:, 0, 0].double().numpy(); xs = xs - xs.mean()
    ac = np.array([(xs[:S - k] * xs[k:]).sum() / (S - k) for k in range(S)]); ac /= ac[0]
    ac[np.where(ac < 0)[0][0]:] = 0
    direct = S / (-1 + 2 * ((S - np.arange(S)) / S * ac).sum())
    assert abs(direct - util.effective_sample_size(x[:, :1, :1]).item()) < 1e-2 * direct
    # batching over chains does not change the result
    a = util.effective_sample_size(x, max_chains_per_batch=7)
    b = util.effective_sample_size(x, max_chains_per_batch=200)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-3)


def test_get_min_ess():
    ess = [np.array([[3.0, 2.0], [5.0, np.nan]]), np.array([1.5, 4.0])]   # parts [C,2] and [C]
    m, s = util.get_min_ess(ess)
    assert abs(m - np.mean([1.5, 0.0])) < 1e-12       # nan -> 0 (np.nan_to_num), min per chain
    assert abs(s - np.std([1.5, 0.0]) / np.sqrt(2)) < 1e-12


def test_variational_inits_and_step_sizes():
    sp = helpers.spec("8schools")
    params = {"mu_loc": 1.0, "mu_scale": 0.5, "log_tau_loc": -1.0, "log_tau_scale": 0.1,
              "theta_loc": list(np.arange(8.0)), "theta_scale": [0.2] * 8}
    init = util.variational_inits_from_params(params, sp.part_names, 5000, seed=1)
    assert list(init.keys()) == sp.part_names and init["theta"].shape == (5000, 8) and init["mu"].dtype == np.float32
    assert abs(init["mu"].mean() - 1.0) < 0.03 and abs(init["theta"][:, 3].std() - 0.2) < 0.01
    steps = util.get_approximate_step_size(params, num_leapfrog_steps=2)
    assert len(steps) == 3 and abs(steps[0] - 0.125) < 1e-12


def test_ab_from_reparam_fallbacks():
    sp = helpers.spec("radon_MN")
    a, b = sp.ab_from_reparam({"mua_a": 0.3, "b1_a": 1, "b2_a": 1, "m_a": [0.5] * 85,
                               "m_prior_mean": [0.0] * 85})   # spurious key ignored, missing _b -> 1
    assert a[0] == np.float32(0.3) and (a[3:] == 0.5).all() and (b == 1).all()
    a, b = sp.ab_from_reparam("NCP")
    assert (a == 0).all() and (b == 0).all()


def test_save_hmc_results_appends(tmp_path):
    from autoreparam_amd import main as cli
    p = os.path.join(str(tmp_path), "CP_tied.json")
    json.dump({"elbo": 1.0}, open(p, "w"))
    cli.save_hmc_results(file_path=p, tuning_runs={"num_leapfrog_steps": 4, "ess_min": 1.0})
    cli.save_hmc_results(file_path=p, tuning_runs={"num_leapfrog_steps": 8, "ess_min": 3.0})
    cli.save_hmc_results(file_path=p, ess_min=2.0)
    r = json.load(open(p))
    assert r["elbo"] == 1.0 and len(r["tuning_runs"]) == 2 and r["ess_min"] == [2.0]
    assert cli.get_best_num_leapfrog_steps_from_tuning_runs(r["tuning_runs"]) == 8


def test_streaming_stats_match_whole_trace():
    """Chunked moments are exact; batch-means ESS agrees with the FFT estimator on AR(1)."""
    from autoreparam_amd.inference import StreamingStats
    torch.manual_seed(1)
    S, C, D, rho = 6000, 40, 3, 0.6
    x = torch.randn(S, C, D)
    for t in range(1, S):
        x[t] = rho * x[t - 1] + (1 - rho ** 2) ** 0.5 * x[t]
    st = StreamingStats(C, D, batch=150, device="cpu")
    for lo in range(0, S, 700):                 # ragged chunks: batches straddle chunk boundaries
        st.update(x[lo:lo + 700])
    assert st.n == S
    assert torch.allclose(st.mean(), x.double().mean(0), atol=1e-10)
    assert torch.allclose(st.var(), x.double().var(0, unbiased=True), rtol=1e-9)
    e_bm = st.ess().mean().item() / S
    e_fft = (util.effective_sample_size(x) / S).mean().item()
    assert abs(e_bm - (1 - rho) / (1 + rho)) < 0.04 and abs(e_bm - e_fft) < 0.04


def test_analyze_reports(tmp_path):
    from autoreparam_amd import analyze
    d = os.path.join(str(tmp_path), "radon_MN"); os.makedirs(d)
    json.dump({"elbo": -1164.4, "estimated_elbo_std": 0.31, "variational_fit_time_secs": 2.0,
               "tuning_runs": [{"num_leapfrog_steps": 4, "ess_min": 1.0}, {"num_leapfrog_steps": 8, "ess_min": 2.0}],
               "ess_min": [3.5], "sem_min": [0.1], "mcmc_time_sec": [1.5]}, open(os.path.join(d, "CP_tied.json"), "w"))
    json.dump({"elbo": -1170.0, "estimated_elbo_std": 0.5, "learned_reparam": {"m_a": [0.2, 0.9]}},
              open(os.path.join(d, "cVIP_eig_tied.json"), "w"))
    json.dump({"num_leapfrog_steps": [4], "ess_min": [5.0], "sem_min": [0.2], "mcmc_time_sec": [2.0]},
              open(os.path.join(d, "i_tied.json"), "w"))
    res = analyze.load(str(tmp_path), "radon_MN")
    assert analyze.leapfrog_steps(res["CP_tied"]) == 8 and analyze.leapfrog_steps(res["i_tied"]) == 4
    assert any("-1164.4000 +/- 0.31" in l for l in analyze.report_elbos(res))
    assert any("m_a" in l for l in analyze.report_reparams(res))
    lines = analyze.report_ess(res, normalize_times=True)
    assert len(lines) == 2 and "8 leapfrog steps" in lines[0]
    analyze.main(["--results_dir", str(tmp_path), "--elbos", "--ess", "--reparams"])


def test_untied_parameter_shapes_follow_the_reference():
    """--notied_pparams: `<rv>_a` has the shape of the variable's loc, `<rv>_b` of its scale
    (program_transformations.py:486-533): vector variables with a scalar loc / scale share one value."""
    import helpers
    from autoreparam_amd import flags as flags_mod, graphs, models
    sp = helpers.spec("election")
    ag, bg = sp.untied_groups()
    k = sp.part_names.index("a"); lo, hi = sp.offsets[k], sp.offsets[k + 1]
    assert (ag[lo:hi] == lo).all() and (ag[:lo] == np.arange(lo)).all() and (bg == np.arange(sp.D)).all()
    assert sp.untied_shape("a", "a") == () and sp.untied_shape("a", "b") == (51,)
    sp = helpers.spec("electric")
    ag, bg = sp.untied_groups()
    for name in ("mua", "sigma_y", "b"):
        k = sp.part_names.index(name); assert (ag[sp.offsets[k]:sp.offsets[k + 1]] == sp.offsets[k]).all()
    k = sp.part_names.index("a"); assert (bg[sp.offsets[k]:sp.offsets[k + 1]] == sp.offsets[k]).all()
    sp = helpers.spec("radon_MN")
    ag, bg = sp.untied_groups()
    assert (ag == np.arange(sp.D)).all() and (bg == np.arange(sp.D)).all()      # m: loc and scale are both [J]
    # the cVIP graph carries those shapes; the tied graph the broadcast (per element) ones
    cfg = models.get_model_by_name("german_credit_lognormalcentered")
    f = flags_mod.FlagValues()
    _, _, _, _, init = graphs.make_cvip_graph(cfg, tied_pparams=False, flags=f)
    assert np.shape(init["beta_log_scales_a"]) == () and np.shape(init["beta_log_scales_b"]) == (62,)
    _, _, _, _, init = graphs.make_cvip_graph(cfg, tied_pparams=True, flags=f)
    assert np.shape(init["beta_log_scales_a"]) == (62,) and "beta_log_scales_b" not in init
    # a scalar entry broadcasts over its part when the parameterisation is applied
    a, b = cfg.model.ab_from_reparam({"overall_log_scale_a": 0.3, "beta_log_scales_a": np.float32(0.25), "beta_a": np.full(62, 0.5)})
    assert (a[1:63] == 0.25).all() and (b == 1).all()


def test_unsupported_flag_fails_loudly():
    from autoreparam_amd import flags as flags_mod, main as cli
    with pytest.raises(NotImplementedError):
        cli.main(["--model=8schools", "--reparameterise_variational"], flags=flags_mod.FlagValues())


def test_streaming_ess_chain_subset_is_keyed_by_global_chain_id(monkeypatch):
    """inference._ess_subset: which chains of a streaming run keep their whole trace for the autocorrelation ESS -- the
    chains with GLOBAL id < --ess_chains, so the reported figure does not depend on how the job is split over ranks;
    halved until [S, k, D] fits 40 % of the device; 0 (batch means only) when asked for or when not even 64 chains fit."""
    import types
    import torch
    from autoreparam_amd import inference
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda dev: types.SimpleNamespace(total_memory=288 * 2 ** 30))
    S, D = 50000, 125
    # one rank: the first 1 024 chains
    assert inference._ess_subset(S, 16384, D, "cuda:0", 0, 1024) == (1024, 1024)
    # two ranks of 8 192: the subset lies on rank 0 alone
    assert inference._ess_subset(S, 8192, D, "cuda:0", 0, 1024) == (1024, 1024)
    assert inference._ess_subset(S, 8192, D, "cuda:0", 8192, 1024) == (1024, 0)
    # a subset larger than the first rank's block spills into the second
    assert inference._ess_subset(S, 512, D, "cuda:0", 0, 1024) == (1024, 512)
    assert inference._ess_subset(S, 512, D, "cuda:0", 512, 1024) == (1024, 512)
    assert inference._ess_subset(S, 512, D, "cuda:0", 1024, 1024) == (1024, 0)
    # memory: 8 192 chains x 50 000 x 125 x 4 B = 205 GB > 40 % of 288 GiB -> halved to 4 096 (102 GB)
    assert inference._ess_subset(S, 16384, D, "cuda:0", 0, 8192)[0] == 4096
    # --ess_chains=0, or a device on which not even 64 chains fit: batch means only
    assert inference._ess_subset(S, 16384, D, "cuda:0", 0, 0) == (0, 0)
    monkeypatch.setattr(torch.cuda, "get_device_properties", lambda dev: types.SimpleNamespace(total_memory=2 ** 30))
    assert inference._ess_subset(S, 16384, D, "cuda:0", 0, 1024) == (0, 0)


def test_debug_switches_need_arp_debug(monkeypatch, capsys):
    from autoreparam_amd import util
    monkeypatch.setenv("ARP_SHARE_GPU", "1")
    monkeypatch.delenv("ARP_DEBUG", raising=False)
    assert util.debug_switch("ARP_SHARE_GPU") is None and "IGNORED" in capsys.readouterr().err
    monkeypatch.setenv("ARP_DEBUG", "1")
    assert util.debug_switch("ARP_SHARE_GPU") == "1" and "DEBUG SWITCH" in capsys.readouterr().err
    monkeypatch.delenv("ARP_SHARE_GPU")
    assert util.debug_switch("ARP_SHARE_GPU") is None and capsys.readouterr().err == ""


def test_result_files_are_replaced_atomically(tmp_path):
    """save_hmc_results goes through a temporary file + os.replace: a reader never sees a half-written JSON, and no
    temporary file is left behind."""
    import json, os
    from autoreparam_amd import main as cli
    p = str(tmp_path / "r.json")
    cli.save_hmc_results(file_path=p, ess_min=1.0, ess_estimator="autocorrelation")
    cli.save_hmc_results(file_path=p, ess_min=2.0, ess_estimator="autocorrelation")
    assert json.load(open(p)) == {"ess_min": [1.0, 2.0], "ess_estimator": ["autocorrelation", "autocorrelation"]}
    assert os.listdir(str(tmp_path)) == ["r.json"]


def test_leapfrog_count_is_settled_as_the_reference_does(capsys):
    """main.py:315-336: a tuning run insists on --num_leapfrog_steps and is skipped when that count is already recorded; a
    sampling run without it takes the best tuning run's; --count_in_leapfrog_steps divides the three schedule lengths
    by it (truncating)."""
    from autoreparam_amd import main as cli
    runs = [{"num_leapfrog_steps": 2, "ess_min": 1.0}, {"num_leapfrog_steps": 8, "ess_min": 3.5},
            {"num_leapfrog_steps": 4, "ess_min": 2.0}]
    assert cli.get_best_num_leapfrog_steps_from_tuning_runs(runs) == 8
    f = flags_mod.FlagValues(); f.parse(["--inference=HMCtuning"])
    with pytest.raises(ValueError, match="number of leapfrog steps"):
        cli._settle_leapfrog_count({"tuning_runs": runs}, True, f)
    f = flags_mod.FlagValues(); f.parse(["--inference=HMCtuning", "--num_leapfrog_steps=4"])
    assert cli._settle_leapfrog_count({"tuning_runs": runs}, True, f) is False          # already there: nothing to do
    f = flags_mod.FlagValues(); f.parse(["--inference=HMCtuning", "--num_leapfrog_steps=16"])
    assert cli._settle_leapfrog_count({"tuning_runs": runs}, True, f) is True and f.num_leapfrog_steps == 16
    f = flags_mod.FlagValues(); f.parse(["--inference=HMC"])
    assert cli._settle_leapfrog_count({"tuning_runs": runs}, False, f) is True and f.num_leapfrog_steps == 8
    assert (f.num_samples, f.num_burnin_steps, f.num_adaptation_steps) == (50000, 10000, 6000)
    f = flags_mod.FlagValues(); f.parse(["--inference=HMC", "--count_in_leapfrog_steps", "--num_leapfrog_steps=3",
                                         "--num_samples=1000", "--num_burnin_steps=500", "--num_adaptation_steps=100"])
    assert cli._settle_leapfrog_count({"tuning_runs": runs}, False, f) is True
    assert (f.num_samples, f.num_burnin_steps, f.num_adaptation_steps) == (333, 166, 33)
    f = flags_mod.FlagValues(); f.parse(["--inference=HMC"])
    with pytest.raises(KeyError):                                                     # no tuning run, no count: the reference's KeyError
        cli._settle_leapfrog_count({}, False, f)
