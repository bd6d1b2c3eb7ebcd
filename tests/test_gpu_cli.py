"""GPU: the reference's CLI flow end to end on the engine (BASELINE config 1 shape:
8schools, CP, 4 chains, 4 leapfrog steps; plus cVIP -> dVIP and the interleaved run),
checking the files and JSON keys of reference main.py:277-290, 376-391, 512-521."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(args, out=None):
    from autoreparam_amd import flags as flags_mod
    from autoreparam_amd import main as cli
    f = flags_mod.FlagValues()
    return cli.main(args, flags=f, out=out)


def test_config1_eight_schools_cp(gpu, tmp_path):
    d = str(tmp_path)
    common = ["--model=8schools", "--method=CP", "--results_dir=" + d, "--num_chains=4"]
    _run(common + ["--inference=VI", "--num_optimization_steps=600"])
    r = json.load(open(os.path.join(d, "CP_tied.json")))
    for k in ("elbo", "variational_fit_time_secs", "actual_num_variational_steps", "estimated_elbo_std",
              "learning_rate", "initial_step_size", "learned_reparam", "learned_variational_params"):
        assert k in r
    assert r["actual_num_variational_steps"] == 600 and -45 < r["elbo"] < -30
    assert set(r["learned_variational_params"]) == {"mu_loc", "mu_scale", "log_tau_loc", "log_tau_scale",
                                                    "theta_loc", "theta_scale"}
    assert len(r["initial_step_size"]) == 3 and len(r["initial_step_size"][2]) == 8
    # VI is skipped when the file exists
    assert _run(common + ["--inference=VI", "--num_optimization_steps=600"]) is None
    hm = ["--num_samples=1000", "--num_burnin_steps=1000", "--num_adaptation_steps=600"]
    _run(common + ["--inference=HMCtuning", "--num_leapfrog_steps=4"] + hm)
    _run(common + ["--inference=HMCtuning", "--num_leapfrog_steps=8"] + hm)
    _run(common + ["--inference=HMCtuning", "--num_leapfrog_steps=8"] + hm)   # already recorded: skipped
    r = json.load(open(os.path.join(d, "CP_tied.json")))
    assert [t["num_leapfrog_steps"] for t in r["tuning_runs"]] == [4, 8]
    for t in r["tuning_runs"]:
        assert set(t) == {"num_leapfrog_steps", "ess_min", "sem_min", "acceptance_rate", "mcmc_time", "num_samples",
                          "num_burnin_steps"}
        assert 30 < t["acceptance_rate"] <= 100 and t["ess_min"] > 0
    _run(common + ["--inference=HMC", "--num_chains_to_save=2"] + hm)         # L from the tuning runs
    r = json.load(open(os.path.join(d, "CP_tied.json")))
    for k in ("ess_min", "sem_min", "acceptance_rate", "mcmc_time_sec"):
        assert isinstance(r[k], list) and len(r[k]) == 1
    ess = np.load(os.path.join(d, "CP_tied_ess.npz"))
    assert ess["theta"].shape == (4, 8) and ess["mu"].shape == (4,)
    tr = np.load(os.path.join(d, "CP_tied_traces.npz"))
    assert tr["theta"].shape == (1000, 2, 8)
    assert os.path.exists(os.path.join(d, "CP_tied_ess.txt"))
    # posterior sanity (8 schools): E[mu] ~ 4.4
    assert 2.0 < tr["mu"].mean() < 7.0


def test_cvip_then_dvip_and_interleaved(gpu, tmp_path):
    d = str(tmp_path)
    base = ["--model=radon", "--dataset=MN", "--results_dir=" + d, "--num_chains=64", "--num_optimization_steps=400",
            "--learning_rates=0.05,0.1"]
    with pytest.raises(Exception):
        _run(base + ["--inference=VI", "--method=dVIP"])                      # needs cVIP first
    _run(base + ["--inference=VI", "--method=cVIP"])
    r = json.load(open(os.path.join(d, "cVIP_eig_tied.json")))
    assert set(r["learned_reparam"]) == {"mua_a", "b1_a", "b2_a", "m_a"} and len(r["learned_reparam"]["m_a"]) == 85
    _run(base + ["--inference=VI", "--method=dVIP"])
    r = json.load(open(os.path.join(d, "dVIP_eig_tied.json")))
    assert set(np.unique(r["learned_reparam"]["m_a"])) <= {0.0, 1.0}
    hm = ["--num_samples=300", "--num_burnin_steps=300", "--num_adaptation_steps=200"]
    _run(base + ["--inference=HMCtuning", "--method=dVIP", "--num_leapfrog_steps=4"] + hm)
    r = json.load(open(os.path.join(d, "dVIP_eig_tied.json")))
    assert r["tuning_runs"][0]["acceptance_rate"] > 40
    with pytest.raises(Exception):
        _run(base + ["--inference=HMC", "--method=i"] + hm)                   # needs CP and NCP runs first
    for m in ("CP", "NCP"):
        _run(base + ["--inference=VI", "--method=" + m])
        _run(base + ["--inference=HMCtuning", "--method=" + m, "--num_leapfrog_steps=4"] + hm)
    _run(base + ["--inference=HMC", "--method=i"] + hm)
    r = json.load(open(os.path.join(d, "i_tied.json")))
    for k in ("initial_step_size_ncp", "initial_step_size_cp", "num_leapfrog_steps", "ess_min", "sem_min",
              "acceptance_rate_cp", "acceptance_rate_ncp", "mcmc_time_sec"):
        assert k in r and len(r[k]) == 1
    assert r["num_leapfrog_steps"] == [4] and r["acceptance_rate_cp"][0] > 30 and r["acceptance_rate_ncp"][0] > 30


def test_streaming_trace_mode_equals_whole_trace(gpu):
    """inference.hmc in streaming mode (--trace_chunk_rows) draws the same samples as the whole-trace run, and its ESS
    is the SAME estimator -- tfp's autocorrelation ESS, here on the chains with global id < --ess_chains: bitwise the
    whole-trace figures of those chains; the batch-means ESS of all chains rides along, it does not replace it."""
    from autoreparam_amd import flags as flags_mod, graphs, inference, models, util
    cfg = models.get_model_by_name("radon", "MN")
    sp = cfg.model
    f = flags_mod.FlagValues()
    f.num_chains, f.num_samples, f.num_burnin_steps, f.num_adaptation_steps, f.num_leapfrog_steps = 96, 400, 100, 80, 4
    f.num_chains_to_save = 5
    target, *_ = graphs.make_cp_graph(cfg, flags=f)
    rs = np.random.RandomState(0)
    init = [0.1 * rs.randn(96, *s).astype(np.float32) for s in sp.part_shapes]
    step = [0.15] * 3 + [np.full(85, 0.3)]
    _, kr_a, st_a, ess_a = inference.hmc(target, cfg, step, init, "CP", flags=f)
    assert kr_a.ess_info.estimator == "autocorrelation" and kr_a.ess_info.chains == 96 and kr_a.moments is None
    for k_ess in (1024, 70, 0):
        f2 = f.copy(); f2.trace_chunk_rows = 96; f2.ess_chains = k_ess
        so, kr_b, st_b, ess_b = inference.hmc(target, cfg, step, init, "CP", flags=f2)
        info = kr_b.ess_info
        assert so is None and info.batch_means.shape == (96, sp.D) and info.batch == 12
        k = min(k_ess, 96)
        for a, b in zip(st_a, st_b):
            assert b.shape[1] == max(k, 5) and np.array_equal(np.asarray(a[:, :5]), np.asarray(b[:, :5]))   # bitwise: same chains, same streams
        assert np.sum(kr_a.inner_results.is_accepted) == np.sum(kr_b.inner_results.is_accepted)
        if k_ess:
            assert info.estimator == "autocorrelation" and info.chains == k
            for a, b in zip(ess_a, ess_b):
                assert b.shape[0] == k and np.array_equal(np.asarray(a)[:k], np.asarray(b))
        else:   # --ess_chains=0: batch means only, and the estimator's name says so
            assert info.estimator == "batch_means(12)" and ess_b[0].shape[0] == 96


def test_batch_means_agree_with_autocorrelation_where_both_fit(gpu):
    """One streaming run yields both estimators on the same samples (--ess_chains >= C): with batches much longer than
    the autocorrelation time, the integrated autocorrelation time tau = S / ESS per element, averaged over chains, agrees
    between tfp's autocorrelation estimator (arp_ess on the kept trace) and the kernels' batch means to 10 %."""
    import torch
    from autoreparam_amd import flags as flags_mod, graphs, inference, models
    cfg = models.get_model_by_name("radon", "MN")
    sp = cfg.model
    f = flags_mod.FlagValues()
    C, S = 512, 8192
    f.num_chains, f.num_samples, f.num_burnin_steps, f.num_adaptation_steps, f.num_leapfrog_steps = C, S, 600, 500, 4
    f.trace_chunk_rows, f.ess_chains = 2048, C                       # batches of 256 samples: 32 of them
    target, *_ = graphs.make_cp_graph(cfg, flags=f)
    rs = np.random.RandomState(1)
    init = [0.1 * rs.randn(C, *s).astype(np.float32) for s in sp.part_shapes]
    step = [0.15] * 3 + [np.full(85, 0.3)]
    _, kr, st, ess = inference.hmc(target, cfg, step, init, "CP", flags=f)
    info = kr.ess_info
    assert info.estimator == "autocorrelation" and info.chains == C and info.batch == 256
    ac = torch.as_tensor(sp.pack([np.asarray(e) for e in ess])).double()      # [C, D]
    bm = info.batch_means.cpu().double()
    tau_ac, tau_bm = (S / ac).mean(dim=0), (S / bm).mean(dim=0)
    assert tau_ac.max() < 40.0                                       # batches of 256 are long against every series
    ratio = (tau_bm / tau_ac).numpy()
    assert np.abs(ratio - 1.0).max() < 0.10, (ratio.min(), ratio.max())


def test_config3_full_size_german_dvip(gpu, tmp_path):
    """BASELINE configs[2] at its real size through the CLI flow: german_credit_lognormalcentered, cVIP fit ->
    dVIP (thresholded parameterisation) -> HMC with dual averaging on 16 384 chains, the statistics accumulated
    inside the kernels (the [S, C, D] trace of the full schedule would be 410 TB).  Short schedule; the pooled
    posterior means of the centred coordinates against the long float64 oracle run (tests/golden/posterior_golden.npz)."""
    import torch
    from autoreparam_amd import flags as flags_mod, inference
    d = str(tmp_path)
    base = ["--model=german_credit_lognormalcentered", "--results_dir=" + d, "--num_chains=16384", "--seed=2"]
    _run(base + ["--inference=VI", "--method=cVIP", "--num_optimization_steps=1500"])
    _run(base + ["--inference=VI", "--method=dVIP", "--num_optimization_steps=1500"])
    r = json.load(open(os.path.join(d, "dVIP_eig_tied.json")))
    assert set(np.unique(np.concatenate([np.ravel(v) for k, v in r["learned_reparam"].items() if k.endswith("_a")]))) <= {0.0, 1.0}
    S, burn = 400, 1500
    hm = ["--num_samples=%d" % S, "--num_burnin_steps=%d" % burn, "--num_adaptation_steps=1200", "--num_leapfrog_steps=8",
          "--trace_chunk_rows=64", "--num_chains_to_save=4"]
    sink = {}
    res = _run(base + ["--inference=HMC", "--method=dVIP"] + hm, out=sink)
    info = sink["kernel_results"].ess_info
    assert info.estimator == "autocorrelation" and info.chains == 1024 and info.batch_means.shape == (16384, 125)
    ess_min, sem_min, acc, mcmc_time = res
    assert 55 < acc < 95 and ess_min > 0
    mean_c, var_c = sink["kernel_results"].moments             # [C, D] per-chain moments from the kernels' accumulators
    assert mean_c.shape == (16384, 125) and torch.isfinite(mean_c).all()
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "posterior_golden.npz"))
    mean_g, sd_g, mcse_g = gold["german/mean"], gold["german/sd"], gold["german/mcse"]
    cm = mean_c.cpu().numpy()
    mean = cm.mean(axis=0)
    mcse = cm.std(axis=0, ddof=1) / np.sqrt(cm.shape[0])
    z = np.abs(mean - mean_g) / (np.sqrt(mcse ** 2 + mcse_g ** 2) + 0.02 * sd_g)
    assert z.max() < 5.0, (int(z.argmax()), float(z.max()))
    sd = np.sqrt(var_c.cpu().numpy().mean(axis=0) + cm.var(axis=0))
    assert np.abs(sd / sd_g - 1).max() < 0.12
    r = json.load(open(os.path.join(d, "dVIP_eig_tied.json")))
    assert len(r["ess_min"]) == 1 and len(r["mcmc_time_sec"]) == 1
    # the reference's key holds the reference's estimator; the batch-means figure of all chains sits next to it
    assert r["ess_estimator"] == ["autocorrelation"] and r["ess_chains"] == [1024] and r["ess_min_batch_means"][0] > 0
    assert np.load(os.path.join(d, "dVIP_eig_tied_ess.npz"))["beta"].shape == (1024, 62)
    tr = np.load(os.path.join(d, "dVIP_eig_tied_traces.npz"))
    assert tr["beta"].shape == (S, 4, 62)


@pytest.mark.parametrize("ds", ["MA", "AZ"])
def test_small_radon_states_through_the_cli(gpu, oracle_lib, tmp_path, ds):
    """`--model=radon --dataset=MA | AZ` (README.md:22; 13 and 15 counties): VI, one tuning run and a sampling run per method,
    the interleaved sampler included; its pooled posterior means against the closed form (radon is Gaussian)."""
    import helpers
    d = str(tmp_path)
    base = ["--model=radon", "--dataset=" + ds, "--results_dir=" + d, "--num_chains=2048", "--seed=3"]
    short = ["--num_samples=400", "--num_burnin_steps=600", "--num_adaptation_steps=400", "--num_chains_to_save=2048"]
    for m in ("CP", "NCP"):
        _run(base + ["--inference=VI", "--method=" + m, "--num_optimization_steps=900"])
        _run(base + ["--inference=HMCtuning", "--method=" + m, "--num_leapfrog_steps=4"] + short)
    res = _run(base + ["--inference=HMC", "--method=CP"] + short)
    assert res[0] > 0 and 50 < res[2] < 99
    res = _run(base + ["--inference=HMC", "--method=i"] + short)
    assert np.isfinite(res[0]) and res[0] > 0
    r = json.load(open(os.path.join(d, "i_tied.json")))
    assert len(r["ess_min"]) == 1 and r["acceptance_rate_cp"][0] > 40 and r["acceptance_rate_ncp"][0] > 40
    sp = helpers.spec("radon_" + ds)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "CP")
    _, g0 = orc.logp_grad(np.zeros((1, sp.D)), a, b)
    _, gI = orc.logp_grad(np.eye(sp.D), a, b)
    P = -(gI - g0)
    mean = np.linalg.solve(P, g0[0]); sd = np.sqrt(np.diag(np.linalg.inv(P)))
    tr = np.load(os.path.join(d, "i_tied_traces.npz"))
    got = np.concatenate([tr[k].reshape(tr[k].shape[0], tr[k].shape[1], -1) for k in ("mua", "b1", "b2", "m")], axis=2)
    assert got.shape == (400, 2048, sp.D)
    m = got.mean(axis=(0, 1))
    assert np.abs((m - mean) / sd).max() < 0.04, np.abs((m - mean) / sd).max()      # ~ 2 048 x 400 correlated samples


def test_radon_stddvs_small_state_through_the_cli(gpu, tmp_path):
    """`--model=radon_stddvs --dataset=AZ`: the cVIP fit, the thresholded dVIP run and the interleaved sampler."""
    d = str(tmp_path)
    base = ["--model=radon_stddvs", "--dataset=AZ", "--results_dir=" + d, "--num_chains=512", "--seed=5"]
    short = ["--num_samples=200", "--num_burnin_steps=400", "--num_adaptation_steps=300", "--num_leapfrog_steps=4"]
    for m in ("CP", "NCP", "cVIP", "dVIP"):
        _run(base + ["--inference=VI", "--method=" + m, "--num_optimization_steps=600"])
    for m in ("CP", "NCP"):
        _run(base + ["--inference=HMCtuning", "--method=" + m] + short)
    for m in ("dVIP", "i"):
        res = _run(base + ["--inference=HMC", "--method=" + m] + short)
        assert np.isfinite(res[0]) and res[0] > 0


def test_config5_full_size_election_cvip(gpu, tmp_path):
    """BASELINE configs[4] at its real size through the CLI flow: election, cVIP fit (the learned continuous parameterisation) ->
    HMCtuning sweep over the leapfrog count -> HMC with dual averaging on 131 072 chains with the tuned count, statistics
    accumulated inside the kernels.  Short schedule; the pooled posterior means of the centred coordinates against the long
    float64 oracle run (tests/golden/posterior_golden.npz), and the 1-based one-hot quirk's prior-only slot: `a[0]` has
    no observation (models.py:978, 985-988), so its posterior is its prior given (mua, sigma_a)."""
    import torch
    from autoreparam_amd import inference
    d = str(tmp_path)
    base = ["--model=election", "--method=cVIP", "--results_dir=" + d, "--num_chains=131072", "--seed=4"]
    _run(base + ["--inference=VI", "--num_optimization_steps=1500"])
    r = json.load(open(os.path.join(d, "cVIP_eig_tied.json")))
    a_learned = np.concatenate([np.ravel(v) for k, v in r["learned_reparam"].items() if k.endswith("_a")])
    assert ((a_learned > 0) & (a_learned < 1)).all() and np.ptp(a_learned) > 0.05        # continuous, and it moved
    small = ["--num_samples=100", "--num_burnin_steps=600", "--num_adaptation_steps=500"]
    for L in (2, 4, 8):                          # the reference's sweep: one tuning run per count (main.py:315-323)
        _run(base[:3] + ["--num_chains=4096", "--seed=4", "--inference=HMCtuning", "--num_leapfrog_steps=%d" % L] + small)
    r = json.load(open(os.path.join(d, "cVIP_eig_tied.json")))
    tried = sorted(t["num_leapfrog_steps"] for t in r["tuning_runs"])
    assert tried == [2, 4, 8]
    S, burn = 300, 1500
    sink = {}
    res = _run(base + ["--inference=HMC", "--num_samples=%d" % S, "--num_burnin_steps=%d" % burn, "--num_adaptation_steps=1200",
                       "--trace_chunk_rows=64", "--num_chains_to_save=2"], out=sink)
    ess_min, sem_min, acc, mcmc_time = res
    assert 55 < acc < 95 and ess_min > 0
    info = sink["kernel_results"].ess_info
    assert info.estimator == "autocorrelation" and info.chains == 1024 and info.batch_means.shape == (131072, 55)
    mean_c, var_c = sink["kernel_results"].moments
    assert mean_c.shape == (131072, 55) and torch.isfinite(mean_c).all()
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "posterior_golden.npz"))
    mean_g, sd_g, mcse_g = gold["election/mean"], gold["election/sd"], gold["election/mcse"]
    cm = mean_c.cpu().numpy()
    mean = cm.mean(axis=0)
    mcse = cm.std(axis=0, ddof=1) / np.sqrt(cm.shape[0])
    z = np.abs(mean - mean_g) / (np.sqrt(mcse ** 2 + mcse_g ** 2) + 0.02 * sd_g)
    assert z.max() < 5.0, (int(z.argmax()), float(z.max()))
    sd = np.sqrt(var_c.cpu().numpy().mean(axis=0) + cm.var(axis=0))
    assert np.abs(sd / sd_g - 1).max() < 0.12
    # "posterior means within 1 %" where the two runs' Monte-Carlo errors allow the statement
    err = np.sqrt(mcse ** 2 + mcse_g ** 2)
    big = (np.abs(mean_g) > 5 * sd_g) & (4 * err < 0.01 * np.abs(mean_g))
    assert big.sum() >= 1 and (np.abs(mean[big] / mean_g[big] - 1) < 0.01).all()
    r = json.load(open(os.path.join(d, "cVIP_eig_tied.json")))
    assert len(r["ess_min"]) == 1 and r["ess_chains"] == [1024]
    assert np.load(os.path.join(d, "cVIP_eig_tied_ess.npz"))["a"].shape == (1024, 51)
    assert np.load(os.path.join(d, "cVIP_eig_tied_traces.npz"))["a"].shape == (S, 2, 51)


def test_every_model_and_method_runs_through_the_cli():
    """The reference's README model list x its five methods (CP, NCP, cVIP, dVIP, i), each VI -> HMCtuning -> HMC through
    main.py at 256 chains (tools/cli_matrix.py): 45 runs, finite ESS and acceptance rates, the file sequencing the
    reference demands (cVIP's fit before dVIP, CP and NCP before the interleaved sampler)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "cli_matrix.py")], cwd=root, timeout=1500,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    lines = [l for l in r.stdout.splitlines() if "ess_min/1000 grad" in l]
    assert r.returncode == 0 and len(lines) == 45 and "cells failed: 0" in r.stdout, r.stdout[-3000:]


def test_analyze_reports_a_real_results_directory(gpu, tmp_path, capsys):
    """The analyze.py-compatible report over files the engine really wrote: radon MN under all five methods
    through the CLI, then every table of autoreparam_amd.analyze."""
    from autoreparam_amd import analyze
    d = os.path.join(str(tmp_path), "radon_MN")
    base = ["--model=radon", "--dataset=MN", "--results_dir=" + d, "--num_chains=128", "--seed=4", "--num_optimization_steps=400"]
    hm = ["--num_samples=200", "--num_burnin_steps=300", "--num_adaptation_steps=250"]
    for method in ("CP", "NCP", "cVIP", "dVIP"):
        _run(base + ["--method=" + method, "--inference=VI"])
        for L in (2, 4):
            _run(base + ["--method=" + method, "--inference=HMCtuning", "--num_leapfrog_steps=%d" % L] + hm)
        _run(base + ["--method=" + method, "--inference=HMC"] + hm)
    _run(base + ["--method=i", "--inference=HMC"] + hm)
    res = analyze.load(str(tmp_path), "radon_MN")
    assert sorted(res) == ["CP_tied", "NCP_tied", "cVIP_eig_tied", "dVIP_eig_tied", "i_tied"]
    elbos = analyze.report_elbos(res)
    assert len(elbos) == 4 and all("+/-" in l for l in elbos)
    assert any("m_a" in l for l in analyze.report_reparams(res))
    for norm in (False, True):
        lines = analyze.report_ess(res, normalize_times=norm)
        assert len(lines) == 5, lines
    # a plain HMC run's leapfrog count comes from its best tuning run, the interleaved run's from its own key
    assert analyze.leapfrog_steps(res["CP_tied"]) in (2, 4) and analyze.leapfrog_steps(res["i_tied"]) in (4, 8)
    analyze.main(["--results_dir", str(tmp_path), "--model", "radon_MN", "--elbos", "--ess", "--reparams"])
    out = capsys.readouterr().out
    assert "CP_tied" in out and "i_tied" in out
