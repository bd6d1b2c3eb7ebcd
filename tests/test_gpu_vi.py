"""GPU: the single-launch mean-field VI kernel against the float32 oracle
restatement of find_best_learning_rate's optimisation loop (same draws)."""
import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu
VI_LANES = {"8schools": 8, "radon_MN": 16, "election": 16, "german": 4, "radon_sd_MN": 16, "funnel": 1, "electric": 16, "time_series": 16}


@pytest.mark.parametrize("mname", ["8schools", "radon_MN", "election", "german", "radon_sd_MN", "funnel", "electric", "time_series"])
@pytest.mark.parametrize("kind", ["CP", "NCP"])
def test_vi_timeline_matches_oracle(oracle_lib, gpu, mname, kind):
    from autoreparam_amd import engine
    sp = helpers.spec(mname)
    eng = engine.Engine(sp, gpu)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, kind)
    eng.set_param(0, (a, b))
    lrs = [0.02, 0.1]
    rs = np.random.RandomState(0)
    loc0 = (1e-2 * rs.randn(2, sp.D)).astype(np.float32)
    rho0 = np.full((2, sp.D), -2.0, np.float32)
    n_steps, n_mc = 150, 256
    loc = torch.as_tensor(loc0.copy(), device=gpu); rho = torch.as_tensor(rho0.copy(), device=gpu)
    elbo = eng.vi_run(lrs, loc, rho, n_steps, n_mc, seed=5).cpu().numpy()
    lo, ro = loc0.copy(), rho0.copy()
    elbo_o = orc.vi_run(a, b, lrs, lo, ro, None, n_steps, n_mc, seed=5, lanes=VI_LANES[mname])
    assert np.isfinite(elbo).all()
    # same draws, float32 both sides: the first ELBO estimates agree to rounding, later ones drift slowly
    np.testing.assert_allclose(elbo[:, :5], elbo_o[:, :5], rtol=2e-5, atol=2e-2)
    tail = slice(n_steps - 32, n_steps)
    np.testing.assert_allclose(elbo[:, tail].mean(1), elbo_o[:, tail].mean(1), rtol=2e-3, atol=0.5)
    np.testing.assert_allclose(loc.cpu().numpy(), lo, rtol=0, atol=0.05 * (np.abs(lo).max() + 1))
    # the ELBO improved
    assert (elbo[:, -32:].mean(1) > elbo[:, :8].mean(1)).all()


@pytest.mark.parametrize("mname", ["radon_MN", "election", "german"])
def test_vi_fit_is_bitwise_reproducible(gpu, mname):
    """The same fit twice (same seed, same starting point): every ELBO of the timeline and the fitted parameters equal
    bit for bit -- the eight waves of a learning rate's workgroup add their partial gradients in wave order, not in
    arrival order, so a whole `--inference=VI` + `--inference=HMC` flow repeats itself run after run."""
    from autoreparam_amd import engine
    sp = helpers.spec(mname)
    eng = engine.Engine(sp, gpu)
    eng.set_param(0, "NCP")
    rs = np.random.RandomState(0)
    loc0 = (1e-2 * rs.randn(2, sp.D)).astype(np.float32)
    outs = []
    for _ in range(3):
        loc = torch.as_tensor(loc0.copy(), device=gpu); rho = torch.full((2, sp.D), -2.0, device=gpu)
        elbo = eng.vi_run([0.02, 0.1], loc, rho, 200, 256, seed=11)
        outs.append((elbo.cpu().numpy(), loc.cpu().numpy(), rho.cpu().numpy()))
    for o in outs[1:]:
        for x, y in zip(outs[0], o):
            assert np.array_equal(x, y)


@pytest.mark.parametrize("mname", ["radon_PA", "election", "german"])
def test_vi_cooperative_and_plain_launch_give_the_same_fit(gpu, mname):
    """arp_model_set_option("vi_launch"): the same kernel started by hipLaunchCooperativeKernel (the default where the device
    supports it: the runtime guarantees that the groups' workgroups are resident together) and by an ordinary launch (one at
    a time per process, residency by the occupancy arithmetic) -- same geometry, same bits; two threads fitting at once
    under the cooperative launch (no process-wide mutex there) each get the fit they get alone."""
    import threading
    from autoreparam_amd import engine
    sp = helpers.spec(mname)
    rs = np.random.RandomState(0)
    loc0 = (1e-2 * rs.randn(5, sp.D)).astype(np.float32)
    lrs = [0.02, 0.05, 0.1, 0.2, 0.4]

    def fit(eng, seed, stream=None):
        ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream(gpu))
        with ctx:
            loc = torch.as_tensor(loc0.copy(), device=gpu); rho = torch.full((5, sp.D), -2.0, device=gpu)
            elbo = eng.vi_run(lrs, loc, rho, 150, 256, seed=seed)
            out = [t.cpu().numpy() for t in (elbo, loc, rho)]
        return out, eng.vi_geometry()

    res = {}
    for mode in ("plain", "cooperative", "auto"):
        eng = engine.Engine(sp, gpu)
        eng.set_param(0, "NCP")
        eng.set_option("vi_launch", mode)
        res[mode] = fit(eng, 11)
    assert res["plain"][1] == res["cooperative"][1] == res["auto"][1]
    assert res["plain"][1]["sample_groups"] * res["plain"][1]["row_parts"] > 1        # the hand-offs are really in play
    for mode in ("cooperative", "auto"):
        for x, y in zip(res["plain"][0], res[mode][0]):
            assert np.array_equal(x, y, equal_nan=True), mode
    with pytest.raises(RuntimeError, match="vi_launch"):
        eng.set_option("vi_launch", "sometimes")
    # two threads, one handle and one stream each, cooperative launches in flight together
    alone = [fit(engine.Engine(sp, gpu), 20 + k)[0] for k in range(2)]
    got, errs = [None, None], []

    def work(k):
        try:
            e = engine.Engine(sp, gpu)
            got[k] = fit(e, 20 + k, torch.cuda.Stream(device=gpu))[0]
        except Exception as ex:     # noqa: BLE001
            errs.append(repr(ex))
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    for t in ts: t.start()
    for t in ts: t.join(timeout=300)
    assert not any(t.is_alive() for t in ts) and not errs, errs
    for k in range(2):
        for x, y in zip(alone[k], got[k]):
            assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("mname,learn", [("radon_PA", False), ("german", True)])
def test_vi_launch_that_times_out_is_taken_again_from_its_starting_point(gpu, monkeypatch, mname, learn):
    """arp_vi_run keeps the parameters a launch starts from until its in-launch hand-offs are known to have gone through: a
    launch that reports a time-out (here by decree: ARP_VI_FAULT_ATTEMPTS, a test hook) is taken again from them with four
    times the bound, and the fit that comes out is bit for bit the undisturbed one; after three failures the call returns an
    error with a message instead of hanging or trapping."""
    from autoreparam_amd import engine
    sp = helpers.spec(mname)
    eng = engine.Engine(sp, gpu)
    eng.set_param(0, (np.full(sp.D, 0.5, np.float32), np.ones(sp.D, np.float32)) if learn else "NCP")
    rs = np.random.RandomState(0)
    loc0 = (1e-2 * rs.randn(7, sp.D)).astype(np.float32)        # seven learning rates: German credit takes them in two launches
    lrs = [0.01, 0.02, 0.05, 0.1, 0.2, 0.3, 0.4]

    def fit():
        loc = torch.as_tensor(loc0.copy(), device=gpu); rho = torch.full((7, sp.D), -2.0, device=gpu)
        w = torch.zeros(7, sp.D, device=gpu) if learn else None
        elbo = eng.vi_run(lrs, loc, rho, 60, 256, w=w, seed=4)
        return [t.cpu().numpy() for t in (elbo, loc, rho)] + ([w.cpu().numpy()] if learn else [])

    ref = fit()
    assert eng.vi_attempts() == 1
    monkeypatch.setenv("ARP_DEBUG", "1")
    monkeypatch.setenv("ARP_VI_FAULT_ATTEMPTS", "2")
    got = fit()
    assert eng.vi_attempts() == 3
    for x, y in zip(ref, got):
        assert np.array_equal(x, y, equal_nan=True)
    monkeypatch.setenv("ARP_VI_FAULT_ATTEMPTS", "3")
    with pytest.raises(RuntimeError, match="timed out three times"):
        fit()
    monkeypatch.delenv("ARP_VI_FAULT_ATTEMPTS")
    again = fit()
    for x, y in zip(ref, again):
        assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize("mname", ["8schools", "radon_MN"])
def test_cvip_learns_parameterisation(oracle_lib, gpu, mname):
    from autoreparam_amd import engine
    sp = helpers.spec(mname)
    eng = engine.Engine(sp, gpu)
    orc = oracle_lib.OracleModel(sp)
    a = np.full(sp.D, 0.5, np.float32); b = np.ones(sp.D, np.float32)
    eng.set_param(0, (a, b))
    rs = np.random.RandomState(1)
    loc0 = (1e-2 * rs.randn(1, sp.D)).astype(np.float32); rho0 = np.full((1, sp.D), -2.0, np.float32)
    w0 = np.zeros((1, sp.D), np.float32)
    loc, rho, w = (torch.as_tensor(v.copy(), device=gpu) for v in (loc0, rho0, w0))
    elbo = eng.vi_run([0.05], loc, rho, 200, 256, w=w, seed=9).cpu().numpy()
    lo, ro, wo = loc0.copy(), rho0.copy(), w0.copy()
    elbo_o = orc.vi_run(a, b, [0.05], lo, ro, wo, 200, 256, learn_a=True, seed=9, lanes=VI_LANES[mname])
    np.testing.assert_allclose(elbo[:, :5], elbo_o[:, :5], rtol=2e-5, atol=2e-2)
    np.testing.assert_allclose(elbo[:, -32:].mean(1), elbo_o[:, -32:].mean(1), rtol=2e-3, atol=0.5)
    wg = w.cpu().numpy()
    assert np.abs(wg).max() > 0.05                      # the parameterisation moved
    np.testing.assert_allclose(wg, wo, rtol=0, atol=0.1 * (np.abs(wo).max() + 0.5))


def test_discrete_prior_term(oracle_lib, gpu):
    """--discrete_prior (reference main.py:244-253, inference.py:50-54): the mixture's log density on the learnable
    parameters is added to the objective.  Same run with and without it, against the oracle; the prior term is
    analytic, so its finite-difference derivative pins the kernel's formula."""
    from autoreparam_amd import engine, inference
    sp = helpers.spec("8schools")
    eng = engine.Engine(sp, gpu)
    orc = oracle_lib.OracleModel(sp)
    a = np.full(sp.D, 0.5, np.float32); b = np.ones(sp.D, np.float32)
    eng.set_param(0, (a, b))
    rs = np.random.RandomState(2)
    loc0 = (1e-2 * rs.randn(1, sp.D)).astype(np.float32); rho0 = np.full((1, sp.D), -2.0, np.float32)
    out = {}
    for prior in (False, True):
        loc, rho, w = (torch.as_tensor(v.copy(), device=gpu) for v in (loc0, rho0, np.zeros((1, sp.D), np.float32)))
        eng.vi_run([0.05], loc, rho, 300, 256, w=w, seed=4, a_prior=prior)
        lo, ro, wo = loc0.copy(), rho0.copy(), np.zeros((1, sp.D), np.float32)
        orc.vi_run(a, b, [0.05], lo, ro, wo, 300, 256, learn_a=True, seed=4, lanes=VI_LANES["8schools"], a_prior=prior)
        # mu and log_tau have no parent, so their `a` sees no likelihood gradient: under the prior a = 1/2 is an
        # unstable equilibrium there (any rounding tips it to an end) -- compare the theta parameters only
        wg = w.cpu().numpy()[:, 2:]; wo = wo[:, 2:]
        np.testing.assert_allclose(wg, wo, rtol=0, atol=0.1 * (np.abs(wo).max() + 0.5))
        out[prior] = 1.0 / (1.0 + np.exp(-wg))
    # the term is active (near 1/2 the mixture is almost flat -- weight e^5 on the uniform -- so the shift is small)
    assert np.abs(out[True] - out[False]).max() > 1e-4
    # analytic derivative used by the kernel == finite difference of DiscretePrior.log_prob
    p = inference.DiscretePrior()
    x = np.linspace(0.02, 0.98, 49); h = 1e-5
    fd = (p.log_prob(x + h) - p.log_prob(x - h)) / (2 * h)
    l0, l1 = 5 * np.exp(-10 * x), 5 * np.exp(-10 * (1 - x))
    np.testing.assert_allclose(10 * (l1 - l0) / (l0 + np.exp(5.0) + l1), fd, rtol=1e-5, atol=1e-8)


@pytest.mark.parametrize("mname", ["election", "german"])
def test_untied_cvip_shares_scalar_shaped_parameters(oracle_lib, gpu, mname):
    """--notied_pparams: the reference creates `<rv>_a` with the shape of the variable's loc and `<rv>_b` with the
    shape of its scale (program_transformations.py:486-533), so election's a[51] and german's beta_log_scales[62]
    -- vector variables with a scalar loc -- learn ONE shared a.  Same run against the oracle; the shared elements
    stay equal; the kernel's gradient of the shared variable is the sum over the part (checked against ed2_ref's
    autograd with a broadcast a)."""
    from autoreparam_amd import engine
    sp = helpers.spec(mname)
    eng = engine.Engine(sp, gpu)
    orc = oracle_lib.OracleModel(sp)
    ag, bg = sp.untied_groups()
    part = {"election": "a", "german": "beta_log_scales"}[mname]
    k = sp.part_names.index(part); lo_, hi_ = sp.offsets[k], sp.offsets[k + 1]
    assert (ag[lo_:hi_] == lo_).all() and (bg == np.arange(sp.D)).all() and sp.untied_shape(part, "a") == ()
    a = np.full(sp.D, 0.5, np.float32); b = np.full(sp.D, 0.5, np.float32)
    eng.set_param(0, (a, b))
    rs = np.random.RandomState(3)
    loc0 = (1e-2 * rs.randn(1, sp.D)).astype(np.float32); rho0 = np.full((1, sp.D), -2.0, np.float32)
    z = np.zeros((1, sp.D), np.float32)
    loc, rho, w, wb = (torch.as_tensor(v.copy(), device=gpu) for v in (loc0, rho0, z, z))
    n = 120
    elbo = eng.vi_run([0.05], loc, rho, n, 256, w=w, wb=wb, seed=6, a_group=ag, b_group=bg).cpu().numpy()
    lo, ro, wo, wbo = loc0.copy(), rho0.copy(), z.copy(), z.copy()
    elbo_o = orc.vi_run(a, b, [0.05], lo, ro, wo, n, 256, learn_a=True, seed=6, lanes=VI_LANES[mname], wb=wbo,
                        a_group=ag, b_group=bg)
    np.testing.assert_allclose(elbo[:, :5], elbo_o[:, :5], rtol=2e-5, atol=2e-2)
    np.testing.assert_allclose(elbo[:, -32:].mean(1), elbo_o[:, -32:].mean(1), rtol=2e-3, atol=0.5)
    wg = w.cpu().numpy()[0]
    assert np.ptp(wg[lo_:hi_]) == 0.0 and abs(wg[lo_]) > 1e-3          # one shared value, and it moved
    if mname == "election":    # b keeps the scale's (vector) shape; german's beta_log_scales has unit scale, so its b is inert
        assert np.ptp(wb.cpu().numpy()[0][lo_:hi_]) > 0.0
    np.testing.assert_allclose(wg, wo[0], rtol=0, atol=0.1 * (np.abs(wo).max() + 0.5))
    # d logp / d(shared a) == sum over the part of the per-element derivative, against float64 autograd of the
    # Edward2 restatement evaluated with a broadcast scalar a
    import oracle.ed2_ref as ed2
    x = helpers.states(sp, 1, seed=9, scale=0.2).astype(np.float64)[0]
    av = rs.rand(sp.D).astype(np.float32).astype(np.float64); av[lo_:hi_] = av[lo_]
    bv = rs.rand(sp.D).astype(np.float32).astype(np.float64)
    da, _ = orc.dparam(x[None], av.astype(np.float32), bv.astype(np.float32))
    h = 1e-4

    def lj(scalar_a):      # the reference's shapes: `<part>_a` is a scalar that broadcasts over the part
        ab = ed2.ab_dict(sp, av, bv)
        ab[part + "_a"] = np.float64(scalar_a)
        return ed2.log_joint(sp, ab, x)[0]
    fd = (lj(av[lo_] + h) - lj(av[lo_] - h)) / (2 * h)
    assert abs(da[0, lo_:hi_].sum() - fd) <= 1e-6 * (abs(fd) + 1.0)


def test_discrete_prior_ranks_on_elbo_plus_prior(gpu, tmp_path):
    """find_best_learning_rate with --discrete_prior: the timeline it returns and ranks the learning rates on is
    elbo + prior, the value it reports is that minus the prior (reference inference.py:50-54, 120-150)."""
    from autoreparam_amd import flags as flags_mod, graphs, inference, models
    cfg = models.get_model_by_name("8schools")
    f = flags_mod.FlagValues()
    f.num_optimization_steps, f.learning_rates = 200, [0.05, 0.1]
    _, _, elbo, vp, lp = graphs.make_cvip_graph(cfg, tied_pparams=True, flags=f)
    e1, tl1, lr1, _, _, rp1 = inference.find_best_learning_rate(elbo, vp, inference.DiscretePrior(), lp, flags=f)
    e0, tl0, lr0, _, _, rp0 = inference.find_best_learning_rate(elbo, vp, None, lp, flags=f)
    prior = inference.DiscretePrior()
    lp_final = sum(float(np.sum(prior.log_prob(v))) for k, v in rp1.items() if k.endswith("_a"))
    # the reported value is the mean of (elbo + prior) over the last 32 steps minus the mean prior over them; the prior
    # of the final parameters is within the drift of those 32 steps of it
    assert abs((np.mean(tl1[-32:]) - e1) - lp_final) < 0.2 * abs(lp_final) + 0.05
    assert abs(np.mean(tl1[-32:]) - e1) > 1e-3          # the prior term is in the timeline (its density is ~1.02 near
    #                                                      the ends of (0, 1) and ~0.99 in the middle: small either way)
    assert abs(np.mean(tl0[-32:]) - e0) < 1e-9           # no prior: the timeline is the ELBO itself


def _fit(eng, gpu, sp, lrs, n_steps, n_mc, seed, learn=False):
    rs = np.random.RandomState(0)
    loc0 = (1e-2 * rs.randn(len(lrs), sp.D)).astype(np.float32)
    loc = torch.as_tensor(loc0.copy(), device=gpu); rho = torch.full((len(lrs), sp.D), -2.0, device=gpu)
    w = torch.zeros(len(lrs), sp.D, device=gpu) if learn else None
    elbo = eng.vi_run(lrs, loc, rho, n_steps, n_mc, w=w, seed=seed)
    return elbo.cpu().numpy(), loc.cpu().numpy(), rho.cpu().numpy(), eng.vi_geometry()


@pytest.mark.parametrize("mname,knob,values", [("radon_MN", "ARP_VI_G", (4, 8, 32)), ("election", "ARP_VI_G", (4, 16)),
                                               ("german", "ARP_VI_R", (1, 2, 8))])
def test_vi_fit_does_not_depend_on_the_launch_geometry(gpu, monkeypatch, mname, knob, values):
    """The draws belong to the sampler's specification, not to the launch: a learning rate's 256 draws split over 4, 8 or
    32 sample groups (a lane then takes 8, 4 or 1 of them per step and skips the words of the others' turns -- by
    stepping, or by one jump of the generator) and German credit's observations over 1, 2 or 8 row parts give the same
    fit up to the order of the sums; every geometry is bitwise reproducible."""
    from autoreparam_amd import engine
    sp = helpers.spec(mname)
    eng = engine.Engine(sp, gpu)
    eng.set_param(0, "NCP")
    monkeypatch.setenv("ARP_DEBUG", "1")
    out = []
    for v in values:
        monkeypatch.setenv(knob, str(v))
        a = _fit(eng, gpu, sp, [0.02, 0.1], 120, 256, seed=21)
        b = _fit(eng, gpu, sp, [0.02, 0.1], 120, 256, seed=21)
        assert a[3]["sample_groups" if knob == "ARP_VI_G" else "row_parts"] == v
        for x, y in zip(a[:3], b[:3]):
            assert np.array_equal(x, y)
        out.append(a)
    for o in out[1:]:
        np.testing.assert_allclose(o[0][:, :3], out[0][0][:, :3], rtol=2e-6, atol=2e-3)      # same draws: rounding only
        np.testing.assert_allclose(o[0][:, -32:].mean(1), out[0][0][:, -32:].mean(1), rtol=1e-3, atol=0.3)
        np.testing.assert_allclose(o[1], out[0][1], rtol=0, atol=0.03 * (np.abs(out[0][1]).max() + 1))


@pytest.mark.parametrize("mname,n_mc", [("radon_MN", 37), ("radon_MN", 600), ("election", 100), ("german", 100),
                                        ("8schools", 1000)])
def test_vi_ragged_draw_counts_match_oracle(oracle_lib, gpu, mname, n_mc):
    """num_mc_samples that do not fill the draw layout's last turn, a workgroup or a wave: same timelines as the oracle's
    sequential draws (orc_vi_run: stream s mod (512 / lanes), turn s / (512 / lanes))."""
    from autoreparam_amd import engine
    sp = helpers.spec(mname)
    eng = engine.Engine(sp, gpu)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "NCP")
    eng.set_param(0, (a, b))
    lrs = [0.05]
    rs = np.random.RandomState(0)
    loc0 = (1e-2 * rs.randn(1, sp.D)).astype(np.float32); rho0 = np.full((1, sp.D), -2.0, np.float32)
    loc = torch.as_tensor(loc0.copy(), device=gpu); rho = torch.as_tensor(rho0.copy(), device=gpu)
    n_steps = 60
    elbo = eng.vi_run(lrs, loc, rho, n_steps, n_mc, seed=8).cpu().numpy()
    lo, ro = loc0.copy(), rho0.copy()
    elbo_o = orc.vi_run(a, b, lrs, lo, ro, None, n_steps, n_mc, seed=8, lanes=VI_LANES[mname])
    np.testing.assert_allclose(elbo[:, :5], elbo_o[:, :5], rtol=2e-5, atol=2e-2)
    np.testing.assert_allclose(elbo[:, -16:].mean(1), elbo_o[:, -16:].mean(1), rtol=2e-3, atol=0.5)


def test_vi_more_learning_rates_than_fit_on_the_device(gpu):
    """German credit, 12 learning rates: 12 groups of 32 workgroups at one workgroup per CU do not fit on 256 CUs, so
    the library runs them in several launches -- each learning rate's fit is bitwise the one it gets alone."""
    from autoreparam_amd import engine
    sp = helpers.spec("german")
    eng = engine.Engine(sp, gpu)
    eng.set_param(0, "NCP")
    lrs = [0.01 * (k + 1) for k in range(12)]
    e_all, loc_all, rho_all, g = _fit(eng, gpu, sp, lrs, 40, 256, seed=3)
    assert g["learning_rates_per_launch"] < 12
    assert g["sample_groups"] * g["row_parts"] * g["learning_rates_per_launch"] <= g["workgroups_per_cu"] * 256
    assert np.isfinite(e_all).all()
    rs = np.random.RandomState(0)
    loc0 = (1e-2 * rs.randn(12, sp.D)).astype(np.float32)
    for k in (0, 7, 11):
        # alone: the stream id carries the learning rate's INDEX, so it runs as index k of a list of the same length
        loc = torch.as_tensor(loc0.copy(), device=gpu); rho = torch.full((12, sp.D), -2.0, device=gpu)
        lr_k = [1e-9] * 12
        lr_k[k] = lrs[k]
        e_k = eng.vi_run(lr_k, loc, rho, 40, 256, seed=3).cpu().numpy()
        assert np.array_equal(e_k[k], e_all[k]) and np.array_equal(loc.cpu().numpy()[k], loc_all[k])


def test_vi_hand_offs_under_uneven_load(gpu):
    """The in-launch hand-offs between a learning rate's workgroups (8-byte {epoch, value} granules, agent-scope relaxed
    atomics) with the device busy on another stream -- workgroups arrive unevenly, lines are warm in the other XCDs' L2:
    the fit stays bit for bit the quiet one, cVIP's four sums per parameter included."""
    from autoreparam_amd import engine
    for mname in ("election", "german"):
        sp = helpers.spec(mname)
        eng = engine.Engine(sp, gpu)
        eng.set_param(0, (np.full(sp.D, 0.5, np.float32), np.ones(sp.D, np.float32)))
        quiet = _fit(eng, gpu, sp, [0.02, 0.05, 0.1], 150, 256, seed=13, learn=True)
        side = torch.cuda.Stream(device=gpu)
        x = torch.randn(2048, 2048, device=gpu)
        stop = torch.cuda.Event()
        with torch.cuda.stream(side):
            for _ in range(200):
                x = torch.tanh(x @ x * 1e-3)
            stop.record()
        busy = _fit(eng, gpu, sp, [0.02, 0.05, 0.1], 150, 256, seed=13, learn=True)
        still_busy = not stop.query()
        torch.cuda.synchronize()
        for a, b in zip(quiet[:3], busy[:3]):
            assert np.array_equal(a, b)
        assert np.isfinite(busy[0]).all()
        del still_busy


@pytest.mark.parametrize("mname,kind,idx", [("8schools", "CP", 1), ("german", "NCP", 0)])
def test_vi_nan_gradients_are_zeroed(oracle_lib, gpu, mname, kind, idx):
    """inference.py:62-65: a NaN gradient becomes 0 before Adam sees it.  A scale parameter that starts at e^100
    overflows float32: the ELBO estimates are NaN from the first step on and the gradients of (nearly) every variational
    parameter with them -- those parameters stay where they started, bit for bit, the ones whose gradient is still a
    number move as the oracle's do, and nothing becomes NaN.  (Only cases where HIP path and oracle put NaN -- not +-inf,
    which Adam turns into a NaN parameter on both sides, as TF's would -- in the same entries: past float32 overflow that
    class depends on the order of the algebra, DESIGN.md section 9, tests/diagnostics/nonfinite_probe.py.)"""
    from autoreparam_amd import engine
    sp = helpers.spec(mname)
    eng = engine.Engine(sp, gpu)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, kind)
    eng.set_param(0, (a, b))
    rs = np.random.RandomState(0)
    loc0 = (1e-2 * rs.randn(1, sp.D)).astype(np.float32); rho0 = np.full((1, sp.D), -2.0, np.float32)
    loc0[0, idx] = 100.0 if mname != "german" else 95.0
    loc = torch.as_tensor(loc0.copy(), device=gpu); rho = torch.as_tensor(rho0.copy(), device=gpu)
    elbo = eng.vi_run([0.05], loc, rho, 20, 64, seed=3).cpu().numpy()
    lo, ro = loc0.copy(), rho0.copy()
    elbo_o = orc.vi_run(a, b, [0.05], lo, ro, None, 20, 64, seed=3, lanes=VI_LANES[mname])
    assert np.isnan(elbo_o).all() and np.isnan(elbo).all()
    lg, rg = loc.cpu().numpy(), rho.cpu().numpy()
    assert np.isfinite(lg).all() and np.isfinite(rg).all() and np.isfinite(lo).all()
    still = (lo == loc0) & (ro == rho0)
    assert still.sum() >= sp.D - 2 and still[0, idx]
    assert np.array_equal(lg[still], loc0[still]) and np.array_equal(rg[still], rho0[still])
    np.testing.assert_allclose(lg[~still], lo[~still], rtol=1e-3, atol=1e-4)
    np.testing.assert_allclose(rg[~still], ro[~still], rtol=1e-3, atol=1e-4)
