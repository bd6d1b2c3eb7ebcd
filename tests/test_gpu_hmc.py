"""GPU: the fused HMC kernel against the float32 C oracle (same RNG
specification, so trajectories agree to float32 tolerance), chunk invariance,
adaptation, the trace schedule and posterior moments at BASELINE sizes."""
import numpy as np
import pytest
import torch

import helpers
import parity

pytestmark = pytest.mark.gpu
LANES = {"8schools": [1, 8], "radon_MN": [4, 8, 16], "radon_PA": [4, 8, 16], "election": [4, 8, 16], "german": [4, 8, 16], "radon_sd_MN": [8, 16], "funnel": [1], "electric": [8, 16], "time_series": [4, 8, 16]}


def _eng(mname, gpu):
    from autoreparam_amd import engine
    return engine.Engine(helpers.spec(mname), gpu)


def _eps0(oracle_lib, sp, a, b, x, frac):
    """a stable per-element step: frac / sqrt(|diag Hessian|) from finite differences of the oracle gradient"""
    orc = oracle_lib.OracleModel(sp)
    x0 = x[:1].astype(np.float64)
    _, g0 = orc.logp_grad(x0, a, b)
    h = 1e-4
    diag = np.zeros(sp.D)
    for d in range(sp.D):
        xp = x0.copy(); xp[0, d] += h
        diag[d] = -(orc.logp_grad(xp, a, b)[1][0, d] - g0[0, d]) / h
    return (frac / np.sqrt(np.abs(diag) + 1.0)).astype(np.float32)


def _compare(oracle_lib, gpu, mname, kind, lanes, adapt_kind, frac, L, n, n_adapt=0, sp=None, Cn=96, state_tol=1e-4, options=()):
    """Run the HIP kernel and the float32 oracle on the same seeds, twice: once on the recording schedule under test
    (burn-in 2, every third transition: trace rows, accept flags) and once recording EVERY transition (state in sampler
    coordinates + accept flag + the oracle's Metropolis margins), which is what `helpers.explain_divergence` needs to
    hold every chain either to step-by-step agreement or to a decision that sat at its threshold.  Returns a dict."""
    from autoreparam_amd import engine
    if sp is None:
        sp = helpers.spec(mname)
        eng = _eng(mname, gpu)
    else:
        eng = engine.Engine(sp, gpu)
    for key, value in options:
        eng.set_option(key, value)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, kind)
    eng.set_param(0, (a, b))
    q0 = helpers.states(sp, Cn, seed=2, scale=0.1)
    eps0 = _eps0(oracle_lib, sp, a, b, q0, frac)
    kw = dict(seed=9, chain_offset=1000, adapt_kind=adapt_kind, n_adapt=n_adapt, lanes=lanes)
    # run A: the schedule
    st = engine.ChainState(torch.as_tensor(q0, device=gpu))
    tr = torch.zeros(4, Cn, sp.D, device=gpu); ta = torch.zeros(4, Cn, dtype=torch.uint8, device=gpu)
    eng.hmc_run(st, eps0, L, n, n_burnin=2, thin=3, trace=tr, trace_accept=ta, trace_centered=True, **kw)
    so = oracle_lib.new_state(q0, np.float32)
    tro = np.zeros((4, Cn, sp.D), np.float32); tao = np.zeros((4, Cn), np.uint8)
    orc.hmc_run(so, a, b, eps0, L, n, n_burnin=2, thin=3, trace=tro, trace_accept=tao, trace_centered=True, **kw)
    # run B: every transition
    r = parity.hmc_every_step(oracle_lib, eng, orc, (a, b), q0, eps0, L, n, state_tol, "%s %s lanes=%d" % (mname, kind, lanes), **kw)
    # what is recorded does not change the chain
    assert torch.equal(st.q, r["st"].q) and torch.equal(st.rng, r["st"].rng) and np.array_equal(so["q"], r["so"]["q"])
    scale, clean, first, mg, es = r["scale"], r["clean"], r["first"], r["margin"], r["escale"]
    err = np.abs(st.q.cpu().numpy() - so["q"]).max(axis=1) / scale
    # trace rows are in centred coordinates, whose magnitude can exceed the sampler coordinates'
    # (time_series accumulates its latents along the chain): scale each chain's rows by their own size
    terr = np.abs(tr.cpu().numpy() - tro).max(axis=(0, 2)) / np.maximum(scale, np.abs(tro).max(axis=(0, 2)) + 1.0)
    return dict(clean=clean, first=first, err=err, terr=terr, st=st, so=so, ta=ta.cpu().numpy(), tao=tao, margin=mg,
                escale=es)


@pytest.mark.parametrize("mname", ["8schools", "radon_MN", "radon_PA", "election", "german", "radon_sd_MN", "funnel", "electric", "time_series"])
@pytest.mark.parametrize("kind", ["CP", "NCP", "VIP"])
def test_trajectories_match_oracle_fixed_step(oracle_lib, gpu, mname, kind):
    """Fixed step (the parity configuration of north_star): after 12 transitions EVERY chain either agrees with the
    float32 oracle step by step to float32 tolerance, or parts from it at a Metropolis test whose margin in the oracle's
    run was within rounding of zero (helpers.explain_divergence); the chains that never branched are then compared on
    the final state, the trace schedule, acceptance and the log density, the random streams bitwise on all chains."""
    for lanes in LANES[mname]:
        r = _compare(oracle_lib, gpu, mname, kind, lanes, 0, 0.05, 4, 12)
        ok, st, so = r["clean"], r["st"], r["so"]
        assert (r["err"][ok] <= 1e-4).all(), (lanes, np.sort(r["err"][ok])[-5:])
        # the map to centred coordinates can amplify a state difference that is inside the tolerance
        # (time_series sums 60 latents along the chain): nearly all rows to 1e-4, every row to 1e-2
        terr = r["terr"]
        assert (terr[ok] <= 1e-4).mean() >= 0.98 and (terr[ok] <= 1e-2).all(), (lanes, np.sort(terr[ok])[-3:])
        assert np.array_equal(st.accept_count.cpu().numpy()[ok], so["accept_count"][ok])
        assert np.array_equal(r["ta"][:, ok], r["tao"][:, ok])
        # time_series: |logp| ~ 1e9 with gradients ~ 1e9 per unit at these states, so a state difference inside the
        # tolerance moves logp by more than float32 resolution
        lp_tol = 1e-4 if mname == "time_series" else 2e-5
        assert np.abs(st.logp.cpu().numpy()[ok] - so["logp"][ok]).max() <= lp_tol * np.abs(so["logp"]).max() + 2e-3
        # the streams are part of the specification: states must be bitwise equal (branched chains included)
        assert np.array_equal(st.rng.cpu().numpy().view(np.uint32)[:, :lanes], so["rng"][:, :lanes])


@pytest.mark.parametrize("math", ["bf16x3", "f32"])
@pytest.mark.parametrize("n_obs", [100, 300, 640])
def test_german_trajectories_on_one_three_and_five_tiles(oracle_lib, gpu, n_obs, math):
    """German credit on the matrix cores -- bf16 with three-piece operands (64-observation tiles: 2, 5 and 10 of them) and
    f32 (128-observation tiles: 1, 3 and 5) -- with the data set cut short: an odd number of tiles flips the buffer the
    first tile of a gradient lands in, and the trace rows are staged in the LDS area that the next gradient's first tile
    overwrites -- trajectories, acceptance and trace rows against the float32 oracle, same tolerances for both."""
    import copy
    full = helpers.spec("german")
    sp = copy.copy(full)
    sp.raw = dict(full.raw); sp.raw["X"] = full.raw["X"][:n_obs]; sp.raw["y"] = full.raw["y"][:n_obs]
    sp.observed = {"y": sp.raw["y"][None]}
    r = _compare(oracle_lib, gpu, "german", "NCP", 4, 0, 0.05, 4, 12, sp=sp, options=(("german_math", math),))
    ok = r["clean"]
    assert (r["err"][ok] <= 1e-4).all()
    assert (r["terr"][ok] <= 1e-4).mean() >= 0.98 and (r["terr"][ok] <= 1e-2).all()
    assert np.array_equal(r["st"].accept_count.cpu().numpy()[ok], r["so"]["accept_count"][ok])
    assert np.array_equal(r["ta"][:, ok], r["tao"][:, ok])


@pytest.mark.parametrize("mname", ["election", "radon_PA", "german", "time_series"])
def test_trajectories_match_oracle_b_equal_one(oracle_lib, gpu, mname):
    """a free, b = 1 (the parameterisation tied cVIP / dVIP runs execute): election and time_series have a
    compile-time form for it, the other models take the general path; both against the oracle."""
    for lanes in LANES[mname]:
        r = _compare(oracle_lib, gpu, mname, "B1", lanes, 0, 0.05, 4, 12)
        ok = r["clean"]
        assert (r["err"][ok] <= 1e-4).all(), (lanes, np.sort(r["err"][ok])[-5:])
        assert (r["terr"][ok] <= 1e-4).mean() >= 0.98 and (r["terr"][ok] <= 1e-2).all()
        assert np.array_equal(r["st"].accept_count.cpu().numpy()[ok], r["so"]["accept_count"][ok])
        assert np.array_equal(r["st"].rng.cpu().numpy().view(np.uint32)[:, :lanes], r["so"]["rng"][:, :lanes])


def test_adaptation_recurrence_on_scripted_acceptance(oracle_lib, gpu):
    """SURVEY 8c known answer (7): the dual-averaging and simple step-size recurrences on a SCRIPTED sequence of log
    acceptance ratios, through the ABI (arp_adapt_probe runs the kernels' own adapt_update), against the oracle's
    libm restatement to 1e-5 -- independent of any trajectory, so the hardware exp/log/rcp approximations are the
    only difference.  Covers the adapting phase, the hand-over to the averaged step and a chunked continuation."""
    from autoreparam_amd import engine
    eng = _eng("8schools", gpu)
    rs = np.random.RandomState(0)
    n_steps, n, n_adapt = 60, 257, 40
    la = np.minimum(rs.randn(n_steps, n) * 1.5 - 0.3, 5.0).astype(np.float32)
    la[3, :7] = -np.inf                                   # a rejected divergent proposal: alpha = 0
    la[5, 7:11] = 30.0                                    # alpha clipped at 1
    for kind in (1, 2):
        ad = torch.zeros(n, 4, device=gpu); ad[:, 0] = 1.0
        k1 = eng.adapt_probe(la[:25], ad, kind, n_adapt)                       # two launches: state round-trips
        k2 = eng.adapt_probe(la[25:], ad, kind, n_adapt, step_base=25)
        kap = torch.cat([k1, k2]).cpu().numpy()
        import ctypes as C
        upd = oracle_lib.lib().orc_adapt_update_f32       # the oracle's recurrence, float32 arithmetic, libm functions
        ref = np.zeros((n_steps, n)); st = np.zeros((n, 3), np.float32); st[:, 0] = 1.0
        for i in range(n):
            s3 = (C.c_float * 3)(1.0, 0.0, 0.0)
            for t in range(n_steps):
                upd(kind, C.c_longlong(t + 1), n_adapt, C.c_float(0.75), C.c_float(0.05), C.c_float(float(la[t, i])), s3)
                ref[t, i] = s3[0]
            st[i] = (s3[0], s3[1], s3[2])
        assert np.isfinite(kap).all()
        np.testing.assert_allclose(kap, ref, rtol=1e-5, atol=0)
        np.testing.assert_allclose(ad.cpu().numpy()[:, :3], st, rtol=1e-5, atol=2e-6)
        if kind == 1:   # after the adaptation steps the step is frozen at the averaged value
            assert np.ptp(kap[n_adapt:], axis=0).max() == 0.0 and (np.abs(np.log(kap[n_adapt]) - st[:, 2]) < 1e-5).all()


@pytest.mark.parametrize("mname,kind", [("8schools", "NCP"), ("radon_MN", "NCP"), ("election", "CP"), ("german", "NCP"),
                                        ("radon_PA", "CP")])
@pytest.mark.parametrize("adapt", [1, 2])
def test_adaptation_matches_oracle(oracle_lib, gpu, mname, kind, adapt):
    """Dual-averaging / simple step-size adaptation inside the kernels, 10 adapting + 4 frozen transitions, EVERY chain
    at EVERY transition (parity.hmc_teacher_forced): the oracle is restarted from the HIP path's own state before each
    transition, because dual averaging explores up to the integrator's stability limit, where free-running float32
    trajectories part within a few steps whatever computes them.  Each transition's decision, new state, cached log
    density, random streams and adaptation state must agree within the float32 rounding of the energies compared.
    Simple adaptation is also compared free-running (explained divergence, as in the fixed-step test)."""
    lanes = LANES[mname][0] if mname in ("german", "radon_PA") else LANES[mname][-1]   # the lane counts the configs run
    frac = 0.002 if mname in ("election", "german") else 0.02
    sp = helpers.spec(mname)
    eng = _eng(mname, gpu)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, kind)
    eng.set_param(0, (a, b))
    q0 = helpers.states(sp, 96, seed=2, scale=0.1)
    eps0 = _eps0(oracle_lib, sp, a, b, q0, frac)
    flipped = parity.hmc_teacher_forced(oracle_lib, eng, orc, (a, b), q0, eps0, 3, 14, "%s %s lanes=%d adapt=%d" % (
        mname, kind, lanes, adapt), adapt, 10, seed=9, chain_offset=1000, lanes=lanes)
    assert flipped <= 4, flipped     # sanity: decisions at their threshold are rare (96 chains x 14 transitions)
    if adapt == 2:
        r = _compare(oracle_lib, gpu, mname, kind, lanes, adapt, frac, 3, 14, n_adapt=10)
        ok, st, so = r["clean"], r["st"], r["so"]
        np.testing.assert_allclose(st.adapt.cpu().numpy()[ok, 0], so["adapt"][ok, 0], rtol=1e-6)
        assert np.array_equal(st.accept_count.cpu().numpy()[ok], so["accept_count"][ok])


@pytest.mark.parametrize("mname", ["radon_MN", "election"])
def test_chunked_run_is_bitwise_single_run(gpu, mname):
    from autoreparam_amd import engine, _lib
    sp = helpers.spec(mname)
    eng = _eng(mname, gpu)
    eng.set_param(0, "CP")
    q0 = helpers.states(sp, 300, seed=5, scale=0.1)
    eps0 = np.full(sp.D, 0.02, np.float32)
    kw = dict(seed=3, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=6, n_burnin=1, thin=2)
    one = engine.ChainState(torch.as_tensor(q0, device=gpu)); t1 = torch.zeros(5, 300, sp.D, device=gpu)
    eng.hmc_run(one, eps0, 3, 11, trace=t1, **kw)
    two = engine.ChainState(torch.as_tensor(q0, device=gpu)); t2 = torch.zeros(5, 300, sp.D, device=gpu)
    for n in (1, 4, 6):
        eng.hmc_run(two, eps0, 3, n, trace=t2, **kw)
    for k in ("q", "grad", "logp", "adapt", "rng", "accept_count"):
        assert torch.equal(getattr(one, k), getattr(two, k)), k
    assert torch.equal(t1, t2)


def test_chain_offset_keys_the_streams(gpu):
    """Sharding invariance: chains [64,128) run alone with chain_offset=64 equal the
    same chains inside a 128-chain launch (what one rank of a multi-GPU run does)."""
    from autoreparam_amd import engine
    sp = helpers.spec("radon_PA")
    eng = _eng("radon_PA", gpu)
    eng.set_param(0, "CP")
    q0 = helpers.states(sp, 128, seed=8, scale=0.1)
    eps0 = np.full(sp.D, 0.02, np.float32)
    full = engine.ChainState(torch.as_tensor(q0, device=gpu))
    eng.hmc_run(full, eps0, 4, 6, seed=1, lanes=8)
    half = engine.ChainState(torch.as_tensor(q0[64:], device=gpu))
    eng.hmc_run(half, eps0, 4, 6, seed=1, chain_offset=64, lanes=8)
    assert torch.equal(full.q[64:], half.q)


def test_radon_posterior_means_config2(gpu, oracle_lib):
    """BASELINE config 2 shape: radon MN, CP, 4096 chains, 4 leapfrog steps; posterior
    means within 1 % of the closed-form Gaussian answer."""
    from autoreparam_amd import engine, _lib
    sp = helpers.spec("radon_MN")
    eng = _eng("radon_MN", gpu)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "CP")
    eng.set_param(0, "CP")
    D = sp.D
    _, g0 = orc.logp_grad(np.zeros((1, D)), a, b)
    _, gI = orc.logp_grad(np.eye(D), a, b)
    P = -(gI - g0)
    mean = np.linalg.solve(P, g0[0]); sd = np.sqrt(np.diag(np.linalg.inv(P)))
    Cn, S = 4096, 200
    rs = np.random.RandomState(0)
    q0 = (mean + 1.5 * sd * rs.randn(Cn, D)).astype(np.float32)
    st = engine.ChainState(torch.as_tensor(q0, device=gpu))
    tr = torch.zeros(S, Cn, D, device=gpu)
    eng.hmc_run(st, sd.astype(np.float32), 4, 1 + 200 + 2 * (S - 1), seed=4, adapt_kind=_lib.ADAPT_DUAL, n_adapt=150,
                n_burnin=200, thin=2, trace=tr, trace_centered=True)
    m = tr.double().mean(dim=(0, 1)).cpu().numpy()
    v = tr.double().reshape(-1, D).std(dim=0).cpu().numpy()
    acc = st.accept_count.double().mean().item() / st.step
    assert 0.6 < acc < 0.95
    assert np.abs((m - mean) / sd).max() < 0.02          # every coordinate within 2 % of a posterior sd
    assert np.abs(m[:3] - mean[:3]).max() <= 0.01 * np.abs(mean[:3]).max()   # "means within 1 %"
    assert np.abs(v / sd - 1).max() < 0.03


def test_full_size_headline_invariants(gpu):
    """radon PA at 65 536 chains (BASELINE headline size): finite states, sane
    acceptance, pooled means equal to the closed form within Monte-Carlo error."""
    from autoreparam_amd import engine, _lib
    sp = helpers.spec("radon_PA")
    eng = _eng("radon_PA", gpu)
    eng.set_param(0, "CP")
    Cn = 65536
    q0 = helpers.states(sp, Cn, seed=1, scale=0.5)
    st = engine.ChainState(torch.as_tensor(q0, device=gpu))
    eps0 = np.full(sp.D, 0.1, np.float32); eps0[2] = 0.01
    eng.hmc_run(st, eps0, 8, 300, seed=2, adapt_kind=_lib.ADAPT_DUAL, n_adapt=200)
    assert torch.isfinite(st.q).all()
    acc = st.accept_count.double().mean().item() / st.step
    assert 0.55 < acc < 0.95
    m = st.q.double().mean(dim=0).cpu().numpy()
    np.testing.assert_allclose(m[:3], (1.3389, -0.0480, -0.1026), atol=0.01)


def test_full_size_interleaved_posterior(gpu, oracle_lib):
    """The headline kernel itself (interleaved CP/NCP, radon PA, 65 536 chains, 4 + 4 leapfrog steps, simple
    adaptation, gradient carried across the coordinate change): pooled posterior means and sds of all 71
    coordinates against the closed-form Gaussian answer."""
    from autoreparam_amd import engine, _lib
    sp = helpers.spec("radon_PA")
    eng = _eng("radon_PA", gpu)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "CP")
    eng.set_param(0, "CP"); eng.set_param(1, "NCP")
    D = sp.D
    _, g0 = orc.logp_grad(np.zeros((1, D)), a, b)
    _, gI = orc.logp_grad(np.eye(D), a, b)
    P = -(gI - g0)
    mean = np.linalg.solve(P, g0[0]); sd = np.sqrt(np.diag(np.linalg.inv(P)))
    Cn, S, burn = 65536, 8, 400
    rs = np.random.RandomState(3)
    q0 = (mean + 1.5 * sd * rs.randn(Cn, D)).astype(np.float32)
    st = engine.ChainState(torch.as_tensor(q0, device=gpu))
    tr = torch.zeros(S, Cn, D, device=gpu)
    e0 = (0.5 * sd).astype(np.float32)
    # NCP scales: the county effects are unit-scale residuals there
    x_ncp = orc.transform(mean[None], *helpers.params(sp, "NCP"), to_centered=False)[0]
    e1 = e0.copy(); e1[3:] = 0.5
    eng.interleaved_run(st, e0, e1, 4, 4, 1 + burn + 25 * (S - 1), seed=21, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=300,
                        n_burnin=burn, thin=25, trace=tr)
    assert torch.isfinite(st.q).all() and np.isfinite(x_ncp).all()
    acc0 = st.accept_count.double().mean().item() / st.step
    acc1 = st.accept_count1.double().mean().item() / st.step
    assert 0.55 < acc0 < 0.95 and 0.55 < acc1 < 0.95, (acc0, acc1)
    m = tr.double().mean(dim=(0, 1)).cpu().numpy()
    v = tr.double().reshape(-1, D).std(dim=0).cpu().numpy()
    # S * Cn = 524 288 nearly independent draws: Monte-Carlo error of a mean ~ sd / 700
    assert np.abs((m - mean) / sd).max() < 0.02
    assert np.abs(m[:3] - mean[:3]).max() <= 0.01 * np.abs(mean[:3]).max()
    assert np.abs(v / sd - 1).max() < 0.02


@pytest.mark.parametrize("mname", ["8schools", "radon_PA", "election", "german", "radon_sd_MN", "funnel", "electric", "time_series"])
def test_interleaved_matches_oracle(oracle_lib, gpu, mname):
    """Interleaved CP/NCP kernel (interleaved.py:113-155) against the float32 oracle, EVERY lanes-per-chain split the
    library instantiates (radon_PA at 4 lanes is the instantiation bench.py times): fixed small step sizes, simple
    adaptation on, the GPU run cut into two launches.  96 chains fill whole waves (radon: the compile-time row-store
    branch), 70 leave a ragged last wave (the general branch).  Every chain agrees step by step or parts at a
    Metropolis test at its threshold; trace rows, both accept arrays and both adaptation states are compared."""
    from autoreparam_amd import engine, _lib
    sp = helpers.spec(mname)
    eng = _eng(mname, gpu)
    orc = oracle_lib.OracleModel(sp)
    cp, ncp = helpers.params(sp, "CP"), helpers.params(sp, "NCP")
    eng.set_param(0, cp); eng.set_param(1, ncp)
    for lanes in LANES[mname]:
        for Cn in (96, 70):
            q0 = helpers.states(sp, Cn, seed=6, scale=0.1)
            e0 = _eps0(oracle_lib, sp, cp[0], cp[1], q0, 0.03)
            q0n = orc.transform(q0, ncp[0], ncp[1], to_centered=False).astype(np.float32)
            e1 = _eps0(oracle_lib, sp, ncp[0], ncp[1], q0n, 0.03)
            kw = dict(seed=13, chain_offset=77, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=4, lanes=lanes)
            r = parity.interleaved_every_step(oracle_lib, eng, orc, cp, ncp, q0, e0, e1, 3, 2, 6, 2e-4,
                                              "%s lanes=%d C=%d" % (mname, lanes, Cn), chunks=[3, 3], **kw)
            ok, so, scale = r["clean"], r["so"], r["scale"]
            # the recording schedule under test: burn-in 1, every second step, the run cut into two launches
            st = engine.ChainState(torch.as_tensor(q0, device=gpu))
            tr = torch.zeros(3, Cn, sp.D, device=gpu)
            t0 = torch.zeros(3, Cn, dtype=torch.uint8, device=gpu); t1 = torch.zeros(3, Cn, dtype=torch.uint8, device=gpu)
            for _ in range(2):
                eng.interleaved_run(st, e0, e1, 3, 2, 3, trace=tr, trace_accept0=t0, trace_accept1=t1, n_burnin=1, thin=2,
                                    trace_centered=False, **kw)   # parameterisation-0 coordinates, as inference.hmc_interleaved records
            assert torch.equal(st.q, r["st"].q) and torch.equal(st.rng, r["st"].rng)     # recording does not change the chain
            assert torch.equal(tr, r["x"][1::2]) and np.array_equal(t0.cpu().numpy(), r["acc"][1::2, 0]) \
                and np.array_equal(t1.cpu().numpy(), r["acc"][1::2, 1])
            err = np.abs(st.q.cpu().numpy() - so["q"]).max(axis=1) / scale
            assert (err[ok] <= 2e-4).all(), (lanes, Cn, np.sort(err[ok])[-4:])
            assert np.array_equal(st.accept_count.cpu().numpy()[ok], so["accept_count"][ok])
            assert np.array_equal(st.accept_count1.cpu().numpy()[ok], so["accept_count1"][ok])
            assert np.abs(r["x"].cpu().numpy()[:, ok] - r["xo"][:, ok]).max() <= 2e-4 * scale
            assert np.array_equal(r["acc"][:, :, ok], r["acco"][:, :, ok])
            np.testing.assert_allclose(st.adapt.cpu().numpy()[ok, 0], so["adapt"][ok, 0], rtol=1e-5)
            np.testing.assert_allclose(st.adapt1.cpu().numpy()[ok, 0], so["adapt1"][ok, 0], rtol=1e-5)
            assert np.array_equal(st.rng.cpu().numpy().view(np.uint32)[:, :lanes], so["rng"][:, :lanes])


def _schools_quadrature():
    """8 schools posterior moments by marginalising theta analytically and integrating
    (mu, log_tau) on a grid: a known answer that does not involve the sampler."""
    y = np.array([28, 8, -3, 7, -1, 1, 18, 12.0]); s = np.array([15, 10, 16, 11, 9, 11, 10, 18.0])
    mu = np.linspace(-40, 50, 1801)[:, None, None]
    lt = np.linspace(-22, 9, 1241)[None, :, None]
    v = np.exp(2 * lt) + s ** 2
    lp = (-0.5 * (y - mu) ** 2 / v - 0.5 * np.log(v)).sum(-1) - 0.5 * (mu[..., 0] / 5) ** 2 - 0.5 * (lt[..., 0] / 5) ** 2
    w = np.exp(lp - lp.max()); w /= w.sum()
    th = (y / s ** 2 + mu / np.exp(2 * lt)) / (1 / s ** 2 + 1 / np.exp(2 * lt))
    e_mu, e_lt = (w * mu[..., 0]).sum(), (w * lt[..., 0]).sum()
    sd_mu = np.sqrt((w * (mu[..., 0] - e_mu) ** 2).sum()); sd_lt = np.sqrt((w * (lt[..., 0] - e_lt) ** 2).sum())
    e_th = (w[..., None] * th).sum((0, 1))
    return e_mu, sd_mu, e_lt, sd_lt, e_th


def test_eight_schools_posterior_against_quadrature(gpu):
    """BASELINE config 1 model, non-Gaussian funnel geometry: NCP HMC (4 leapfrog steps, dual
    averaging) reproduces the quadrature posterior means of mu, log_tau and theta."""
    from autoreparam_amd import engine, _lib
    e_mu, sd_mu, e_lt, sd_lt, e_th = _schools_quadrature()
    sp = helpers.spec("8schools")
    eng = _eng("8schools", gpu)
    eng.set_param(0, "NCP")
    Cn, S = 8192, 250
    rs = np.random.RandomState(0)
    q0 = (0.5 * rs.randn(Cn, sp.D)).astype(np.float32)
    st = engine.ChainState(torch.as_tensor(q0, device=gpu))
    tr = torch.zeros(S, Cn, sp.D, device=gpu)
    eps0 = np.full(sp.D, 0.25, np.float32)
    eng.hmc_run(st, eps0, 4, 1 + 600 + 2 * (S - 1), seed=6, adapt_kind=_lib.ADAPT_DUAL, n_adapt=500, n_burnin=600,
                thin=2, trace=tr, trace_centered=True, lanes=8)
    acc = st.accept_count.double().mean().item() / st.step
    assert 0.55 < acc < 0.95
    m = tr.double().mean(dim=(0, 1)).cpu().numpy()
    assert abs(m[0] - e_mu) < 0.05 * sd_mu + 0.05, (m[0], e_mu)
    assert abs(m[1] - e_lt) < 0.05 * sd_lt + 0.05, (m[1], e_lt)
    assert np.abs(m[2:] - e_th).max() < 0.25, (m[2:], e_th)


@pytest.mark.parametrize("mname,kind,L,Cn", [("election", "CP", 8, 2048), ("german", "NCP", 8, 768), ("electric", "NCP", 8, 1024),
                                              ("radon_sd_MN", "CP", 8, 1024), ("time_series", "CP", 64, 512)])
def test_posterior_moments_against_long_cpu_run(gpu, mname, kind, L, Cn):
    """election / german credit / electric / radon_stddvs / time_series have no closed-form posterior: the known answer is a long float64
    oracle run committed with Monte-Carlo error bars (tests/golden/posterior_golden.npz)."""
    import os
    from autoreparam_amd import engine, _lib
    gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "posterior_golden.npz"))
    mean_g, sd_g, mcse_g = gold[mname + "/mean"], gold[mname + "/sd"], gold[mname + "/mcse"]
    sc, mode = gold[mname + "/step_scale"], gold[mname + "/mode"]
    sp = helpers.spec(mname)
    eng = _eng(mname, gpu)
    eng.set_param(0, kind)
    rs = np.random.RandomState(1)
    q0 = (mode + 0.5 * sc * rs.randn(Cn, sp.D)).astype(np.float32)     # mode / scale are in `kind` coordinates
    st = engine.ChainState(torch.as_tensor(q0, device=gpu))
    burn, S = 1500, (600 if mname == "german" else 300)   # german's log-scale coordinates mix slowly
    if mname == "time_series":                            # 123 latents chained in time: long trajectories, long burn-in
        burn, S = 8000, 600
    tr = torch.zeros(S, Cn, sp.D, device=gpu)
    eng.hmc_run(st, (0.5 * sc).astype(np.float32), L, 1 + burn + 2 * (S - 1), seed=77, adapt_kind=_lib.ADAPT_DUAL,
                n_adapt=burn - 200, n_burnin=burn, thin=2, trace=tr, trace_centered=True)
    acc = st.accept_count.double().mean().item() / st.step
    assert 0.55 < acc < 0.95, acc
    cm = tr.double().mean(dim=0).cpu().numpy()
    mean = cm.mean(axis=0)
    mcse = cm.std(axis=0, ddof=1) / np.sqrt(Cn)
    sd = tr.double().reshape(-1, sp.D).std(dim=0).cpu().numpy()
    z = np.abs(mean - mean_g) / (np.sqrt(mcse ** 2 + mcse_g ** 2) + 0.01 * sd_g)
    assert z.max() < 5.0, (int(z.argmax()), z.max())
    assert np.abs(sd / sd_g - 1).max() < 0.10   # both runs estimate the marginal sds from finite samples
    # "posterior means within 1 %" on the coordinates whose mean is well away from zero
    # (where the Monte-Carlo error of the two runs allows a 1 % statement)
    err = np.sqrt(mcse ** 2 + mcse_g ** 2)
    big = (np.abs(mean_g) > 5 * sd_g) & (4 * err < 0.01 * np.abs(mean_g))
    if big.any():
        assert (np.abs(mean[big] / mean_g[big] - 1) < 0.01).all()


@pytest.mark.parametrize("mname", ["radon_PA", "election"])
def test_general_ab_packed_kernels_sample_the_same_posterior(gpu, oracle_lib, mname):
    """The packed general-(a, b) forms (round 4: pk_hmc_kernel<RadonPk / ElectionPk, kModeVIP>, what cVIP / dVIP and untied
    runs execute): the posterior of the CENTRED coordinates does not depend on the parameterisation -- radon PA against
    its closed form, election against the long float64 fixture -- from states started in VIP coordinates."""
    import os
    from autoreparam_amd import engine, _lib
    sp = helpers.spec(mname)
    eng = _eng(mname, gpu)
    a, b = helpers.params(sp, "VIP", seed=3)
    eng.set_param(0, (a, b))
    D = sp.D
    rs = np.random.RandomState(2)
    if mname == "radon_PA":
        orc = oracle_lib.OracleModel(sp)
        ac, bc = helpers.params(sp, "CP")
        _, g0 = orc.logp_grad(np.zeros((1, D)), ac, bc)
        _, gI = orc.logp_grad(np.eye(D), ac, bc)
        Pm = -(gI - g0)
        mean_g = np.linalg.solve(Pm, g0[0]); sd_g = np.sqrt(np.diag(np.linalg.inv(Pm))); mcse_g = np.zeros(D)
        Cn, L, burn, S = 8192, 8, 400, 200
        x0 = mean_g + 1.5 * sd_g * rs.randn(Cn, D)
        eps0 = (0.6 * sd_g).astype(np.float32)
    else:
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "posterior_golden.npz"))
        mean_g, sd_g, mcse_g = gold["election/mean"], gold["election/sd"], gold["election/mcse"]
        Cn, L, burn, S = 2048, 8, 1500, 300
        x0 = mean_g + 0.5 * sd_g * rs.randn(Cn, D)
        eps0 = (0.25 * sd_g).astype(np.float32)
    q0 = eng.transform(x0.astype(np.float32), which=0, to_centered=False)          # centred -> VIP coordinates
    back = eng.transform(q0, which=0, to_centered=True).cpu().numpy()
    np.testing.assert_allclose(back, x0, rtol=2e-4, atol=2e-4 * np.abs(x0).max())
    st = engine.ChainState(q0)
    tr = torch.zeros(S, Cn, D, device=gpu)
    eng.hmc_run(st, eps0, L, 1 + burn + 2 * (S - 1), seed=21, adapt_kind=_lib.ADAPT_DUAL, n_adapt=burn - 100, n_burnin=burn,
                thin=2, trace=tr, trace_centered=True)
    acc = st.accept_count.double().mean().item() / st.step
    assert 0.55 < acc < 0.95, acc
    cm = tr.double().mean(dim=0).cpu().numpy()
    mean = cm.mean(axis=0)
    mcse = cm.std(axis=0, ddof=1) / np.sqrt(Cn)
    sd = tr.double().reshape(-1, D).std(dim=0).cpu().numpy()
    z = np.abs(mean - mean_g) / (np.sqrt(mcse ** 2 + mcse_g ** 2) + 0.01 * sd_g)
    assert z.max() < 5.0, (int(z.argmax()), float(z.max()))
    assert np.abs(sd / sd_g - 1).max() < 0.10
    if mname == "radon_PA":
        assert np.abs(mean[:3] - mean_g[:3]).max() <= 0.01 * np.abs(mean_g[:3]).max()      # "posterior means within 1 %"


@pytest.mark.parametrize("mname,lanes", [("radon_PA", 4), ("radon_PA", 8), ("election", 4), ("german", 4), ("8schools", 2)])
def test_in_kernel_statistics_match_trace_and_oracle(oracle_lib, gpu, mname, lanes):
    """arp_hmc_io.stats / rec_accept_count / trace_chains: the statistics the kernel accumulates while sampling
    equal (i) the oracle's, (ii) the moments and batch-means ESS of the full trace of the same run, over a
    chunked run with a ragged chain count; the partial trace equals the first chains of the full one."""
    from autoreparam_amd import engine, inference
    sp = helpers.spec(mname)
    eng = _eng(mname, gpu)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "NCP")
    eng.set_param(0, (a, b))
    Cn, S, batch, keep, burn, thin = 75, 24, 4, 5, 3, 2
    q0 = helpers.states(sp, Cn, seed=4, scale=0.1)
    eps0 = _eps0(oracle_lib, sp, a, b, q0, 0.05)
    total = 1 + burn + thin * (S - 1)
    kw = dict(seed=8, n_burnin=burn, thin=thin, trace_centered=True, lanes=lanes)
    # full trace run
    st_a = engine.ChainState(torch.as_tensor(q0, device=gpu))
    tr_a = torch.zeros(S, Cn, sp.D, device=gpu); ta_a = torch.zeros(S, Cn, dtype=torch.uint8, device=gpu)
    eng.hmc_run(st_a, eps0, 3, total, trace=tr_a, trace_accept=ta_a, **kw)
    # statistics run, chunked into uneven launches, partial trace of the first `keep` chains
    st_b = engine.ChainState(torch.as_tensor(q0, device=gpu))
    stats = torch.zeros(6, Cn, sp.D, device=gpu); racc = torch.zeros(Cn, dtype=torch.int32, device=gpu)
    tr_b = torch.zeros(S, keep, sp.D, device=gpu)
    for n in (5, 1, 17, total - 23):
        eng.hmc_run(st_b, eps0, 3, n, stats=stats, stats_batch=batch, n_samples=S, trace=tr_b, trace_chains=keep,
                    rec_accept=racc, **kw)
    assert torch.equal(st_a.q, st_b.q) and torch.equal(st_a.rng, st_b.rng)
    assert torch.equal(tr_b, tr_a[:, :keep])
    assert torch.equal(racc.long(), ta_a.long().sum(dim=0))
    mean, var, ess = engine.stats_summary(stats, S, batch)
    x = tr_a.double()
    scale = x.abs().amax(dim=0) + 1
    assert ((mean - x.mean(dim=0)).abs() / scale).max() < 2e-6
    assert ((var - x.var(dim=0, unbiased=True)).abs() / (scale * scale)).max() < 2e-6
    ss = inference.StreamingStats(Cn, sp.D, batch, gpu); ss.update(tr_a)
    ok = torch.isfinite(ss.ess()) & (ss.ess() < S)
    assert torch.allclose(ess.float()[ok], ss.ess()[ok], rtol=2e-3)
    # oracle, float32, same schedule; chains are compared unless they parted from the oracle at a Metropolis test at
    # its threshold (every-step run of the same seeds)
    so = oracle_lib.new_state(q0, np.float32)
    stats_o = np.zeros((6, Cn, sp.D), np.float32); racc_o = np.zeros(Cn, np.uint32)
    orc.hmc_run(so, a, b, eps0, 3, total, stats=stats_o, stats_batch=batch, n_samples=S, rec_accept=racc_o, **kw)
    r = parity.hmc_every_step(oracle_lib, eng, orc, (a, b), q0, eps0, 3, total, 1e-4, "%s lanes=%d stats" % (mname, lanes),
                              seed=8, lanes=lanes)
    same = r["clean"]
    assert torch.equal(r["st"].q, st_b.q)
    assert (np.abs(st_b.q.cpu().numpy() - so["q"]).max(axis=1)[same] <= 1e-4 * (np.abs(so["q"]).max() + 1)).all()
    mean_o, var_o, ess_o = engine.stats_summary(torch.as_tensor(stats_o), S, batch)
    sc, sm = scale.cpu(), torch.as_tensor(same)
    assert ((mean.cpu() - mean_o).abs() / sc)[sm].max() < 2e-5
    assert ((var.cpu() - var_o).abs() / (sc * sc))[sm].max() < 2e-5
    fin = torch.isfinite(ess_o) & (ess_o < S) & ok.cpu() & sm[:, None]
    assert torch.allclose(ess.cpu()[fin], ess_o[fin], rtol=5e-3)
    assert np.array_equal(racc.cpu().numpy()[same], racc_o[same].astype(np.int64))


def test_in_kernel_statistics_interleaved(oracle_lib, gpu):
    """The same for the interleaved kernel (carried-gradient radon form and a re-bootstrapping model)."""
    from autoreparam_amd import engine
    for mname, lanes in (("radon_PA", 4), ("election", 8)):
        sp = helpers.spec(mname)
        eng = _eng(mname, gpu)
        eng.set_param(0, "CP"); eng.set_param(1, "NCP")
        Cn, S, batch, burn, thin = 70, 12, 3, 2, 2
        q0 = helpers.states(sp, Cn, seed=5, scale=0.1)
        e = np.full(sp.D, 1e-3, np.float32)
        total = 1 + burn + thin * (S - 1)
        kw = dict(seed=3, n_burnin=burn, thin=thin, trace_centered=False, lanes=lanes)
        st_a = engine.ChainState(torch.as_tensor(q0, device=gpu))
        tr = torch.zeros(S, Cn, sp.D, device=gpu)
        t0 = torch.zeros(S, Cn, dtype=torch.uint8, device=gpu); t1 = torch.zeros(S, Cn, dtype=torch.uint8, device=gpu)
        eng.interleaved_run(st_a, e, e, 2, 2, total, trace=tr, trace_accept0=t0, trace_accept1=t1, **kw)
        st_b = engine.ChainState(torch.as_tensor(q0, device=gpu))
        stats = torch.zeros(6, Cn, sp.D, device=gpu)
        r0 = torch.zeros(Cn, dtype=torch.int32, device=gpu); r1 = torch.zeros(Cn, dtype=torch.int32, device=gpu)
        for n in (4, total - 4):
            eng.interleaved_run(st_b, e, e, 2, 2, n, stats=stats, stats_batch=batch, n_samples=S, rec_accept0=r0,
                                rec_accept1=r1, **kw)
        assert torch.equal(st_a.q, st_b.q)
        assert torch.equal(r0.long(), t0.long().sum(dim=0)) and torch.equal(r1.long(), t1.long().sum(dim=0))
        mean, var, _ = engine.stats_summary(stats, S, batch)
        x = tr.double(); scale = x.abs().amax(dim=0) + 1
        assert ((mean - x.mean(dim=0)).abs() / scale).max() < 2e-6
        assert ((var - x.var(dim=0, unbiased=True)).abs() / (scale * scale)).max() < 2e-6


@pytest.mark.parametrize("chains,lanes", [(65536, 0), (8192, 0), (1000, 4), (4097, 8)])
def test_relay_segments_equal_the_unsegmented_launch(gpu, monkeypatch, chains, lanes):
    """radon's interleaved launch cut into segments of steps (a workgroup per segment and chain block, the state relayed through
    HBM inside the launch; arp_api.hip: arp_interleaved_run, radon_fast.h): states, gradients, adaptation, counters, trace
    rows, acceptance flags and in-kernel statistics bit for bit those of the launch with one workgroup per block -- for the
    library's own choice (4 or 8 segments where the blocks are a round or more, none below), for 2, 3 and 8 segments, with steps that do not divide evenly and a
    burn-in that ends inside a segment, twice in a row (the flags' epochs)."""
    from autoreparam_amd import engine, _lib
    sp = helpers.spec("radon_PA")
    eng = _eng("radon_PA", gpu)
    eng.set_param(0, "CP"); eng.set_param(1, "NCP")
    q0 = helpers.states(sp, chains, seed=2, scale=0.1)
    e = np.full(sp.D, 0.06, np.float32); e[2] = 0.015
    monkeypatch.setenv("ARP_DEBUG", "1")

    def run(segs, T, with_stats):
        if segs is None:
            monkeypatch.delenv("ARP_SEGMENTS", raising=False)
        else:
            monkeypatch.setenv("ARP_SEGMENTS", str(segs))
        st = engine.ChainState(torch.as_tensor(q0, device=gpu))
        n_burn = 101
        S = 2 * ((2 * T - n_burn) // 2 // 2 + 1)
        tr = torch.zeros(S, chains, sp.D, device=gpu)
        a0 = torch.zeros(S, chains, dtype=torch.uint8, device=gpu); a1 = torch.zeros_like(a0)
        extra = dict(stats=torch.zeros(6, chains, sp.D, device=gpu), stats_batch=3, n_samples=S) if with_stats else {}
        for _ in range(2):          # two launches: the second starts from the first one's state (step_base > 0)
            eng.interleaved_run(st, e, e, 4, 4, T, seed=9, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=150, n_burnin=n_burn, thin=2,
                                trace=tr, trace_accept0=a0, trace_accept1=a1, trace_centered=False, lanes=lanes, **extra)
        torch.cuda.synchronize()
        out = [st.q, st.grad, st.logp, st.adapt, st.adapt1, st.accept_count, st.accept_count1, st.rng, tr, a0, a1]
        if with_stats:
            out.append(extra["stats"])
        return [t.cpu().numpy() for t in out]

    for T, with_stats in ((333, False), (256, True), (520, False)):     # library's choice: 4, 4 and 8 segments
        ref = run(1, T, with_stats)
        # the small launch repeats its 8-segment run: a hand-over that does not publish EVERY wave's stores before the flag goes
        # up hands over a stale cache line once in ~ 10 such runs (tests/diagnostics/relay_race_probe.py; round 6)
        for segs in (None, 2, 3, 8) + ((8,) * 24 if chains == 8192 and with_stats else ()):
            got = run(segs, T, with_stats)
            for k, (x, y) in enumerate(zip(ref[:11], got[:11])):
                assert np.array_equal(x, y, equal_nan=True), (segs, T, with_stats, k)
            if with_stats:
                # the accumulators fold their partial batch where a segment ends, as they do where a launch ends: the sums
                # are the same numbers added in another grouping (float32 rounding of the sums)
                n = ref[8].shape[0]
                for a_, b_ in zip(engine.stats_summary(torch.as_tensor(ref[11]), n, 3)[:2], engine.stats_summary(torch.as_tensor(got[11]), n, 3)[:2]):
                    scale = a_.abs().max(dim=0).values + 1e-3
                    assert ((a_ - b_).abs() / scale).max() < 1e-5, (segs, T)


@pytest.mark.parametrize("mname,kind,chains,sampler", [
    ("german", "NCP", 16384, "hmc"),          # hmc_kernel<GermanLane<4,16,4,false,true>>: 256 workgroups = one round at one per CU
    ("german", "VIP", 4099, "hmc"),
    ("election", "NCP", 131072, "hmc"),       # pk_hmc_kernel<ElectionPk<4,13>, NCP>
    ("election", "B1", 70001, "hmc"),
    ("radon_MN", "CP", 65536, "hmc"),         # pk_hmc_kernel<RadonPk<4,22>, CP>
    ("election", None, 65536, "interleaved"),   # pk_interleaved_kernel (re-bootstraps)
    ("electric", None, 65536, "interleaved"),   # the generic interleaved_kernel
    ("radon_sd_MN", "NCP", 65536, "hmc"),     # the generic hmc_kernel at two waves per SIMD
])
def test_relay_segments_in_every_chain_kernel(gpu, monkeypatch, mname, kind, chains, sampler):
    """The relay (kernels.h: relay_begin / relay_end) in the generic and the packed chain kernels: the library's own choice of
    segments and a forced three against the launch with one workgroup per chain block -- states, gradients, log densities,
    adaptation, counters, generator states, trace rows and acceptance flags bit for bit, over two launches."""
    from autoreparam_amd import engine, _lib
    sp = helpers.spec(mname)
    eng = _eng(mname, gpu)
    if sampler == "hmc":
        eng.set_param(0, helpers.params(sp, kind, seed=3))
    else:
        eng.set_param(0, "CP"); eng.set_param(1, "NCP")
    q0 = helpers.states(sp, chains, seed=4, scale=0.05)
    e = np.full(sp.D, 2e-3 if mname != "time_series" else 1e-4, np.float32)
    T = 300 if mname != "german" else 260
    keep = min(chains, 4096)
    monkeypatch.setenv("ARP_DEBUG", "1")

    def run(segs):
        if segs is None:
            monkeypatch.delenv("ARP_SEGMENTS", raising=False)
        else:
            monkeypatch.setenv("ARP_SEGMENTS", str(segs))
        st = engine.ChainState(torch.as_tensor(q0, device=gpu))
        S = T      # rows: two launches of T steps, every second one recorded
        tr = torch.zeros(S, keep, sp.D, device=gpu)
        a0 = torch.zeros(S, chains, dtype=torch.uint8, device=gpu); a1 = torch.zeros_like(a0)
        for _ in range(2):
            if sampler == "hmc":
                eng.hmc_run(st, e, 3, T, seed=21, adapt_kind=_lib.ADAPT_DUAL, n_adapt=350, n_burnin=57, thin=2, trace=tr,
                            trace_accept=a0, trace_chains=keep)
            else:
                eng.interleaved_run(st, e, e, 2, 2, T, seed=21, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=350, n_burnin=57, thin=2,
                                    trace=tr, trace_accept0=a0, trace_accept1=a1, trace_chains=keep)
        torch.cuda.synchronize()
        out = [st.q, st.grad, st.logp, st.adapt, st.accept_count, st.rng, tr, a0, a1]
        if sampler != "hmc":
            out += [st.adapt1, st.accept_count1]
        return [t.cpu().numpy() for t in out]

    ref = run(1)
    assert np.isfinite(ref[0]).all() and ref[4].sum() > 0
    for segs in (None, 3):
        got = run(segs)
        for k, (x, y) in enumerate(zip(ref, got)):
            assert np.array_equal(x, y, equal_nan=True), (segs, k)


def test_relay_rule(gpu, monkeypatch):
    """Which launches the library cuts into relay segments (host_common.h: relay_plan), read back through arp_relay_geometry: 8
    from 512 steps per launch on and 4 from 256 where the chain blocks are at least one round of the kernel's resident
    workgroups; none for shorter launches, for fewer blocks, or for kernels that hold a whole CU per workgroup."""
    from autoreparam_amd import engine, _lib
    monkeypatch.delenv("ARP_SEGMENTS", raising=False)

    def segs(mname, chains, T, sampler="i", lanes=0):
        sp = helpers.spec(mname)
        eng = _eng(mname, gpu)
        eng.set_param(0, "CP"); eng.set_param(1, "NCP")
        st = engine.ChainState(torch.as_tensor(helpers.states(sp, chains, seed=1, scale=0.05), device=gpu))
        e = np.full(sp.D, 1e-3, np.float32)
        if sampler == "i":
            eng.interleaved_run(st, e, e, 1, 1, T, seed=1, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=0, lanes=lanes)
        else:
            eng.hmc_run(st, e, 1, T, seed=1, adapt_kind=_lib.ADAPT_NONE, lanes=lanes)
        torch.cuda.synchronize()
        return eng.relay_geometry()

    g = segs("radon_PA", 65536, 1024)
    assert g["segments"] == 8 and g["chain_blocks"] == 1024 and g["workgroups_per_cu"] == 2        # the headline launch
    assert segs("radon_PA", 65536, 300)["segments"] == 4
    assert segs("radon_PA", 65536, 255)["segments"] == 1
    assert segs("radon_PA", 32768, 512)["segments"] == 8          # exactly one round
    assert segs("radon_PA", 16384, 512)["segments"] == 1          # half a round (4 lanes per chain: 256 blocks)
    assert segs("radon_PA", 8192, 512)["segments"] == 1           # the 8-GPU shard (8 lanes per chain: 256 blocks)
    assert segs("election", 131072, 512, "hmc")["segments"] == 8
    g = segs("german", 16384, 512, "hmc")                          # 256 blocks: under the two-per-CU round, not even asked
    assert g["segments"] == 1 and g["chain_blocks"] == 256
    g = segs("time_series", 65536, 512, "hmc")                     # 1 024 blocks, but one workgroup holds a CU: stays whole
    assert g["segments"] == 1 and g["workgroups_per_cu"] == 1 and g["chain_blocks"] == 1024
