"""CPU: properties of the oracle's HMC restatement (the TFP internals are not
under /root/reference, so they are pinned by algorithmic properties and by an
independent restatement of the published recurrences)."""
import ctypes as C

import numpy as np
import pytest

import helpers


def _leapfrog(oracle, orc, a, b, L, eps, q, p):
    q = np.ascontiguousarray(q, np.float64).copy(); p = np.ascontiguousarray(p, np.float64).copy()
    eps = np.ascontiguousarray(eps, np.float64)
    lp = C.c_double(0)
    oracle.lib().orc_leapfrog_f64(orc._h, oracle._p(a), oracle._p(b), L, oracle._p(eps), oracle._p(q), oracle._p(p),
                                  C.byref(lp))
    return q, p, lp.value


@pytest.mark.parametrize("mname", ["radon_MN", "8schools", "election"])
def test_leapfrog_reversible_and_second_order(oracle_lib, mname):
    sp = helpers.spec(mname)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "VIP", seed=1)
    rs = np.random.RandomState(0)
    q0 = 0.1 * rs.randn(sp.D); p0 = rs.randn(sp.D)
    lp0, g0 = orc.logp_grad(q0[None], a, b)
    scale = 1.0 / np.sqrt(np.abs(g0[0]).max() + 1.0)
    errs = []
    for h in (1e-2, 5e-3, 2.5e-3):
        eps = np.full(sp.D, h * scale)
        q1, p1, lp1 = _leapfrog(oracle_lib, orc, a, b, 8, eps, q0, p0)
        # reversibility: flip the momentum and integrate back
        q2, p2, _ = _leapfrog(oracle_lib, orc, a, b, 8, eps, q1, -p1)
        np.testing.assert_allclose(q2, q0, rtol=0, atol=1e-9)
        np.testing.assert_allclose(-p2, p0, rtol=0, atol=1e-8)
        errs.append(abs((lp1 - 0.5 * p1 @ p1) - (lp0[0] - 0.5 * p0 @ p0)))
    # energy error shrinks ~4x per halving of the step (second-order integrator)
    assert errs[0] / errs[1] > 3.0 and errs[1] / errs[2] > 3.0


def _run(oracle, sp, orc, kind, q0, eps0, L, n, dtype=np.float64, **kw):
    a, b = helpers.params(sp, kind)
    st = oracle.new_state(q0, dtype)
    orc.hmc_run(st, a, b, eps0, L, n, **kw)
    return st


def test_chunked_run_equals_single_run(oracle_lib):
    sp = helpers.spec("radon_MN")
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "CP")
    q0 = helpers.states(sp, 16, seed=3)
    eps0 = np.full(sp.D, 0.05, np.float32)
    one = oracle_lib.new_state(q0)
    orc.hmc_run(one, a, b, eps0, 4, 12, seed=5, adapt_kind=1, n_adapt=8, lanes=4)
    two = oracle_lib.new_state(q0)
    orc.hmc_run(two, a, b, eps0, 4, 5, seed=5, adapt_kind=1, n_adapt=8, lanes=4)
    orc.hmc_run(two, a, b, eps0, 4, 7, seed=5, adapt_kind=1, n_adapt=8, lanes=4)
    for k in ("q", "grad", "logp", "adapt", "rng", "accept_count"):
        assert np.array_equal(one[k], two[k]), k


def test_trace_schedule_matches_sample_chain(oracle_lib):
    """tfp.mcmc.sample_chain(num_results=S, num_burnin_steps=B, num_steps_between_results=1):
    result r is the state after transition 1 + B + 2r."""
    sp = helpers.spec("8schools")
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "CP")
    q0 = helpers.states(sp, 4, seed=1)
    eps0 = np.full(sp.D, 0.1, np.float32)
    B, S, thin = 3, 4, 2
    total = 1 + B + thin * (S - 1)
    st = oracle_lib.new_state(q0)
    trace = np.zeros((S, 4, sp.D)); acc = np.zeros((S, 4), np.uint8)
    orc.hmc_run(st, a, b, eps0, 3, total, seed=2, n_burnin=B, thin=thin, trace=trace, trace_accept=acc,
                trace_centered=False, lanes=1)
    for r in range(S):
        ref = oracle_lib.new_state(q0)
        orc.hmc_run(ref, a, b, eps0, 3, 1 + B + thin * r, seed=2, lanes=1)
        assert np.array_equal(ref["q"], trace[r])
    assert st["step"] == total


def test_radon_posterior_moments(oracle_lib):
    sp = helpers.spec("radon_MN")
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "CP")
    D = sp.D
    _, g0 = orc.logp_grad(np.zeros((1, D)), a, b)
    _, gI = orc.logp_grad(np.eye(D), a, b)
    P = -(gI - g0)
    mean = np.linalg.solve(P, g0[0]); sd = np.sqrt(np.diag(np.linalg.inv(P)))
    Cn, S = 256, 150
    rs = np.random.RandomState(0)
    q0 = mean + sd * rs.randn(Cn, D)
    eps0 = (0.35 * sd).astype(np.float32)
    st = oracle_lib.new_state(q0)
    trace = np.zeros((S, Cn, D))
    orc.hmc_run(st, a, b, eps0, 8, 50 + S, seed=11, n_burnin=49, thin=1, trace=trace, trace_centered=True, lanes=4)
    acc = st["accept_count"].mean() / st["step"]
    assert 0.6 < acc <= 1.0
    m = trace.mean(axis=(0, 1)); v = trace.reshape(-1, D).std(axis=0)
    # Monte-Carlo error: chains are independent, within-chain samples correlated -> be generous
    assert np.abs((m - mean) / sd).max() < 0.1
    assert np.abs(v / sd - 1).max() < 0.1
    # "posterior means within 1 %" (BASELINE.json) on the three regression coefficients' scale
    assert np.abs(m[:3] - mean[:3]).max() < 0.01 * np.abs(mean[:3]).max() + 0.05 * sd[:3].max()


def _tfp_dual_averaging(eps0, alphas, n_adapt, target=0.75, shrink=0.05, smoothing=10.0, decay=0.75):
    """Published dual-averaging recurrence as TFP's DualAveragingStepSizeAdaptation
    applies it (absolute log step; shrinkage target log(10 eps0)); returns the step
    size used for each transition."""
    log_target = np.log(10.0 * eps0)
    err, log_avg, step = 0.0, 0.0, eps0
    used = []
    for t, al in enumerate(alphas, start=1):
        used.append(step)
        if t <= n_adapt:
            err += target - al
            log_step = log_target - err * np.sqrt(t) / ((t + smoothing) * shrink)
            eta = t ** (-decay)
            log_avg = eta * log_step + (1 - eta) * log_avg
            step = np.exp(log_step)
        else:
            step = np.exp(log_avg)
    return np.array(used)


def test_dual_averaging_recurrence(oracle_lib):
    rs = np.random.RandomState(3)
    alphas = rs.rand(40)
    eps0, n_adapt = 0.037, 25
    ref = _tfp_dual_averaging(eps0, alphas, n_adapt)
    st = (C.c_double * 3)(1.0, 0.0, 0.0)
    used = []
    for n, al in enumerate(alphas, start=1):
        used.append(eps0 * st[0])
        oracle_lib.lib().orc_adapt_update_f64(1, C.c_longlong(n), n_adapt, C.c_double(0.75), C.c_double(0.05),
                                              C.c_double(np.log(al)), st)
    np.testing.assert_allclose(used, ref, rtol=1e-12)


def test_simple_adaptation_recurrence(oracle_lib):
    rs = np.random.RandomState(4)
    alphas = rs.rand(30)
    st = (C.c_double * 3)(1.0, 0.0, 0.0)
    k = 1.0
    for n, al in enumerate(alphas, start=1):
        oracle_lib.lib().orc_adapt_update_f64(2, C.c_longlong(n), 20, C.c_double(0.75), C.c_double(0.05),
                                              C.c_double(np.log(al)), st)
        if n <= 20:
            k = k * 1.05 if al > 0.75 else k / 1.05
        assert abs(st[0] - k) < 1e-12


def test_non_finite_energy_rejects(oracle_lib):
    """A proposal whose energy is not finite must be rejected (TFP safe_sum; inference.py:323-324)."""
    sp = helpers.spec("8schools")
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "CP")
    q0 = helpers.states(sp, 8, seed=1)
    eps0 = np.full(sp.D, 50.0, np.float32)   # absurd step: exp(log_tau) overflows
    st = oracle_lib.new_state(q0, np.float32)
    orc.hmc_run(st, a, b, eps0, 4, 5, seed=3, lanes=1)
    assert np.isfinite(st["q"]).all()
    assert (st["accept_count"] <= 5).all()


def test_interleaved_step_structure(oracle_lib):
    """One interleaved step == CP transition, to_ncp, NCP transition, to_cp, with
    logp/grad re-bootstrapped after each change of coordinates (interleaved.py:113-155).
    Rebuilt here from single hmc_run calls that share the same RNG streams."""
    sp = helpers.spec("radon_MN")
    orc = oracle_lib.OracleModel(sp)
    cp, ncp = helpers.params(sp, "CP"), helpers.params(sp, "NCP")
    q0 = helpers.states(sp, 6, seed=4)
    e0 = np.full(sp.D, 0.05, np.float32); e1 = np.full(sp.D, 0.08, np.float32)
    st = oracle_lib.new_state(q0)
    orc.interleaved_run(st, cp, ncp, e0, e1, 3, 2, 2, seed=21, adapt_kind=2, n_adapt=1, lanes=4)
    # manual: each half step is a fresh single-transition run whose rng/adapt state is carried over by hand
    man = oracle_lib.new_state(q0)
    rng = None; ad = [np.zeros((6, 4)), np.zeros((6, 4))]; ad[0][:, 0] = 1; ad[1][:, 0] = 1
    q = q0.astype(np.float64)
    for n in (1, 2):
        for w, (ab, eps, L) in enumerate(((cp, e0, 3), (ncp, e1, 2))):
            h = oracle_lib.new_state(q)
            h["step"] = n - 1           # transition index drives the adaptation window
            if n - 1 > 0 or w > 0:
                # not the first call: continue the streams; logp/grad must be re-bootstrapped
                h["rng"] = rng
                h["logp"], h["grad"] = orc.logp_grad(q, ab[0], ab[1])
                h["adapt"] = ad[w].copy()
                if n - 1 == 0:          # step_base 0 would reseed: emulate by shifting the window
                    h["step"] = 0
            if h["step"] == 0 and (n - 1 > 0 or w > 0):
                # run with step_base = 1 and an adaptation window one longer (same decisions)
                h["step"] = 1
                orc.hmc_run(h, ab[0], ab[1], eps, L, 1, seed=21, adapt_kind=2, n_adapt=2, lanes=4)
            else:
                orc.hmc_run(h, ab[0], ab[1], eps, L, 1, seed=21, adapt_kind=2, n_adapt=1 if n == 1 else 0, lanes=4)
            rng = h["rng"]; ad[w] = h["adapt"]
            xc = orc.transform(h["q"], ab[0], ab[1], True)
            other = ncp if w == 0 else cp
            q = orc.transform(xc, other[0], other[1], False)
    np.testing.assert_allclose(st["q"], q, rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(st["adapt"][:, 0], ad[0][:, 0], rtol=1e-12)
    np.testing.assert_allclose(st["adapt1"][:, 0], ad[1][:, 0], rtol=1e-12)


def test_streaming_statistics_equal_trace_moments(oracle_lib):
    """arp_hmc_io.stats as the oracle restates it: shifted first / second moments and batch means accumulated
    during the run equal those of the recorded trace; a chunked run equals a single one; partial traces and the
    accepted-among-recorded counters follow the same schedule."""
    sp = helpers.spec("radon_MN")
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "VIP", seed=3)
    C, S, batch, burn, thin, keep = 9, 20, 4, 3, 2, 2
    q0 = helpers.states(sp, C, seed=1, scale=0.1).astype(np.float64)
    eps0 = np.full(sp.D, 0.02, np.float32)
    total = 1 + burn + thin * (S - 1)
    kw = dict(seed=5, n_burnin=burn, thin=thin, trace_centered=True, lanes=8)
    st = oracle_lib.new_state(q0, np.float64)
    tr = np.zeros((S, C, sp.D)); ta = np.zeros((S, C), np.uint8)
    orc.hmc_run(st, a, b, eps0, 3, total, trace=tr, trace_accept=ta, **kw)
    st2 = oracle_lib.new_state(q0, np.float64)
    stats = np.zeros((6, C, sp.D)); racc = np.zeros(C, np.uint32); trk = np.zeros((S, keep, sp.D))
    for n in (2, 11, total - 13):
        orc.hmc_run(st2, a, b, eps0, 3, n, stats=stats, stats_batch=batch, n_samples=S, trace=trk, trace_chains=keep,
                    rec_accept=racc, **kw)
    assert np.array_equal(st["q"], st2["q"]) and np.array_equal(trk, tr[:, :keep])
    assert np.array_equal(racc, ta.sum(axis=0))
    ref, s1, s2, cur, sb1, sb2 = stats
    np.testing.assert_allclose(ref, tr[0])
    np.testing.assert_allclose(ref + s1 / S, tr.mean(axis=0), rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose((s2 - s1 * s1 / S) / (S - 1), tr.var(axis=0, ddof=1), rtol=1e-9, atol=1e-12)
    bm = (tr - ref).reshape(S // batch, batch, C, sp.D).mean(axis=1)
    np.testing.assert_allclose(sb1, bm.sum(axis=0), rtol=1e-10, atol=1e-12)
    np.testing.assert_allclose(sb2, (bm * bm).sum(axis=0), rtol=1e-10, atol=1e-12)


def test_metropolis_margins_explain_the_decisions(oracle_lib):
    """The diagnostics the GPU parity tests lean on: margin = log u - log alpha of every Metropolis test (negative iff
    the proposal was accepted) and escale = the largest energy term that test compared, for orc_hmc_run and for both
    kernels of orc_interleaved_run; the float32 and float64 runs agree on every decision whose margin is not tiny."""
    sp = helpers.spec("radon_PA")
    orc = oracle_lib.OracleModel(sp)
    cp, ncp = helpers.params(sp, "CP"), helpers.params(sp, "NCP")
    Cn, n = 40, 9
    q0 = helpers.states(sp, Cn, seed=3, scale=0.1)
    eps0 = np.full(sp.D, 0.05, np.float32)
    runs = {}
    for dt in (np.float32, np.float64):
        st = oracle_lib.new_state(q0, dt)
        mg = np.full((n, Cn), np.nan, dt); es = np.full((n, Cn), np.nan, dt); ta = np.zeros((n, Cn), np.uint8)
        orc.hmc_run(st, cp[0], cp[1], eps0, 4, n, seed=2, lanes=4, n_burnin=0, thin=1, trace_accept=ta, margin=mg, escale=es)
        assert np.isfinite(es).all() and not np.isnan(mg).any()
        assert np.array_equal(mg < 0, ta.astype(bool))
        assert (es >= np.abs(st["logp"]).min() * 0).all() and es.max() > 1.0
        runs[dt] = (mg, ta, es)
    (m32, a32, _), (m64, a64, e64) = runs[np.float32], runs[np.float64]
    differ = (a32 != a64).any(axis=0)
    # float32 rounding of the energies moves log alpha by a few ulps of their size: only a decision that close can differ
    tol = 1e-3 + 64 * np.finfo(np.float32).eps * e64
    for c in np.where(differ)[0]:
        s = int(np.argmax(a32[:, c] != a64[:, c]))
        assert abs(m64[s, c]) < tol[s, c], (c, s, m64[s, c])
    same = ~differ
    assert (np.abs(m32 - m64)[:, same] <= 8 * tol[:, same]).all()   # trajectories agree; rounding accumulates over the run
    # interleaved: [n_steps, 2, C], kernel 0 = parameterisation 0
    st = oracle_lib.new_state(q0, np.float64)
    mg = np.full((n, 2, Cn), np.nan); es = np.full((n, 2, Cn), np.nan)
    t0 = np.zeros((n, Cn), np.uint8); t1 = np.zeros((n, Cn), np.uint8)
    orc.interleaved_run(st, cp, ncp, eps0, eps0, 3, 2, n, seed=4, lanes=4, n_burnin=0, thin=1, trace_acc0=t0, trace_acc1=t1,
                        margin=mg, escale=es)
    assert np.array_equal(mg[:, 0] < 0, t0.astype(bool)) and np.array_equal(mg[:, 1] < 0, t1.astype(bool))
    assert np.isfinite(es).all()


def test_time_series_stream_partition_in_whole_time_steps(oracle_lib):
    """The random-stream partition of time_series: a slot owns whole (alpha_t, mu_t) time steps, ceil(120 / lanes)
    rounded up to even elements (30 / 16 / 8 at 4 / 8 / 16 lanes per chain, oracle.c: orc_per_lane).  Every latent gets
    a momentum draw whatever the split (one tiny leapfrog step moves EVERY latent), and the splits are different
    streams."""
    sp = helpers.spec("time_series")
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "NCP")
    q0 = helpers.states(sp, 6, seed=3, scale=0.05)
    ends = {}
    for lanes in (4, 8, 16):
        st = oracle_lib.new_state(q0, np.float64)
        orc.hmc_run(st, a, b, np.full(sp.D, 1e-6, np.float64), 1, 1, seed=9, lanes=lanes)
        moved = st["q"] - q0
        assert (st["accept_count"] == 1).all()                 # an (almost) exact integrator step is accepted
        assert (np.abs(moved) > 0).all(), lanes                 # every latent received a momentum
        ends[lanes] = moved
    assert not np.allclose(ends[4], ends[8]) and not np.allclose(ends[8], ends[16])
