"""GPU test helper: run the HIP path and the float32 oracle on the same seeds recording EVERY step (state, accept
decisions, the oracle's Metropolis margins) and hold every chain to `helpers.explain_divergence`: step-by-step agreement,
or a first differing decision that sat within rounding of its threshold.  No percentage of chains is waved through."""
import numpy as np
import torch

import helpers


def hmc_every_step(oracle_lib, eng, orc, ab, q0, eps0, L, n, state_tol, what, margin_extra=0.0, **kw):
    """Plain HMC, n transitions from q0 (parameterisation slot 0 of `eng` already set to `ab`).  kw: seed, chain_offset,
    adapt_kind, n_adapt, lanes.  Returns dict(clean, first, st, so, x, xo, acc, acco, margin, escale)."""
    from autoreparam_amd import engine
    gpu = eng.device
    Cn, D = q0.shape
    st = engine.ChainState(torch.as_tensor(q0, device=gpu))
    xs = torch.zeros(n, Cn, D, device=gpu); xa = torch.zeros(n, Cn, dtype=torch.uint8, device=gpu)
    eng.hmc_run(st, eps0, L, n, n_burnin=0, thin=1, trace=xs, trace_accept=xa, trace_centered=False, **kw)
    so = oracle_lib.new_state(q0, np.float32)
    xso = np.zeros((n, Cn, D), np.float32); xao = np.zeros((n, Cn), np.uint8)
    mg = np.zeros((n, Cn), np.float32); es = np.zeros((n, Cn), np.float32)
    orc.hmc_run(so, ab[0], ab[1], eps0, L, n, n_burnin=0, thin=1, trace=xso, trace_accept=xao, trace_centered=False,
                margin=mg, escale=es, **kw)
    scale = np.abs(so["q"]).max() + 1.0
    clean, first = helpers.explain_divergence(xs.cpu().numpy(), xso, xa.cpu().numpy()[:, None], xao[:, None], mg[:, None],
                                              es[:, None], state_tol * scale, extra=margin_extra, what=what)
    return dict(clean=clean, first=first, st=st, so=so, x=xs, xo=xso, acc=xa.cpu().numpy(), acco=xao, margin=mg,
                escale=es, scale=scale)


def interleaved_every_step(oracle_lib, eng, orc, cp, ncp, q0, e0, e1, L0, L1, n, state_tol, what, chunks=None, **kw):
    """Interleaved sampler, n steps from q0 (slots 0 / 1 of `eng` already set to cp / ncp); the GPU run may be cut into
    `chunks` launches.  kw: seed, chain_offset, adapt_kind, n_adapt, lanes."""
    from autoreparam_amd import engine
    gpu = eng.device
    Cn, D = q0.shape
    st = engine.ChainState(torch.as_tensor(q0, device=gpu))
    xs = torch.zeros(n, Cn, D, device=gpu)
    a0 = torch.zeros(n, Cn, dtype=torch.uint8, device=gpu); a1 = torch.zeros(n, Cn, dtype=torch.uint8, device=gpu)
    for m in (chunks or [n]):
        eng.interleaved_run(st, e0, e1, L0, L1, m, n_burnin=0, thin=1, trace=xs, trace_accept0=a0, trace_accept1=a1,
                            trace_centered=False, **kw)
    assert st.step == n
    so = oracle_lib.new_state(q0, np.float32)
    xso = np.zeros((n, Cn, D), np.float32); a0o = np.zeros((n, Cn), np.uint8); a1o = np.zeros((n, Cn), np.uint8)
    mg = np.zeros((n, 2, Cn), np.float32); es = np.zeros((n, 2, Cn), np.float32)
    orc.interleaved_run(so, cp, ncp, e0, e1, L0, L1, n, n_burnin=0, thin=1, trace=xso, trace_acc0=a0o, trace_acc1=a1o,
                        trace_centered=False, margin=mg, escale=es, **kw)
    scale = np.abs(so["q"]).max() + 1.0
    acc = np.stack([a0.cpu().numpy(), a1.cpu().numpy()], axis=1)
    acco = np.stack([a0o, a1o], axis=1)
    clean, first = helpers.explain_divergence(xs.cpu().numpy(), xso, acc, acco, mg, es, state_tol * scale, what=what)
    return dict(clean=clean, first=first, st=st, so=so, x=xs, xo=xso, acc=acc, acco=acco, margin=mg, escale=es, scale=scale)


def _snapshot(st, dtype=np.float32):
    """The oracle-side copy of an engine.ChainState (same layout as oracle.new_state)."""
    f = lambda t: np.ascontiguousarray(t.cpu().numpy())
    return dict(q=f(st.q).astype(dtype), grad=f(st.grad).astype(dtype), logp=f(st.logp).astype(dtype),
                adapt=f(st.adapt).astype(dtype), adapt1=f(st.adapt1).astype(dtype),
                rng=f(st.rng).view(np.uint32).copy(), accept_count=f(st.accept_count).view(np.uint32).copy(),
                accept_count1=f(st.accept_count1).view(np.uint32).copy(), step=int(st.step))


def hmc_teacher_forced(oracle_lib, eng, orc, ab, q0, eps0, L, n, what, adapt_kind, n_adapt, state_tol=1e-4, **kw):
    """Adaptive runs cannot be compared as free-running trajectories: dual averaging feeds exp(log alpha) back into the
    step size and explores up to the integrator's stability limit, where rounding differences grow from step to step.
    Here the oracle is RESTARTED FROM THE HIP PATH'S OWN STATE before every transition (state, cached gradient and log
    density, adaptation state, random streams), so each transition is compared in isolation, for EVERY chain:
      * the accept decision is equal, or its margin in the oracle's run is within `helpers.margin_tol` (= tau, the float32
        rounding of the energies compared);
      * with equal decisions the new state agrees to `state_tol` (bitwise the old one after a rejection), and the cached
        log density to tau;
      * the adaptation state follows the recurrence within what tau allows: |d error-sum| <= tau,
        |d log kappa| <= tau sqrt(t) / (0.05 (t + 10)) (dual averaging), kappa to 1e-6 relative (simple adaptation: a
        product of exact factors, decided by the same comparison as the Metropolis test's margin).
    Returns the number of decisions that differed (all explained)."""
    from autoreparam_amd import engine
    gpu = eng.device
    Cn, D = q0.shape
    st = engine.ChainState(torch.as_tensor(q0, device=gpu))
    flipped = 0
    for s in range(n):
        so = _snapshot(st)
        q_before = so["q"].copy()
        eng.hmc_run(st, eps0, L, 1, adapt_kind=adapt_kind, n_adapt=n_adapt, **kw)
        mg = np.zeros((1, Cn), np.float32); es = np.zeros((1, Cn), np.float32); la = np.zeros((1, Cn), np.float32)
        orc.hmc_run(so, ab[0], ab[1], eps0, L, 1, adapt_kind=adapt_kind, n_adapt=n_adapt, margin=mg, escale=es,
                    log_alpha=la, **kw)
        tau = helpers.margin_tol(es[0])
        sg = _snapshot(st)
        acc_g = sg["accept_count"].astype(np.int64) - (0 if s == 0 else prev_acc)
        acc_o = so["accept_count"].astype(np.int64) - (0 if s == 0 else prev_acc)
        prev_acc = sg["accept_count"].astype(np.int64)
        same = acc_g == acc_o
        unexplained = ~same & ~(np.abs(mg[0]) <= tau)
        assert not unexplained.any(), "%s step %d: decisions differ with margins %s > %s" % (
            what, s, mg[0][unexplained][:4], tau[unexplained][:4])
        flipped += int((~same).sum())
        scale = np.abs(so["q"]).max() + 1.0
        err = np.abs(sg["q"] - so["q"]).max(axis=1)
        assert (err[same] <= state_tol * scale).all(), (what, s, np.sort(err[same])[-3:])
        rej = same & (acc_g == 0)
        assert np.array_equal(sg["q"][rej], q_before[rej]), (what, s)            # a rejection leaves the state bitwise alone
        assert (np.abs(sg["logp"] - so["logp"])[same] <= tau[same] + 1e-5 * np.abs(so["logp"][same])).all(), (what, s)
        assert np.array_equal(sg["rng"][:, :kw.get("lanes", 0) or 16], so["rng"][:, :kw.get("lanes", 0) or 16]), (what, s)
        t = float(s + 1)
        ad_g, ad_o = sg["adapt"].astype(np.float64), so["adapt"].astype(np.float64)
        if adapt_kind == 1:
            gain = np.sqrt(t) / (0.05 * (t + 10.0)) if t <= n_adapt else 0.0
            assert (np.abs(ad_g[:, 1] - ad_o[:, 1]) <= tau + 1e-6).all(), (what, s, "error sum")
            dl = np.abs(np.log(ad_g[:, 0]) - np.log(ad_o[:, 0]))
            assert (dl <= tau * max(gain, 1.0) + 3e-5).all(), (what, s, "log kappa", dl.max(), (tau * gain).max())
            assert (np.abs(ad_g[:, 2] - ad_o[:, 2]) <= tau * max(gain, 1.0) + 3e-5).all(), (what, s, "log-averaged kappa")
        elif adapt_kind == 2:
            # kappa moves by an exact factor chosen by log alpha > log(target): a comparison like the Metropolis one,
            # held to the same tolerance
            other = np.abs(ad_g[:, 0] / ad_o[:, 0] - 1) > 1e-6
            assert (np.abs(la[0] - np.log(0.75))[other] <= tau[other]).all(), (what, s, la[0][other], tau[other])
    return flipped
