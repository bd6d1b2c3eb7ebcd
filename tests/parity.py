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
