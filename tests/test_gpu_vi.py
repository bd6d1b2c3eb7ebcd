"""GPU: the single-launch mean-field VI kernel against the float32 oracle
restatement of find_best_learning_rate's optimisation loop (same draws)."""
import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu
VI_LANES = {"8schools": 8, "radon_MN": 16, "election": 16, "german": 4, "radon_sd_MN": 16, "funnel": 1, "electric": 16, "time_series": 4}


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
