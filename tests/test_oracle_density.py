"""CPU: the C oracle's analytic densities against (1) the committed golden vectors
from the reference-shaped float64 autograd restatement, (2) finite differences,
(3) the converter properties the reference's models_test.py intends to pin."""
import os

import numpy as np
import pytest

import helpers

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "density_golden.npz")
MODELS = list(helpers.MODEL_SPECS)
KINDS = ["CP", "NCP", "VIP"]


@pytest.fixture(scope="module")
def gold():
    with np.load(GOLD) as z:
        return {k: z[k] for k in z.files}


@pytest.mark.parametrize("mname", MODELS)
@pytest.mark.parametrize("kind", KINDS)
def test_oracle_matches_golden(oracle_lib, gold, mname, kind):
    sp = helpers.spec(mname)
    orc = oracle_lib.OracleModel(sp)
    x = gold[mname + "/x"]
    a, b = gold["%s/%s/a" % (mname, kind)], gold["%s/%s/b" % (mname, kind)]
    lp, g = orc.logp_grad(x, a, b, dtype=np.float64)
    lp_ref = gold["%s/%s/logp" % (mname, kind)]
    g_ref = gold["%s/%s/grad" % (mname, kind)]
    # sufficient statistics are held in float32 in the oracle, hence 1e-6 relative
    np.testing.assert_allclose(lp + orc.logp_const(b), lp_ref, rtol=1e-7, atol=1e-4)
    scale = np.abs(g_ref).max()
    assert np.abs(g - g_ref).max() <= 2e-6 * scale
    xc = orc.transform(x, a, b, to_centered=True)
    np.testing.assert_allclose(xc, gold["%s/%s/centred" % (mname, kind)], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("mname", MODELS)
def test_float32_oracle_close_to_float64(oracle_lib, mname):
    sp = helpers.spec(mname)
    orc = oracle_lib.OracleModel(sp)
    x = helpers.states(sp, 8, seed=3)
    for kind in KINDS:
        a, b = helpers.params(sp, kind)
        lp64, g64 = orc.logp_grad(x, a, b, dtype=np.float64)
        lp32, g32 = orc.logp_grad(x, a, b, dtype=np.float32)
        assert np.abs(g32 - g64).max() <= 2e-4 * max(1.0, np.abs(g64).max())
        assert np.abs(lp32 - lp64).max() <= 1e-5 * max(1.0, np.abs(lp64).max())


@pytest.mark.parametrize("mname", ["8schools", "radon_MN", "election"])
def test_gradient_finite_differences(oracle_lib, mname):
    sp = helpers.spec(mname)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "VIP", seed=2)
    x = helpers.states(sp, 2, seed=5).astype(np.float64)
    _, g = orc.logp_grad(x, a, b)
    h = 1e-5
    for d in range(0, sp.D, max(1, sp.D // 12)):
        xp, xm = x.copy(), x.copy()
        xp[:, d] += h; xm[:, d] -= h
        fd = (orc.logp_grad(xp, a, b)[0] - orc.logp_grad(xm, a, b)[0]) / (2 * h)
        np.testing.assert_allclose(fd, g[:, d], rtol=2e-5, atol=2e-4)


# --- converter properties (reference models_test.py:29-60) -------------------
@pytest.mark.parametrize("mname", MODELS)
def test_cp_converter_is_identity(oracle_lib, mname):
    sp = helpers.spec(mname)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "CP")
    x = helpers.states(sp, 5, seed=1).astype(np.float64)
    np.testing.assert_allclose(orc.transform(x, a, b, True), x, rtol=0, atol=1e-14)


@pytest.mark.parametrize("mname", MODELS)
@pytest.mark.parametrize("kind", ["NCP", "VIP"])
def test_converter_round_trip_and_determinism(oracle_lib, mname, kind):
    sp = helpers.spec(mname)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, kind)
    x = helpers.states(sp, 5, seed=2).astype(np.float64)
    xt = orc.transform(x, a, b, to_centered=False)
    xt2 = orc.transform(x, a, b, to_centered=False)
    assert np.array_equal(xt, xt2)
    back = orc.transform(xt, a, b, to_centered=True)
    np.testing.assert_allclose(back, x, rtol=1e-6, atol=1e-6)  # (a,b) are float32 inside the oracle


@pytest.mark.parametrize("mname", ["radon_MN", "radon_PA", "8schools"])
def test_density_is_invariant_up_to_jacobian_free_shift(oracle_lib, mname):
    """For radon every latent scale is 1, so the VIP map is a shear with unit
    Jacobian: logp_CP(x) == logp_VIP(xt).  (8 schools is checked on the
    theta-only shear with b = 1.)"""
    sp = helpers.spec(mname)
    orc = oracle_lib.OracleModel(sp)
    a1, b1 = helpers.params(sp, "CP")
    a, _ = helpers.params(sp, "VIP", seed=4)
    b = b1.copy()
    x = helpers.states(sp, 4, seed=9).astype(np.float64)
    xt = orc.transform(x, a, b, to_centered=False)
    lp_cp, _ = orc.logp_grad(x, a1, b1)
    lp_vip, _ = orc.logp_grad(xt, a, b)
    np.testing.assert_allclose(lp_cp, lp_vip, rtol=1e-9, atol=1e-6)


def test_radon_closed_form_posterior(oracle_lib):
    """radon is exactly Gaussian (sigma_y = 1): the posterior mean solves P mean = grad(0).
    Known answers from SURVEY.md 8c / BASELINE.md."""
    expect = {"radon_PA": ((1.3389, -0.0480, -0.1026), (0.4509, 0.5190, 0.0244), 933.9),
              "radon_MN": ((1.4931, -0.0156, -0.6721), (0.1203, 0.2960, 0.0954), 84.6)}
    for mname, (mean_e, sd_e, cond_e) in expect.items():
        sp = helpers.spec(mname)
        orc = oracle_lib.OracleModel(sp)
        a, b = helpers.params(sp, "CP")
        D = sp.D
        _, g0 = orc.logp_grad(np.zeros((1, D)), a, b)
        _, gI = orc.logp_grad(np.eye(D), a, b)
        P = -(gI - g0)
        mean = np.linalg.solve(P, g0[0])
        sd = np.sqrt(np.diag(np.linalg.inv(P)))
        np.testing.assert_allclose(mean[:3], mean_e, atol=6e-5)
        np.testing.assert_allclose(sd[:3], sd_e, atol=6e-5)
        assert abs(np.linalg.cond(P) - cond_e) < 0.06


@pytest.mark.parametrize("mname", MODELS)
def test_parameter_gradients_by_finite_differences(oracle_lib, mname):
    """d logp / d a, d logp / d b (what cVIP optimises) against central differences."""
    sp = helpers.spec(mname)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "VIP", seed=6)
    x = helpers.states(sp, 1, seed=8).astype(np.float64)
    da, db = orc.dparam(x, a, b)
    h = 1e-3   # a, b are float32 inside the oracle
    for d in range(0, sp.D, max(1, sp.D // 10)):
        for which, ref in ((0, da), (1, db)):
            vp, vm = [a.copy(), b.copy()], [a.copy(), b.copy()]
            vp[which][d] += h; vm[which][d] -= h
            step = float(vp[which][d]) - float(vm[which][d])
            cp = orc.logp_grad(x, vp[0], vp[1])[0][0] + orc.logp_const(vp[1])
            cm = orc.logp_grad(x, vm[0], vm[1])[0][0] + orc.logp_const(vm[1])
            fd = (cp - cm) / step
            assert abs(fd - ref[0, d]) <= 2e-3 * (abs(ref[0, d]) + 1.0), (d, which, fd, ref[0, d])
