"""GPU: HIP logp/grad and state converters, called through the C ABI, against the
C oracle (float64) and the committed golden vectors."""
import os

import numpy as np
import pytest

import helpers

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "density_golden.npz")
MODELS = ["8schools", "radon_MN", "radon_PA", "election", "german", "radon_sd_MN", "funnel", "electric", "time_series"]
LANES = {"8schools": [1, 2, 4, 8], "radon_MN": [4, 8, 16], "radon_PA": [4, 8, 16], "election": [4, 8, 16], "german": [4, 8, 16], "radon_sd_MN": [8, 16], "funnel": [1], "electric": [8, 16], "time_series": [4, 8, 16]}


@pytest.fixture(scope="module")
def engines(gpu):
    from autoreparam_amd import engine
    cache = {}

    def get(mname):
        if mname not in cache:
            cache[mname] = engine.Engine(helpers.spec(mname), gpu)
        return cache[mname]
    return get


def _tol(g):
    return 3e-5 * max(1.0, float(np.abs(g).max()))


@pytest.mark.parametrize("mname", MODELS)
@pytest.mark.parametrize("kind", ["CP", "NCP", "VIP"])
def test_logp_grad_matches_oracle(oracle_lib, engines, mname, kind):
    sp = helpers.spec(mname)
    eng = engines(mname)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, kind)
    eng.set_param(0, (a, b))
    for n in (1, 7, 130):  # ragged: fewer chains than a wave, not a multiple of the group size
        x = helpers.states(sp, n, seed=n)
        lp_o, g_o = orc.logp_grad(x, a, b, dtype=np.float64)
        for lanes in LANES[mname]:
            lp, g = eng.logp_grad(x, which=0, lanes=lanes)
            lp, g = lp.cpu().numpy(), g.cpu().numpy()
            # float32 arithmetic on |logp| ~ 1e3..1e5: a few ulp of the largest term
            assert np.abs(lp - lp_o).max() <= 2e-6 * max(1.0, np.abs(lp_o).max()) + 1e-3, (lanes, n)
            assert np.abs(g - g_o).max() <= _tol(g_o), (lanes, n)
            assert abs(eng.logp_const(0) - orc.logp_const(b)) < 1e-6 * abs(orc.logp_const(b)) + 1e-9


@pytest.mark.parametrize("mname", MODELS)
def test_against_golden_vectors(engines, mname):
    sp = helpers.spec(mname)
    eng = engines(mname)
    with np.load(GOLD) as z:
        x = z[mname + "/x"]
        for kind in ("CP", "NCP", "VIP"):
            a, b = z["%s/%s/a" % (mname, kind)], z["%s/%s/b" % (mname, kind)]
            eng.set_param(0, (a, b))
            lp, g = eng.logp_grad(x.astype(np.float32))
            lp_ref, g_ref = z["%s/%s/logp" % (mname, kind)], z["%s/%s/grad" % (mname, kind)]
            assert np.abs(lp.cpu().numpy() + eng.logp_const(0) - lp_ref).max() <= 3e-6 * np.abs(lp_ref).max() + 1e-3
            assert np.abs(g.cpu().numpy() - g_ref).max() <= _tol(g_ref)
            xc = eng.transform(x.astype(np.float32), which=0, to_centered=True).cpu().numpy()
            np.testing.assert_allclose(xc, z["%s/%s/centred" % (mname, kind)], rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("mname", MODELS)
def test_converter_properties(engines, mname):
    """models_test.py:29-60: CP map is the identity, maps are deterministic, round trip."""
    sp = helpers.spec(mname)
    eng = engines(mname)
    x = helpers.states(sp, 257, seed=4)
    eng.set_param(0, "CP")
    # identity up to float32 rounding of mu + (x - mu), as in the reference's own float32 graph
    np.testing.assert_allclose(eng.transform(x, 0, True).cpu().numpy(), x, rtol=0, atol=2e-7 * 4)
    for kind in ("NCP", "VIP"):
        eng.set_param(1, helpers.params(sp, kind))
        t1 = eng.transform(x, 1, to_centered=False)
        t2 = eng.transform(x, 1, to_centered=False)
        assert np.array_equal(t1.cpu().numpy(), t2.cpu().numpy())
        back = eng.transform(t1, 1, to_centered=True).cpu().numpy()
        # time_series: the trend is a double cumulative sum over 60 steps, so the ~1e-6 relative error of
        # exp(-log S) exp(log S) per step arrives 20 - 30 times larger at the last steps (measured on the worst chain of
        # this set: 1.5e-5 / 2.1e-5 / 2.7e-5 at 4 / 8 / 16 lanes per chain -- this call takes the 16-lane instantiation)
        tol = 5e-5 if mname == "time_series" else 2e-5
        np.testing.assert_allclose(back, x, rtol=tol, atol=tol)


def test_missing_param_and_bad_lanes_fail(engines):
    eng = engines("radon_MN")
    x = helpers.states(helpers.spec("radon_MN"), 4)
    with pytest.raises(RuntimeError):
        eng.logp_grad(x, lanes=3)
    with pytest.raises(RuntimeError):
        eng.logp_grad(x, lanes=1)  # no 1-lane instantiation for radon


@pytest.mark.parametrize("n_obs", [1, 17, 64, 65, 128, 129, 300, 513, 1000])
def test_german_observation_tiles(oracle_lib, gpu, n_obs):
    """German credit's matrix-core likelihoods (4 lanes per chain) stream the design matrix in tiles through two LDS
    buffers -- 64 observations per tile on bf16 matrix cores with three-piece operands (the default where the data allow
    it: `german_math`), 128 on f32 matrix cores: one tile, an odd and an even number of tiles, tiles that end inside a
    16-row block -- log density and gradient against the float64 oracle on the truncated data set, SAME tolerances for
    both, and the 8- and 16-lane paths beside them."""
    import copy
    from autoreparam_amd import engine
    full = helpers.spec("german")
    sp = copy.copy(full)
    sp.raw = dict(full.raw); sp.raw["X"] = full.raw["X"][:n_obs]; sp.raw["y"] = full.raw["y"][:n_obs]
    sp.observed = {"y": sp.raw["y"][None]}
    eng = engine.Engine(sp, gpu)
    orc = oracle_lib.OracleModel(sp)
    for kind in ("NCP", "VIP"):
        a, b = helpers.params(sp, kind)
        eng.set_param(0, (a, b))
        x = helpers.states(sp, 37, seed=n_obs)
        lp_o, g_o = orc.logp_grad(x, a, b, dtype=np.float64)
        for lanes, math in ((4, "bf16x3"), (4, "f32"), (8, "auto"), (16, "auto")):
            eng.set_option("german_math", math)
            lp, g = eng.logp_grad(x, which=0, lanes=lanes)
            lp, g = lp.cpu().numpy(), g.cpu().numpy()
            assert np.abs(lp - lp_o).max() <= 2e-6 * max(1.0, np.abs(lp_o).max()) + 1e-3, (lanes, math, kind)
            assert np.abs(g - g_o).max() <= _tol(g_o), (lanes, math, kind)


def test_german_math_option(oracle_lib, gpu):
    """arp_model_set_option("german_math", ...): auto picks the bf16 x 3 likelihood for the reference's data (7 columns
    need three pieces), "f32" keeps the f32 matrix cores; a design matrix with more than 8 non-bf16 columns has no bf16
    image (auto = f32, "bf16x3" refused); unknown keys / values / models fail loudly.  Both forms meet the oracle, and the
    bf16 form is at least as close to it as the f32 one (its products are exact, only the accumulation rounds)."""
    import copy
    from autoreparam_amd import engine
    sp = helpers.spec("german")
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "VIP")
    x = helpers.states(sp, 256, seed=5, scale=0.3)
    lp_o, g_o = orc.logp_grad(x, a, b, dtype=np.float64)
    err = {}
    eng = engine.Engine(sp, gpu)
    eng.set_param(0, (a, b))
    for math in ("f32", "bf16x3", "auto"):
        eng.set_option("german_math", math)
        lp, g = eng.logp_grad(x, lanes=4)
        err[math] = np.abs(g.cpu().numpy() - g_o).max() / np.abs(g_o).max()
        assert err[math] <= 2e-6
    assert err["auto"] == err["bf16x3"] and err["bf16x3"] <= 1.5 * err["f32"]
    with pytest.raises(RuntimeError):
        eng.set_option("german_math", "fp8")
    with pytest.raises(RuntimeError):
        eng.set_option("no_such_key", "1")
    with pytest.raises(RuntimeError):
        engine.Engine(helpers.spec("radon_MN"), gpu).set_option("german_math", "f32")
    # a dense design matrix: every column needs three pieces
    dense = copy.copy(sp)
    dense.raw = dict(sp.raw)
    rs = np.random.RandomState(0)
    dense.raw["X"] = (sp.raw["X"] + 0.01 * rs.randn(*sp.raw["X"].shape)).astype(np.float32)
    e2 = engine.Engine(dense, gpu)
    e2.set_param(0, (a, b))
    with pytest.raises(RuntimeError):
        e2.set_option("german_math", "bf16x3")
    o2 = oracle_lib.OracleModel(dense)
    lp2, g2 = e2.logp_grad(x, lanes=4)                         # auto: the f32 matrix cores
    _, g2o = o2.logp_grad(x, a, b, dtype=np.float64)
    assert np.abs(g2.cpu().numpy() - g2o).max() <= 2e-6 * np.abs(g2o).max()


@pytest.mark.parametrize("ds", ["IN", "MO", "ND", "MA", "AZ"])
def test_other_radon_datasets(oracle_lib, gpu, ds):
    """The reference's remaining radon data sets (91, 115, 53, 13 and 15 counties: every state of srrs2.dat that README.md:22
    lists; main.py --dataset; the two small ones run the generic lane kernels at 8 / 16 lanes per chain): every lanes-per-chain split
    the library instantiates for them against the float64 oracle, the library's own choice included, and a short
    interleaved run against the float32 oracle."""
    import torch
    from autoreparam_amd import engine, _lib
    sp = helpers.spec("radon_" + ds)
    eng = engine.Engine(sp, gpu)
    orc = oracle_lib.OracleModel(sp)
    x = helpers.states(sp, 70, seed=3)
    served = []
    for kind in ("CP", "NCP", "VIP"):
        a, b = helpers.params(sp, kind)
        eng.set_param(0, (a, b))
        lp_o, g_o = orc.logp_grad(x, a, b, dtype=np.float64)
        for lanes in (0, 4, 8, 16):
            try:
                lp, g = eng.logp_grad(x, which=0, lanes=lanes)
            except RuntimeError:
                assert lanes != 0          # the default must always be served
                continue
            served.append(lanes)
            assert np.abs(lp.cpu().numpy() - lp_o).max() <= 2e-6 * max(1.0, np.abs(lp_o).max()) + 1e-3, (lanes, kind)
            assert np.abs(g.cpu().numpy() - g_o).max() <= _tol(g_o), (lanes, kind)
    assert len(set(served) - {0}) >= 2, served
    # interleaved CP / NCP, 6 steps, against the float32 oracle on the same seeds and the same lanes per chain
    cp, ncp = helpers.params(sp, "CP"), helpers.params(sp, "NCP")
    eng.set_param(0, cp); eng.set_param(1, ncp)
    q0 = helpers.states(sp, 64, seed=5, scale=0.1)
    e = np.full(sp.D, 0.02, np.float32)
    lanes = sorted(set(served) - {0})[0]
    import parity
    r = parity.interleaved_every_step(oracle_lib, eng, orc, cp, ncp, q0, e, e, 3, 3, 6, 2e-4, "radon_%s lanes=%d" % (ds, lanes),
                                      seed=11, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=4, lanes=lanes)
    ok = r["clean"]
    err = np.abs(r["st"].q.cpu().numpy() - r["so"]["q"]).max(axis=1) / r["scale"]
    assert (err[ok] <= 2e-4).all(), np.sort(err[ok])[-5:]
    assert np.array_equal(r["st"].accept_count.cpu().numpy()[ok], r["so"]["accept_count"][ok])


def test_radon_stddvs_on_a_small_state(oracle_lib, gpu):
    """radon_stddvs --dataset=AZ (15 counties: one or two per lane): density and gradient at both lane splits against the
    float64 oracle, and a VI fit against the oracle's timeline (the widest split, one county per lane)."""
    import torch
    from autoreparam_amd import engine
    sp = helpers.spec("radon_sd_AZ")
    eng = engine.Engine(sp, gpu)
    orc = oracle_lib.OracleModel(sp)
    x = helpers.states(sp, 70, seed=3)
    for kind in ("CP", "NCP", "VIP"):
        a, b = helpers.params(sp, kind)
        eng.set_param(0, (a, b))
        lp_o, g_o = orc.logp_grad(x, a, b, dtype=np.float64)
        for lanes in (0, 8, 16):
            lp, g = eng.logp_grad(x, which=0, lanes=lanes)
            assert np.abs(lp.cpu().numpy() - lp_o).max() <= 2e-6 * max(1.0, np.abs(lp_o).max()) + 1e-3, (lanes, kind)
            assert np.abs(g.cpu().numpy() - g_o).max() <= _tol(g_o), (lanes, kind)
    a, b = helpers.params(sp, "NCP")
    eng.set_param(0, (a, b))
    rs = np.random.RandomState(0)
    loc0 = (1e-2 * rs.randn(1, sp.D)).astype(np.float32); rho0 = np.full((1, sp.D), -2.0, np.float32)
    loc = torch.as_tensor(loc0.copy(), device=gpu); rho = torch.as_tensor(rho0.copy(), device=gpu)
    elbo = eng.vi_run([0.05], loc, rho, 80, 256, seed=4).cpu().numpy()
    lo, ro = loc0.copy(), rho0.copy()
    elbo_o = orc.vi_run(a, b, [0.05], lo, ro, None, 80, 256, seed=4, lanes=16)
    np.testing.assert_allclose(elbo[:, :5], elbo_o[:, :5], rtol=2e-5, atol=2e-2)
    np.testing.assert_allclose(elbo[:, -16:].mean(1), elbo_o[:, -16:].mean(1), rtol=2e-3, atol=0.5)


def test_german_bf16x3_on_adversarial_design_matrices(oracle_lib, gpu):
    """The bf16 x 3 likelihood away from the reference's data: EIGHT split columns (the most its extra K = 32 step holds)
    with scales from 1e-3 to 1e3, columns that are exact in one bf16 piece without being 0/1 (0.5, 1.5, -2, 96), an
    all-zero column, 333 observations (five whole tiles and 13 rows) -- log density and gradient against the float64
    oracle at the usual tolerances, beside the f32 matrix-core form; nine split columns: no bf16 image."""
    import copy
    from autoreparam_amd import engine
    full = helpers.spec("german")
    rs = np.random.RandomState(7)
    N, F = 333, full.raw["X"].shape[1]
    X = (rs.rand(N, F) < 0.3).astype(np.float32)                                  # zeros and ones
    X[:, 0] = 1.0
    scales = [1e-3, 0.05, 1.0, 3.7, 37.5, 250.0, 1e3, 0.3]
    for q, s_ in enumerate(scales):
        X[:, 1 + q] = (s_ * rs.randn(N)).astype(np.float32)                       # eight columns that need three pieces
    X[:, 20] = rs.choice([0.5, 1.5, -2.0, 96.0], N)                              # exact in one piece, not 0/1
    X[:, 21] = 0.0
    y = (rs.rand(N) < 0.4).astype(np.float32)
    sp = copy.copy(full)
    sp.raw = dict(full.raw); sp.raw["X"] = X; sp.raw["y"] = y
    sp.observed = {"y": y[None]}
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "VIP")
    x = helpers.states(sp, 64, seed=11, scale=0.05)          # small coefficients: the logits stay out of saturation for most rows
    lp_o, g_o = orc.logp_grad(x, a, b, dtype=np.float64)
    eng = engine.Engine(sp, gpu)
    eng.set_param(0, (a, b))
    for math in ("bf16x3", "f32"):
        eng.set_option("german_math", math)
        lp, g = eng.logp_grad(x, lanes=4)
        lp, g = lp.cpu().numpy(), g.cpu().numpy()
        assert np.abs(lp - lp_o).max() <= 2e-6 * max(1.0, np.abs(lp_o).max()) + 1e-3, math
        assert np.abs(g - g_o).max() <= _tol(g_o), math
    X9 = X.copy(); X9[:, 30] = rs.randn(N).astype(np.float32)                     # a ninth split column
    sp9 = copy.copy(sp); sp9.raw = dict(sp.raw); sp9.raw["X"] = X9
    e9 = engine.Engine(sp9, gpu)
    with pytest.raises(RuntimeError):
        e9.set_option("german_math", "bf16x3")
