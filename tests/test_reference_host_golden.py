"""CPU: the package's data fixtures and host bookkeeping against output of the REFERENCE'S OWN CODE.

tests/golden/reference_host_golden.npz was written by tests/golden/make_reference_host_golden.py, which AST-extracts the
TensorFlow-free functions of /root/reference (loaders models.py:706-760, 860-881; util.py:65-88, 271-276, 308-331,
394-460; main.py:292-294; every flags.DEFINE_*) and executes them in the build container.  This is the part of SURVEY.md
8(c) that can be pinned without TF/TFP: what the reference's models are GIVEN (so that the oracle and the HIP kernels
evaluate the reference's posterior, not a look-alike) and what its CLI does around the sampler.  Integer and float32
data bit for bit; the TFP arithmetic of the hot path is NOT covered here (tests/test_reference_golden.py, skipped
without TF)."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
DATA = os.path.join(ROOT, "autoreparam_amd", "data")


@pytest.fixture(scope="module")
def ref():
    return np.load(os.path.join(HERE, "golden", "reference_host_golden.npz"))


def same(a, b):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(a, b), np.abs(a.astype(np.float64) - b.astype(np.float64)).max()


@pytest.mark.parametrize("state", ["MN", "PA", "IN", "MO", "ND", "MA", "AZ"])
def test_radon_fixture_is_the_reference_loaders_output(ref, state):
    d = np.load(os.path.join(DATA, "radon_%s.npz" % state))
    c = ref["data/radon_%s/c" % state]
    same(d["county"], c.astype(np.int32))
    assert int(c.max()) + 1 == len(d["u"]) == len(np.unique(c))
    same(d["u"], ref["data/radon_%s/u" % state])                         # float32, log uranium with the look-up quirk
    same(d["x"], ref["data/radon_%s/x" % state])
    same(d["y"], ref["data/radon_%s/data" % state].reshape(-1))          # the loader returns it as [N, 1]
    assert ref["data/radon_%s/data" % state].dtype == np.float32 and d["y"].dtype == np.float32


def test_german_fixture_is_the_reference_loaders_output(ref):
    d = np.load(os.path.join(DATA, "german_credit.npz"))
    num, cat, status = ref["data/german/numericals"], ref["data/german/categoricals"], ref["data/german/status"]
    same(d["numericals"], num.astype(np.float32))                         # models.py:889: numericals.astype(np.float32)
    same(d["categoricals"], cat.astype(np.int32))
    same(d["y"], status)
    same(d["cat_sizes"], cat.max(axis=0) + 1)
    # the design matrix the model builds from them (models.py:889-891: numerics, then one one-hot block of depth
    # c.max()+1 per categorical column, in column order)
    blocks = [num.astype(np.float32)] + [np.eye(int(c.max()) + 1, dtype=np.float32)[c] for c in cat.T]
    same(d["X"], np.concatenate(blocks, axis=1))
    assert d["X"].shape == (1000, 62)


def test_election_and_electric_fixtures_are_the_reference_data_modules(ref):
    d = np.load(os.path.join(DATA, "election88.npz"))
    assert int(d["n_state"]) == int(ref["data/election88/n_state"]) and len(d["y"]) == int(ref["data/election88/N"])
    for k in ("state", "female", "black", "y"):
        same(d[k].astype(np.float64), ref["data/election88/" + k].astype(np.float64))
    d = np.load(os.path.join(DATA, "electric.npz"))
    for k in ("n_pair", "n_grade", "n_grade_pair"):
        assert int(d[k]) == int(ref["data/electric/" + k])
    assert len(d["y"]) == int(ref["data/electric/N"])
    for k in ("pair", "grade", "grade_pair"):
        same(d[k].astype(np.int64), ref["data/electric/" + k].astype(np.int64))
    for k in ("treatment", "y"):
        same(d[k], ref["data/electric/" + k].astype(np.float32))         # the graph holds them as float32 constants


def test_eight_schools_and_time_series_fixtures_are_the_reference_constants(ref):
    d = np.load(os.path.join(DATA, "eight_schools.npz"))
    same(d["y"], ref["data/eight_schools/treatment_effects"])
    same(d["sigma"], ref["data/eight_schools/treatment_stddevs"])
    d = np.load(os.path.join(DATA, "time_series.npz"))
    same(d["x"], ref["data/time_series/x"].astype(np.float32))
    same(d["y"], ref["data/time_series/y"].astype(np.float32))


def _vp(ref):
    import collections
    vp = collections.OrderedDict()
    for k in ("mu", "log_tau", "theta", "m"):
        vp[k + "_loc"] = ref["util/vp/%s_loc" % k]
        vp[k + "_scale"] = ref["util/vp/%s_scale" % k]
    return vp


def test_step_size_helpers(ref):
    from autoreparam_amd import util
    vp = _vp(ref)
    for L in (1, 4, 7):
        got = util.get_approximate_step_size(vp, L)
        assert len(got) == 4
        for i, g in enumerate(got):
            same(g, ref["util/approx_step/L%d/%d" % (L, i)])
        got = util.stddvs_to_mcmc_step_sizes(vp, L)
        for i, g in enumerate(got):
            same(g, ref["util/stddvs_step/L%d/%d" % (L, i)])


def test_variational_inits_draw_the_reference_population(ref):
    """util.py:394-410 with numpy's global generator seeded: same draws in the same order, float32"""
    from autoreparam_amd import util
    vp = _vp(ref)
    names = ["mu", "log_tau", "theta", "m"]
    np.random.seed(7)
    got = util.variational_inits_from_params(vp, names, 6)               # unseeded form = the reference's global RNG
    got_seeded = util.variational_inits_from_params(vp, names, 6, seed=7)
    assert list(got) == names
    for k in names:
        want = ref["util/inits/seed7_n6/" + k]
        assert got[k].dtype == np.float32
        same(got[k], want)
        same(got_seeded[k], want)                                        # RandomState(7) is the same stream


def test_ess_summaries(ref):
    from autoreparam_amd import util
    ess = [ref["util/ess_in/%d" % i] for i in range(3)]
    same(np.asarray(util.get_min_ess([e.copy() for e in ess]), np.float64), ref["util/get_min_ess"])
    by_chain = [[e[c] for e in ess] for c in range(ess[0].shape[0])]
    same(np.asarray(util.get_min_ess_other(by_chain), np.float64), ref["util/get_min_ess_other"])
    same(util.reject_outliers(ref["util/reject_outliers/in"]), ref["util/reject_outliers/out"])
    groups = [[ref["util/true_mean/in/%d/%d" % (g, j)] for j in range(2)] for g in range(3)]
    tm = util.estimate_true_mean(groups, list(ref["util/true_mean/esss"]))
    for g in range(3):
        np.testing.assert_allclose(np.asarray(tm[g], np.float64), ref["util/true_mean/out/%d" % g], rtol=1e-15, atol=0)


def test_two_variable_hierarchy_formulas(ref):
    from autoreparam_amd import util
    qv = ref["util/qv"]
    for name in ("compute_V_cp", "compute_V_ncp", "condition_number_cp", "condition_number_ncp"):
        got = np.array([getattr(util, name)(q, v) for q, v in qv])
        np.testing.assert_allclose(got, ref["util/" + name], rtol=1e-12, atol=0)      # closed forms, float64


def test_best_leapfrog_count_from_tuning_runs(ref):
    from autoreparam_amd import main
    runs = [{"num_leapfrog_steps": int(L), "ess_min": float(e)} for L, e in ref["main/tuning_runs"]]
    assert main.get_best_num_leapfrog_steps_from_tuning_runs(runs) == int(ref["main/best_num_leapfrog_steps"])
    # ties go to the first run in file order, as max() does (main.py:292-294)
    assert main.get_best_num_leapfrog_steps_from_tuning_runs(runs[::-1]) == int(ref["main/best_num_leapfrog_steps_reversed"])


def test_cli_flags_are_the_references(ref):
    """every flag the reference's CLI defines exists here under the same name, kind and default"""
    from autoreparam_amd import flags
    table = json.load(open(os.path.join(HERE, "golden", "reference_host_flags.json")))
    mine = {n: (t, d) for n, t, d, _ in flags._DEFS}
    kinds = {"string": str, "boolean": bool, "integer": int, "list": list}
    assert len(table) == 20
    for name, (kind, default, _) in table.items():
        assert name in mine, name
        assert mine[name][0] is kinds[kind], name
        assert mine[name][1] == default, (name, mine[name][1], default)
