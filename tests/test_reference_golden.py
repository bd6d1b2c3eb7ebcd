"""CPU: the oracle against output of the REFERENCE itself (tests/golden/reference_golden.npz, written by
tests/golden/make_reference_golden.py on a machine that has TensorFlow 1.x + TFP).  The file cannot be produced in the
build container, so these tests are skipped until someone commits it; they are what turns the
"parity unpinned" caveat into "pinned" -- every TFP-internal restatement (densities through the interceptors,
dual averaging, simple adaptation, the sample_chain schedule, effective_sample_size) has its check here."""
import ctypes as C
import os

import numpy as np
import pytest

import helpers

PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "reference_golden.npz")
pytestmark = pytest.mark.skipif(not os.path.exists(PATH), reason="reference_golden.npz has not been generated "
                                "(needs TensorFlow 1.x + TFP: tests/golden/make_reference_golden.py)")


@pytest.fixture(scope="module")
def ref():
    return np.load(PATH)


@pytest.mark.parametrize("mname", list(helpers.MODEL_SPECS))
@pytest.mark.parametrize("kind", ["CP", "NCP", "VIP"])
def test_density_gradient_converters(oracle_lib, ref, mname, kind):
    gold = np.load(os.path.join(os.path.dirname(PATH), "density_golden.npz"))
    sp = helpers.spec(mname)
    orc = oracle_lib.OracleModel(sp)
    x = gold[mname + "/x"].astype(np.float32).astype(np.float64)
    a, b = gold["%s/%s/a" % (mname, kind)], gold["%s/%s/b" % (mname, kind)]
    lp, g = orc.logp_grad(x, a, b)
    lp = lp + orc.logp_const(b)
    xc = orc.transform(x, a, b, to_centered=True)
    np.testing.assert_allclose(lp, ref["density/%s/%s/logp" % (mname, kind)], rtol=2e-5, atol=2e-2)     # float32 reference
    gr = ref["density/%s/%s/grad" % (mname, kind)]
    np.testing.assert_allclose(g, gr, rtol=2e-4, atol=2e-4 * (np.abs(gr).max() + 1))
    cr = ref["density/%s/%s/centred" % (mname, kind)]
    np.testing.assert_allclose(xc, cr, rtol=1e-5, atol=1e-5 * (np.abs(cr).max() + 1))


@pytest.mark.parametrize("tag,kind", [("dual", 1), ("simple", 2)])
def test_step_size_adaptation(oracle_lib, ref, tag, kind):
    la, eps0, sizes = ref[tag + "/log_accept"], ref[tag + "/eps0"], ref[tag + "/step_size"]
    n_adapt = int(ref[tag + "/num_adaptation_steps"])
    upd = oracle_lib.lib().orc_adapt_update_f64
    got = np.zeros_like(sizes, dtype=np.float64)
    for i in range(la.shape[1]):
        st = (C.c_double * 3)(1.0, 0.0, 0.0)
        for t in range(la.shape[0]):
            upd(kind, C.c_longlong(t + 1), n_adapt, C.c_double(0.75), C.c_double(0.05), C.c_double(float(la[t, i])), st)
            got[t, i] = eps0[i] * st[0]
    np.testing.assert_allclose(got, sizes, rtol=1e-5)


def test_sample_chain_schedule(ref):
    S, B = int(ref["schedule/num_results"]), int(ref["schedule/num_burnin_steps"])
    thin = 2                                              # num_steps_between_results = 1 (inference.py:234)
    assert np.array_equal(ref["schedule/kept_steps"].astype(np.int64), 1 + B + thin * np.arange(S))


def test_effective_sample_size(ref):
    """the oracle's ESS (oracle/ess_ref.py) against tfp.mcmc.effective_sample_size's own output; the product's GPU kernel
    and CPU form are held to the oracle in tests/test_gpu_edges.py and tests/test_oracle_ess.py"""
    from oracle import ess_ref
    ess = ess_ref.ess_fft(np.asarray(ref["ess/series"]))    # [S, D] -> [D]
    np.testing.assert_allclose(ess, ref["ess/ess"], rtol=1e-3)


@pytest.mark.parametrize("mname", ["radon_MN", "8schools", "election"])
def test_leapfrog_with_injected_momentum(oracle_lib, ref, mname):
    if "leapfrog/%s/x" % mname not in ref.files:
        pytest.skip("the generator could not patch this TFP version's momentum draw")
    sp = helpers.spec(mname)
    orc = oracle_lib.OracleModel(sp)
    a, b = sp.ab_from_reparam("CP")
    q0 = ref["leapfrog/%s/x" % mname].astype(np.float64)
    p0 = ref["leapfrog/%s/p" % mname].astype(np.float64)
    eps = np.ascontiguousarray(ref["leapfrog/%s/eps" % mname], np.float64)
    lp0, _ = orc.logp_grad(q0[None], a, b)
    q, p = q0.copy(), p0.copy()
    lp1 = C.c_double(0)
    oracle_lib.lib().orc_leapfrog_f64(orc._h, oracle_lib._p(a), oracle_lib._p(b), 4, oracle_lib._p(eps), oracle_lib._p(q),
                                      oracle_lib._p(p), C.byref(lp1))          # in place: 4 leapfrog steps
    np.testing.assert_allclose(q, ref["leapfrog/%s/proposed" % mname], rtol=1e-4, atol=1e-5)
    la = (lp1.value - lp0[0]) + 0.5 * ((p0 ** 2).sum() - (p ** 2).sum())
    assert abs(la - float(np.ravel(ref["leapfrog/%s/log_accept_ratio" % mname])[0])) < 2e-3
