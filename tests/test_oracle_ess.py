"""CPU: the oracle's effective sample size (oracle/ess_ref.py, float64, tfp.mcmc.effective_sample_size defaults as
called at inference.py:240, 327) against the definition of its lag sums, the AR(1) known answer
and util.py:445-460; and the PRODUCT's CPU form (autoreparam_amd.util.effective_sample_size_fft) against the oracle.
The GPU kernel `arp_ess` is held to the same oracle in tests/test_gpu_edges.py."""
import numpy as np
import torch

from oracle import ess_ref
from autoreparam_amd import util


def test_fft_route_equals_the_definition():
    x = ess_ref.ar1(257, (6, 3), [[0.0, 0.5, 0.9]] * 6, seed=1) * [1.0, 30.0, 0.01] + [0.0, 1e3, -7.0]
    a, b = ess_ref.ess_fft(x), ess_ref.ess_direct(x)
    np.testing.assert_allclose(a, b, rtol=1e-9)
    # the cut: everything from the first negative auto-correlation on is dropped, so appending a strongly
    # anti-correlated tail lag cannot change the sum before it
    assert (a > 0).all() and (a <= 257 * 1.5).all()


def test_ar1_known_answer_and_white_noise():
    S = 20000
    for rho in (0.0, 0.3, 0.7, 0.9):
        x = ess_ref.ar1(S, (64,), rho, seed=3)
        e = ess_ref.ess_fft(x) / S
        assert abs(e.mean() - (1 - rho) / (1 + rho)) < 0.03 * (1 - rho) / (1 + rho) + 2 * e.std() / 8, (rho, e.mean())
    # negative correlation: the first lag is negative, the sum stops at lag 0 -> ESS = S exactly (TFP's truncation)
    x = ess_ref.ar1(2000, (8,), -0.5, seed=4)
    np.testing.assert_allclose(ess_ref.ess_fft(x), 2000.0, rtol=1e-12)


def test_constant_series_and_two_samples():
    assert np.isnan(ess_ref.ess_fft(np.ones((50, 3)))).all()
    x = np.array([[1.0], [2.0]])            # S = 2: rho_1 = -1 -> dropped, ESS = S
    np.testing.assert_allclose(ess_ref.ess_fft(x), 2.0)


def test_min_ess_follows_the_reference_summary():
    ess = [np.array([[3.0, 2.0], [5.0, np.nan]]), np.array([1.5, 4.0])]   # parts [C, 2] and [C]
    m, s = ess_ref.min_ess(ess)
    assert abs(m - np.mean([1.5, 0.0])) < 1e-12 and abs(s - np.std([1.5, 0.0]) / np.sqrt(2)) < 1e-12
    assert np.allclose(util.get_min_ess(ess), (m, s))


def test_product_cpu_form_matches_the_oracle():
    """autoreparam_amd.util.effective_sample_size_fft (float32 torch FFT; what the CLI uses off the GPU) against the
    float64 oracle on series with offsets and scales, a drifting start and a short series."""
    x = ess_ref.ar1(600, (37, 5), [0.0, 0.3, 0.6, 0.9, -0.4], seed=5) * [1.0, 10.0, 0.1, 3.0, 1.0] + [0.0, 100.0, -5.0, 1e3, 0.0]
    got = util.effective_sample_size_fft(torch.as_tensor(x, dtype=torch.float32)).numpy()
    np.testing.assert_allclose(got, ess_ref.ess_fft(x.astype(np.float32)), rtol=2e-3)
    short = x[:9]
    np.testing.assert_allclose(util.effective_sample_size_fft(torch.as_tensor(short)).numpy(), ess_ref.ess_fft(short), rtol=1e-4)
    g = util.effective_sample_size_fft(torch.as_tensor(x[:, :4]), max_chains_per_batch=3).numpy()
    np.testing.assert_allclose(g, ess_ref.ess_fft(x[:, :4]), rtol=1e-4)
