"""CPU oracle for the effective sample size -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

float64 numpy restatement of `tfp.mcmc.effective_sample_size(states)` with its defaults
(filter_threshold=0., filter_beyond_lag=None), the call the reference makes at inference.py:240 and
inference.py:327 on the [S, C, *event] trace, followed by util.py:445-460 (`get_min_ess`).

PARITY UNPINNED: TensorFlow-Probability is not importable where this repository is built
(SURVEY.md 8c), so this follows the published definition of the estimator (restated from the public
TFP sources of the 0.7 line, not checked against a run of them):

    auto_corr = stats.auto_correlation(states, axis=0)        centred, every lag k = 0..S-1:
        c_k   = sum_{t < S-k} (x_t - mean)(x_{t+k} - mean) / (S - k)      (FFT of the zero-padded series)
        rho_k = c_k / c_0
    mask      = cumsum(rho < 0) == 0                          every lag from the first negative one on is dropped
    ESS       = S / (-1 + 2 sum_k (S - k)/S rho_k mask_k)

`ess_fft` is that, lag products by FFT as TFP forms them; `ess_direct` forms the same sums by their definition
(O(S^2), small S only) so the FFT route has a check that shares nothing with it.
tests/golden/make_reference_golden.py records TFP's own output for recorded series; tests/test_reference_golden.py
holds `ess_fft` to it once that fixture exists.
"""
import numpy as np


def _finish(rho, S):
    """rho [S, ...] normalised auto-correlations -> ESS, TFP's masking and (S - k)/S weights."""
    keep = np.cumsum(rho < 0.0, axis=0) == 0
    k = np.arange(S, dtype=np.float64).reshape((S,) + (1,) * (rho.ndim - 1))
    with np.errstate(divide="ignore", invalid="ignore"):
        return S / (-1.0 + 2.0 * np.sum((S - k) / S * rho * keep, axis=0))


def ess_fft(states):
    """states [S, ...] -> ESS [...] (float64).  A constant series gives nan (0 / 0), as in TFP."""
    x = np.asarray(states, np.float64)
    S = x.shape[0]
    x = x - x.mean(axis=0, keepdims=True)
    n_fft = 1 << int(np.ceil(np.log2(2 * S)))
    f = np.fft.rfft(x, n=n_fft, axis=0)
    ac = np.fft.irfft(f * np.conj(f), n=n_fft, axis=0)[:S]
    k = np.arange(S, dtype=np.float64).reshape((S,) + (1,) * (x.ndim - 1))
    ac = ac / (S - k)
    with np.errstate(divide="ignore", invalid="ignore"):
        rho = ac / ac[:1]
    return _finish(rho, S)


def ess_direct(states):
    """The same statistic from the definition of the lag sums (no FFT): O(S^2), for small S."""
    x = np.asarray(states, np.float64)
    S = x.shape[0]
    x = x - x.mean(axis=0, keepdims=True)
    ac = np.stack([(x[:S - k] * x[k:]).sum(axis=0) / (S - k) for k in range(S)])
    with np.errstate(divide="ignore", invalid="ignore"):
        rho = ac / ac[:1]
    return _finish(rho, S)


def min_ess(ess_parts):
    """reference util.py:445-460: per chain the minimum over every element of every part (nan -> 0), then the mean and
    the standard error over chains.  `ess_parts`: list of [C, *event] arrays."""
    parts = [np.nan_to_num(np.asarray(e, np.float64)).reshape(len(e), -1) for e in ess_parts]
    per_chain = np.concatenate(parts, axis=1).min(axis=1)
    return per_chain.mean(), per_chain.std() / np.sqrt(len(per_chain))


def ar1(S, shape, rho, seed=0):
    """Stationary AR(1) series x_t = rho x_{t-1} + sqrt(1 - rho^2) e_t, [S, *shape] float64 (rho broadcasts over shape);
    its asymptotic ESS / S is (1 - rho) / (1 + rho)."""
    rs = np.random.RandomState(seed)
    rho = np.broadcast_to(np.asarray(rho, np.float64), shape)
    e = rs.randn(S, *shape)
    x = np.empty_like(e)
    x[0] = e[0]
    for t in range(1, S):
        x[t] = rho * x[t - 1] + np.sqrt(1.0 - rho ** 2) * e[t]
    return x
