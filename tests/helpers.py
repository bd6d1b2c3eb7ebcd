"""Shared test helpers: model specs, seeded parameterisations and states."""
import numpy as np

from autoreparam_amd import models

MODEL_SPECS = {
    "8schools": lambda: models._spec_eight_schools(),
    "radon_MN": lambda: models._spec_radon("MN"),
    "radon_PA": lambda: models._spec_radon("PA"),
    "radon_IN": lambda: models._spec_radon("IN"),
    "radon_MO": lambda: models._spec_radon("MO"),
    "radon_ND": lambda: models._spec_radon("ND"),
    "radon_MA": lambda: models._spec_radon("MA"),          # 13 counties
    "radon_AZ": lambda: models._spec_radon("AZ"),          # 15 counties
    "radon_sd_AZ": lambda: models._spec_radon_stddvs("AZ"),
    "german": lambda: models._spec_german(),
    "radon_sd_MN": lambda: models._spec_radon_stddvs("MN"),
    "funnel": lambda: models._spec_funnel(),
    "election": lambda: models._spec_election(),
    "electric": lambda: models._spec_electric(),
    "time_series": lambda: models._spec_time_series(),
}
_cache = {}


def spec(name):
    if name not in _cache:
        _cache[name] = MODEL_SPECS[name]()
    return _cache[name]


def params(sp, kind, seed=0):
    """(a, b) float32 [D] for 'CP', 'NCP', a seeded 'VIP' (a and b free) or 'B1' (a free, b = 1)."""
    if kind in ("CP", "NCP"):
        return sp.ab_from_reparam(kind)
    rs = np.random.RandomState(1000 + seed)
    if kind == "B1":   # a free, b = 1: what the reference's tied cVIP / dVIP runs execute
        return rs.rand(sp.D).astype(np.float32), np.ones(sp.D, np.float32)
    return rs.rand(sp.D).astype(np.float32), rs.rand(sp.D).astype(np.float32)


def states(sp, n, seed=0, scale=0.3):
    rs = np.random.RandomState(seed)
    return (scale * rs.randn(n, sp.D)).astype(np.float32)


# ---------------------------------------------------------------------------
# Trajectory parity with explained divergence.  The HIP path and the float32 oracle run the same algorithm on the same
# random streams, but sum in different orders, so their log acceptance ratios differ by a few float32 ulps of the
# energies compared.  A Metropolis test whose margin |log u - log alpha| is smaller than that can fall the other way,
# after which the two chains are different (both correct) chains.  `explain_divergence` accepts exactly that and
# nothing else: every chain must agree step by step until its FIRST differing accept decision, and that decision must
# have sat within `margin_tol` of its threshold in the oracle's run; a chain that differs anywhere else fails.
# ---------------------------------------------------------------------------
EPS32 = float(np.finfo(np.float32).eps)
# measured (tests/diagnostics/margin_probe.py, profiles/r03_margin_probe.txt: every model x parameterisation x lanes, fixed step and
# simple adaptation): every decision that differed had |log u - log alpha| <= 1.7e-3 or <= 1.0 ulp of its energies
MARGIN_ABS, MARGIN_ULPS = 2e-3, 4.0


def margin_tol(escale, ulps=MARGIN_ULPS, extra=0.0):
    """How far from its threshold a float32 Metropolis test can sit and still flip between two correct
    implementations: an absolute floor plus `ulps` float32 ulps of the largest energy term it compared."""
    return MARGIN_ABS + extra + ulps * EPS32 * np.asarray(escale, np.float64)


def explain_divergence(x_hip, x_orc, acc_hip, acc_orc, margin, escale, state_tol, ulps=MARGIN_ULPS, extra=0.0, what=""):
    """x_* [n, C, D]: the state after every step; acc_* [n, K, C]: every accept decision (K kernels per step);
    margin / escale [n, K, C]: the oracle's diagnostics.  Returns (clean [C] bool, first [C] int): chains that never
    branched, and the step at which the others did (n for clean chains).  Raises on an unexplained chain."""
    n, Cn, _ = x_orc.shape
    K = acc_orc.shape[1]
    tol = margin_tol(escale, ulps, extra)
    err = np.abs(np.asarray(x_hip, np.float64) - x_orc).max(axis=2)          # [n, C]
    diff = (np.asarray(acc_hip) != np.asarray(acc_orc)).reshape(n * K, Cn)   # decision order: step-major, kernel-minor
    first_dec = np.where(diff.any(axis=0), diff.argmax(axis=0), n * K)
    first = first_dec // K
    clean = first_dec == n * K
    bad = []
    for c in range(Cn):
        upto = n if clean[c] else first[c]            # states recorded before the branching step must agree
        if upto > 0 and err[:upto, c].max() > state_tol:
            bad.append((c, "state differs by %.3g at step %d before any decision differs" %
                        (err[:upto, c].max(), int(err[:upto, c].argmax()))))
            continue
        if not clean[c]:
            s, k = divmod(int(first_dec[c]), K)
            if not abs(margin[s, k, c]) <= tol[s, k, c]:
                bad.append((c, "decision (step %d, kernel %d) differs with margin %.3g > tolerance %.3g (energies ~ %.3g)" %
                            (s, k, margin[s, k, c], tol[s, k, c], escale[s, k, c])))
    assert not bad, "%s: %d of %d chains differ from the oracle without a Metropolis test at its threshold: %s" % (
        what, len(bad), Cn, bad[:6])
    return clean, first
