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
