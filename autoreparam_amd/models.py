"""Model definitions for the hot path: the four hierarchical models named by
BASELINE.json plus radon_stddvs, neals_funnel, electric and time_series, as frozen
data + a model id the HIP engine understands.

Mirrors the reference's ``models.get_model_by_name(name, dataset) -> ModelConfig``
(models.py:51-54, 1144-1175).  In the reference ``ModelConfig.model`` is an
Edward2 program; here it is a :class:`ModelSpec` (same role: it defines the joint
density, evaluated natively by the engine instead of by TensorFlow).
"""
import collections
import os

import numpy as np

from . import _lib

DATA_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

ModelConfig = collections.namedtuple(
    "ModelConfig", ("model", "model_args", "observed_data", "to_centered",
                    "to_noncentered", "make_to_centered",
                    "make_to_partially_noncentered", "bijectors_fn"))


class ModelSpec(object):
    """Joint density of one model: id, latent parts in trace order, raw inputs."""

    def __init__(self, name, model_id, part_names, part_shapes, raw, observed, scalar_loc=(), scalar_scale=()):
        # vector parts whose reference random variable has a SCALAR loc / scale: with --notied_pparams the
        # reference gives `<rv>_a` the loc's shape and `<rv>_b` the scale's (program_transformations.py:486-533),
        # i.e. one shared value for the part
        self.scalar_loc = set(scalar_loc)
        self.scalar_scale = set(scalar_scale)
        self.name = name
        self.model_id = model_id
        self.part_names = list(part_names)
        self.part_shapes = [tuple(s) for s in part_shapes]
        self.part_sizes = [int(np.prod(s)) if len(s) else 1 for s in self.part_shapes]
        self.offsets = np.concatenate([[0], np.cumsum(self.part_sizes)]).astype(int)
        self.D = int(self.offsets[-1])
        self.raw = raw            # dict of numpy arrays, as the reference's model_args hold them
        self.observed = observed  # dict name -> array, as the reference's observed_data

    # ---- flat [C, D] <-> list of [C, *event] parts (reference layout) ----
    def pack(self, parts, xp=np):
        cols = []
        for p, shp in zip(parts, self.part_shapes):
            p = xp.asarray(p) if xp is np else p
            cols.append(p.reshape(p.shape[0], -1))
        return xp.concatenate(cols, axis=1) if xp is np else xp.cat(cols, dim=1)

    def unpack(self, flat):
        out = []
        for k, shp in enumerate(self.part_shapes):
            sl = flat[..., self.offsets[k]:self.offsets[k + 1]]
            out.append(sl.reshape(tuple(flat.shape[:-1]) + shp))
        return out

    def ab_from_reparam(self, reparam):
        """Per-element VIP parameters (a, b), float32 [D].

        'CP' -> a=b=1, 'NCP' -> a=b=0, otherwise a dict with keys ``<rv>_a`` and
        optionally ``<rv>_b`` (missing ``_b`` means 1, exactly as the reference's
        get_or_init falls back at program_transformations.py:495-500; other keys
        such as ``*_prior_mean`` are ignored).
        """
        a = np.ones(self.D, np.float32)
        b = np.ones(self.D, np.float32)
        if isinstance(reparam, str):
            if reparam == "CP":
                return a, b
            if reparam == "NCP":
                return np.zeros(self.D, np.float32), np.zeros(self.D, np.float32)
            raise ValueError("unknown parameterisation %r" % (reparam,))
        for k, name in enumerate(self.part_names):
            lo, hi = self.offsets[k], self.offsets[k + 1]
            if name + "_a" not in reparam:
                raise KeyError("parameterisation has no entry for %s_a" % name)
            a[lo:hi] = np.broadcast_to(np.asarray(reparam[name + "_a"], np.float32).reshape(-1), (hi - lo,))
            if name + "_b" in reparam:
                b[lo:hi] = np.broadcast_to(np.asarray(reparam[name + "_b"], np.float32).reshape(-1), (hi - lo,))
        return a, b

    def untied_groups(self):
        """(a_group, b_group) int32 [D] for arp_vi_io: leader element of every element's untied a / b variable."""
        ag, bg = np.arange(self.D, dtype=np.int32), np.arange(self.D, dtype=np.int32)
        for k, name in enumerate(self.part_names):
            lo, hi = self.offsets[k], self.offsets[k + 1]
            if name in self.scalar_loc:
                ag[lo:hi] = lo
            if name in self.scalar_scale:
                bg[lo:hi] = lo
        return ag, bg

    def untied_shape(self, name, which):
        """shape of the reference's untied `<name>_a` (which='a') / `<name>_b` variable"""
        k = self.part_names.index(name)
        shared = name in (self.scalar_loc if which == "a" else self.scalar_scale)
        return () if shared else self.part_shapes[k]

    def dataset(self):
        """(ctypes Dataset, keep-alive list) for arp_model_create."""
        keep = []

        def f32(x):
            arr = np.ascontiguousarray(x, dtype=np.float32)
            keep.append(arr)
            return arr.ctypes.data_as(_lib._f32p)

        def i32(x):
            arr = np.ascontiguousarray(x, dtype=np.int32)
            keep.append(arr)
            return arr.ctypes.data_as(_lib._i32p)

        d = _lib.Dataset()
        d.model = self.model_id
        r = self.raw
        if self.model_id in (_lib.MODEL_RADON, _lib.MODEL_RADON_STDDVS):
            d.n_obs = len(r["y"]); d.n_groups = len(r["u"])
            d.group_host = i32(r["county"]); d.u_host = f32(r["u"])
            d.x_host = f32(r["x"]); d.y_host = f32(r["y"])
        elif self.model_id == _lib.MODEL_EIGHT_SCHOOLS:
            d.n_obs = 8; d.n_groups = 8
            d.u_host = f32(r["sigma"]); d.y_host = f32(r["y"])
        elif self.model_id == _lib.MODEL_ELECTION:
            d.n_obs = len(r["y"]); d.n_groups = int(r["n_state"])
            d.group_host = i32(r["state"]); d.x_host = f32(r["female"]); d.x2_host = f32(r["black"])
            d.y_host = f32(r["y"])
        elif self.model_id == _lib.MODEL_NEALS_FUNNEL:
            pass
        elif self.model_id == _lib.MODEL_TIME_SERIES:
            d.n_obs = len(r["y"]); d.x_host = f32(r["x"]); d.y_host = f32(r["y"])
        elif self.model_id == _lib.MODEL_ELECTRIC:
            if int(r["n_grade"]) != int(r["n_grade_pair"]):
                raise ValueError("electric: n_grade and n_grade_pair must agree")
            d.n_obs = len(r["y"]); d.n_groups = int(r["n_pair"]); d.n_features = int(r["n_grade"])
            d.group_host = i32(r["pair"]); d.group2_host = i32(r["grade"]); d.group3_host = i32(r["grade_pair"])
            d.x_host = f32(r["treatment"]); d.y_host = f32(r["y"])
        elif self.model_id == _lib.MODEL_GERMAN_CREDIT:
            d.n_obs = r["X"].shape[0]; d.n_features = r["X"].shape[1]
            d.X_host = f32(r["X"]); d.y_host = f32(r["y"])
        else:
            raise ValueError("unknown model id")
        return d, keep


def _load(fname):
    path = os.path.join(DATA_DIR, fname)
    if not os.path.exists(path):
        raise IOError("frozen dataset %s not found (tools/freeze_data.py writes it)" % path)
    with np.load(path) as z:
        return {k: z[k] for k in z.files}


def _spec_eight_schools():
    r = _load("eight_schools.npz")
    return ModelSpec("8schools", _lib.MODEL_EIGHT_SCHOOLS, ["mu", "log_tau", "theta"],
                     [(), (), (8,)], r, {"y": r["y"]})


def _spec_radon(state_code):
    r = _load("radon_%s.npz" % state_code)
    J = len(r["u"])
    return ModelSpec("radon", _lib.MODEL_RADON, ["mua", "b1", "b2", "m"], [(), (), (), (J,)],
                     r, {"y": r["y"].reshape(-1, 1)})


def _spec_radon_stddvs(state_code):
    """radon with inferred per-county observation scales (reference models.py:763-806)."""
    r = _load("radon_%s.npz" % state_code)
    J = len(r["u"])
    return ModelSpec("radon_stddvs", _lib.MODEL_RADON_STDDVS, ["mua", "b1", "b2", "m", "log_m_stddv"],
                     [(), (), (), (J,), (J,)], r, {"y": r["y"].reshape(-1, 1)})


def _spec_funnel():
    return ModelSpec("neals_funnel", _lib.MODEL_NEALS_FUNNEL, ["x1", "x2"], [(), ()], {}, {})


def _spec_electric():
    """electric company (reference models.py:1011-1066); `a` keeps the reference's [n_pair, 1] shape."""
    r = _load("electric.npz")
    P, G = int(r["n_pair"]), int(r["n_grade"])
    return ModelSpec("electric", _lib.MODEL_ELECTRIC, ["mua", "sigma_y", "a", "b"],
                     [(int(r["n_grade_pair"]),), (G,), (P, 1), (G,)], r, {"y": r["y"]},
                     scalar_loc=("mua", "sigma_y", "b"), scalar_scale=("a",))


def _spec_time_series():
    """local linear trend (reference models.py:1069-1141): every latent is its own scalar random variable, in
    trace order sigma_alpha, sigma_mu, alpha0, mu0, alpha1, mu1, ..., beta."""
    r = _load("time_series.npz")
    T = len(r["y"])
    names = ["sigma_alpha", "sigma_mu"] + [n for t in range(T) for n in ("alpha%d" % t, "mu%d" % t)] + ["beta"]
    return ModelSpec("time_series", _lib.MODEL_TIME_SERIES, names, [()] * len(names), r, {"y": r["y"]})


def _spec_german():
    r = _load("german_credit.npz")
    F = r["X"].shape[1]
    return ModelSpec("german_credit_lognormalcentered", _lib.MODEL_GERMAN_CREDIT,
                     ["overall_log_scale", "beta_log_scales", "beta"], [(), (F,), (F,)],
                     r, {"y": r["y"][np.newaxis, ...]}, scalar_loc=("beta_log_scales",))


def _spec_election():
    r = _load("election88.npz")
    S = int(r["n_state"])
    return ModelSpec("election", _lib.MODEL_ELECTION, ["mua", "log_sigma_a", "a", "b1", "b2"],
                     [(), (), (S,), (), ()], r, {"y": r["y"].reshape(-1, 1)}, scalar_loc=("a",))


def get_model_by_name(model_name, dataset=None):
    """Reference: models.py:1144-1175 (the four BASELINE models and the scalar-Normal hierarchical models of
    SURVEY 8f-3; the GP / MVN / Wishart models are out of scope)."""
    if model_name == "8schools":
        spec = _spec_eight_schools()
    elif model_name == "radon":
        spec = _spec_radon(dataset if dataset else "MN")
    elif model_name == "neals_funnel":
        spec = _spec_funnel()
    elif model_name == "radon_stddvs":
        spec = _spec_radon_stddvs(dataset if dataset else "MN")
    elif model_name == "electric":
        spec = _spec_electric()
    elif model_name == "time_series":
        spec = _spec_time_series()
    elif model_name == "german_credit_lognormalcentered":
        spec = _spec_german()
    elif model_name in ("election", "election88"):
        spec = _spec_election()
    else:
        raise Exception("unknown model {} (this build covers 8schools, radon, radon_stddvs, "
                        "neals_funnel, electric, time_series, german_credit_lognormalcentered, election)".format(model_name))
    from . import engine  # deferred: converters run on the device

    varnames = spec.part_names
    noncentered = {p: 0. for v in varnames for p in (v + "_a", v + "_b", v + "_c")}
    make_to_centered = engine.build_make_to_centered(spec)
    make_to_partially_noncentered = engine.build_make_to_partially_noncentered(spec)
    to_centered = make_to_centered(**noncentered)
    to_noncentered = make_to_partially_noncentered(**noncentered)
    return ModelConfig(spec, [], spec.observed, to_centered, to_noncentered, make_to_centered,
                       make_to_partially_noncentered, None)
