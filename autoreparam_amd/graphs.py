"""Target / ELBO construction with the reference's entry points (graphs.py:14-213).

The reference builds TensorFlow graphs; here ``make_*_graph`` return light objects
that name the same things -- the target density (model + parameterisation), the
ELBO to optimise and the variational / learnable parameters -- and the numbers are
produced by the HIP engine when ``inference.*`` runs them.
"""
import collections

import numpy as np

from . import engine as _engine
from .flags import FLAGS


class Target(object):
    """target_log_prob_fn of one parameterisation (reference closures target_cp /
    target_ncp / target_vip, graphs.py:37-44, 84-91, 139-145, 197-203)."""

    def __init__(self, spec, reparam):
        self.spec = spec
        self.reparam = reparam          # 'CP', 'NCP' or a dict of <rv>_a / <rv>_b
        self.ab = spec.ab_from_reparam(reparam)

    def __call__(self, *parts):
        """Log joint at one state (parts without a chain axis) or a batch (leading chain axis);
        reference-valued (all additive constants included)."""
        spec = self.spec
        single = all(np.ndim(p) == len(s) for p, s in zip(parts, spec.part_shapes))
        if single:
            parts = [np.asarray(p, np.float32)[np.newaxis] for p in parts]
        eng = _engine.engine_for(spec)
        eng.set_param(0, self.ab)
        lp, _ = eng.logp_grad(spec.pack(parts))
        out = lp.cpu().numpy().astype(np.float64) + eng.logp_const(0)
        return out[0] if single else out


class Elbo(object):
    """Mean-field ELBO of `target` (util.get_mean_field_elbo, util.py:232-268)."""

    def __init__(self, target, num_mc_samples, learn_a=False, tied=True):
        self.target = target
        self.num_mc_samples = num_mc_samples
        self.learn_a = learn_a
        self.tied = tied


def _variational_parameters(spec):
    """name_loc / name_scale with the reference's initial values
    (program_transformations.py:207-215: loc0 = 1e-2 N(0,1) drawn at session start,
    scale0 = softplus(-2)); the entries here carry shapes, the draws happen per run."""
    vp = collections.OrderedDict()
    for name, shp in zip(spec.part_names, spec.part_shapes):
        vp[name + "_loc"] = np.zeros(shp, np.float32)
        vp[name + "_scale"] = np.full(shp, np.log1p(np.exp(-2.0)), np.float32)
    return vp


def make_cp_graph(model_config, flags=FLAGS):
    spec = model_config.model
    target = Target(spec, "CP")
    return target, spec, Elbo(target, flags.num_mc_samples), _variational_parameters(spec), None


def make_ncp_graph(model_config, flags=FLAGS):
    spec = model_config.model
    target = Target(spec, "NCP")
    return target, spec, Elbo(target, flags.num_mc_samples), _variational_parameters(spec), None


def make_cvip_graph(model_config, parameterisation_type="exp", tied_pparams=False, flags=FLAGS):
    """cVIP: a = sigmoid(w), w initialised at 0 (program_transformations.py:507-510).

    tied_pparams=True reproduces what the reference *executes*: `b` equals `a` only
    on the trace that creates the variable and falls back to 1 on every later trace
    (program_transformations.py:495-500, 513-514), so the target and
    the ELBO see b = 1.  tied_pparams=False learns a separate b = sigmoid(w_b), each with the reference's
    variable shapes.
    """
    spec = model_config.model
    init = collections.OrderedDict()
    for name, shp in zip(spec.part_names, spec.part_shapes):
        if tied_pparams:
            init[name + "_a"] = np.full(shp, 0.5, np.float32)      # broadcast shape of loc and scale
        else:
            # untied: `a` takes the shape of the variable's loc, `b` of its scale (program_transformations.py:486-533);
            # a vector variable with a scalar loc (german beta_log_scales, election a) learns ONE shared a
            init[name + "_a"] = np.full(spec.untied_shape(name, "a"), 0.5, np.float32)
            init[name + "_b"] = np.full(spec.untied_shape(name, "b"), 0.5, np.float32)
    target = Target(spec, init)
    elbo = Elbo(target, flags.num_mc_samples, learn_a=True, tied=tied_pparams)
    return target, spec, elbo, _variational_parameters(spec), init


def make_dvip_graph(model_config, reparam, parameterisation_type="exp", flags=FLAGS):
    spec = model_config.model
    target = Target(spec, reparam)
    return target, spec, Elbo(target, flags.num_mc_samples), _variational_parameters(spec), None
