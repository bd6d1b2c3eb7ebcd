"""Float64 restatement of the reference's model programs and program
transformations, in the reference's own *shape*: generative programs that call
``Normal`` / ``Bernoulli`` random variables in trace order, one-hot gathers done as
dense matmuls, gradients by reverse-mode autodiff (torch.autograd standing in for
tf.gradients).  It is deliberately slow and literal; it pins the analytic,
sufficient-statistic formulas used by oracle.c and by the HIP kernels.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED (the reference itself cannot be run
here).

Restated from (behaviour, not code):
  models.py:139-147 (schools), 826-837 (radon), 888-904 (german credit),
  969-982 (election), 1013-1035 (electric), 1071-1094 (time_series);
  program_transformations.py:110-139 (log joint = sum of rv.log_prob over all
  RVs and elements), 262-279 (ncp: every Normal not named y*), 486-533 + 555-600
  (VIP: xt ~ N(a mu, sigma^b), x = mu + sigma/sigma^b (xt - a mu); missing `_b`
  means b = 1);
  models.py:56-128 (state converters run the program in trace order).
"""
import math

import numpy as np
import torch

F64 = torch.float64
HALF_LOG_2PI = 0.5 * math.log(2.0 * math.pi)


def _t(x):
    return torch.as_tensor(x, dtype=F64)


class Run(object):
    """One execution of a model program.

    mode 'logjoint' : `values[name]` are the latent values in the parameterised
                      coordinates; accumulates the log joint; hands centred values
                      to the rest of the program.
    mode 'invert'   : `values[name]` are centred values; records the
                      parameterised values (to_noncentered / to_partially_...).
    In both modes `centred` and `param` record every latent in trace order.
    """

    def __init__(self, mode, values, observed, ab):
        self.mode, self.values, self.observed, self.ab = mode, values, observed, ab
        self.logp = _t(0.0)
        self.centred, self.param, self.order = {}, {}, []

    def _ab(self, name, like):
        if self.ab == "CP":
            return 1.0, 1.0
        if self.ab == "NCP":
            return 0.0, 0.0
        a = _t(self.ab[name + "_a"])
        b = _t(self.ab.get(name + "_b", 1.0))
        return a, b

    def normal(self, name, loc, scale):
        loc, scale = _t(loc), _t(scale)
        if name.startswith("y"):  # observed (the reference keys on the name, p_t.py:262)
            v = _t(self.observed[name])
            z = (v - loc) / scale
            self.logp = self.logp + (-0.5 * z * z - torch.log(scale) - HALF_LOG_2PI).expand(v.shape).sum()
            return v
        a, b = self._ab(name, loc)
        loc_t = a * loc
        scale_t = scale ** b
        ratio = scale / scale_t
        shape = torch.broadcast_shapes(loc.shape, scale.shape)
        if self.mode == "logjoint":
            xt = self.values[name]
            x = loc + ratio * (xt - loc_t)
        else:
            x = self.values[name]
            xt = loc_t + (x - loc) / ratio
        z = (xt - loc_t) / scale_t
        self.logp = self.logp + (-0.5 * z * z - torch.log(scale_t) - HALF_LOG_2PI).expand(shape).sum()
        self.centred[name], self.param[name] = x, xt
        self.order.append(name)
        return x

    def bernoulli(self, name, logits):
        v = _t(self.observed[name])
        lp = v * logits - torch.nn.functional.softplus(logits)
        self.logp = self.logp + lp.sum()
        return v


def one_hot(idx, depth):
    """tf.one_hot semantics: an index outside [0, depth) gives an all-zero row."""
    idx = np.asarray(idx)
    out = np.zeros((idx.shape[0], depth))
    ok = (idx >= 0) & (idx < depth)
    out[np.nonzero(ok)[0], idx[ok]] = 1.0
    return _t(out)


# --- the four programs ------------------------------------------------------
def schools_program(r, raw):
    mu = r.normal("mu", 0.0, 5.0)
    log_tau = r.normal("log_tau", 0.0, 5.0)
    theta = r.normal("theta", mu * torch.ones(8, dtype=F64), torch.exp(log_tau) * torch.ones(8, dtype=F64))
    r.normal("y", theta, _t(raw["sigma"]))


def radon_program(r, raw):
    J = len(raw["u"])
    mua = r.normal("mua", 0.0, 1.0)
    b1 = r.normal("b1", 0.0, 1.0)
    b2 = r.normal("b2", 0.0, 1.0)
    m = r.normal("m", mua + _t(raw["u"]) * b1, torch.ones(J, dtype=F64))
    Cm = one_hot(raw["county"], J)
    y_mu = Cm @ m.unsqueeze(1) + _t(raw["x"]).unsqueeze(1) * b2
    r.normal("y", y_mu, 1.0)


def radon_stddvs_program(r, raw):
    J = len(raw["u"])
    mua = r.normal("mua", 0.0, 1.0)
    b1 = r.normal("b1", 0.0, 1.0)
    b2 = r.normal("b2", 0.0, 1.0)
    m = r.normal("m", mua + _t(raw["u"]) * b1, torch.ones(J, dtype=F64))
    Cm = one_hot(raw["county"], J)
    lms = r.normal("log_m_stddv", torch.zeros(J, dtype=F64), torch.ones(J, dtype=F64))
    y_mu = Cm @ m.unsqueeze(1) + _t(raw["x"]).unsqueeze(1) * b2
    y_sd = Cm @ torch.exp(lms).unsqueeze(1)
    r.normal("y", y_mu, y_sd)


def funnel_program(r, raw):
    x1 = r.normal("x1", 0.0, 3.0)
    r.normal("x2", 0.0, torch.exp(x1 / 2.0))


def german_program(r, raw):
    X = _t(raw["X"])
    F = X.shape[1]
    ols = r.normal("overall_log_scale", 0.0, 10.0)
    bls = r.normal("beta_log_scales", ols, torch.ones(F, dtype=F64))
    beta = r.normal("beta", torch.zeros(F, dtype=F64), torch.exp(bls))
    logits = torch.einsum("nd,md->mn", X, beta.unsqueeze(0))
    r.bernoulli("y", logits)


def election_program(r, raw):
    S = int(raw["n_state"])
    mua = r.normal("mua", 0.0, 100.0)
    lsa = r.normal("log_sigma_a", 0.0, 10.0)
    a = r.normal("a", mua, torch.ones(S, dtype=F64) * torch.exp(lsa))
    b1 = r.normal("b1", 0.0, 100.0)
    b2 = r.normal("b2", 0.0, 100.0)
    Cm = one_hot(raw["state"], S)  # the reference feeds the 1-based state index
    y_hat = Cm @ a.unsqueeze(1) + _t(raw["female"]).unsqueeze(1) * b2 + _t(raw["black"]).unsqueeze(1) * b1
    r.bernoulli("y", y_hat)


def electric_program(r, raw):
    # models.py:1013-1035: pair, grade and grade_pair are all fed 1-based to tf.one_hot
    n_pair, n_grade, n_gp = int(raw["n_pair"]), int(raw["n_grade"]), int(raw["n_grade_pair"])
    C_pair = one_hot(raw["pair"], n_pair)
    C_grade = one_hot(raw["grade"], n_grade)
    C_gp = one_hot(raw["grade_pair"], n_gp)
    mua = r.normal("mua", 0.0, torch.ones(n_gp, dtype=F64))
    mua_hat = 100.0 * (C_gp @ mua.unsqueeze(1))                     # [n_pair, 1]
    sigma_y = r.normal("sigma_y", 0.0, torch.ones(n_grade, dtype=F64))
    sigma_y_hat = C_grade @ sigma_y.unsqueeze(1)                    # [N, 1]
    a = r.normal("a", mua_hat, 1.0)                                 # [n_pair, 1]
    b = r.normal("b", 0.0, 100.0 * torch.ones(n_grade, dtype=F64))
    y_hat_a = (C_pair @ a).reshape(-1)
    y_hat_b = (C_grade @ b.unsqueeze(1)).reshape(-1)
    y_hat = y_hat_a + y_hat_b * _t(raw["treatment"])
    r.normal("y", y_hat, torch.exp(sigma_y_hat.reshape(-1)))


def time_series_program(r, raw):
    # models.py:1071-1094: every latent is its own scalar RV; scales are softplus of the sigma latents
    T = len(raw["y"])
    sp = torch.nn.functional.softplus
    sigma_alpha = r.normal("sigma_alpha", 0.0, 1.0)
    sigma_mu = r.normal("sigma_mu", 0.0, 1.0)
    alpha = [r.normal("alpha0", 0.0, sp(sigma_alpha))]
    mu = [r.normal("mu0", 0.0, sp(sigma_mu))]
    for t in range(1, T):
        alpha.append(r.normal("alpha%d" % t, alpha[t - 1] + mu[t - 1], sp(sigma_alpha)))
        mu.append(r.normal("mu%d" % t, mu[t - 1], sp(sigma_mu)))
    beta = r.normal("beta", 0.0, 1.0)
    r.normal("y", torch.stack(alpha) + beta * _t(raw["x"]), 0.12)


PROGRAMS = {"time_series": time_series_program, "electric": electric_program, "8schools": schools_program, "radon": radon_program, "radon_stddvs": radon_stddvs_program, "neals_funnel": funnel_program,
            "german_credit_lognormalcentered": german_program, "election": election_program}


def _observed(spec):
    return {k: np.asarray(v, np.float64) for k, v in spec.observed.items()}


def ab_dict(spec, a, b):
    """{name_a, name_b} dict from flat per-element arrays."""
    d = {}
    for k, name in enumerate(spec.part_names):
        lo, hi = spec.offsets[k], spec.offsets[k + 1]
        shp = spec.part_shapes[k]
        d[name + "_a"] = np.asarray(a[lo:hi], np.float64).reshape(shp)
        d[name + "_b"] = np.asarray(b[lo:hi], np.float64).reshape(shp)
    return d


def log_joint(spec, ab, flat_state):
    """(logp, grad) at one flat state [D] (float64), reference-valued (all constants)."""
    x = _t(flat_state).clone().requires_grad_(True)
    vals = {n: x[spec.offsets[k]:spec.offsets[k + 1]].reshape(spec.part_shapes[k])
            for k, n in enumerate(spec.part_names)}
    r = Run("logjoint", vals, _observed(spec), ab)
    PROGRAMS[spec.name](r, spec.raw)
    assert r.order == spec.part_names
    (g,) = torch.autograd.grad(r.logp, x)
    return float(r.logp.detach()), g.numpy()


def convert(spec, ab, flat_state, to_centered):
    """State converter: parameterised -> centred (to_centered) or the inverse."""
    x = _t(flat_state)
    vals = {n: x[spec.offsets[k]:spec.offsets[k + 1]].reshape(spec.part_shapes[k])
            for k, n in enumerate(spec.part_names)}
    r = Run("logjoint" if to_centered else "invert", vals, _observed(spec), ab)
    PROGRAMS[spec.name](r, spec.raw)
    src = r.centred if to_centered else r.param
    return torch.cat([src[n].reshape(-1) for n in spec.part_names]).numpy()
