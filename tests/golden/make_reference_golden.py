#!/usr/bin/env python3
"""Pin the oracle to the REFERENCE itself.  Build-container tool, never shipped to the GPU box.

Everything else under tests/golden/ is generated from this repository's own restatements, because the reference
(/root/reference, TensorFlow 1.14 + TensorFlow-Probability 0.7.0 + absl, README.md:40) cannot be imported where
this repository is built (parity unpinned).  This script is the route to "pinned": on a machine that
HAS TensorFlow 1.x, TFP and absl it imports the reference's own modules from REFERENCE_DIR (default
/root/reference) and records, with no random number generator in the loop,

  density/<model>/<kind>/{logp, grad, centred}   target_* log joints + tf.gradients and the state converters at the
                                                  states, (a, b) and model list of density_golden.npz
                                                  (graphs.make_cp_graph / make_ncp_graph / make_dvip_graph,
                                                  models.build_make_to_centered)
  dual/{log_accept, step_size}, simple/{...}      tfp.mcmc.DualAveragingStepSizeAdaptation / SimpleStepSizeAdaptation
                                                  (inference.py:224-226, 288-306) wrapped around a scripted inner
                                                  kernel: the step sizes TFP derives from a given sequence of log
                                                  acceptance ratios -- what arp_adapt_probe / orc_adapt_update replay
  schedule/kept_steps                             which kernel steps tfp.mcmc.sample_chain(num_results, num_burnin_steps,
                                                  num_steps_between_results=1) returns (inference.py:228-236),
                                                  with a step-counting kernel
  ess/{series, ess}                               tfp.mcmc.effective_sample_size with its defaults (inference.py:240)
                                                  on recorded AR(1) series
  leapfrog/...                                    best effort: one HamiltonianMonteCarlo.one_step per model with the
                                                  momentum draw patched to a recorded array (proposed state and
                                                  log_accept_ratio); skipped with a message if TFP's internals differ

into tests/golden/reference_golden.npz.  tests/test_reference_golden.py (skipped while that file is absent) holds
the oracle -- and through it the HIP path -- to these numbers.  Nothing of the reference is copied: the file holds
inputs and outputs only.

Usage:  REFERENCE_DIR=/root/reference python tests/golden/make_reference_golden.py
Exit code 3 with a message when TensorFlow / TFP / absl are not importable (the case in the build container).
Record the versions it ran against in the npz ("versions") and in DESIGN.md section 5.
"""
import collections
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("REFERENCE_DIR", "/root/reference")

try:
    import tensorflow.compat.v1 as tf
    import tensorflow_probability as tfp
    import absl  # noqa: F401
except Exception as e:  # pragma: no cover - the build container takes this branch
    sys.stderr.write("make_reference_golden.py: TensorFlow 1.x / tensorflow_probability / absl are required and not "
                     "importable here (%r).\nRun it on a machine with the reference's dependencies "
                     "(README.md:40: TF 1.14, TFP 0.7.0); nothing was written.\n" % (e,))
    sys.exit(3)

sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, REF)
import helpers  # noqa: E402  (this repository: model list, seeded states and (a, b))

tf.disable_v2_behavior()
import main as ref_main  # noqa: E402,F401  (the reference's main.py defines the absl flags the other modules read)
import graphs as ref_graphs  # noqa: E402
import models as ref_models  # noqa: E402
from absl import flags as absl_flags  # noqa: E402

absl_flags.FLAGS(["make_reference_golden"])
mcmc = tfp.mcmc

# this repository's test names -> the reference's (model, dataset)
REF_NAME = {"8schools": ("8schools", None), "radon_MN": ("radon", "MN"), "radon_PA": ("radon", "PA"),
            "radon_IN": ("radon", "IN"), "radon_MO": ("radon", "MO"), "radon_ND": ("radon", "ND"),
            "radon_MA": ("radon", "MA"), "radon_AZ": ("radon", "AZ"), "radon_sd_AZ": ("radon_stddvs", "AZ"),
            "german": ("german_credit_lognormalcentered", None), "radon_sd_MN": ("radon_stddvs", "MN"),
            "funnel": ("neals_funnel", None), "election": ("election", None), "electric": ("electric", None),
            "time_series": ("time_series", None)}
out = {"versions": np.array("tensorflow %s, tensorflow_probability %s" % (tf.__version__, tfp.__version__))}


def reparam_dict(sp, a, b):
    d = collections.OrderedDict()
    for k, name in enumerate(sp.part_names):
        lo, hi = sp.offsets[k], sp.offsets[k + 1]
        d[name + "_a"] = np.asarray(a[lo:hi], np.float32).reshape(sp.part_shapes[k])
        d[name + "_b"] = np.asarray(b[lo:hi], np.float32).reshape(sp.part_shapes[k])
    return d


def density_section():
    gold = np.load(os.path.join(HERE, "density_golden.npz"))
    for mname, (rname, dataset) in REF_NAME.items():
        sp = helpers.spec(mname)
        x = gold[mname + "/x"].astype(np.float32)
        for kind in ("CP", "NCP", "VIP"):
            a, b = gold["%s/%s/a" % (mname, kind)], gold["%s/%s/b" % (mname, kind)]
            cfg = ref_models.get_model_by_name(rname, dataset=dataset)
            if kind == "CP":
                target = ref_graphs.make_cp_graph(cfg)[0]
                to_centered = lambda parts: parts
            elif kind == "NCP":
                target = ref_graphs.make_ncp_graph(cfg)[0]
                to_centered = cfg.to_centered
            else:
                rp = reparam_dict(sp, a, b)
                target = ref_graphs.make_dvip_graph(cfg, rp)[0]
                to_centered = cfg.make_to_centered(**rp)
            lps, grads, cents = [], [], []
            for i in range(x.shape[0]):
                parts = [tf.constant(x[i, sp.offsets[k]:sp.offsets[k + 1]].reshape(sp.part_shapes[k]))
                         for k in range(len(sp.part_names))]
                lp = target(*parts)
                g = tf.gradients(lp, parts)
                c = to_centered(parts)
                with tf.Session() as sess:
                    lp_, g_, c_ = sess.run((lp, g, c))
                lps.append(float(np.sum(lp_)))
                grads.append(np.concatenate([np.ravel(v) for v in g_]))
                cents.append(np.concatenate([np.ravel(v) for v in c_]))
            out["density/%s/%s/logp" % (mname, kind)] = np.array(lps)
            out["density/%s/%s/grad" % (mname, kind)] = np.array(grads)
            out["density/%s/%s/centred" % (mname, kind)] = np.array(cents)
            print("density", mname, kind, lps[0])


class ScriptedKernel(mcmc.TransitionKernel):
    """An inner 'HMC-like' kernel whose log acceptance ratio at step t is script[t]: the adaptation wrappers only read
    `log_accept_ratio` and the step size of the results they are given."""
    Results = collections.namedtuple("ScriptedResults", ["log_accept_ratio", "step_size", "t", "target_log_prob",
                                                         "is_accepted"])

    def __init__(self, script, step_size):
        self._script = tf.constant(script, tf.float32)
        self._step = tf.constant(step_size, tf.float32)
        self._parameters = dict(script=script, step_size=step_size)

    @property
    def is_calibrated(self):
        return True

    @property
    def parameters(self):
        return self._parameters

    def one_step(self, current_state, previous_kernel_results):
        t = previous_kernel_results.t
        la = tf.gather(self._script, t)
        return current_state, previous_kernel_results._replace(log_accept_ratio=la, t=t + 1)

    def bootstrap_results(self, init_state):
        return self.Results(log_accept_ratio=tf.zeros_like(self._script[0]), step_size=self._step, t=tf.constant(0),
                            target_log_prob=tf.zeros_like(self._script[0]), is_accepted=tf.constant(True))


def adaptation_section():
    rs = np.random.RandomState(0)
    n_steps, n_adapt, n = 60, 40, 7
    script = np.minimum(rs.randn(n_steps, n) * 1.5 - 0.3, 5.0).astype(np.float32)
    eps0 = np.full(n, 0.037, np.float32)
    for tag, wrap in (("dual", lambda k: mcmc.DualAveragingStepSizeAdaptation(k, num_adaptation_steps=n_adapt,
                                                                               step_size_setter_fn=lambda kr, s: kr._replace(step_size=s),
                                                                               step_size_getter_fn=lambda kr: kr.step_size,
                                                                               log_accept_prob_getter_fn=lambda kr: kr.log_accept_ratio)),
                      ("simple", lambda k: mcmc.SimpleStepSizeAdaptation(k, num_adaptation_steps=n_adapt, adaptation_rate=0.05,
                                                                          target_accept_prob=0.75,
                                                                          step_size_setter_fn=lambda kr, s: kr._replace(step_size=s),
                                                                          step_size_getter_fn=lambda kr: kr.step_size,
                                                                          log_accept_prob_getter_fn=lambda kr: kr.log_accept_ratio))):
        tf.reset_default_graph()
        kern = wrap(ScriptedKernel(script, eps0))
        state = tf.zeros([n])
        res = kern.bootstrap_results(state)
        sizes = []
        for _ in range(n_steps):
            state, res = kern.one_step(state, res)
            sizes.append(res.inner_results.step_size if hasattr(res, "inner_results") else res.new_step_size)
        with tf.Session() as sess:
            sess.run(tf.global_variables_initializer())
            out_sizes = sess.run(sizes)
        out[tag + "/log_accept"] = script
        out[tag + "/eps0"] = eps0
        out[tag + "/num_adaptation_steps"] = np.array(n_adapt)
        out[tag + "/step_size"] = np.array(out_sizes)       # [t] = the step size in force AFTER the update of step t+1
        print(tag, np.array(out_sizes)[:3, 0])


class CountingKernel(mcmc.TransitionKernel):
    def __init__(self):
        self._parameters = {}

    @property
    def is_calibrated(self):
        return True

    @property
    def parameters(self):
        return self._parameters

    def one_step(self, current_state, previous_kernel_results):
        return current_state + 1.0, previous_kernel_results

    def bootstrap_results(self, init_state):
        return ()


def schedule_section():
    tf.reset_default_graph()
    S, B = 11, 5
    states = mcmc.sample_chain(num_results=S, num_burnin_steps=B, current_state=tf.constant(0.0), kernel=CountingKernel(),
                               num_steps_between_results=1, trace_fn=None)
    with tf.Session() as sess:
        kept = sess.run(states)
    out["schedule/num_results"] = np.array(S); out["schedule/num_burnin_steps"] = np.array(B)
    out["schedule/kept_steps"] = np.asarray(kept)           # number of kernel steps taken when each result was recorded
    print("schedule", kept)


def ess_section():
    rs = np.random.RandomState(1)
    S, n = 2000, 6
    rho = np.array([0.0, 0.3, 0.6, 0.9, -0.4, 0.97])
    x = np.zeros((S, n), np.float32)
    for t in range(1, S):
        x[t] = rho * x[t - 1] + rs.randn(n) * np.sqrt(1 - rho ** 2)
    tf.reset_default_graph()
    ess = mcmc.effective_sample_size(tf.constant(x))
    with tf.Session() as sess:
        out["ess/ess"] = sess.run(ess)
    out["ess/series"] = x
    print("ess", out["ess/ess"])


def leapfrog_section():
    """One HMC step with the momentum draw replaced by a recorded array (best effort: patches tf.random.normal while the
    kernel graph is built; TFP 0.7's hmc.py draws momenta with tf.random.normal)."""
    gold = np.load(os.path.join(HERE, "density_golden.npz"))
    for mname in ("radon_MN", "8schools", "election"):
        rname, dataset = REF_NAME[mname]
        sp = helpers.spec(mname)
        cfg = ref_models.get_model_by_name(rname, dataset=dataset)
        target = ref_graphs.make_cp_graph(cfg)[0]
        x = gold[mname + "/x"].astype(np.float32)[0]
        rs = np.random.RandomState(5)
        p = rs.randn(sp.D).astype(np.float32)
        eps = np.full(sp.D, 1e-3, np.float32)
        parts = [tf.constant(x[sp.offsets[k]:sp.offsets[k + 1]].reshape(sp.part_shapes[k])) for k in range(len(sp.part_names))]
        steps = [tf.constant(eps[sp.offsets[k]:sp.offsets[k + 1]].reshape(sp.part_shapes[k])) for k in range(len(sp.part_names))]
        draws = [p[sp.offsets[k]:sp.offsets[k + 1]].reshape(sp.part_shapes[k]) for k in range(len(sp.part_names))]
        it = iter(draws)
        real_normal = tf.random.normal
        try:
            tf.random.normal = lambda shape, *a, **kw: tf.constant(next(it))
            kern = mcmc.HamiltonianMonteCarlo(target_log_prob_fn=target, step_size=steps, num_leapfrog_steps=4)
            res0 = kern.bootstrap_results(parts)
            _, res = kern.one_step(parts, res0)
            with tf.Session() as sess:
                prop, lar = sess.run((res.proposed_state, res.log_accept_ratio))
            out["leapfrog/%s/x" % mname] = x; out["leapfrog/%s/p" % mname] = p; out["leapfrog/%s/eps" % mname] = eps
            out["leapfrog/%s/proposed" % mname] = np.concatenate([np.ravel(v) for v in prop])
            out["leapfrog/%s/log_accept_ratio" % mname] = np.asarray(lar)
            print("leapfrog", mname, lar)
        except Exception as e:   # TFP internals differ from what this patch assumes: say so, keep the rest
            print("leapfrog section skipped for %s: %r" % (mname, e))
        finally:
            tf.random.normal = real_normal
        tf.reset_default_graph()


if __name__ == "__main__":
    density_section()
    adaptation_section()
    schedule_section()
    ess_section()
    leapfrog_section()
    np.savez_compressed(os.path.join(HERE, "reference_golden.npz"), **out)
    print("wrote", len(out), "arrays to reference_golden.npz")
