#!/usr/bin/env python3
"""Pin what CAN be pinned to the reference itself: the TensorFlow-free parts of /root/reference.

The reference's hot path (TF 1.14 / TFP 0.7) cannot run in the build container, so the oracle's TFP arithmetic stays
unpinned.  But the inputs of that path and the host bookkeeping around it are plain numpy /
pandas code in the reference, and THOSE are executed here, from the reference's own source text, and their outputs
frozen as numbers:

  data/<name>/...        what the reference's loaders hand to its models
                         load_radon_data(state) for the seven states          (models.py:706-760)
                         load_german_credit_data()                            (models.py:860-881)
                         data/election88.py, data/electric.py `data` dicts    (models.py:984-989, 1037-1045)
                         eight-schools constants, the time-series lists       (models.py:134-137, 1096-1112)
  util/...               get_approximate_step_size, stddvs_to_mcmc_step_sizes, variational_inits_from_params (numpy's
                         global RNG seeded before the call), get_min_ess, get_min_ess_other/reject_outliers,
                         estimate_true_mean, compute_V_* / condition_number_*  (util.py:65-88, 271-276, 308-331, 394-460)
  main/...               get_best_num_leapfrog_steps_from_tuning_runs         (main.py:292-294)
  flags                  every flags.DEFINE_* name and default of main.py / program_transformations.py (the CLI surface)

How: the functions are AST-EXTRACTED from the reference's files -- the modules are never imported (their top level
imports TensorFlow) -- and exec'd in a namespace that holds numpy, pandas, collections, a FLAGS stand-in and an
`open_data_file` bound to plain `open` on /root/reference/data (the reference's one goes through tf.io.gfile).  The two
data modules are pure Python literals and are loaded as they are.  Nothing of the reference's text is stored: the output
tests/golden/reference_host_golden.npz (+ .json for the flag table) holds inputs and outputs only.

Usage (build container only; /root/reference does not exist on the GPU box):
    python tests/golden/make_reference_host_golden.py            # REFERENCE_DIR=/root/reference
tests/test_reference_host_golden.py holds autoreparam_amd/data/*.npz, autoreparam_amd/util.py, main.py and flags.py to it.
"""
import ast
import collections
import contextlib
import importlib.util
import json
import os
import sys
import types

import numpy as np
import pandas as pd

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("REFERENCE_DIR", "/root/reference")
STATES = ("MN", "PA", "IN", "MO", "ND", "MA", "AZ")


def _tree(fname):
    with open(os.path.join(REF, fname)) as f:
        return ast.parse(f.read(), filename=fname)


def extract(fname, names, ns):
    """exec the top-level `def`s called `names` of reference file `fname` in namespace `ns`"""
    tree = _tree(fname)
    found = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    missing = set(names) - {n.name for n in found}
    if missing:
        raise SystemExit("%s: no top-level def %s" % (fname, sorted(missing)))
    mod = ast.Module(body=found, type_ignores=[])
    exec(compile(mod, os.path.join(REF, fname), "exec"), ns)
    return ns


def local_literals(fname, func, names, ns):
    """values of the simple assignments `name = <expr>` inside def `func` of `fname` (expressions over numpy literals)"""
    tree = _tree(fname)
    fn = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == func][0]
    out = {}
    for node in fn.body:
        if isinstance(node, ast.Assign) and len(node.targets) == 1 and isinstance(node.targets[0], ast.Name) \
                and node.targets[0].id in names:
            out[node.targets[0].id] = eval(compile(ast.Expression(node.value), fname, "eval"), dict(ns))
    assert set(out) == set(names), (func, sorted(set(names) - set(out)))
    return out


def flag_table():
    """name -> [kind, default] of every flags.DEFINE_*(name, default=...) call of the CLI's files"""
    table = collections.OrderedDict()
    for fname in ("main.py", "program_transformations.py", "interleaved.py", "inference.py", "graphs.py", "util.py",
                  "models.py"):
        for node in ast.walk(_tree(fname)):
            if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr.startswith("DEFINE_") \
                    and isinstance(node.func.value, ast.Name) and node.func.value.id == "flags":
                args = list(node.args)
                kw = {k.arg: k.value for k in node.keywords}
                name = ast.literal_eval(args[0] if args else kw["name"])
                default = kw.get("default", args[1] if len(args) > 1 else None)
                table[name] = [node.func.attr[len("DEFINE_"):], ast.literal_eval(default) if default is not None else None,
                               fname]
    return table


def data_module(name):
    spec = importlib.util.spec_from_file_location("_ref_data_" + name, os.path.join(REF, "data", name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.data


def main():
    import warnings
    warnings.simplefilter("ignore", FutureWarning)      # pandas deprecations inside the reference's loaders
    if not os.path.isdir(REF):
        raise SystemExit("make_reference_host_golden.py: %s is not there (build container only)" % REF)
    out = {}

    # ---- loaders -------------------------------------------------------------------------------------------------
    @contextlib.contextmanager
    def open_data_file(fname):                       # the reference's goes through tf.io.gfile; same file, plain open
        with open(os.path.join(REF, "data", os.path.basename(fname)), "r") as f:
            yield f

    ns = {"np": np, "pd": pd, "os": os, "collections": collections, "open_data_file": open_data_file}
    extract("models.py", ["load_radon_data", "load_german_credit_data"], ns)
    for st in STATES:
        c, u, x, data = ns["load_radon_data"](st)
        out["data/radon_%s/c" % st] = np.asarray(c)
        out["data/radon_%s/u" % st] = np.asarray(u)
        out["data/radon_%s/x" % st] = np.asarray(x)
        out["data/radon_%s/data" % st] = np.asarray(data)
    numericals, categoricals, status = ns["load_german_credit_data"]()
    out["data/german/numericals"] = np.asarray(numericals)
    out["data/german/categoricals"] = np.stack([np.asarray(c) for c in categoricals], axis=1)
    out["data/german/status"] = np.asarray(status)
    for mod, keys in (("election88", ("N", "n_state", "black", "female", "state", "y")),
                      ("electric", ("N", "n_pair", "n_grade", "n_grade_pair", "grade", "grade_pair", "pair", "treatment", "y"))):
        d = data_module(mod)
        for k in keys:
            out["data/%s/%s" % (mod, k)] = np.asarray(d[k])
    lit = local_literals("models.py", "get_eight_schools", ["treatment_effects", "treatment_stddevs"], {"np": np})
    out["data/eight_schools/treatment_effects"] = lit["treatment_effects"]
    out["data/eight_schools/treatment_stddevs"] = lit["treatment_stddevs"]
    lit = local_literals("models.py", "get_time_series", ["x", "y"], {})
    out["data/time_series/x"] = np.asarray(lit["x"], np.float64)
    out["data/time_series/y"] = np.asarray(lit["y"], np.float64)

    # ---- util.py / main.py bookkeeping -----------------------------------------------------------------------------
    FLAGS = types.SimpleNamespace(num_chains=0)

    def quiet_print(*a, **k):
        pass
    uns = {"np": np, "collections": collections, "FLAGS": FLAGS, "print": quiet_print}
    extract("util.py", ["get_approximate_step_size", "stddvs_to_mcmc_step_sizes", "variational_inits_from_params",
                        "reject_outliers", "get_min_ess_other", "get_min_ess", "estimate_true_mean", "compute_V_cp",
                        "compute_V_ncp", "condition_number_cp", "condition_number_ncp"], uns)
    rs = np.random.RandomState(20261003)
    # a fitted mean-field posterior the way main.py stores it: <name>_loc / <name>_scale per model part
    shapes = collections.OrderedDict([("mu", ()), ("log_tau", ()), ("theta", (8,)), ("m", (5, 3))])
    vp = collections.OrderedDict()
    for k, shp in shapes.items():
        vp[k + "_loc"] = np.asarray(rs.randn(*shp), np.float32)
        vp[k + "_scale"] = np.asarray(np.exp(0.3 * rs.randn(*shp)), np.float32)
    for k, v in vp.items():
        out["util/vp/" + k] = np.asarray(v)
    for L in (1, 4, 7):
        for i, v in enumerate(uns["get_approximate_step_size"](vp, L)):
            out["util/approx_step/L%d/%d" % (L, i)] = np.asarray(v)
        for i, v in enumerate(uns["stddvs_to_mcmc_step_sizes"](vp, L)):
            out["util/stddvs_step/L%d/%d" % (L, i)] = np.asarray(v)
    np.random.seed(7)
    inits = uns["variational_inits_from_params"](vp, list(shapes), 6)
    for k, v in inits.items():
        out["util/inits/seed7_n6/" + k] = np.asarray(v)
    # ESS summaries: a list of per-part [C, *event] arrays, NaNs included (nan_to_num: they count as 0)
    C = 37
    ess = [np.abs(rs.randn(C)).astype(np.float32) * 100, np.abs(rs.randn(C, 8)).astype(np.float32) * 100,
           np.abs(rs.randn(C, 5, 3)).astype(np.float32) * 100]
    ess[1][3, 2] = np.nan
    ess[2][11, 4, 1] = np.inf
    for i, e in enumerate(ess):
        out["util/ess_in/%d" % i] = e
    FLAGS.num_chains = C
    out["util/get_min_ess"] = np.asarray(uns["get_min_ess"]([e.copy() for e in ess]), np.float64)
    ess_by_chain = [[e[c] for e in ess] for c in range(C)]
    out["util/get_min_ess_other"] = np.asarray(uns["get_min_ess_other"](ess_by_chain), np.float64)
    d = rs.randn(200)
    d[:3] = 25.0
    out["util/reject_outliers/in"] = d
    out["util/reject_outliers/out"] = np.asarray(uns["reject_outliers"](d))
    groups = [[rs.randn(50) + g, rs.randn(50, 2) - g] for g in range(3)]
    esss = [10.0, 30.0, 60.0]
    tm = uns["estimate_true_mean"](groups, esss)
    for g in range(3):
        for j in range(2):
            out["util/true_mean/in/%d/%d" % (g, j)] = groups[g][j]
        out["util/true_mean/out/%d" % g] = np.asarray(tm[g], np.float64)
    out["util/true_mean/esss"] = np.asarray(esss)
    qv = np.array([[0.1, 0.5], [1.0, 1.0], [3.0, 0.2], [10.0, 100.0]])
    out["util/qv"] = qv
    out["util/compute_V_cp"] = np.stack([uns["compute_V_cp"](q, v) for q, v in qv])
    out["util/compute_V_ncp"] = np.stack([uns["compute_V_ncp"](q, v) for q, v in qv])
    out["util/condition_number_cp"] = np.array([uns["condition_number_cp"](q, v) for q, v in qv])
    out["util/condition_number_ncp"] = np.array([uns["condition_number_ncp"](q, v) for q, v in qv])

    mns = {}
    extract("main.py", ["get_best_num_leapfrog_steps_from_tuning_runs"], mns)
    runs = [{"num_leapfrog_steps": int(L), "ess_min": float(e)} for L, e in
            zip([1, 2, 4, 8, 16, 32, 4], [3.1, 9.7, 22.0, 33.2, 33.2, 12.0, 21.0])]
    out["main/tuning_runs"] = np.array([[r["num_leapfrog_steps"], r["ess_min"]] for r in runs])
    out["main/best_num_leapfrog_steps"] = np.asarray(mns["get_best_num_leapfrog_steps_from_tuning_runs"](runs))
    out["main/best_num_leapfrog_steps_reversed"] = np.asarray(mns["get_best_num_leapfrog_steps_from_tuning_runs"](runs[::-1]))

    path = os.path.join(HERE, "reference_host_golden.npz")
    np.savez_compressed(path, **out)
    with open(os.path.join(HERE, "reference_host_flags.json"), "w") as f:
        json.dump(flag_table(), f, indent=1)
    print("wrote %s: %d arrays, %.1f KB; %d flags" % (path, len(out), os.path.getsize(path) / 1024.0, len(flag_table())))


if __name__ == "__main__":
    sys.exit(main())
