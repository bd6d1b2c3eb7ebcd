#!/usr/bin/env python3
"""Freeze the reference's datasets into small binary fixtures.

Runs ONLY in the build container (it reads /root/reference/data); the GPU box
and the package itself only ever see the frozen ``autoreparam_amd/data/*.npz``.

What is restated (behaviour, not code) and where it comes from:
  * radon      -- reference models.py:706-760 (``load_radon_data``), including
                  the uranium look-up quirk (row number of the cty/srrs merge is
                  indexed by the srrs first-appearance order of the county).
  * german     -- reference models.py:860-881 (``load_german_credit_data``) and
                  the design matrix built at models.py:889-892 (intercept,
                  standardised numerics with pandas' ddof=1 std, then one-hot
                  blocks per categorical column, in column order).
  * election88 -- reference models.py:984-989 reading data/election88.py.
  * 8schools   -- constants at reference models.py:134-137.
  * time_series -- the two constant lists at reference models.py:1096-1112 (years 1959-2018 and the
                  annual series they index).
  * electric   -- reference models.py:1038-1046 reading data/electric.py (pair, grade and
                  grade_pair stay 1-based, exactly as they are fed to tf.one_hot).

Pandas semantics are version sensitive, which is why
the result is frozen rather than re-derived per run.
"""
import argparse
import importlib.util
import os

import numpy as np
import pandas as pd

REF = "/root/reference/data"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..",
                   "autoreparam_amd", "data")


def radon(state_code):
    srrs = pd.read_csv(os.path.join(REF, "srrs2.dat"))
    srrs.columns = [c.strip() for c in srrs.columns]
    srrs["fips"] = srrs.stfips * 1000 + srrs.cntyfips
    st = srrs[srrs.state == state_code].copy()
    st["county"] = st.county.str.strip()

    cty = pd.read_csv(os.path.join(REF, "cty.dat"))
    cty_st = cty[cty.st == state_code].copy()
    cty_st["fips"] = 1000 * cty_st.stfips + cty_st.ctfips

    # first-appearance order of (county, fips) pairs in the survey table
    pairs = st[["county", "fips"]].drop_duplicates()
    first_seen = {}
    for pos, name in enumerate(pairs["county"]):
        first_seen[name] = pos          # later duplicates overwrite, as a dict does
    # uranium column of the county-table x pairs join, ordered like the county table
    uppm_joined = cty_st.merge(pairs, on="fips")["Uppm"].reset_index(drop=True)

    obs = st.merge(cty_st[["fips", "Uppm"]], on="fips")
    obs = obs.drop_duplicates(subset="idnum")
    names = obs.county.str.strip()
    uniq = list(pd.unique(names))
    code_of = {n: i for i, n in enumerate(uniq)}
    county = names.map(code_of).to_numpy().astype(np.int32)

    J = obs.groupby(names)["idnum"].count().shape[0]
    assert J == len(uniq)
    u_raw = np.zeros(J, dtype=np.float32)
    for name, j in code_of.items():
        u_raw[j] = uppm_joined[first_seen[name]]
    u = np.array([np.log(v) if v > 0.0 else 0.0 for v in u_raw], dtype=np.float32)

    x = obs.floor.to_numpy().astype(np.float32)
    y = np.log(obs.activity.to_numpy() + 0.1).astype(np.float32)
    return dict(county=county, u=u, x=x, y=y)


def german():
    df = pd.read_csv(os.path.join(REF, "german.data"), sep=r"\s+", header=None)
    cols_num = [np.ones(len(df))]
    cols_cat = []
    for c in df.columns[:-1]:
        col = df[c]
        if col.dtype == "O":
            levels = {lv: i for i, lv in enumerate(np.unique(col))}
            cols_cat.append(np.array([levels[v] for v in col], dtype=np.int32))
        else:
            cols_num.append(((col - col.mean()) / col.std()).to_numpy())
    numer = np.array(cols_num).T.astype(np.float32)          # [N, 8]
    blocks = [numer]
    for c in cols_cat:
        blocks.append(np.eye(c.max() + 1, dtype=np.float32)[c])
    X = np.concatenate(blocks, axis=1)
    yv = (df[20] == 1).to_numpy().astype(np.int32)
    cats = np.stack(cols_cat, axis=1).astype(np.int32)        # [N, 13]
    ncat = np.array([c.max() + 1 for c in cols_cat], dtype=np.int32)
    return dict(X=X, y=yv, numericals=numer, categoricals=cats, cat_sizes=ncat)


def election():
    spec = importlib.util.spec_from_file_location(
        "_e88", os.path.join(REF, "election88.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = mod.data
    return dict(n_state=np.int32(d["n_state"]),
                state=np.asarray(d["state"], dtype=np.int32),      # 1-based, kept as is
                female=np.asarray(d["female"], dtype=np.float32),
                black=np.asarray(d["black"], dtype=np.float32),
                y=np.asarray(d["y"], dtype=np.int32))


def electric():
    spec = importlib.util.spec_from_file_location(
        "_el", os.path.join(REF, "electric.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = mod.data
    assert d["N"] == len(d["y"]) and d["n_pair"] == len(d["grade_pair"])
    return dict(n_pair=np.int32(d["n_pair"]), n_grade=np.int32(d["n_grade"]),
                n_grade_pair=np.int32(d["n_grade_pair"]),
                pair=np.asarray(d["pair"], dtype=np.int32),              # 1-based, kept as is
                grade=np.asarray(d["grade"], dtype=np.int32),            # 1-based
                grade_pair=np.asarray(d["grade_pair"], dtype=np.int32),  # 1-based
                treatment=np.asarray(d["treatment"], dtype=np.float32),
                y=np.asarray(d["y"], dtype=np.float32))


def time_series():
    x = np.arange(1959, 2019, dtype=np.float32)
    y = np.array([
        315.97, 316.91, 317.64, 318.45, 318.99, 319.62, 320.04, 321.38, 322.16, 323.04, 324.62, 325.68, 326.32,
        327.45, 329.68, 330.18, 331.11, 332.04, 333.83, 335.4, 336.84, 338.75, 340.11, 341.45, 343.05, 344.65,
        346.12, 347.42, 349.19, 351.57, 353.12, 354.39, 355.61, 356.45, 357.1, 358.83, 360.82, 362.61, 363.73,
        366.7, 368.38, 369.55, 371.14, 373.28, 375.8, 377.52, 379.8, 381.9, 383.79, 385.6, 387.43, 389.9, 391.65,
        393.85, 396.52, 398.65, 400.83, 404.24, 406.55, 408.52], dtype=np.float32)
    assert len(x) == len(y) == 60
    return dict(x=x, y=y)


def schools():
    return dict(y=np.array([28, 8, -3, 7, -1, 1, 18, 12], dtype=np.float32),
                sigma=np.array([15, 10, 16, 11, 9, 11, 10, 18], dtype=np.float32))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=OUT)
    args = ap.parse_args()
    os.makedirs(args.out, exist_ok=True)
    for code in ("MN", "PA", "IN", "MO", "ND", "MA", "AZ"):      # the states of srrs2.dat (README.md:22: MA, IN, PA, MO, ND, AZ; MN the default)
        try:
            d = radon(code)
        except Exception as e:  # states whose county tables do not join
            print("radon", code, "skipped:", repr(e))
            continue
        print("radon", code, "N=%d J=%d" % (d["y"].shape[0], d["u"].shape[0]))
        np.savez_compressed(os.path.join(args.out, "radon_%s.npz" % code), **d)
    g = german()
    print("german X", g["X"].shape, "positives", int(g["y"].sum()))
    np.savez_compressed(os.path.join(args.out, "german_credit.npz"), **g)
    e = election()
    print("election N", e["y"].shape[0])
    np.savez_compressed(os.path.join(args.out, "election88.npz"), **e)
    np.savez_compressed(os.path.join(args.out, "eight_schools.npz"), **schools())
    np.savez_compressed(os.path.join(args.out, "time_series.npz"), **time_series())
    el = electric()
    print("electric N", el["y"].shape[0], "pairs", int(el["n_pair"]))
    np.savez_compressed(os.path.join(args.out, "electric.npz"), **el)


if __name__ == "__main__":
    main()
