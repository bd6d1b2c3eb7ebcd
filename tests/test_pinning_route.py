"""CPU: the route to PINNED parity stays runnable.

tests/golden/make_reference_golden.py records the reference's own numbers on a machine with TensorFlow 1.x / TFP, and
tests/test_reference_golden.py holds the oracle to them.  Neither can run here, so what CAN be checked is checked: the
generator's model list, states and (a, b) are exactly density_golden.npz's and helpers.MODEL_SPECS' (the first run on a
TensorFlow machine needs no edits), every key the consumer reads is a key the generator writes, the model names exist in
the reference, and the generator declines cleanly (exit code 3, nothing written) where its dependencies are missing."""
import ast
import os
import re
import subprocess
import sys

import numpy as np
import pytest

import helpers

HERE = os.path.dirname(os.path.abspath(__file__))
GEN = os.path.join(HERE, "golden", "make_reference_golden.py")
CONSUMER = os.path.join(HERE, "test_reference_golden.py")


def _ref_name_table():
    tree = ast.parse(open(GEN).read())
    for node in ast.walk(tree):
        if isinstance(node, ast.Assign) and any(getattr(t, "id", None) == "REF_NAME" for t in node.targets):
            return ast.literal_eval(node.value)
    raise AssertionError("REF_NAME not found in make_reference_golden.py")


def test_generator_covers_exactly_the_golden_model_list():
    ref_name = _ref_name_table()
    gold = np.load(os.path.join(HERE, "golden", "density_golden.npz"))
    in_gold = sorted(set(k.split("/")[0] for k in gold.files))
    assert sorted(ref_name) == in_gold == sorted(helpers.MODEL_SPECS)
    # the states and parameterisations the generator feeds the reference ARE the golden file's (it reads them from it)
    src = open(GEN).read()
    assert 'gold[mname + "/x"]' in src and 'gold["%s/%s/a" % (mname, kind)]' in src and 'gold["%s/%s/b" % (mname, kind)]' in src
    for m in in_gold:
        sp = helpers.spec(m)
        assert gold[m + "/x"].shape[1] == sp.D
        for kind in ("CP", "NCP", "VIP"):
            a, b = gold["%s/%s/a" % (m, kind)], gold["%s/%s/b" % (m, kind)]
            assert a.shape == b.shape == (sp.D,)
            ha, hb = helpers.params(sp, kind)
            assert np.array_equal(a, ha) and np.array_equal(b, hb), (m, kind)   # the seeded (a, b) of every other test


def test_reference_model_names_exist_in_the_reference():
    ref = os.environ.get("REFERENCE_DIR", "/root/reference")
    path = os.path.join(ref, "models.py")
    if not os.path.exists(path):
        pytest.skip("the reference tree is not on this machine")
    src = open(path).read()
    body = src[src.index("def get_model_by_name"):]
    known = set(re.findall(r"model == '([a-z0-9_]+)'", body))
    for mname, (rname, dataset) in _ref_name_table().items():
        assert rname in known, (mname, rname)
        # (a radon `dataset` is a state code the loader filters srrs2.dat by, models.py:706-719: any state in the file)
        assert dataset is None or (rname.startswith("radon") and re.fullmatch(r"[A-Z]{2}", dataset)), (mname, dataset)


def test_consumer_reads_only_keys_the_generator_writes():
    gen, con = open(GEN).read(), open(CONSUMER).read()
    written = set(re.findall(r'out\["([a-z_/%]+)"', gen)) | {"density/%s/%s/logp", "density/%s/%s/grad", "density/%s/%s/centred"}
    written |= {t + k for t in ("dual", "simple") for k in ("/log_accept", "/eps0", "/num_adaptation_steps", "/step_size")}
    read = set(re.findall(r'ref\["([a-z_/%]+)"', con))
    read |= {m.replace("tag + ", "") for m in re.findall(r'ref\[(tag \+ "[a-z_/]+")\]', con)}
    norm = lambda k: k.strip('"')
    missing = []
    for k in read:
        k = norm(k)
        if k.startswith("tag"):
            continue
        if k.startswith("/"):      # ref[tag + "/x"]
            if not any(w.endswith(k) for w in written):
                missing.append(k)
        elif k not in written:
            missing.append(k)
    assert not missing, missing
    # the consumer's parametrisations: every model the generator's density section writes, the leapfrog models it writes
    assert "list(helpers.MODEL_SPECS)" in con
    leap_gen = re.search(r'for mname in \(([^)]*)\):\s*\n\s*rname, dataset = REF_NAME\[mname\]', gen).group(1)
    leap_con = re.search(r'parametrize\("mname", \[([^\]]*)\]\)\s*\ndef test_leapfrog', con).group(1)
    assert sorted(re.findall(r'"(\w+)"', leap_gen)) == sorted(re.findall(r'"(\w+)"', leap_con))


def test_generator_declines_cleanly_without_tensorflow(tmp_path):
    try:
        import tensorflow  # noqa: F401
        pytest.skip("TensorFlow is importable here: run the generator itself")
    except Exception:
        pass
    before = set(os.listdir(os.path.join(HERE, "golden")))
    r = subprocess.run([sys.executable, GEN], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
    assert r.returncode == 3 and "nothing was written" in r.stderr
    assert set(os.listdir(os.path.join(HERE, "golden"))) == before
