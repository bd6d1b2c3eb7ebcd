"""CPU: the C-ABI library builds, loads and exports every symbol that
include/autoreparam.h declares; no compute calls (there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# every way a Python file can pull in the oracle package: `import oracle`, `import x, oracle`, `from oracle[.x] import ...`,
# importlib.import_module("oracle") / __import__("oracle")
ORACLE_IMPORT = re.compile(r"^[ \t]*import[ \t]+(?:[\w.]+(?:[ \t]+as[ \t]+\w+)?[ \t]*,[ \t]*)*oracle\b(?![\w])"
                           r"|^[ \t]*from[ \t]+oracle(?:\.[\w.]+)?[ \t]+import\b"
                           r"|(?:import_module|__import__)\([ \t]*['\"]oracle['\"]", re.M)


@pytest.fixture(scope="module")
def built():
    import __graft_entry__ as ge
    ge.build()
    from autoreparam_amd import _lib
    return _lib


def _declared():
    src = open(os.path.join(ROOT, "include", "autoreparam.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(arp_[a-z_0-9]+)\s*\(", src)))


def test_header_symbols_exported(built):
    L = built.lib()
    names = _declared()
    assert len(names) >= 12
    for n in names:
        assert hasattr(L, n), n
    assert sorted(names) == sorted(built.SYMBOLS)
    assert L.arp_version() == 2


def test_struct_layout_matches_header(built):
    # sizes the C compiler gives the ABI structs (host compile of the header)
    import subprocess, tempfile
    code = '#include <stdio.h>\n#include "autoreparam.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu\\n",' \
           'sizeof(arp_dataset),sizeof(arp_hmc_config),sizeof(arp_hmc_io),sizeof(arp_interleaved_io),' \
           'sizeof(arp_vi_config),sizeof(arp_vi_io));return 0;}\n'
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(code)
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "t.c"), "-o",
                               os.path.join(d, "t")])
        out = subprocess.check_output([os.path.join(d, "t")]).decode().split()
    got = [ctypes.sizeof(x) for x in (built.Dataset, built.HmcConfig, built.HmcIO, built.InterleavedIO,
                                      built.ViConfig, built.ViIO)]
    assert [int(v) for v in out] == got


def test_bad_arguments_fail_loudly(built):
    L = built.lib()
    assert L.arp_model_create(None, None) != 0
    assert b"null" in L.arp_last_error()
    assert L.arp_model_dim(None) == -1


def test_product_path_does_not_import_oracle():
    pkg = os.path.join(ROOT, "autoreparam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".h", ".hip")):
                txt = open(os.path.join(dirpath, f)).read()
                assert not ORACLE_IMPORT.search(txt), (dirpath, f)
                assert "liboracle" not in txt, (dirpath, f)


def test_the_import_guard_sees_every_spelling():
    for line in ("import oracle", "  import oracle as o", "import helpers, oracle", "import a, oracle.x, b", "from oracle import build",
                 "from oracle.ess_ref import ess", "m = importlib.import_module('oracle')", 'x = __import__("oracle")'):
        assert ORACLE_IMPORT.search(line), line
    for line in ("import oracle_free_module", "# import oracle", "import helpers, oracles", "from oracles import x",
                 "the oracle is imported elsewhere"):
        assert not ORACLE_IMPORT.search(line), line


def test_only_the_checkers_import_the_oracle():
    """Outside tests/ the oracle is imported in exactly two places, both as the checker: bench.py's CPU-baseline leg and
    __graft_entry__ (build() compiles it, smoke() checks one small run against it).  The measurement scripts under tools/
    do not; the ones that compare against the oracle live under tests/diagnostics/."""
    hits = []
    for dirpath, dirs, files in os.walk(ROOT):
        dirs[:] = [d for d in dirs if d not in (".git", "gpurun_out", "tests", "oracle", "__pycache__", "build")]
        for f in files:
            if f.endswith(".py") or f.endswith(".sh"):
                txt = open(os.path.join(dirpath, f)).read()
                if ORACLE_IMPORT.search(txt):
                    hits.append(os.path.relpath(os.path.join(dirpath, f), ROOT))
    assert sorted(hits) == ["__graft_entry__.py", "bench.py"], hits
    src = open(os.path.join(ROOT, "bench.py")).read()
    for m in re.finditer(r"^(\s*)(import|from)\s+oracle\b", src, flags=re.M):
        assert len(m.group(1)) >= 4, "bench.py imports the oracle at module level"     # inside cpu_baseline() only


def test_enum_values_match_header(built):
    src = open(os.path.join(ROOT, "include", "autoreparam.h")).read()
    enums = {k: int(v) for k, v in re.findall(r"\b(ARP_[A-Z_0-9]+)\s*=\s*(\d+)", src)}
    assert len([k for k in enums if k.startswith("ARP_MODEL_")]) == 8
    for k, v in enums.items():
        if k.startswith("ARP_MODEL_"):
            assert getattr(built, k[4:]) == v, k          # _lib.MODEL_*
        elif k.startswith("ARP_ADAPT_"):
            assert getattr(built, k[4:]) == v, k          # _lib.ADAPT_*
