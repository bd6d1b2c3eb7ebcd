"""Build libautoreparam_hip.so in-tree with hipcc for gfx950 (no JIT cache: the
built library travels with the source tree)."""
import concurrent.futures
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
TAG = os.environ.get("ARP_BUILD_TAG", "")   # experiments only: separate objects and library name
OUT = os.path.join(HERE, "libautoreparam_hip%s.so" % TAG)
OBJ = os.path.join(HERE, "build" + TAG)
ARCH = "gfx950"
FLAGS = ["-O3", "-fno-slp-vectorize", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-Wall", "-Wno-unused-function"]
FLAGS += os.environ.get("ARP_HIPCC_FLAGS", "").split()   # experiments only (e.g. -DNAME for a timing variant)
# per-file flags.  German credit's matrix-core likelihood: let the MFMAs write VGPRs (the forward product's result is
# consumed by VALU instructions at once; from AGPRs every element costs a v_accvgpr_read first: 601 -> 460 instructions
# per 128-row tile)
FILE_FLAGS = {"inst_german.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form"]}


def _sources():
    return sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "autoreparam.h"))
    jobs = []
    objs = []
    for src in _sources():
        s = os.path.join(CSRC, src)
        o = os.path.join(OBJ, src[:-4] + ".o")
        objs.append(o)
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc] + FLAGS + FILE_FLAGS.get(src, []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd))
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        return r.stdout

    with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        for out in ex.map(run, jobs):
            if verbose and out.strip():
                print(out)
    if jobs or force or _stale(OUT, objs):
        run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC"] + objs + ["-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
