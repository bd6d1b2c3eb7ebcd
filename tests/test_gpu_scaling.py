"""GPU: the strong-scaling shard (BASELINE configs[3]: 65 536 chains over 8 GPUs = 8 192 per GPU) and the
multi-rank CLI path (one process per GPU, RCCL for the end-of-run statistics)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _time_interleaved(eng, sp, gpu, chains, lanes, T=64, reps=4):
    from autoreparam_amd import engine, _lib
    q0 = torch.as_tensor(helpers.states(sp, chains, seed=1, scale=0.1), device=gpu)
    st = engine.ChainState(q0)
    e = np.full(sp.D, 0.08, np.float32); e[2] = 0.02
    tr = torch.empty(T // 2, chains, sp.D, device=gpu)

    def run():
        eng.interleaved_run(st, e, e, 4, 4, T, seed=3, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10 ** 9, n_burnin=st.step,
                            thin=2, trace=tr, trace_centered=False, lanes=lanes)
    run(); run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        run()
    e1.record(); torch.cuda.synchronize()
    used = int((st.rng.view(st.rng.shape[0], -1, 4)[0, :, :2].abs().sum(dim=1) != 0).sum().item())   # RNG slots in use = lanes
    return e0.elapsed_time(e1) / reps, used


@pytest.mark.parametrize("chains", [8192, 65536])
def test_default_lanes_per_chain_is_the_fastest(gpu, chains):
    """arp_api.hip: select_ops -- the lanes-per-chain the library picks for the headline job (65 536 chains) and for
    its 8-GPU shard (8 192 chains) is the fastest of the instantiated splits {4, 8, 16}."""
    from autoreparam_amd import engine
    sp = helpers.spec("radon_PA")
    eng = engine.Engine(sp, gpu)
    eng.set_param(0, "CP"); eng.set_param(1, "NCP")
    t = {k: _time_interleaved(eng, sp, gpu, chains, k)[0] for k in (4, 8, 16)}
    t0, k0 = _time_interleaved(eng, sp, gpu, chains, 0)
    best = min(t, key=t.get)
    assert k0 in (4, 8, 16)
    assert t[k0] <= 1.05 * t[best], (k0, t)     # 5 %: run-to-run spread of sub-millisecond launches
    assert t0 <= 1.10 * t[best], (t0, t)


def _free_port():
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _two_rank_env():
    """Two GPUs: one rank per GPU over RCCL, as in production.  One GPU (the test box): both ranks share it and the
    end-of-run exchange runs over gloo (ARP_SHARE_GPU / ARP_DIST_BACKEND, test-only switches of main.py and bench.py) --
    everything but the transport is the production path: the launcher, the sharding, the streams keyed by the global
    chain id, the kernels, the gathers and the files rank 0 writes."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    if torch.cuda.device_count() < 2:
        env.update(ARP_DEBUG="1", ARP_SHARE_GPU="1", ARP_DIST_BACKEND="gloo")
    return env


def test_two_rank_cli_equals_one_rank(tmp_path):
    """python -m torch.distributed.run --nproc-per-node 2 -m autoreparam_amd.main --inference=HMC: chains sharded over two
    ranks (streams keyed by the global chain id), per-chain minimum ESS all-gathered and acceptance counts
    all-reduced -- the rank-0 JSON and ESS files equal the single-process run bitwise."""
    env = _two_rank_env()
    out = {}
    for tag, launcher in (("one", [sys.executable, "-m", "autoreparam_amd.main"]),
                          ("two", [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                                   "--master-addr=127.0.0.1", "--master-port=%d" % _free_port(), "-m", "autoreparam_amd.main"])):
        d = str(tmp_path / tag)
        base = ["--model=radon", "--dataset=MN", "--method=CP", "--results_dir=" + d, "--num_chains=128", "--seed=3"]
        hm = ["--num_samples=200", "--num_burnin_steps=200", "--num_adaptation_steps=150", "--num_leapfrog_steps=4"]
        subprocess.check_call([sys.executable, "-m", "autoreparam_amd.main"] + base +
                              ["--inference=VI", "--num_optimization_steps=300"], env=env, cwd=ROOT)
        subprocess.check_call(launcher + base + ["--inference=HMC"] + hm, env=env, cwd=ROOT, timeout=600)
        out[tag] = (json.load(open(os.path.join(d, "CP_tied.json"))), np.load(os.path.join(d, "CP_tied_ess.npz")))
    j1, j2 = out["one"][0], out["two"][0]
    assert j1["learned_variational_params"] == j2["learned_variational_params"]      # same VI fit (rank 0 runs it)
    for k in ("ess_min", "sem_min", "acceptance_rate"):
        assert j1[k] == j2[k], (k, j1[k], j2[k])
    # rank 0 saves the per-element ESS of ALL chains (gathered from the ranks): the single-process arrays, bitwise
    for k in out["one"][1].files:
        a, b = out["one"][1][k], out["two"][1][k]
        assert a.shape == b.shape and np.array_equal(a, b), k


def test_two_rank_vi_runs_on_rank_zero_only(tmp_path):
    """--inference=VI under torch.distributed.run with two ranks: rank 0 fits and writes the JSON, rank 1 leaves at once,
    no process group is created (nothing to exchange) and the launcher returns cleanly; the fit equals the
    single-process fit bit for bit."""
    env = _two_rank_env()
    base = ["--model=radon", "--dataset=MN", "--method=CP", "--num_chains=64", "--seed=3", "--inference=VI",
            "--num_optimization_steps=300"]
    d1, d2 = str(tmp_path / "one"), str(tmp_path / "two")
    subprocess.check_call([sys.executable, "-m", "autoreparam_amd.main"] + base + ["--results_dir=" + d1], env=env, cwd=ROOT)
    subprocess.check_call([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                           "--master-addr=127.0.0.1", "--master-port=%d" % _free_port(), "-m", "autoreparam_amd.main"] + base +
                          ["--results_dir=" + d2], env=env, cwd=ROOT, timeout=600)
    j1, j2 = (json.load(open(os.path.join(d, "CP_tied.json"))) for d in (d1, d2))
    for k in ("elbo", "learning_rate", "learned_variational_params", "initial_step_size"):
        assert j1[k] == j2[k], k


@pytest.mark.parametrize("scaling", ["weak", "strong"])
def test_two_rank_bench_line(scaling):
    """bench.py as the driver launches it for N = 2 (torch.distributed.run, one rank per GPU): one JSON line from rank 0
    with the whole-job value, both ranks' kernel times and the end-of-run exchange."""
    env = _two_rank_env()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr=127.0.0.1",
           "--master-port=%d" % _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
           "--chains", "4096", "--transitions", "32", "--scaling", scaling]
    r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    j = check_driver_line("\n".join(lines))
    assert j["n_gpus"] == 2 and j["ranks"] == 2 and j["scaling"] == scaling and j["steps"] == 3
    # the other reading of the job, timed in the same run: strong (4 096 chains in total) beside weak and the reverse
    o = j["scaling_strong" if scaling == "weak" else "scaling_weak"]
    assert o["chains_per_gpu"] == (2048 if scaling == "weak" else 4096) and o["chains_total"] == 2 * o["chains_per_gpu"]
    assert o["leapfrog_steps_per_s"] > 0 and o["speedup_vs_1gpu_projected"] > 0
    # the transport is reported as what it was: RCCL on a two-GPU box, gloo when both ranks share the test box's GPU
    if torch.cuda.device_count() >= 2:
        assert j["dist_backend"] == "nccl" and j["rccl_ranks"] == 2
    else:
        assert j["dist_backend"] == "gloo" and "rccl_ranks" not in j
    per_gpu = 4096 if scaling == "weak" else 2048
    assert j["config"]["chains_per_gpu"] == per_gpu and j["config"]["chains_total"] == 2 * per_gpu
    assert len(j["kernel_ms_per_rank"]) == 2 and min(j["kernel_ms_per_rank"]) > 0
    # the whole-job value: all ranks' leapfrog steps over the slowest rank's wall time
    assert abs(j["value"] - 2 * per_gpu * 32 * 8 * 3 / (j["ms_per_step"] * 3e-3)) <= 1e-6 * j["value"]
    assert 0.3 < j["accept_rate"] < 1.0


_RCCL_SCRIPT = r"""
import json, os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, os.environ["ARP_ROOT"])
from autoreparam_amd import parallel
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
try:
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
except TypeError:
    dist.init_process_group("nccl", rank=0, world_size=1)
assert parallel.live() and parallel.world() == (0, 1) and dist.get_backend() == "nccl"
rs = np.random.RandomState(0)
# all-gather of a per-chain statistic: fixed-length blocks (float32) and variable-length ones (sizes gathered first, int64)
x = rs.rand(1000).astype(np.float32)
g = parallel.all_gather_chains(torch.as_tensor(x), 1000, dev)
assert g.is_cuda and np.array_equal(g.cpu().numpy(), x)
g = parallel.all_gather_chains(torch.as_tensor(x[:37]), None, dev)
assert g.is_cuda and np.array_equal(g.cpu().numpy(), x[:37])
g2 = parallel.all_gather_chains(torch.as_tensor(x.reshape(250, 4)), 250)        # device defaulted to the current GPU
assert g2.is_cuda and np.array_equal(g2.cpu().numpy(), x.reshape(250, 4))
# float64 all-reduce (acceptance counts)
t = parallel.all_reduce_sum(12345678.25, dev)
assert t.is_cuda and t.dtype == torch.float64 and float(t.item()) == 12345678.25
# the summaries main.py forms from them
ess = [rs.rand(64, 1).astype(np.float32) + 1.0, rs.rand(64, 5).astype(np.float32) + 1.0]
acc = rs.rand(20, 64) < 0.7
e, s, a, mins = parallel.summarize(ess, acc, 20, 64, device=dev)
want = np.minimum(ess[0].min(1), ess[1].min(1))
assert np.array_equal(mins, want) and abs(e - want.mean()) < 1e-6 and abs(a - 100.0 * acc.sum() / (20 * 64)) < 1e-9
parts = parallel.gather_parts(ess, 64, dev)
assert all(np.array_equal(p, q) for p, q in zip(parts, ess))
parallel.barrier()
torch.cuda.synchronize()
print(json.dumps({"dist_backend": dist.get_backend(), "ranks": dist.get_world_size(), "ok": True}))
dist.destroy_process_group()
"""


def test_rccl_collectives_run_at_world_size_one():
    """backend="nccl" IS RCCL: a process group of ONE rank on the box's GPU drives every collective of parallel.py --
    the padded float32 all-gather, the int64 gather of block lengths, the float64 all-reduce, gather_parts, summarize,
    the barrier -- through real RCCL calls on device tensors (a group of one takes no short-cut: parallel.live())."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0",
               ARP_ROOT=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, "-c", _RCCL_SCRIPT], env=env, cwd=ROOT, timeout=600, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j == {"dist_backend": "nccl", "ranks": 1, "ok": True}


def test_one_rank_bench_under_the_launcher_reports_rccl():
    """bench.py under torch.distributed.run with ONE rank: the process group is created (RCCL), the end-of-run exchange
    of the timed run goes through it, and the line says so (`dist_backend: "nccl"`, `rccl_ranks: 1`)."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr=127.0.0.1",
           "--master-port=%d" % _free_port(), os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1",
           "--chains", "4096", "--transitions", "32", "--headline-only", "--no-cpu-baseline", "--no-ess"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert j["dist_backend"] == "nccl" and j["rccl_ranks"] == 1 and j["n_gpus"] == 1 and j["ranks"] == 1
    assert j["stats_allgather_s"] > 0 and 0.3 < j["accept_rate"] < 1.0


REQUIRED_LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                      "vs_baseline", "dtype", "data", "config", "roofline", "ess_per_sec", "ranks", "dist_backend")
REQUIRED_ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "kernel_ms",
                          "algorithmic_flop_per_leapfrog", "hbm")


def check_driver_line(stdout, combined=None):
    """What the round-end driver needs from `python bench.py`: stdout is ONE line, shorter than 4 KB, JSON, with the
    contract's keys; and it is also the last line of stdout + stderr read as one stream."""
    lines = stdout.splitlines()
    assert len(lines) == 1, stdout[:2000]
    assert len(lines[0]) < 4096, len(lines[0])
    j = json.loads(lines[0])
    for k in REQUIRED_LINE_KEYS:
        assert k in j, k
    for k in REQUIRED_ROOFLINE_KEYS:
        assert k in j["roofline"], k
    assert "workload" in j["config"] and "model" not in j["config"]
    if combined is not None:
        last = [l for l in combined.splitlines() if l.strip()][-1]
        assert json.loads(last) == j
    return j


def test_bench_stdout_is_one_json_line(tmp_path):
    """`python bench.py` the way the driver runs the N = 1 bench: stdout holds the one short JSON line and nothing else; the
    CLI flows it drives (VI fits, tuning runs, the ESS run) are silent, so the line is also the last line of the combined
    stream; the long secondary figures are in the extras file the line names."""
    env = dict(os.environ, PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    ex = str(tmp_path / "extras.json")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "2", "--warmup", "1", "--chains", "4096", "--transitions", "32",
           "--extras", ex]
    r = subprocess.run(cmd, env=env, cwd=ROOT, timeout=900, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    j = check_driver_line(r.stdout)
    assert j["n_gpus"] == 1 and j["ess_per_sec"] > 0
    assert "finished optimization" not in r.stderr and "ESS" not in r.stderr
    cb = j["cpu_baseline"]
    assert cb["value"] > 0 and cb["cores"] >= 1 and cb["kind"] == "port" and cb["unit"] == "leapfrog-steps/s" and cb["sample"]
    full = json.load(open(ex))
    assert j["extras"] == ex
    for k in ("other_models", "vi_kernel", "german_credit", "election", "strong_shard", "ess", "ess_kernel"):
        assert k in full, k
    assert "profile" not in j["roofline"]
    # both streams as the driver's log has them
    r2 = subprocess.run(cmd + ["--no-cpu-baseline", "--headline-only"], env=env, cwd=ROOT, timeout=900, stdout=subprocess.PIPE,
                        stderr=subprocess.STDOUT, text=True)
    assert r2.returncode == 0
    last = [l for l in r2.stdout.splitlines() if l.strip()][-1]
    assert json.loads(last)["n_gpus"] == 1 and len(last) < 4096
