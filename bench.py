#!/usr/bin/env python3
"""Headline benchmark: leapfrog-steps/sec over all chains (+ ESS/sec), radon(PA), 65 536 chains
per GPU (BASELINE.json metric; workload = configs[3]: radon --dataset=PA --method=i, interleaved
CP/NCP, 4 + 4 = "8 leapfrog steps" per step, main.py:493), on the fused HIP kernels.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one launch of the hot path over the rank's whole chain batch: `--transitions`
(default 1 024) sampler steps for every chain -- 13 ms per step, so that a handful of warm-up steps cover
the ~30 ms an idle MI355X takes to reach the clock it then holds (profiles/r03_clock_ramp.txt: the first
launches after idle take 5.7, 4.1, 3.8, 3.6, 3.55, 3.5, 3.4 ms per 256 steps, the following hundreds
3.36; with 256-step launches the driver's `--warmup 5` ended inside the ramp).  A sampler step (--method i) is one interleaved step:
num_ls leapfrogs in CP coordinates, to_ncp, num_ls leapfrogs in NCP coordinates, to_cp, two
Metropolis tests, two simple step-size adaptations (counted as 2*num_ls leapfrog steps; the
reference's two bootstrap gradient evaluations per step are not needed -- the kernel carries the
gradient across the change of coordinates -- and are not counted).  A CP trace row is written
every `--thin`-th step (default 2 = the reference's sample_chain(num_steps_between_results=1),
inference.py:228-236); `trace_every_step` re-times round 1's harsher variant (a row every step,
32 steps per launch).  Chains are independent, so ranks shard them with no data-path collective
(`--scaling weak`: the per-GPU batch is fixed; `--scaling strong`: `--chains` is the job total and
is split over the ranks, BASELINE configs[3] read literally); one RCCL all-gather of the per-chain
statistics runs after the timed region.

Prints ONE JSON line.  `roofline` prices the dominant kernel against the resource that binds it,
FP32 vector issue: achieved = SURVEY 8(d) algorithmic flops / kernel time
from HIP events of this run.  The HBM side is reported next to it (`hbm_algorithmic_GBps` is the
SURVEY 8(d) byte model, which the fused kernel does not perform; `traffic` is measured FETCH_SIZE +
WRITE_SIZE of a profiled pass of the SAME configuration, copied from profiles/ with its source).
"""
import argparse
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

FP32_PEAK_TFLOPS = 157.3      # MI355X vector FP32 (guides/MI355X_MICROARCH.md); the f32 MFMA peak is the same
HBM_PEAK_GBPS = 8000.0


def algorithmic_bytes_per_transition(D, trace=True):
    # read q, grad, logp, 3 adaptation scalars; write the same + accept byte (+ trace row)
    return 4 * ((5 if trace else 4) * D + 8) + 1


def radon_flop_per_leapfrog(J, D):
    # ~30 J + 20 per logp+grad, 4 D for the leapfrog update
    return 30.0 * J + 20.0 + 4.0 * D


# Vector-issue bound of the election kernels.  Their FMA count understates them: every state costs one v_exp and four
# v_rcp per gradient (eight v_log more in the closing pass), and transcendentals issue at a quarter of the FMA rate.
# Instruction counts per WAVE (16 chains at 4 lanes per chain) from the ISA of pk_hmc_kernel<ElectionPk<4,13>, MODE>
# (tools/asm_loops.py: interior leapfrog pass, closing pass, momentum draw + first drift + Metropolis + adaptation),
# priced with the issue costs measured on MI355X (tools/valu_bench.hip, cycles per wave instruction at >= 2 waves per
# SIMD): transcendental 8.2, v_pk_*_f32 4.4, other vector 2.6.
ISSUE_CYCLES = {"trans": 8.17, "pk": 4.43, "other": 2.6}     # "other" mixes two- and three-source forms (2.48 / 3.02)
ELECTION_MIX = {
    "NCP": {"interior": {"trans": 73, "pk": 149, "other": 50}, "closing": {"trans": 130, "pk": 238, "other": 80},
            "start": {"trans": 36, "pk": 43, "other": 105}},
    "tied_cVIP_b1": {"interior": {"trans": 74, "pk": 177, "other": 50}, "closing": {"trans": 131, "pk": 266, "other": 85},
                     "start": {"trans": 36, "pk": 43, "other": 105}},
}


# Issue costs of the vector instruction classes: cycles per wave-instruction per SIMD at EXACTLY two resident waves per SIMD
# (256-register kernels, 19 ms runs; tools/pk_vs_fma_2waves.hip, round 4), quoted at 2.4 GHz -- i.e. they are TIMES
# (1 cycle = 1 / 2.4 ns) measured at whatever clock the chip holds under a vector-bound load, so
#     sum(count x cost) / 2.4 GHz / measured time
# is a utilisation that does not depend on the clock and cannot exceed 1 (the costs are each class's best back-to-back
# rate).  Pricing the same costs at the held clock instead double-counts the clock (it read 1.02 for one election form --
# a utilisation above 1); that figure is no longer printed, its derivation stays in profiles/r04_issue_costs.txt.
ISSUE_COST_2W = {"pk": 4.43, "trans": 8.17, "dpp": 4.41, "mad_u64": 4.42, "half_rate": 4.19, "fma3": 3.02, "mov": 2.48, "other": 2.48}
# The headline kernel's own instruction mix: VALU instructions per WAVE (16 chains) and interleaved step (2 x 4 leapfrogs)
# by issue class, from the ISA (tools/asm_ledger.py, profiles/r04_headline_ledger.txt; the hardware's SQ_INSTS_VALU says
# 1 543 against the ledger's 1 568, which counts both arms of the rejection branch).
HEADLINE_COST = ISSUE_COST_2W
HEADLINE_MIX = {"pk": 840.0, "trans": 76.0, "dpp": 76.0, "mad_u64": 37.0, "half_rate": 60.5, "fma3": 65.0, "mov": 77.5, "other": 336.5}


_LEDGER_CLASS = {"pk_f32": "pk", "trans": "trans", "dpp": "dpp", "mad_u64": "mad_u64", "half_rate": "half_rate", "fma3": "fma3",
                 "mov": "mov", "valu_other": "other"}


def headline_mix():
    """The headline kernel's VALU mix per wave and step: the `weighted` row of the newest profiles/rNN_headline_ledger.txt
    (tools/asm_ledger.py over the ISA of the build that is benched); the round-4 constants above only if no ledger parses."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_headline_ledger.txt")), reverse=True):
        try:
            cols, row = None, None
            for ln in open(path):
                w = ln.split()
                if w[:2] == ["block", "instr"]:
                    cols = w[2:]
                elif w and w[0] == "weighted" and cols:
                    row = [float(v) for v in w[2:2 + len(cols)]]
            if cols and row:
                mix = {_LEDGER_CLASS[c]: v for c, v in zip(cols, row) if c in _LEDGER_CLASS}
                if set(mix) == set(HEADLINE_MIX):
                    return mix, "profiles/" + os.path.basename(path)
        except (OSError, ValueError):
            continue
    return dict(HEADLINE_MIX), "bench.py constants (profiles/r04_headline_ledger.txt)"


def headline_issue_bound(clock_ghz, leapfrogs_per_step=8, chains_per_wave=16, mix=None):
    """leapfrog-steps/s if all 1 024 SIMDs issued the headline step's instruction mix back to back"""
    cyc = sum(HEADLINE_COST[k] * v for k, v in (mix or HEADLINE_MIX).items())
    return 256 * 4 * clock_ghz * 1e9 / cyc * chains_per_wave * leapfrogs_per_step, cyc


def election_issue_bound(form, L, clock_ghz=2.4):
    """leapfrog-steps/s if the vector pipes of all 1 024 SIMDs issued this instruction mix back to back"""
    m = ELECTION_MIX[form]
    cyc = lambda b: sum(ISSUE_CYCLES[k] * v for k, v in b.items())
    per_transition = (L - 1) * cyc(m["interior"]) + cyc(m["closing"]) + cyc(m["start"])      # one wave = 16 chains
    return 256 * 4 * clock_ghz * 1e9 * 16 * L / per_transition, per_transition


def _time_launches(fn, n, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


def cpu_baseline(spec, L, n_chains, n_trans, eps0, lanes, inter):
    """The C oracle (oracle/, a port of the same algorithm) timed on this host's cores."""
    import oracle
    orc = oracle.OracleModel(spec)
    cp, ncp = spec.ab_from_reparam("CP"), spec.ab_from_reparam("NCP")
    rs = np.random.RandomState(0)
    q0 = (0.1 * rs.randn(n_chains, spec.D)).astype(np.float32)
    st = oracle.new_state(q0, np.float32)

    def run(n):
        if inter:
            orc.interleaved_run(st, cp, ncp, eps0, eps0, L, L, n, seed=1, adapt_kind=2, n_adapt=10 ** 6, lanes=lanes)
        else:
            orc.hmc_run(st, cp[0], cp[1], eps0, L, n, seed=1, adapt_kind=1, n_adapt=10 ** 6, lanes=lanes)
    t0 = time.time()
    run(8)                                                 # warm-up + calibration
    rate = 8.0 / (time.time() - t0)
    n_trans = int(min(max(n_trans, 12.0 * rate), 4096))   # about 12 s of CPU work
    t0 = time.time()
    run(n_trans)
    dt = time.time() - t0
    cores = len(os.sched_getaffinity(0))
    LL = 2 * L if inter else L
    out = {"value": n_chains * n_trans * LL / dt, "unit": "leapfrog-steps/s", "cores": cores, "kind": "port",
           "sample": "%d chains x %d %s x %d leapfrogs, float32 C oracle (oracle/oracle.c) with OpenMP over chains "
                     "on %d threads, %.1f s" % (n_chains, n_trans, "interleaved steps" if inter else "transitions",
                                                LL, cores, dt)}
    # the same port on ONE core (XLA:CPU's elementwise code is effectively single-threaded), in a child
    # process with OMP_NUM_THREADS=1 -- the thread count of this process (torch's pool included) is left alone
    try:
        import subprocess
        code = ("import sys, time, json, numpy as np; sys.path.insert(0, %r)\n"
                "import oracle\nfrom autoreparam_amd import models\n"
                "spec = models._spec_radon(%r); orc = oracle.OracleModel(spec)\n"
                "cp, ncp = spec.ab_from_reparam('CP'), spec.ab_from_reparam('NCP')\n"
                "q0 = (0.1 * np.random.RandomState(0).randn(256, spec.D)).astype(np.float32)\n"
                "st = oracle.new_state(q0, np.float32); e = np.asarray(%r, np.float32)\n"
                "t0 = time.time()\n"
                "%s\n"
                "print(json.dumps(time.time() - t0))\n") % (
                    ROOT, "PA" if spec.D == 71 else "MN", [float(v) for v in eps0],
                    ("orc.interleaved_run(st, cp, ncp, e, e, %d, %d, 1024, seed=1, adapt_kind=2, n_adapt=10**6, lanes=%d)" % (L, L, lanes))
                    if inter else
                    ("orc.hmc_run(st, cp[0], cp[1], e, %d, 1024, seed=1, adapt_kind=1, n_adapt=10**6, lanes=%d)" % (L, lanes)))
        env = dict(os.environ, OMP_NUM_THREADS="1")
        r1 = subprocess.run([sys.executable, "-c", code], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=120)
        d1 = float(json.loads(r1.stdout.strip().splitlines()[-1]))
        out["single_core"] = {"value": 256 * 1024 * LL / d1, "unit": "leapfrog-steps/s", "cores": 1,
                              "sample": "256 chains x 1024 steps x %d leapfrogs on one thread (child process, OMP_NUM_THREADS=1), %.1f s" % (LL, d1)}
    except Exception as e:
        out["single_core"] = {"value": None, "error": repr(e)}
    return out


def cpu_baseline_reference_shaped(spec, L, n_chains=128, budget_s=6.0, min_transitions=100, cap_s=30.0):
    """SURVEY.md 8(d) baseline (A): what the reference's XLA:CPU path executes, restated in torch on the host --
    float32, the county gather as a dense [N, J] one-hot matmul batched over chains, gradients by reverse-mode
    autodiff, one pass per leapfrog step (CP radon, momentum refresh and Metropolis test included).  Not the
    reference itself (TF 1.14 / TFP cannot be installed here), and not the oracle: a shape-faithful stand-in."""
    torch.manual_seed(0)
    r = spec.raw
    J, N = len(r["u"]), len(r["y"])
    onehot = torch.zeros(N, J); onehot[torch.arange(N), torch.as_tensor(r["county"], dtype=torch.long)] = 1.0
    u, x, y = (torch.as_tensor(np.asarray(r[k], np.float32)) for k in ("u", "x", "y"))

    def logp(q):   # q [C, 3+J]
        mua, b1, b2, m = q[:, 0:1], q[:, 1:2], q[:, 2:3], q[:, 3:]
        prior = -0.5 * (q[:, :3] ** 2).sum(1) - 0.5 * ((m - (mua + u * b1)) ** 2).sum(1)
        yhat = m @ onehot.t() + b2 * x          # [C, N]: the reference's tf.matmul(C_onehot, m) per chain
        return prior - 0.5 * ((y - yhat) ** 2).sum(1)

    def grad(q):
        q = q.detach().requires_grad_(True)
        lp = logp(q)
        (g,) = torch.autograd.grad(lp.sum(), q)
        return lp.detach(), g

    q = 0.1 * torch.randn(n_chains, 3 + J)
    eps = torch.full((3 + J,), 0.02)
    lp, g = grad(q)
    done, t0 = 0, time.time()
    while (time.time() - t0 < budget_s or done < min_transitions) and time.time() - t0 < cap_s:
        p = torch.randn_like(q)
        h0 = -lp + 0.5 * (p * p).sum(1)
        qn, pn = q, p + 0.5 * eps * g
        for l in range(L):
            qn = qn + eps * pn
            lpn, gn = grad(qn)
            pn = pn + (eps if l + 1 < L else 0.5 * eps) * gn
        acc = torch.rand(n_chains).log() < h0 - (-lpn + 0.5 * (pn * pn).sum(1))
        q = torch.where(acc[:, None], qn, q); g = torch.where(acc[:, None], gn, g); lp = torch.where(acc, lpn, lp)
        done += 1
    dt = time.time() - t0
    return {"value": n_chains * done * L / dt, "unit": "leapfrog-steps/s", "cores": torch.get_num_threads(),
            "kind": "reference-shaped stand-in",
            "sample": "%d chains x %d transitions x %d leapfrogs of CP radon(%s), torch CPU float32, dense [N=%d, J=%d] "
                      "one-hot matmul + autograd per leapfrog, %.1f s" % (n_chains, done, L, "PA" if J == 68 else "?",
                                                                          N, J, dt)}


def reference_flow_ess(dataset, chains, dev_index, samples=1000, burnin=1000, adapt=600, tune_chains=4096):
    """ESS/sec from the reference's own flow (main.py:190-231, 292-336, 452-528) on the engine's CLI: mean-field VI
    under CP and NCP -> HMCtuning sweeps (leapfrog counts 2, 4, 8) -> --inference=HMC --method=i at the headline
    chain count with the reference's thinning; ESS = tfp-style autocorrelation ESS of every centred element
    (arp_ess on the device trace), per chain the minimum over elements (util.get_min_ess)."""
    from autoreparam_amd import main as cli
    from autoreparam_amd.flags import FLAGS
    tmp = tempfile.mkdtemp(prefix="arp_bench_")
    try:
        base = ["--model=radon", "--dataset=%s" % dataset, "--results_dir=%s" % tmp, "--seed=1",
                "--device=cuda:%d" % dev_index]
        t0 = time.perf_counter()
        for m in ("CP", "NCP"):
            cli.main(base + ["--inference=VI", "--method=%s" % m], flags=FLAGS.copy())
        t_vi = time.perf_counter() - t0
        t0 = time.perf_counter()
        for m in ("CP", "NCP"):
            for L in (2, 4, 8):
                cli.main(base + ["--inference=HMCtuning", "--method=%s" % m, "--num_leapfrog_steps=%d" % L,
                                 "--num_chains=%d" % tune_chains, "--num_samples=300", "--num_burnin_steps=300",
                                 "--num_adaptation_steps=200"], flags=FLAGS.copy())
        t_tune = time.perf_counter() - t0
        # The sampling run below allocates its [S, C, D] trace (18.6 GB at the headline size) and two [S, C] acceptance
        # arrays.  The first allocation of that size in a process costs 0.01 - 0.5 s depending on the box (driver page
        # tables; it is not first touch by the kernel), which is allocator start-up rather than sampling: take it once
        # here, timed, and leave the block with torch's caching allocator as a long-lived process would have it.
        dev = torch.device("cuda", dev_index)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        warm = torch.empty(samples * chains * (4 * (3 + {"PA": 67, "MN": 85}.get(dataset, 96)) + 2) + (64 << 20),
                           dtype=torch.uint8, device=dev)
        torch.cuda.synchronize(dev)
        t_alloc = time.perf_counter() - t0
        del warm
        fl = FLAGS.copy()
        res = cli.main(base + ["--inference=HMC", "--method=i", "--num_chains=%d" % chains,
                               "--num_samples=%d" % samples, "--num_burnin_steps=%d" % burnin,
                               "--num_adaptation_steps=%d" % adapt], flags=fl)
        ess_norm, sem_norm, acc_cp, acc_ncp, mcmc_time = res[0], res[1], res[2], res[3], res[4]
        # arp_ess on the flow's own trace (the kept candidate's [S, C, D]): HIP events around the last call
        from autoreparam_amd import util as _util
        ess_call = None
        ev = getattr(_util.effective_sample_size, "last_events", None)
        if ev is not None:
            torch.cuda.synchronize(dev)
            ems = float(ev[0].elapsed_time(ev[1]))
            ess_call = {"kernel_ms": ems, "trace_bytes": ev[2], "algorithmic_GBps": ev[2] / (ems * 1e-3) / 1e9,
                        "frac_of_8TBps": ev[2] / (ems * 1e-3) / 1e9 / HBM_PEAK_GBPS}
        with open(os.path.join(tmp, "i_tied.json")) as f:
            saved = json.load(f)
        num_ls = int(saved["num_leapfrog_steps"][-1])
        LL = 2 * num_ls
        ess = float(ess_norm) * samples * LL / 1000.0      # undo main.py:362-366's "per 1000 gradients"
        total_steps = 1 + burnin + 2 * (samples - 1)
        return {"flow": "VI(CP), VI(NCP) -> HMCtuning L in {2,4,8} x {CP,NCP} (%d chains) -> HMC --method=i" % tune_chains,
                "chains": chains, "num_samples": samples, "num_burnin_steps": burnin, "num_adaptation_steps": adapt,
                "thinning": 2, "num_ls_chosen": num_ls, "mean_min_ess_per_chain": ess,
                "ess_min_per_1000_gradients": float(ess_norm), "sem_min_per_1000_gradients": float(sem_norm),
                "acceptance_rate_cp": float(acc_cp), "acceptance_rate_ncp": float(acc_ncp),
                "mcmc_time_sec": float(mcmc_time), "vi_time_sec": t_vi, "tuning_time_sec": t_tune,
                "trace_first_alloc_sec": t_alloc, "arp_ess_on_this_trace": ess_call,
                "ess_per_sec": ess / float(mcmc_time),
                # what a fresh process would see: the first device allocation of the trace block inside the clock
                "ess_per_sec_cold": ess / (float(mcmc_time) + t_alloc),
                "ess_per_sec_all_chains": ess * chains / float(mcmc_time),
                "leapfrog_steps_per_s_end_to_end": chains * total_steps * LL / float(mcmc_time),
                "note": "mcmc_time_sec is main.py's wall clock around the kept candidate's run: sampling, arp_ess over the "
                        "[S, C, D] device trace, host summaries; the trace block comes from the caching allocator "
                        "(trace_first_alloc_sec is what its first allocation cost)"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def load_profile(tag_cfg):
    """The newest profiles/rNN_headline.json (tools/summarize_profile.py), flagged with whether it was taken on this
    configuration."""
    import glob
    found = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_headline.json")))
    if not found:
        return None
    p = found[-1]
    try:
        prof = json.load(open(p))
    except Exception:
        return None
    same = all(prof.get("config", {}).get(k) == v for k, v in tag_cfg.items())
    prof["config_matches_this_run"] = bool(same)
    prof["source"] = "profiles/" + os.path.basename(p)
    return prof


def self_launch(n_gpus, argv):
    """Run this script on `n_gpus` ranks of this node through torch.distributed.run (one process per GPU, RCCL), the way
    the driver launches it for N > 1; returns the launcher's return code."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=65536, help="chains per GPU (weak) / in total (strong)")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    ap.add_argument("--method", default="i", choices=["i", "CP"])
    ap.add_argument("--leapfrog", type=int, default=8, help="leapfrog steps per sampler step (i: split CP/NCP)")
    ap.add_argument("--transitions", type=int, default=1024, help="sampler steps per launch (= per bench step)")
    ap.add_argument("--thin", type=int, default=2, help="a trace row every THIN-th sampler step (reference: 2)")
    ap.add_argument("--dataset", default="PA")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per chain (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the secondary figures: profiler runs, so that every launch of the headline kernel in "
                         "the trace is a timed or warm-up step")
    ap.add_argument("--no-ess", action="store_true", help="skip the reference-flow ESS/sec run")
    ap.add_argument("--verbose", action="store_true", help="let the CLI flows' progress through (to stderr)")
    ap.add_argument("--extras", default=os.path.join(ROOT, "bench_extras.json"),
                    help="where the secondary figures go (the stdout line stays short and names this path)")
    ap.add_argument("--no-trace", action="store_true", help="diagnostic: do not record trace rows")
    ap.add_argument("--stats", action="store_true", help="diagnostic: accumulate the in-kernel streaming statistics "
                                                         "(arp_hmc_io.stats) instead of writing trace rows")
    args = ap.parse_args()

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` outside a launcher: start the N ranks ourselves, as a CHILD process and before
        # anything in this process has touched the GPU (never an exec), relay the child's output (rank 0's one JSON
        # line goes to the inherited stdout) and leave with its return code
        sys.exit(self_launch(args.gpus, sys.argv[1:]))
    # stdout carries the ONE short JSON line and nothing else, and stderr stays quiet, so that the line is also the last
    # line of the combined stream: what the CLI flows print on the way (fits, tuning runs) is dropped (--verbose: to stderr)
    json_out = sys.stdout
    sys.stdout = sys.stderr if args.verbose else open(os.devnull, "w")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    # under a launcher (RANK and MASTER_PORT set) the process group is created for ANY world size: a one-rank launch
    # runs the same end-of-run exchange through RCCL that an N-rank launch does
    under_launcher = "RANK" in os.environ and "MASTER_PORT" in os.environ
    from autoreparam_amd.util import debug_switch
    if debug_switch("ARP_SHARE_GPU") and torch.cuda.device_count():
        local_rank %= torch.cuda.device_count()               # tests only (ARP_DEBUG=1): several ranks on one GPU (gloo backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev                                                # where the tensors of a collective live
    dist = None
    if world > 1 or under_launcher:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = debug_switch("ARP_DIST_BACKEND") or "nccl"  # "nccl" is RCCL; "gloo": tests that share one GPU
        if backend != "nccl":
            dist.init_process_group(backend)
            cdev = torch.device("cpu")
        else:
            try:
                dist.init_process_group("nccl", device_id=dev)   # RCCL; binds the communicator to this rank's GPU
            except TypeError:                                     # older torch: no device_id keyword
                dist.init_process_group("nccl")

    from autoreparam_amd import models, engine, _lib, parallel, util
    spec = models._spec_radon(args.dataset)
    eng = engine.Engine(spec, dev)
    eng.set_param(0, "CP")
    eng.set_param(1, "NCP")
    L, T, D = args.leapfrog, args.transitions, spec.D
    J = D - 3
    if args.scaling == "strong":
        lo, hi = parallel.shard_bounds(args.chains, rank, world)
        C, chain_offset, C_total = hi - lo, lo, args.chains
    else:
        C, chain_offset, C_total = args.chains, rank * args.chains, world * args.chains
    inter = args.method == "i"
    num_ls = L // 2
    thin = max(1, args.thin)

    # synthetic chain population: i.i.d. draws keyed by the global chain id
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    q0 = (0.1 * torch.randn(C, D, generator=g)).to(dev)
    eps0 = np.full(D, 0.08 / (L / 4.0) ** 2, np.float32)
    eps0[2] = 0.02 / (L / 4.0) ** 2
    eps_i = np.full(D, 0.08 / (max(num_ls, 1) / 4.0) ** 2, np.float32)   # interleaved: eps0/(num_ls/4)^2
    eps_i[2] = 0.02 / (max(num_ls, 1) / 4.0) ** 2

    def make_launcher(Tn, thin_n, chains=None, record=True, plain=False, stats=False, lanes=args.lanes, offset=None, q_init=None):
        """A closure that advances a fresh population by Tn sampler steps per call, with its own trace ring."""
        qq = q_init if q_init is not None else (q0 if chains is None else q0[:chains])
        off = chain_offset if offset is None else offset
        Cn = qq.shape[0]
        st = engine.ChainState(qq)
        rows = (Tn + thin_n - 1) // thin_n
        trace = torch.empty(rows, Cn, D, dtype=torch.float32, device=dev) if (record and not stats) else None
        stats_t = torch.zeros(6, Cn, D, dtype=torch.float32, device=dev) if stats else None
        skw = dict(stats=stats_t, stats_batch=64, n_samples=1 << 30) if stats else {}   # the CLI's batches are S/8 samples

        def launch():
            # rows cycle through a bounded buffer (n_burnin = steps done: row 0 is this launch's first sample)
            if inter and not plain:
                eng.interleaved_run(st, eps_i, eps_i, num_ls, num_ls, Tn, seed=7, chain_offset=off,
                                    adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10 ** 9, adapt_target=0.75, adapt_rate=0.05,
                                    n_burnin=st.step if not stats else 0, thin=thin_n, trace=trace,
                                    trace_centered=False, lanes=lanes, **skw)
            else:
                eng.hmc_run(st, eps0, L, Tn, seed=7, chain_offset=off, adapt_kind=_lib.ADAPT_DUAL,
                            n_adapt=10 ** 9, n_burnin=st.step if not stats else 0, thin=thin_n, trace=trace,
                            trace_centered=True, lanes=lanes, **skw)
        return launch, st

    rec = not args.no_trace

    def timed_leg(launch):
        """W untimed warm-up steps, then exactly K steps between barrier + synchronize on both sides; returns this
        rank's wall time and its per-launch HIP-event times."""
        for _ in range(args.warmup):
            launch()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            ev[k][0].record()
            launch()
            ev[k][1].record()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        return dt, [float(a.elapsed_time(b)) for a, b in ev]

    launch, st = make_launcher(T, thin, record=rec, stats=args.stats)
    elapsed, per_launch_ms = timed_leg(launch)
    kern_ms = float(np.mean(per_launch_ms))
    try:
        relay_segments = eng.relay_geometry()["segments"]     # of the last timed launch (arp_relay_geometry)
    except Exception:
        relay_segments = None
    # the shader clock this box holds under a vector-bound load, measured now, while the chip is at the temperature and power
    # state of the timed region (arp_clock_probe: s_memtime / s_memrealtime around 10 ms of packed FMAs on every SIMD)
    try:
        clock_live = engine.clock_probe(dev, 10.0)
    except Exception:
        clock_live = None

    # end-of-run statistics exchange (the only collectives of the path, parallel.py): all-gather of a per-chain
    # statistic (here the acceptance rate; a sampling run gathers the per-chain minimum ESS the same way) and the
    # all-reduce of the acceptance counts
    acc = st.accept_count.float() / st.step
    t_coll = 0.0
    rank_ms = [kern_ms]
    if dist is not None:
        tm = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        elapsed = float(tm.item())
        # every rank's own kernel time (HIP events): the spread is the load imbalance of the sharding
        km = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(km, torch.tensor([kern_ms], dtype=torch.float64, device=cdev))
        rank_ms = [float(t.item()) for t in km]
        torch.cuda.synchronize(); tc = time.perf_counter()
        acc = parallel.all_gather_chains(acc, C_total, dev)
        tot = parallel.all_reduce_sum(float(st.accept_count.sum().item()), dev)
        torch.cuda.synchronize(); t_coll = time.perf_counter() - tc
        assert abs(float(tot.item()) - float(acc.double().sum().item()) * st.step) < 1e-3 * max(1.0, float(tot.item()))
    assert torch.isfinite(st.q).all(), "non-finite chain state"
    accept_rate = float(acc.mean().item())

    LL = 2 * num_ls if inter else L
    extras = {}
    secondary = world == 1 and not args.headline_only

    # N > 1: the OTHER reading of the job, timed in the same run (same W and K, same barriers).  The headline leg is
    # `--scaling` (default weak: --chains per GPU); the second leg is strong (--chains in total, C/N per rank: BASELINE
    # configs[3] read literally) or, under --scaling strong, weak.  In the weak leg every GPU runs exactly the 1-GPU job,
    # so weak_rate / N is this node's own 1-GPU rate and the strong leg's speed-up over it needs no second run.
    other_leg = None
    if world > 1:
        if args.scaling == "weak":
            lo2, hi2 = parallel.shard_bounds(args.chains, rank, world)
            C2, off2, C2_total = hi2 - lo2, lo2, args.chains
        else:
            C2, off2, C2_total = args.chains, rank * args.chains, world * args.chains
        if C2 > 0:
            q2 = q0[:C2] if C2 <= q0.shape[0] else (0.1 * torch.randn(C2, D, generator=g)).to(dev)
            launch2, st2 = make_launcher(T, thin, record=rec, offset=off2, q_init=q2)
            el2, ms2 = timed_leg(launch2)
            del launch2, st2, q2
        else:                                                  # a rank that owns no chain still meets the barriers
            el2, ms2 = timed_leg(lambda: None)
        tm2 = torch.tensor([el2], dtype=torch.float64, device=cdev)
        dist.all_reduce(tm2, op=dist.ReduceOp.MAX)
        el2 = float(tm2.item())
        rate_main = C_total * T * args.steps * LL / elapsed
        rate2 = C2_total * T * args.steps * LL / el2
        weak_rate, strong_rate = (rate_main, rate2) if args.scaling == "weak" else (rate2, rate_main)
        other_leg = {"chains_per_gpu": C2, "chains_total": C2_total, "leapfrog_steps_per_s": rate2,
                     "ms_per_step": 1e3 * el2 / args.steps, "kernel_ms_this_rank": float(np.mean(ms2)) if ms2 else None,
                     # strong leg over this node's 1-GPU rate (= weak rate / N; projected only in that the 1-GPU job ran
                     # on N GPUs at once rather than alone)
                     "speedup_vs_1gpu_projected": strong_rate / (weak_rate / world)}
    if secondary and inter:
        # round 1's workload for continuity: a trace row EVERY step, 32 steps per launch
        l32, _ = make_launcher(32, 1)
        ms = _time_launches(l32, 20, 3)
        extras["trace_every_step"] = {"transitions_per_launch": 32, "thin": 1, "kernel_ms": ms,
                                      "leapfrog_steps_per_s": C * 32 * LL / (ms * 1e-3)}
        lnt, _ = make_launcher(T, thin, record=False)
        ms = _time_launches(lnt, 5, 2)
        extras["no_trace"] = {"transitions_per_launch": T, "kernel_ms": ms, "leapfrog_steps_per_s": C * T * LL / (ms * 1e-3)}
        if D == 71 and num_ls == 4 and args.lanes in (0, 4):
            # the same instruction mix minus the row stores (~ 20 instructions per step): how much of the launch the rows
            # account for beyond their instructions
            extras["no_trace"]["issue_bound_frac"] = extras["no_trace"]["leapfrog_steps_per_s"] / headline_issue_bound(2.4, mix=headline_mix()[0])[0]
        lst, _ = make_launcher(T, thin, stats=True)
        ms = _time_launches(lst, 5, 2)
        extras["in_kernel_stats"] = {"transitions_per_launch": T, "thin": thin, "stats_batch": 64, "kernel_ms": ms,
                                     "leapfrog_steps_per_s": C * T * LL / (ms * 1e-3),
                                     "overhead_vs_no_trace": ms / extras["no_trace"]["kernel_ms"] - 1.0,
                                     "overhead_vs_headline": ms / kern_ms - 1.0}
        # the plain fused HMC kernel (CP, dual averaging, L leapfrogs, centred trace row every `thin`-th transition)
        lp_, _ = make_launcher(T, thin, plain=True)
        ms = _time_launches(lp_, 5, 2)
        extras["plain_hmc"] = {"kernel": "pk_hmc_kernel<RadonPk<4,17>,CP>", "kernel_ms": ms, "num_leapfrog_steps": L,
                               "leapfrog_steps_per_s": C * T * L / (ms * 1e-3),
                               "fp32_frac": C * T * L * radon_flop_per_leapfrog(J, D) / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS}
        # strong-scaling shards of the 65 536-chain job on ONE GPU (what a rank of a 2 / 4 / 8 GPU job runs)
        shards = {}
        rate1 = C * T * LL / (kern_ms * 1e-3)
        for n_gpu in (2, 4, 8):
            Cs = args.chains // n_gpu
            ls, _ = make_launcher(T, thin, chains=Cs)
            ms = _time_launches(ls, 5, 2)
            shards["%d" % n_gpu] = {"chains_per_gpu": Cs, "kernel_ms": ms,
                                    "leapfrog_steps_per_s_per_gpu": Cs * T * LL / (ms * 1e-3),
                                    "projected_speedup_vs_1gpu": n_gpu * (Cs * T * LL / (ms * 1e-3)) / rate1}
        extras["strong_shard"] = {"total_chains": args.chains, "by_n_gpus": shards,
                                  "note": "per-GPU throughput of a C/N-chain shard measured on this GPU; the projection "
                                          "assumes N identical GPUs and no data-path collective (there is none)"}

    # secondary: the compute-bound model (BASELINE configs[2], german credit, 16 384 chains; SURVEY 8d prices it
    # against the f32 peak: algorithmic flops = 2 products x 2 flop x N x F per gradient).  Two forms of the SAME f32-exact
    # contraction (arp_model_set_option "german_math"): f32 matrix cores, and -- the default since round 5 -- bf16 matrix
    # cores with three-piece operands, which issue 3.7 x the flops on a pipe that is 16 x faster.
    if secondary:
        gspec = models._spec_german()
        Cg, Lg, Tg = 16384, 4, 64   # 64 transitions per launch (the CLI runs up to 4 096): what a launch costs once -- state in and out, the spread between workgroups over few transitions -- is 0.09 ms, 14 % of a 4-transition launch
        Ng, Fg = gspec.raw["X"].shape
        gflop = 4.0 * Ng * Fg
        BF16_PEAK_TFLOPS = 16.0 * FP32_PEAK_TFLOPS          # dense v_mfma_f32_16x16x32_bf16: 1 024 flop per cycle and SIMD
        # matrix-core flops the bf16 form issues per gradient and chain: per 32 observations 2 x 7 (forward) + 14 (backward)
        # instructions of 16 x 16 x 32 x 2 flop, shared by the 16 chains of a wave; tiles of 64 observations
        bf3_flop = ((Ng + 63) // 64) * 2 * 28 * (16 * 16 * 32 * 2) / 16.0
        forms = {}
        for gmath in ("bf16x3", "f32"):
            geng = engine.Engine(gspec, dev)
            geng.set_option("german_math", gmath)
            geng.set_param(0, "NCP")
            rsg = np.random.RandomState(1)
            stg = engine.ChainState(torch.as_tensor((0.1 * rsg.randn(Cg, gspec.D)).astype(np.float32), device=dev))
            epsg = np.full(gspec.D, 0.005, np.float32)
            kwg = dict(seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10 ** 9)
            gms = _time_launches(lambda: geng.hmc_run(stg, epsg, Lg, Tg, **kwg), 3, 1)
            alg_tf = Cg * Tg * Lg * gflop / (gms * 1e-3) / 1e12
            entry = {"kernel_ms": gms, "leapfrog_steps_per_s": Cg * Tg * Lg / (gms * 1e-3),
                     # the model's 4 N F flop per gradient at this speed: a RATE for comparing the two forms, not a utilisation
                     # (the bf16 form does this work on another pipe; its fraction is roofline.frac below)
                     "f32_equivalent_TFLOPs": alg_tf}
            if gmath == "f32":
                entry["kernel"] = "hmc_kernel<GermanLane<4,16>> (v_mfma_f32_16x16x4_f32)"
                entry["roofline"] = {"bound": "mfma", "achieved": alg_tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": alg_tf / FP32_PEAK_TFLOPS, "algorithmic_flop_per_gradient": gflop}
            else:
                ex_tf = Cg * Tg * Lg * bf3_flop / (gms * 1e-3) / 1e12
                entry["kernel"] = "hmc_kernel<GermanLane<4,16,4,false,true>> (v_mfma_f32_16x16x32_bf16, three-piece operands)"
                entry["roofline"] = {"bound": "mfma", "achieved": ex_tf, "peak": BF16_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": ex_tf / BF16_PEAK_TFLOPS, "issued_flop_per_gradient": bf3_flop,
                                     "algorithmic_flop_per_gradient": gflop,
                                     "note": "issued bf16 matrix-core flops against the dense bf16 peak; the wave is bound by "
                                             "instruction issue (one wave per SIMD: every vector, LDS and LDS-DMA instruction "
                                             "costs it 4 - 100 cycles), not by the matrix pipe -- DESIGN.md section 3"}
            forms[gmath] = entry
            del geng, stg
        extras["german_credit"] = dict(forms["bf16x3"], chains=Cg, num_leapfrog_steps=Lg, transitions_per_launch=Tg,
                                       default_math="bf16x3", forms=forms,
                                       speedup_over_f32_matrix_cores=forms["f32"]["kernel_ms"] / forms["bf16x3"]["kernel_ms"])
        # election (BASELINE configs[4]: 131 072 chains): SURVEY 8d ~4.5 kflop per gradient (161 cells x ~25 incl. one exp
        # and one log1p each + 51 x 10) + 4 D for the leapfrog update
        espec = models._spec_election()
        eeng = engine.Engine(espec, dev)
        Ce, Le, Te = 131072, 4, 1024   # launch length as the CLI's (rounds 1 - 3 timed 128-transition launches: 1.77e10 against 1.98e10)
        rse = np.random.RandomState(2)
        eflop = 4500.0 + 4.0 * espec.D
        el = {}
        for name, rp in (("NCP", "NCP"), ("tied_cVIP_b1", None)):
            if rp is None:
                a = np.full(espec.D, 0.5, np.float32); b = np.ones(espec.D, np.float32)
                eeng.set_param(0, (a, b))
            else:
                eeng.set_param(0, rp)
            ste = engine.ChainState(torch.as_tensor((0.05 * rse.randn(Ce, espec.D)).astype(np.float32), device=dev))
            epse = np.full(espec.D, 0.02, np.float32)
            kwe = dict(seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10 ** 9)
            ems_short = _time_launches(lambda: eeng.hmc_run(ste, epse, Le, 128, **kwe), 3, 1)     # rounds 1 - 3's launch length
            ems = _time_launches(lambda: eeng.hmc_run(ste, epse, Le, Te, **kwe), 3, 1)
            rate = Ce * Te * Le / (ems * 1e-3)
            ib, ib_cyc = election_issue_bound(name, Le)
            el[name] = {"kernel_ms": ems, "leapfrog_steps_per_s": rate, "transitions_per_launch": Te,
                        "at_round3_launch_length": {"transitions_per_launch": 128, "kernel_ms": ems_short,
                                                    "leapfrog_steps_per_s": Ce * 128 * Le / (ems_short * 1e-3)},
                        "roofline": {"bound": "valu", "achieved": Ce * Te * Le * eflop / (ems * 1e-3) / 1e12,
                                     "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                                     "frac": Ce * Te * Le * eflop / (ems * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                     "algorithmic_flop_per_leapfrog": eflop},
                        # the ceiling the transcendentals set: the same kernel's instruction mix issued back to back
                        "issue_bound": {"leapfrog_steps_per_s": ib, "frac": rate / ib, "clock_ghz": 2.4,
                                        "frac_note": "time based (costs are times measured on this chip, quoted at 2.4 GHz): the "
                                                     "figure to quote",
                                        "issue_cycles_per_wave_transition": ib_cyc, "mix_per_wave": ELECTION_MIX[name],
                                        "cycles_per_instruction": ISSUE_CYCLES,
                                        "note": "per state and gradient: 1 v_exp + 4 v_rcp (+ 8 v_log per state pair in the "
                                                "closing pass) at a quarter of the FMA issue rate; counts from the ISA"}}
            del ste
        extras["election"] = {"kernel": "pk_hmc_kernel<ElectionPk<4,13>>", "chains": Ce, "num_leapfrog_steps": Le, "forms": el}
        del eeng

    # BASELINE configs[1] (radon MN, CP, 4 096 chains, L = 4) and the SURVEY 8f-3 models on the same fused HMC kernel
    # family, each priced on its own algorithmic flop count (SURVEY 8(d) style: per logp+grad, + 4 D for the leapfrog update)
    if secondary:
        others = {}

        def time_model(tag, mspec, reparam, Cm, Lm, Tm, step, flop_lf, note, T_long=1024):
            """One entry: the fused HMC kernel at `T_long` transitions per launch (what the CLI's launches look like: up to
            4 096 per launch), and -- for continuity with rounds 1 - 3, which timed these models at 16 - 256 transitions per
            launch, i.e. inside the clock ramp and with the state load / store a sixth of the launch -- at `Tm` as well
            (tools/launch_length_sweep.py: radon_stddvs 1.09e10 at 16 transitions per launch, 1.43e10 at 1 024)."""
            try:
                me = engine.Engine(mspec, dev)
                me.set_param(0, reparam)
                res = {}
                for label, Tn in (("short", Tm), ("long", T_long)):
                    rsm = np.random.RandomState(3)
                    stm = engine.ChainState(torch.as_tensor((0.1 * rsm.randn(Cm, mspec.D)).astype(np.float32), device=dev))
                    epm = np.full(mspec.D, step, np.float32)
                    kwm = dict(seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10 ** 9)
                    msm = _time_launches(lambda: me.hmc_run(stm, epm, Lm, Tn, **kwm), 3, 1)
                    res[label] = (Tn, msm, Cm * Tn * Lm / (msm * 1e-3), float(stm.accept_count.float().mean().item() / stm.step))
                    del stm
                Tn, msm, rate, accr = res["long"]
                others[tag] = {"chains": Cm, "num_leapfrog_steps": Lm, "transitions_per_launch": Tn, "kernel_ms": msm,
                               "leapfrog_steps_per_s": rate, "accept_rate": accr,
                               "roofline": {"bound": "valu", "achieved": rate * flop_lf / 1e12, "peak": FP32_PEAK_TFLOPS,
                                            "unit": "TFLOP/s", "frac": rate * flop_lf / 1e12 / FP32_PEAK_TFLOPS,
                                            "algorithmic_flop_per_leapfrog": flop_lf},
                               "at_round3_launch_length": {"transitions_per_launch": res["short"][0], "kernel_ms": res["short"][1],
                                                           "leapfrog_steps_per_s": res["short"][2],
                                                           "frac": res["short"][2] * flop_lf / 1e12 / FP32_PEAK_TFLOPS},
                               "note": note}
                del me
            except Exception as e:
                others[tag] = {"error": repr(e)}

        mn = models._spec_radon("MN")
        Jm = mn.D - 3
        time_model("config2_radon_MN_CP_4096", mn, "CP", 4096, 4, 256, 0.05, radon_flop_per_leapfrog(Jm, mn.D),
                   "BASELINE configs[1]: 4 096 chains fill a quarter of the lanes a 256-CU device wants (16 lanes per chain: one wave per SIMD)")
        time_model("radon_MN_CP_65536", mn, "CP", 65536, 4, 64, 0.05, radon_flop_per_leapfrog(Jm, mn.D),
                   "4 lanes per chain, 22 counties per lane (the packed kernel spills 3 - 21 values, about 1 % of its instructions)")
        sd = models._spec_radon_stddvs("MN")
        time_model("radon_stddvs_MN_NCP_65536", sd, "NCP", 65536, 8, 16, 0.01, 45.0 * Jm + 20.0 + 4.0 * sd.D,
                   "per county 6 sufficient statistics and one exp: ~45 flop per county and gradient; 8 lanes per chain, county tables in LDS, two waves per SIMD")
        el = models._spec_electric()
        time_model("electric_NCP_65536", el, "NCP", 65536, 8, 16, 0.01, 97 * 60.0 + 12 * 10.0 + 4.0 * el.D,
                   "97 groups x ~60 flop (two cells, one-hot grade look-ups as 12 FMAs; exp(-2 s) once per grade) + 12 grade scalars; 8 lanes per chain, tables in LDS")
        ts = models._spec_time_series()
        time_model("time_series_NCP_65536", ts, "NCP", 65536, 8, 16, 0.05, 60 * 80.0 + 4.0 * ts.D,
                   "60 time steps x ~80 flop of the general form (centring recurrence and its adjoint as block scans); the run takes the compile-time non-centred form (no per-step exp, unit block maps), 4 lanes per chain")
        # The general per-element (a, b) form -- what `--tied_pparams=False` cVIP / dVIP runs execute
        # (program_transformations.py:513-533, 555-600) -- on the two BASELINE models whose CP / NCP / b = 1 forms have
        # packed kernels: these runs take the generic float-array kernels (kernels.h: hmc_kernel<Lane, kModeVIP>)
        es2 = models._spec_election()
        rs_ab = np.random.RandomState(7)
        ab_el = (rs_ab.uniform(0.2, 0.8, es2.D).astype(np.float32), rs_ab.uniform(0.2, 0.8, es2.D).astype(np.float32))
        time_model("election_untied_general_ab_131072", es2, ab_el, 131072, 4, 128, 0.02, 4500.0 + 4.0 * es2.D,
                   "untied cVIP: a and b free per element (exp(b log sigma) per state and pass); the packed chain layer's "
                   "general form, pk_hmc_kernel<ElectionPk<4,13>, kModeVIP> (arp_api.hip routes kModeVIP to hmc_vip_pk)")
        pa = models._spec_radon("PA")
        ab_pa = (rs_ab.uniform(0.2, 0.8, pa.D).astype(np.float32), rs_ab.uniform(0.2, 0.8, pa.D).astype(np.float32))
        time_model("radon_PA_general_ab_65536", pa, ab_pa, 65536, 8, 256, 0.02, radon_flop_per_leapfrog(pa.D - 3, pa.D),
                   "cVIP / dVIP with a free per element (m has unit scale: b is inert); pk_hmc_kernel<RadonPk<4,17>, kModeVIP> "
                   "against the packed CP kernel's plain_hmc entry above")
        extras["other_models"] = others

    # The mean-field VI kernel (find_best_learning_rate, inference.py:26-154): all learning rates x all optimisation steps in
    # ONE launch; every learning rate's 256 draws are spread over a group of workgroups (G sample groups x R row parts of
    # German credit's observations) that exchange their partial gradient sums twice per step inside the launch (8-byte
    # {epoch, value} granules, agent-scope atomics, fixed summation order: DESIGN.md section 3).  Every step depends on the
    # last, so the kernel is LATENCY bound: the figure to read is us_per_optimisation_step; the flop fraction is against
    # the whole chip and says how little of it a 256-draw gradient can use.
    if secondary:
        try:
            vflags_lrs = [0.02, 0.05, 0.1, 0.2, 0.4]
            n_opt, n_mc = 3000, 256
            vi = {}
            e88 = models._spec_election()
            for tag, vspec, flop, r04_ms in (("radon_PA_CP", spec, radon_flop_per_leapfrog(J, D) - 4.0 * D, 27.9),
                                             ("german_NCP", models._spec_german(), 4.0 * 1000 * 62, 645.0),
                                             ("election_NCP", e88, 4500.0, 100.0)):
                veng = engine.Engine(vspec, dev)
                veng.set_param(0, "CP" if tag.startswith("radon") else "NCP")
                times = []
                for rep in range(3):
                    loc = torch.zeros(len(vflags_lrs), vspec.D, device=dev)
                    rho = torch.full((len(vflags_lrs), vspec.D), -2.0, device=dev)
                    torch.cuda.synchronize()
                    a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a_.record(); veng.vi_run(vflags_lrs, loc, rho, n_opt, n_mc, seed=1); b_.record(); torch.cuda.synchronize()
                    times.append(a_.elapsed_time(b_))
                ms = min(times)
                geo = veng.vi_geometry()
                wgs = len(vflags_lrs) * geo["sample_groups"] * geo["row_parts"]
                per_cu = max(1, geo["workgroups_per_cu"])
                grads = len(vflags_lrs) * n_opt * n_mc
                vi[tag] = {"kernel_ms": ms, "learning_rates": len(vflags_lrs), "steps": n_opt, "mc_samples": n_mc,
                           "gradients_per_s": grads / (ms * 1e-3), "us_per_optimisation_step": 1e3 * ms / n_opt,
                           "round4_one_workgroup_per_learning_rate_ms": r04_ms,
                           "geometry": geo,
                           "roofline": {"bound": "latency", "achieved": grads * flop / (ms * 1e-3) / 1e12, "peak": FP32_PEAK_TFLOPS,
                                        "unit": "TFLOP/s", "frac": grads * flop / (ms * 1e-3) / 1e12 / FP32_PEAK_TFLOPS,
                                        "workgroups": wgs,
                                        # the dispatcher fills empty CUs first: `wgs` workgroups of which a CU holds `per_cu`
                                        "cus_used_of_256": min(256, wgs) if wgs <= 256 else min(256, (wgs + per_cu - 1) // per_cu),
                                        "note": "G x R workgroups per learning rate; each optimisation step is 256 gradients, two "
                                                "in-launch hand-offs of the partial sums and an Adam update the next step depends "
                                                "on; frac is against the whole chip"}}
                del veng
            extras["vi_kernel"] = vi
        except Exception as e:
            extras["vi_kernel"] = {"error": repr(e)}

    # arp_ess on its own: the [S, C, D] trace of a sampling run at the headline size (1 000 recorded samples = 18.6 GB),
    # priced against HBM: the kernel is a strided stream of the trace (algorithmic bytes = one read of it)
    if secondary and inter and not args.no_ess:
        try:
            S_e = 1000
            tr_e = torch.empty(S_e, C, D, dtype=torch.float32, device=dev)
            st_e = engine.ChainState(q0)
            total_e, done_e = 1 + 1000 + 2 * (S_e - 1), 0
            while done_e < total_e:
                n_e = min(4096, total_e - done_e)
                eng.interleaved_run(st_e, eps_i, eps_i, num_ls, num_ls, n_e, seed=11, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=600,
                                    n_burnin=1000, thin=2, trace=tr_e, trace_centered=False, lanes=args.lanes)
                done_e += n_e
            ess_t = []
            ess_v = util.effective_sample_size(tr_e)
            torch.cuda.synchronize()
            for _ in range(5):
                a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a_.record(); ess_v = util.effective_sample_size(tr_e); b_.record(); torch.cuda.synchronize()
                ess_t.append(a_.elapsed_time(b_))
            ems_ = float(np.median(ess_t))
            nbytes = 4.0 * S_e * C * D
            # measured HBM bytes of a profiled pass of the same call (tools/profile_bench.sh: FETCH_SIZE x 2 + WRITE_SIZE,
            # separate passes), only when that pass ran this very configuration
            import glob as _glob
            ess_prof = sorted(_glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_ess_kernel.json")))
            ess_traffic, ess_src = None, None
            if ess_prof:
                try:
                    ep = json.load(open(ess_prof[-1]))
                    if abs(ep["grid"] - C * D) < 256 and ep["hbm_bytes_per_dispatch_x2_reads"]:
                        # the first dispatches of that run are this trace (bench's own); the last two the flow's candidates
                        own = ep["hbm_bytes_per_dispatch_x2_reads"][:-2] or ep["hbm_bytes_per_dispatch_x2_reads"]
                        ess_traffic, ess_src = float(np.median(own)), "profiles/" + os.path.basename(ess_prof[-1])
                except Exception:
                    pass
            extras["ess_kernel"] = {
                "kernel": "ess_kernel", "samples": S_e, "series": C * D, "kernel_ms": ems_, "kernel_ms_min": float(min(ess_t)),
                "kernel_ms_max": float(max(ess_t)), "mean_min_ess_per_chain": float(ess_v.min(dim=1).values.mean().item()),
                "roofline": {"bound": "hbm", "achieved": nbytes / (ems_ * 1e-3) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                             "frac": nbytes / (ems_ * 1e-3) / 1e9 / HBM_PEAK_GBPS, "traffic": ess_traffic,
                             "traffic_source": ess_src,
                             "traffic_over_algorithmic": (ess_traffic / nbytes) if ess_traffic else None,
                             "algorithmic_bytes": nbytes,
                             "note": "algorithmic bytes = one read of the trace; series still positive at lag 16 are read "
                                     "once more by their wave (64-byte sectors), so the HBM traffic is higher"}}
            del tr_e, st_e, ess_v
        except Exception as e:
            extras["ess_kernel"] = {"error": repr(e)}

    # ESS/sec (second half of the BASELINE metric) from the reference flow at the headline size
    ess_info = None
    if secondary and inter and not args.no_ess:
        try:
            ess_info = reference_flow_ess(args.dataset, args.chains, local_rank)
        except Exception as e:   # the throughput figure must not be lost to a failure of the secondary run
            ess_info = {"error": repr(e)}

    if rank == 0:
        value = C_total * T * args.steps * LL / elapsed
        flop_lf = radon_flop_per_leapfrog(J, D)
        achieved_tf = C * T * LL * flop_lf / (kern_ms * 1e-3) / 1e12
        # SURVEY 8(d) byte model: per interleaved step one transition with a trace row every `thin` steps + one without
        n_rows = (T + thin - 1) // thin if rec else 0
        if inter:
            alg_bytes = C * (T * 2 * algorithmic_bytes_per_transition(D, False) + n_rows * 4 * D)
        else:
            alg_bytes = C * (T * algorithmic_bytes_per_transition(D, False) + n_rows * 4 * D)
        prof = load_profile({"chains": C, "transitions": T, "thin": thin if rec else 0, "method": args.method,
                             "leapfrog": L, "dataset": args.dataset, "lanes": args.lanes})
        traffic = prof["hbm_bytes_per_launch"] if prof and prof.get("config_matches_this_run") else None
        roof = {"bound": "valu", "achieved": achieved_tf, "peak": FP32_PEAK_TFLOPS, "unit": "TFLOP/s",
                "frac": achieved_tf / FP32_PEAK_TFLOPS, "traffic": traffic,
                "kernel": "radon_interleaved_kernel<RadonPk<4,17>>" if inter else "pk_hmc_kernel<RadonPk<4,17>,CP>",
                "kernel_ms": kern_ms, "kernel_ms_min": float(np.min(per_launch_ms)),
                "kernel_ms_median": float(np.median(per_launch_ms)), "kernel_ms_max": float(np.max(per_launch_ms)),
                "algorithmic_flop_per_leapfrog": flop_lf,
                "note": "FP32 vector issue binds this kernel (state in registers; the ONE launch of a step hands its chains from "
                        "workgroup to workgroup up to eight times -- relay segments, DESIGN.md section 3 -- bit for bit the "
                        "unsegmented run); frac = SURVEY 8(d) algorithmic flops / HIP-event kernel time / 157.3 TFLOP/s",
                "hbm": {"algorithmic_bytes_per_launch": alg_bytes,
                        "algorithmic_GBps": alg_bytes / (kern_ms * 1e-3) / 1e9,
                        "algorithmic_frac_of_8TBps": alg_bytes / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS,
                        "measured_bytes_per_launch": traffic,
                        "measured_frac_of_8TBps": (traffic / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                        "note": "the byte model charges every transition a state load/store the fused kernel performs "
                                "once per launch segment (the launch hands its chains from workgroup to workgroup up to eight "
                                "times: DESIGN.md section 3, relay segments); the measured figure is what crosses HBM"}}
        roof["relay_segments"] = relay_segments
        roof["clock_ghz_live"] = clock_live
        roof["clock_live_note"] = ("shader clock held under 10 ms of packed FMAs on every SIMD right after the timed region "
                                   "(arp_clock_probe); the 157.3 TFLOP/s peak assumes 2.4 GHz: frac x 2.4 / clock_ghz_live is the "
                                   "fraction of what THIS box can issue.  An upper estimate of the clock under the headline kernel "
                                   "itself, which holds ~ 7 % less with its LDS traffic and trace stores "
                                   "(profiles/r04_box_to_box.txt)")
        roof["frac_of_peak_at_live_clock"] = (achieved_tf / (FP32_PEAK_TFLOPS * clock_live / 2.4)) if clock_live else None
        if prof:
            roof["profile"] = prof
            # shader clock held under this kernel's load (GRBM_GUI_ACTIVE / 8 XCDs / kernel time of the profiled pass);
            # the 157.3 TFLOP/s peak assumes 2.4 GHz
            clk = prof.get("derived", {}).get("clock_ghz_estimate")
            roof["clock_ghz_estimate"] = clk
            roof["clock_source"] = prof["source"] if clk else None
        if inter and D == 71 and num_ls == 4 and args.lanes in (0, 4):
            # how close the launch comes to issuing its own instruction mix back to back (<= 1 by construction)
            rate1 = C * T * LL / (kern_ms * 1e-3)
            mix, mix_src = headline_mix()
            ib24, cyc = headline_issue_bound(2.4, mix=mix)
            roof["issue_bound"] = {"frac": rate1 / ib24, "leapfrog_steps_per_s": ib24,
                                   "cycles_per_wave_step_at_2.4GHz": cyc, "mix_per_wave_step": mix,
                                   "cost_cycles_at_2.4GHz": HEADLINE_COST,
                                   "source": "%s (ISA ledger), tools/pk_vs_fma_2waves.hip (costs)" % mix_src,
                                   "note": "time based: the costs are times measured on this chip at two resident waves per SIMD "
                                           "(quoted as cycles at 2.4 GHz), so frac = priced time / measured time whatever clock "
                                           "the chip holds; <= 1 by construction"}
        backend = dist.get_backend() if dist is not None else None
        workload = (("radon --dataset=%s --method=i --inference=HMC (interleaved CP/NCP), %d chains%s, num_ls=%d+%d leapfrogs, "
                     "%d interleaved steps per launch, simple step-size adaptation, CP trace row every %s step" % (
                         args.dataset, args.chains, "/GPU" if args.scaling == "weak" else " total", num_ls, num_ls, T,
                         {1: "", 2: "2nd"}.get(thin, "%d-th" % thin))) if inter else
                    ("radon --dataset=%s --method=CP --inference=HMC, %d chains%s, L=%d, %d transitions per launch, "
                     "dual-averaging adaptation, centred trace row every %d transition(s)" % (
                         args.dataset, args.chains, "/GPU" if args.scaling == "weak" else " total", L, T, thin)))
        cpu, cpu_shaped = None, None
        if world == 1 and not args.no_cpu_baseline:
            try:
                cpu = cpu_baseline(spec, num_ls if inter else L, 8192, 64, eps_i if inter else eps0, 8, inter)
            except Exception as e:  # the oracle is a checker; its absence must not hide the GPU number
                cpu = {"value": None, "error": repr(e)}
            if not args.headline_only:
                try:
                    cpu_shaped = cpu_baseline_reference_shaped(spec, L)
                except Exception as e:
                    cpu_shaped = {"value": None, "error": repr(e)}
        # ---- the ONE stdout line: the contract's keys, `roofline`, `cpu_baseline`, nothing long (< 4 KB) ----
        hbm = roof["hbm"]
        line = {
            "metric": "leapfrog-steps/sec (all chains) + ESS/sec, radon(%s) %d chains%s" % (
                args.dataset, args.chains, " per GPU" if args.scaling == "weak" else " in total"),
            "value": value, "unit": "leapfrog-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "ess_per_sec": ess_info.get("ess_per_sec") if ess_info else None,
            "config": {"workload": workload, "chains_per_gpu": C, "chains_total": C_total, "num_leapfrog_steps": LL,
                       "transitions_per_step": T, "trace_thin": thin if rec else 0, "D": D, "lanes_per_chain": args.lanes,
                       "parallelism": "chains sharded, %d rank(s), %s scaling" % (world, args.scaling)},
            "roofline": {"bound": roof["bound"], "achieved": roof["achieved"], "peak": roof["peak"], "unit": roof["unit"],
                         "frac": roof["frac"], "traffic": roof["traffic"], "kernel": roof["kernel"], "kernel_ms": kern_ms,
                         "algorithmic_flop_per_leapfrog": flop_lf,
                         "hbm": {k: hbm[k] for k in ("algorithmic_bytes_per_launch", "algorithmic_frac_of_8TBps",
                                                     "measured_bytes_per_launch", "measured_frac_of_8TBps")},
                         "clock_ghz_live": clock_live, "frac_of_peak_at_live_clock": roof["frac_of_peak_at_live_clock"],
                         "issue_bound_frac": roof.get("issue_bound", {}).get("frac"), "relay_segments": relay_segments},
            "accept_rate": accept_rate, "ranks": world,
            # the transport the end-of-run exchange actually ran on: "nccl" is RCCL; "gloo" only in the tests that put
            # two ranks on one GPU; None for a single rank (no process group)
            "dist_backend": backend, "stats_allgather_s": t_coll,
            "kernel_ms_per_rank": [round(v, 4) for v in rank_ms], "rank_imbalance_max_over_min": max(rank_ms) / min(rank_ms),
        }
        if backend == "nccl":
            line["rccl_ranks"] = world           # only when the collectives really ran over RCCL
        if other_leg is not None:
            line["scaling_strong" if args.scaling == "weak" else "scaling_weak"] = other_leg
        if cpu is not None:
            line["cpu_baseline"] = {k: cpu.get(k) for k in ("value", "unit", "cores", "kind", "sample", "error") if k in cpu}
        # ---- everything else: bench_extras.json ----
        full = dict(line)
        full["roofline"] = roof
        full["ess"] = ess_info
        full.update(extras)
        if cpu is not None:
            full["cpu_baseline"] = cpu
        if cpu_shaped is not None:
            full["cpu_baseline_reference_shaped"] = cpu_shaped
        try:
            with open(args.extras, "w") as f:
                json.dump(full, f, indent=1)
            line["extras"] = os.path.relpath(args.extras, ROOT) if args.extras.startswith(ROOT) else args.extras
        except OSError as e:
            line["extras"] = None
            line["extras_error"] = repr(e)
        text = json.dumps(line)
        assert len(text) < 4096 and "\n" not in text, len(text)
        sys.stderr.flush()
        print(text, file=json_out, flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
