#!/usr/bin/env python3
"""Headline benchmark: leapfrog-steps/sec over all chains, radon(PA), 65 536 chains
per GPU (BASELINE.json metric; workload = configs[3]: radon --dataset=PA --method=i,
interleaved CP/NCP, 4 + 4 = "8 leapfrog steps" per step, main.py:493), on the fused
HIP kernels.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one launch of the hot path over the whole chain batch: `--transitions`
sampler steps for every chain on the rank.  With --method i (default) a sampler
step is one interleaved step: bootstrap + num_ls leapfrogs in CP coordinates,
to_ncp, bootstrap + num_ls leapfrogs in NCP coordinates, to_cp, two Metropolis
tests, two simple step-size adaptations (counted as 2*num_ls leapfrog steps, the
two bootstrap gradient evaluations are extra work that is not counted).  With
--method CP it is one plain HMC transition of `--leapfrog` steps with dual averaging.  Chains are independent, so ranks shard
them with no data-path collective (weak scaling: the per-GPU batch is fixed); one
RCCL all-gather of the acceptance statistics runs after the timed region.

Prints ONE JSON line (see DESIGN.md "Measurement" for the roofline definition).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402


def algorithmic_bytes_per_transition(D, trace=True):
    # read q, grad, logp, 3 adaptation scalars; write the same + accept byte (+ trace row)
    return 4 * ((5 if trace else 4) * D + 8) + 1


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
    return {"value": n_chains * n_trans * LL / dt, "unit": "leapfrog-steps/s", "cores": cores, "kind": "port",
            "sample": "%d chains x %d %s x %d leapfrogs, float32 C oracle (oracle/oracle.c) with OpenMP over chains "
                      "on %d threads, %.1f s" % (n_chains, n_trans, "interleaved steps" if inter else "transitions",
                                                 LL, cores, dt)}


def cpu_baseline_reference_shaped(spec, L, n_chains=1024, budget_s=6.0):
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
    while time.time() - t0 < budget_s:
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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--chains", type=int, default=65536, help="chains per GPU")
    ap.add_argument("--method", default="i", choices=["i", "CP"])
    ap.add_argument("--leapfrog", type=int, default=8, help="leapfrog steps per sampler step (i: split CP/NCP)")
    ap.add_argument("--transitions", type=int, default=32, help="HMC transitions per launch (= per step)")
    ap.add_argument("--dataset", default="PA")
    ap.add_argument("--lanes", type=int, default=0, help="lanes per chain (0 = library default)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--headline-only", action="store_true",
                    help="skip the secondary figures (plain_hmc, german_credit, ess): profiler runs, so that every "
                         "launch of the headline kernel in the trace is a timed or warm-up step")
    ap.add_argument("--no-trace", action="store_true", help="diagnostic: do not record trace rows")
    ap.add_argument("--stats", action="store_true", help="diagnostic: accumulate the in-kernel streaming statistics "
                                                         "(arp_hmc_io.stats) every step instead of writing trace rows")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run)" % (args.gpus, world))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        try:
            dist.init_process_group("nccl", device_id=dev)   # RCCL; binds the communicator to this rank's GPU
        except TypeError:                                     # older torch: no device_id keyword
            dist.init_process_group("nccl")

    from autoreparam_amd import models, engine, _lib
    spec = models._spec_radon(args.dataset)
    eng = engine.Engine(spec, dev)
    eng.set_param(0, "CP")
    eng.set_param(1, "NCP")
    C, L, T, D = args.chains, args.leapfrog, args.transitions, spec.D
    inter = args.method == "i"
    num_ls = L // 2

    # synthetic chain population: i.i.d. draws keyed by the global chain id
    g = torch.Generator(device="cpu").manual_seed(1234 + rank)
    q0 = (0.1 * torch.randn(C, D, generator=g)).to(dev)
    eps0 = np.full(D, 0.08 / (L / 4.0) ** 2, np.float32)
    eps0[2] = 0.02 / (L / 4.0) ** 2
    st = engine.ChainState(q0)
    S = args.steps * T  # every timed transition appends a trace row
    trace = torch.empty(min(S, 64), C, D, dtype=torch.float32, device=dev)  # ring of rows that gets overwritten

    eps_i = np.full(D, 0.08 / (max(num_ls, 1) / 4.0) ** 2, np.float32)   # interleaved: eps0/(num_ls/4)^2
    eps_i[2] = 0.02 / (max(num_ls, 1) / 4.0) ** 2

    stats = torch.zeros(6, C, D, dtype=torch.float32, device=dev) if args.stats else None
    skw = dict(stats=stats, stats_batch=8, n_samples=1 << 30) if args.stats else {}

    def launch(record, plain=False):
        # trace rows cycle through a bounded buffer so a long bench does not need S*C*D floats
        if args.stats and not plain:
            record = False
        if inter and not plain:
            eng.interleaved_run(st, eps_i, eps_i, num_ls, num_ls, T, seed=7, chain_offset=rank * C,
                                adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10 ** 9, adapt_target=0.75, adapt_rate=0.05,
                                n_burnin=st.step if not args.stats else 0, thin=1, trace=trace[:T] if record else None,
                                trace_centered=False, lanes=args.lanes, **skw)
        else:
            eng.hmc_run(st, eps0, L, T, seed=7, chain_offset=rank * C, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10 ** 9,
                        n_burnin=st.step, thin=1, trace=trace[:T] if record else None, trace_centered=True,
                        lanes=args.lanes)

    rec = not args.no_trace
    for _ in range(args.warmup):
        launch(rec)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        ev[k][0].record()
        launch(rec)
        ev[k][1].record()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))

    # end-of-run statistics exchange (the only collective of the path)
    acc = st.accept_count.float() / st.step
    t_coll = 0.0
    if dist is not None:
        tm = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        elapsed = float(tm.item())
        torch.cuda.synchronize(); tc = time.perf_counter()
        gathered = [torch.empty_like(acc) for _ in range(world)]
        dist.all_gather(gathered, acc)
        acc = torch.cat(gathered)
        torch.cuda.synchronize(); t_coll = time.perf_counter() - tc
    assert torch.isfinite(st.q).all(), "non-finite chain state"
    accept_rate = float(acc.mean().item())

    # secondary figure, same run: the plain fused HMC kernel (CP, dual averaging, L leapfrogs)
    plain = None
    if inter and world == 1 and not args.headline_only:
        st2, st = st, engine.ChainState(q0)
        for _ in range(2):
            launch(True, plain=True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            launch(True, plain=True)
        e1.record(); torch.cuda.synchronize()
        pms = e0.elapsed_time(e1) / 5
        plain = {"kernel": "hmc_kernel<RadonLane,CP>", "kernel_ms": pms,
                 "leapfrog_steps_per_s": C * T * L / (pms * 1e-3),
                 "achieved_GBps": C * T * algorithmic_bytes_per_transition(D) / (pms * 1e-3) / 1e9}
        st = st2

    # secondary figure: the one compute-bound model (BASELINE configs[2], german credit, 16 384 chains).
    # SURVEY.md 8d prices it against the f32 peak: algorithmic flops = 2 products x 2 flop x N x F per gradient.
    german = None
    if world == 1 and not args.headline_only:
        gspec = models._spec_german()
        geng = engine.Engine(gspec, dev)
        geng.set_param(0, "NCP")
        Cg, Lg, Tg = 16384, 4, 4
        rsg = np.random.RandomState(1)
        stg = engine.ChainState(torch.as_tensor((0.1 * rsg.randn(Cg, gspec.D)).astype(np.float32), device=dev))
        epsg = np.full(gspec.D, 0.005, np.float32)
        kwg = dict(seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10 ** 9)
        geng.hmc_run(stg, epsg, Lg, Tg, **kwg)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            geng.hmc_run(stg, epsg, Lg, Tg, **kwg)
        e1.record(); torch.cuda.synchronize()
        gms = e0.elapsed_time(e1) / 3
        Ng, Fg = gspec.raw["X"].shape
        gflop = 4.0 * Ng * Fg
        german = {"kernel": "hmc_kernel<GermanLane<4,16>> (v_mfma_f32_16x16x4_f32)", "chains": Cg, "num_leapfrog_steps": Lg,
                  "kernel_ms": gms, "leapfrog_steps_per_s": Cg * Tg * Lg / (gms * 1e-3),
                  "roofline": {"bound": "mfma", "achieved": Cg * Tg * Lg * gflop / (gms * 1e-3) / 1e12, "peak": 157.3,
                               "unit": "TFLOP/s", "frac": Cg * Tg * Lg * gflop / (gms * 1e-3) / 1e12 / 157.3,
                               "algorithmic_flop_per_gradient": gflop}}

    # ESS/sec (second half of the BASELINE metric): a separate short run with the same kernel,
    # S recorded samples at the reference's thinning, ESS by FFT on the device trace
    ess_info = None
    if world == 1 and not args.headline_only:
        from autoreparam_amd import util
        S_ess, burn = 200, 200
        Ce = C   # the headline chain count: the trace is S_ess x C x D x 4 B = 3.7 GB
        st3 = engine.ChainState(q0[:Ce])
        tr3 = torch.empty(S_ess, Ce, D, dtype=torch.float32, device=dev)
        tot = 1 + burn + 2 * (S_ess - 1)
        util.effective_sample_size(tr3[:, :64])          # first-call (module load) cost is not part of the figure
        torch.cuda.synchronize(); te = time.perf_counter()
        if inter:
            eng.interleaved_run(st3, eps_i, eps_i, num_ls, num_ls, tot, seed=11, adapt_kind=_lib.ADAPT_SIMPLE,
                                n_adapt=burn, n_burnin=burn, thin=2, trace=tr3, trace_centered=False, lanes=args.lanes)
        else:
            eng.hmc_run(st3, eps0, L, tot, seed=11, adapt_kind=_lib.ADAPT_DUAL, n_adapt=burn, n_burnin=burn, thin=2,
                        trace=tr3, trace_centered=True, lanes=args.lanes)
        torch.cuda.synchronize(); t_samp = time.perf_counter() - te
        ess = util.effective_sample_size(tr3)
        torch.cuda.synchronize(); t_all = time.perf_counter() - te
        min_ess = ess.nan_to_num().min(dim=1).values
        ess_info = {"chains": Ce, "samples": S_ess, "burnin": burn, "mean_min_ess_per_chain": float(min_ess.mean()),
                    "sampling_s": t_samp, "sampling_plus_ess_s": t_all,
                    "min_ess_per_sec_all_chains": float(min_ess.sum()) / t_all}

    if rank == 0:
        LL = 2 * num_ls if inter else L
        value = world * C * T * args.steps * LL / elapsed
        # interleaved step = one transition with a trace row + one without (per transition)
        BT = (algorithmic_bytes_per_transition(D, True) + algorithmic_bytes_per_transition(D, False)) if inter \
            else algorithmic_bytes_per_transition(D)
        achieved = C * T * BT / (kern_ms * 1e-3) / 1e9
        traffic = None
        tp = os.path.join(ROOT, "profiles", "hbm_traffic.json")
        if os.path.exists(tp):
            try:
                traffic = json.load(open(tp)).get("bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "leapfrog-steps/sec (all chains), radon(PA) 65536 chains per GPU",
            "value": value, "unit": "leapfrog-steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": ("radon --dataset=%s --method=i --inference=HMC (interleaved CP/NCP), %d chains/GPU, "
                                    "num_ls=%d+%d leapfrog steps, %d interleaved steps per launch, simple step-size "
                                    "adaptation on both kernels, CP trace row every step" % (args.dataset, C, num_ls,
                                                                                            num_ls, T)) if inter else
                                   ("radon --dataset=%s --method=CP --inference=HMC, %d chains/GPU, L=%d, "
                                    "%d transitions per launch, dual-averaging adaptation, centred trace row "
                                    "every transition" % (args.dataset, C, L, T)),
                       "chains_per_gpu": C, "num_leapfrog_steps": LL, "transitions_per_step": T, "D": D,
                       "lanes_per_chain": args.lanes, "parallelism": "chains sharded, %d rank(s)" % world},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": 8000.0, "unit": "GB/s",
                         "frac": achieved / 8000.0, "traffic": traffic,
                         "kernel": "interleaved_kernel<RadonLane,CP,NCP>" if inter else "hmc_kernel<RadonLane,CP>",
                         "kernel_ms": kern_ms, "algorithmic_bytes_per_step_per_chain": BT,
                         "algorithmic_bytes_per_launch": C * T * BT},
            "accept_rate": accept_rate, "stats_allgather_s": t_coll, "plain_hmc": plain, "german_credit": german,
            "ess": ess_info,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(spec, num_ls if inter else L, 8192, 64, eps_i if inter else eps0, 8,
                                                   inter)
            except Exception as e:  # the oracle is a checker; its absence must not hide the GPU number
                out["cpu_baseline"] = {"value": None, "error": repr(e)}
            try:
                out["cpu_baseline_reference_shaped"] = cpu_baseline_reference_shaped(spec, L)
            except Exception as e:
                out["cpu_baseline_reference_shaped"] = {"value": None, "error": repr(e)}
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
