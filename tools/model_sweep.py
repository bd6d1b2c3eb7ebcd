"""Throughput of the fused HMC kernel for every model / lanes-per-chain (one line each)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
from autoreparam_amd import models, engine, _lib

def run(name, spec, C, L, lanes, reparam="CP", T=16, eps=0.01):
    eng = engine.Engine(spec, "cuda:0")
    eng.set_param(0, reparam)
    rs = np.random.RandomState(0)
    q0 = torch.as_tensor((0.1 * rs.randn(C, spec.D)).astype(np.float32), device="cuda:0")
    eps0 = np.full(spec.D, eps, np.float32)
    st = engine.ChainState(q0)
    try:
        eng.hmc_run(st, eps0, L, 4, seed=1, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10**9, lanes=lanes)
    except RuntimeError as e:
        print(name, "lanes", lanes, "skip:", e); return
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(3):
        eng.hmc_run(st, eps0, L, T, seed=1, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10**9, lanes=lanes)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 3
    BT = 4 * (4 * spec.D + 8) + 1
    print("%-10s %-4s C=%6d L=%d lanes=%2d  %8.3f ms/%d tr  %.3e leapfrog/s  alg(no trace) %.2f TB/s  acc=%.2f" % (
        name, reparam, C, L, lanes, ms, T, C * T * L / (ms * 1e-3), C * T * BT / (ms * 1e-3) / 1e12,
        st.accept_count.float().mean().item() / st.step), flush=True)

def main():
    only = sys.argv[1:]
    if only:   # e.g. `model_sweep.py electric radon_stddvs`
        if "electric" in only:
            for lanes in (8, 16):
                run("electric", models._spec_electric(), 65536, 8, lanes, "NCP", eps=0.01)
            run("electric", models._spec_electric(), 16384, 8, 16, "NCP", eps=0.01)
        if "election" in only:
            for rep in ("NCP", "CP", "VIP"):
                for lanes in (4, 8, 16):
                    run("election", models._spec_election(), 131072, 8, lanes, rep if rep != "VIP" else {k + "_a": 0.5 for k in ("mua", "log_sigma_a", "a", "b1", "b2")}, eps=0.005)
        if "german" in only:
            for lanes in (4, 8, 16):
                run("german", models._spec_german(), 16384, 4, lanes, "NCP", T=4, eps=0.005)
        if "time_series" in only:
            for lanes in (4, 8, 16):
                run("time_series", models._spec_time_series(), 65536, 8, lanes, "NCP", eps=0.05)
        if "radon_stddvs" in only:
            for lanes in (8, 16):
                run("radon_sd", models._spec_radon_stddvs("MN"), 65536, 8, lanes, "NCP", eps=0.01)
        return
    for lanes in (1, 2, 4, 8):
        run("8schools", models._spec_eight_schools(), 65536, 4, lanes, "NCP", eps=0.1)
    for lanes in (4, 8, 16):
        run("radon_MN", models._spec_radon("MN"), 4096, 4, lanes, eps=0.05)
    for lanes in (4, 8, 16):
        run("radon_MN", models._spec_radon("MN"), 65536, 4, lanes, eps=0.05)
    for lanes in (4, 8, 16):
        run("election", models._spec_election(), 131072, 8, lanes, "NCP", eps=0.005)
    for lanes in (4, 8, 16):
        run("german", models._spec_german(), 16384, 4, lanes, "NCP", T=4, eps=0.005)


if __name__ == "__main__":
    main()
