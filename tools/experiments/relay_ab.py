"""Relay segments on / off (ARP_DEBUG=1 ARP_SEGMENTS=1 forces one workgroup per chain block) for the chain kernels other than the
headline's: ms per launch of 1 024 transitions (German credit: 512)."""
import os, sys, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np, torch
    import helpers
    from autoreparam_amd import engine, _lib
    def timeit(f, n=3):
        f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n
    out = {}
    for mname, kind, C, L, T, sampler in (("german", "NCP", 16384, 4, 512, "hmc"), ("election", "NCP", 131072, 4, 1024, "hmc"),
                                          ("election", "B1", 131072, 4, 1024, "hmc"), ("radon_MN", "CP", 65536, 4, 1024, "hmc"),
                                          ("radon_PA", "CP", 65536, 8, 1024, "hmc"), ("election", None, 131072, 4, 512, "i"),
                                          ("electric", "NCP", 65536, 8, 512, "hmc"), ("radon_sd_MN", "NCP", 65536, 8, 512, "hmc"),
                                          ("time_series", "NCP", 65536, 8, 512, "hmc")):
        sp = helpers.spec(mname); eng = engine.Engine(sp, "cuda:0")
        if sampler == "hmc": eng.set_param(0, helpers.params(sp, kind, seed=1))
        else: eng.set_param(0, "CP"); eng.set_param(1, "NCP")
        st = engine.ChainState(torch.as_tensor(helpers.states(sp, C, seed=1, scale=0.05), device="cuda:0"))
        e = np.full(sp.D, 2e-3 if mname != "time_series" else 1e-4, np.float32)
        if sampler == "hmc":
            ms = timeit(lambda: eng.hmc_run(st, e, L, T, seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=10**9))
        else:
            ms = timeit(lambda: eng.interleaved_run(st, e, e, L, L, T, seed=5, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=10**9))
        out["%s %s %s C=%d L=%d T=%d" % (mname, kind, sampler, C, L, T)] = ms
    print(json.dumps(out))
    sys.exit(0)
res = {}
for tag, env in (("one workgroup per block", {"ARP_DEBUG": "1", "ARP_SEGMENTS": "1"}), ("relay (library's choice)", {})):
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, **env), stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    res[tag] = json.loads(r.stdout.strip().splitlines()[-1])
a, b = res["one workgroup per block"], res["relay (library's choice)"]
for k in a:
    print("%-44s %9.3f ms -> %9.3f ms  (%+.1f %%)" % (k, a[k], b[k], 100 * (b[k] / a[k] - 1)))
