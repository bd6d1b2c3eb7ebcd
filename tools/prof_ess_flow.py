"""cProfile of inference.hmc_interleaved inside a full bench.py run (the ESS/sec flow): where main.py's mcmc_time goes.
usage: prof_ess_flow.py [standalone]"""
import cProfile, pstats, os, sys, io
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from autoreparam_amd import inference
orig = inference.hmc_interleaved
def wrapped(*a, **k):
    pr = cProfile.Profile(); pr.enable()
    try:
        return orig(*a, **k)
    finally:
        torch.cuda.synchronize(); pr.disable()
        s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14); print(s.getvalue()[:4000], file=sys.stderr)
inference.hmc_interleaved = wrapped
if len(sys.argv) > 1 and sys.argv[1] == "standalone":
    r = bench.reference_flow_ess("PA", 65536, 0)
    print({k: r[k] for k in ("mcmc_time_sec", "ess_per_sec", "mean_min_ess_per_chain", "num_ls_chosen")})
else:
    sys.argv = ["bench.py", "--steps", "20", "--warmup", "5", "--no-cpu-baseline"]
    bench.main()
