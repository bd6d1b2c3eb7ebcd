"""cProfile of one full CLI sampling run at the headline size (radon PA --method=i, 65 536 chains, S = 1 000): where main.py's
wall clock goes outside the kernels."""
import cProfile, pstats, io, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import torch
from autoreparam_amd import flags as flags_mod, main as cli
d = tempfile.mkdtemp()
C = sys.argv[1] if len(sys.argv) > 1 else "65536"
base = ["--model=radon", "--dataset=PA", "--results_dir=" + d, "--num_chains=" + C]
short = ["--num_samples=1000", "--num_burnin_steps=1000", "--num_adaptation_steps=600"]
for m in ("CP", "NCP"):
    cli.main(base + ["--inference=VI", "--method=" + m], flags=flags_mod.FlagValues())
    cli.main(base[:3] + ["--num_chains=4096", "--inference=HMCtuning", "--method=" + m, "--num_leapfrog_steps=4"] + short, flags=flags_mod.FlagValues())
cli.main(base + ["--inference=HMC", "--method=i"] + short, flags=flags_mod.FlagValues())     # warm
pr = cProfile.Profile(); pr.enable(); t0 = time.time()
cli.main(base + ["--inference=HMC", "--method=i"] + short, flags=flags_mod.FlagValues())
torch.cuda.synchronize(); dt = time.time() - t0; pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(26); print(s.getvalue()[:6000]); print("wall", dt)
