import cProfile, pstats, io, os, sys, tempfile, time
sys.path.insert(0, os.getcwd())
import torch
from autoreparam_amd import flags as flags_mod, main as cli
d = tempfile.mkdtemp()
base = ["--model=radon", "--dataset=PA", "--results_dir=" + d, "--num_chains=4096", "--method=CP"]
cli.main(base + ["--inference=VI"], flags=flags_mod.FlagValues())
short = ["--num_samples=1000", "--num_burnin_steps=1000", "--num_adaptation_steps=600"]
cli.main(base + ["--inference=HMCtuning", "--num_leapfrog_steps=2"] + short, flags=flags_mod.FlagValues())   # warm
pr = cProfile.Profile(); pr.enable(); t0 = time.time()
cli.main(base + ["--inference=HMCtuning", "--num_leapfrog_steps=4"] + short, flags=flags_mod.FlagValues())
torch.cuda.synchronize(); dt = time.time() - t0; pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(22); print(s.getvalue()[:5000]); print("wall", dt)
