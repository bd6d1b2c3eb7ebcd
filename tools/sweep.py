"""GPU sweep of the headline kernels over lanes-per-chain (prints one line per variant)."""
import os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for method in ("i", "CP"):
    for lanes in (4, 8, 16):
        for chains in (65536, 8192):
            out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--method", method, "--lanes", str(lanes),
                                  "--chains", str(chains), "--steps", "10", "--warmup", "2", "--no-cpu-baseline"],
                                 capture_output=True, text=True)
            try:
                j = json.loads(out.stdout.strip().splitlines()[-1])
                print("method=%s lanes=%2d chains=%6d  %.3e leapfrog/s  frac=%.3f  kernel_ms=%.3f acc=%.2f" % (
                    method, lanes, chains, j["value"], j["roofline"]["frac"], j["roofline"]["kernel_ms"], j["accept_rate"]), flush=True)
            except Exception as e:
                print("FAILED", method, lanes, chains, out.stderr[-500:])
