#!/usr/bin/env python3
"""Why do 4 of 1 024 chains of BASELINE config 3 (german credit, dVIP, L = 4, the reference's full schedule) never move
after burn-in (profiles/r04_ess_reconcile.txt: "chains with a constant series")?  A diagnostic, not a test: it uses the
oracle as the checker, so it lives under tests/.

  1. the HIP run of tools/ess_reconcile.py again, driven in 1 000-transition launches: acceptance history per chain,
     the dual-averaging state when adaptation ends (transition 6 000), the stuck chains' positions;
  2. the float32 and float64 oracle CONTINUED from the HIP run's own state at transition 6 000 (same random streams,
     same frozen step size) for the stuck chains and for controls: does the restated algorithm move from there?
  3. the float64 (and float32) oracle FROM SCRATCH on the same seeds, all chains, whole schedule: how many of ITS chains
     stick, which ones, and how their step sizes compare.

    python tests/diagnostics/stuck_chains.py [chains=1024] [samples=50000] > profiles/r05_stuck_chains.txt
"""
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402  (the checker)
from autoreparam_amd import flags as flags_mod, main as cli, inference, util, models, engine, _lib  # noqa: E402

C = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
S = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
LANES = int(sys.argv[3]) if len(sys.argv) > 3 else 0            # lanes per chain (0: the library's choice)
HIP_ONLY = len(sys.argv) > 4 and sys.argv[4] == "hip"            # only part 1 (the rate of stuck chains per kernel path)
NO_SCRATCH = len(sys.argv) > 4 and sys.argv[4] == "continued"     # parts 1 and 2 only
L = 4
dev = torch.device("cuda", 0)
d = tempfile.mkdtemp(prefix="arp_stuck_")
base = ["--model=german_credit_lognormalcentered", "--results_dir=" + d, "--num_chains=%d" % C]
for m in ("cVIP", "dVIP"):
    cli.main(base + ["--inference=VI", "--method=" + m], flags=flags_mod.FlagValues())
f = flags_mod.FlagValues()
util.print_ = lambda *a, **k: None
f.parse(base + ["--inference=HMC", "--method=dVIP", "--num_leapfrog_steps=%d" % L, "--num_samples=%d" % S,
                "--lanes_per_chain=%d" % LANES])
cfg = models.get_model_by_name("german_credit_lognormalcentered", dataset="")
r = json.load(open(os.path.join(d, "dVIP_eig_tied.json")))
target = cli.create_target_graph(cfg, d, f)[0]
spec = target.spec
a, b = (np.asarray(v, np.float32) for v in target.ab)
init = list(util.variational_inits_from_params(r["learned_variational_params"], param_names=list(spec.part_names),
                                               num_inits=C, seed=f.seed).values())
q0 = spec.pack([np.asarray(p, np.float32) for p in init])
eps0 = inference._flat_step(spec, r["initial_step_size"], L)
B, NA = int(f.num_burnin_steps), int(f.num_adaptation_steps)
total = 1 + B + 2 * (S - 1)
print("german credit dVIP, C = %d, L = %d, S = %d: %d transitions, adaptation ends at %d, burn-in at %d; a = 1 on %d of %d elements"
      % (C, L, S, total, NA, B, int((a > 0.5).sum()), spec.D))

# ---- 1. the HIP run
eng = engine.Engine(spec, dev)
eng.set_param(0, (a, b))
st = engine.ChainState(torch.as_tensor(q0, device=dev))
CH = 1000
hist, snaps = [], {}
kw = dict(seed=f.seed, adapt_kind=_lib.ADAPT_DUAL, n_adapt=NA, adapt_target=0.75, n_burnin=B, thin=2, lanes=f.lanes_per_chain)
t0 = time.time()
prev = np.zeros(C, np.int64)
while st.step < total:
    n = min(CH, total - st.step)
    eng.hmc_run(st, eps0, L, n, **kw)
    acc = st.accept_count.cpu().numpy().astype(np.int64)
    hist.append(acc - prev)
    prev = acc
    if st.step in (NA, 2 * NA):
        snaps[st.step] = {k: getattr(st, k).cpu().numpy().copy() for k in ("q", "grad", "logp", "adapt", "rng", "accept_count")}
torch.cuda.synchronize()
hist = np.array(hist)                                             # [chunks, C]
lanes = int((np.abs(snaps[NA]["rng"].reshape(C, -1, 4)[0, :, :2]).sum(axis=1) != 0).sum())
print("HIP run: %.1f s, %d lanes per chain; acceptance rate after burn-in %.2f %%" % (
    time.time() - t0, lanes, 100.0 * hist[B // CH:].sum() / (C * (total - B))))
post = hist[(2 * NA) // CH:].sum(axis=0)                          # accepted transitions after transition 12 000
stuck = np.where(post == 0)[0]
print("chains with NO accepted transition after transition %d (of %d): %d of %d (%.2f %%) -> ids %s%s" % (
    2 * NA, total - 2 * NA, len(stuck), C, 100.0 * len(stuck) / C, stuck.tolist()[:12], " ..." if len(stuck) > 12 else ""))
never = np.where(hist[B // CH:].sum(axis=0) == 0)[0]
print("chains with NO accepted transition after burn-in (a constant recorded series, ESS = nan): %d of %d" % (len(never), C))
ad = snaps[NA]["adapt"]                                           # [C, 4]: kappa, error sum, log averaged kappa, -
kbar = np.exp(ad[:, 2].astype(np.float64))
print("averaged step multiplier kappa_bar at the end of adaptation: quantiles 1/25/50/75/99 %% = %s" % np.round(
    np.quantile(kbar, [0.01, 0.25, 0.5, 0.75, 0.99]), 4).tolist())
ctrl = [c for c in range(C) if c not in set(stuck.tolist())][:4]
sel = stuck.tolist()[:6] + ctrl
xc = eng.transform(torch.as_tensor(snaps[2 * NA]["q"], device=dev), which=0, to_centered=True).cpu().numpy()
F = (spec.D - 1) // 2
print("\nchain  stuck  kappa_bar  kappa(6000)  rank of kappa_bar  accepted per 1 000 transitions: 0-1k 2-3k 5-6k | 6-7k 7-8k 11-12k | after 12k   "
      "logp(12k)  overall_log_scale  min beta_log_scale (centred, at 12k)")
order = np.argsort(np.argsort(kbar))
for c in sel:
    h = hist[:, c]
    print("%5d  %-5s  %9.4f  %11.4f  %8d / %-6d  %34s | %3d %3d %3d | %6d     %9.2f  %8.3f  %8.3f" % (
        c, "yes" if c in stuck else "no", kbar[c], ad[c, 0], order[c] + 1, C, "%3d %3d %3d" % (h[0], h[2], h[5]),
        h[6], h[7], h[11], post[c], snaps[2 * NA]["logp"][c], xc[c, 0], xc[c, 1:1 + F].min()))

if HIP_ONLY:
    sys.exit(0)
# ---- 2. the oracle continued from the HIP state at the end of adaptation
orc = oracle.OracleModel(spec)
NC = 6000
print("\noracle CONTINUED from the HIP run's state at transition %d (same streams, same frozen step), %d more transitions:" % (NA, NC))
print("chain  stuck(HIP)  HIP accepted in (6k, 12k]   f32 oracle accepted   f64 oracle accepted   f64: median / max log alpha")
for c in sel:
    row = []
    la64 = None
    for dt in (np.float32, np.float64):
        so = dict(q=snaps[NA]["q"][c:c + 1].astype(dt), grad=snaps[NA]["grad"][c:c + 1].astype(dt),
                  logp=snaps[NA]["logp"][c:c + 1].astype(dt), adapt=snaps[NA]["adapt"][c:c + 1].astype(dt),
                  adapt1=np.zeros((1, 4), dt), accept_count1=np.zeros(1, np.uint32),
                  rng=snaps[NA]["rng"][c:c + 1].view(np.uint32).copy(), accept_count=np.zeros(1, np.uint32), step=NA)
        la = np.zeros((NC, 1), dt)
        orc.hmc_run(so, a, b, eps0, L, NC, seed=f.seed, chain_offset=c, adapt_kind=1, n_adapt=NA, adapt_target=0.75,
                    n_burnin=B, thin=2, lanes=lanes, log_alpha=la)
        row.append(int(so["accept_count"][0]))
        if dt == np.float64:
            la64 = la[:, 0]
    fin = la64[np.isfinite(la64)]
    print("%5d  %-10s  %24d   %19d   %19d   %s" % (c, "yes" if c in stuck else "no", hist[NA // CH:(2 * NA) // CH, c].sum(),
                                                    row[0], row[1],
                                                    ("%.1f / %.1f" % (np.median(fin), fin.max())) if len(fin) else "all non-finite"))

if NO_SCRATCH:
    sys.exit(0)
# ---- 3. the oracle from scratch, same seeds, all chains, whole schedule
for dt, name in ((np.float64, "float64"), (np.float32, "float32")):
    t0 = time.time()
    so = oracle.new_state(q0, dt)
    accs = []
    prev = np.zeros(C, np.int64)
    for stop in (NA, 2 * NA, total):
        orc.hmc_run(so, a, b, eps0, L, stop - so["step"], seed=f.seed, chain_offset=0, adapt_kind=1, n_adapt=NA,
                    adapt_target=0.75, n_burnin=B, thin=2, lanes=lanes)
        cur = so["accept_count"].astype(np.int64)
        accs.append(cur - prev)
        prev = cur
        if stop == NA:
            kb_o = np.exp(so["adapt"][:, 2].astype(np.float64))
    stuck_o = np.where(accs[2] == 0)[0]
    print("\n%s oracle from scratch (same seeds, %d chains, %d transitions, %.0f s): chains with no accepted transition after %d: %d -> ids %s"
          % (name, C, total, time.time() - t0, 2 * NA, len(stuck_o), stuck_o.tolist()[:12]))
    print("   in common with the HIP run's: %s;  kappa_bar quantiles 1/25/50/75/99 %% = %s" % (
        sorted(set(stuck_o.tolist()) & set(stuck.tolist())),
        np.round(np.quantile(kb_o, [0.01, 0.25, 0.5, 0.75, 0.99]), 4).tolist()))
    for c in stuck_o.tolist()[:6]:
        print("   chain %5d: kappa_bar %.4f (rank %d / %d), accepted in (6k, 12k]: %d" % (
            c, kb_o[c], int((kb_o < kb_o[c]).sum()) + 1, C, accs[1][c]))
    print("   the HIP run's stuck chains in this run: %s" % ", ".join(
        "%d: kappa_bar %.4f, accepted after 12k %d" % (c, kb_o[c], accs[2][c]) for c in stuck.tolist()[:6]))
