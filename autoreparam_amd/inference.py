"""Inference drivers with the reference's signatures (inference.py:26-356), running
on the HIP engine: find_best_learning_rate (mean-field VI with the learning-rate
sweep), hmc (batched HMC + dual-averaging adaptation + sample_chain thinning) and
hmc_interleaved (CP/NCP interleaving with simple adaptation).  Results come back as
numpy arrays in the reference's layouts instead of lazy TF tensors.
"""
import collections

import numpy as np
import torch

from . import _lib
from . import engine as _engine
from . import util
from .flags import FLAGS

HmcInnerResults = collections.namedtuple("HmcInnerResults", ["is_accepted"])
# ess_info (EssInfo: which estimator the returned ESS is, over how many chains) and moments (per-chain mean / variance from
# the in-kernel accumulators of a streaming run, else None) belong to the call that produced them and travel with its result
KernelResults = collections.namedtuple("KernelResults", ["inner_results", "new_step_size", "step", "ess_info", "moments"],
                                       defaults=(None, None))
InterleavedKernelResults = collections.namedtuple("InterleavedKernelResults", ["cp_results", "ncp_results", "ess_info", "moments"],
                                                  defaults=(None, None))

# transitions per launch: keeps a single launch well under a second at any size
_MAX_STEPS_PER_LAUNCH = 4096


class DiscretePrior(object):
    """The reference's --discrete_prior (main.py:244-253): Mixture(Categorical(logits=[0, 5, 0]),
    [Laplace(0, 0.1), Uniform(0, 1), Laplace(1, 0.1)]) on every learnable parameterisation parameter;
    its log density is added to the VI objective (inference.py:50-54).  Evaluated inside the VI kernel
    (`arp_vi_config.a_prior`); this object only selects it."""

    def log_prob(self, x):
        x = np.asarray(x, np.float64)
        lap0, lap1 = 5.0 * np.exp(-np.abs(x) * 10.0), 5.0 * np.exp(-np.abs(x - 1.0) * 10.0)
        uni = ((x >= 0) & (x <= 1)).astype(np.float64)
        return np.log((lap0 + np.exp(5.0) * uni + lap1) / (2.0 + np.exp(5.0)))


def find_best_learning_rate(elbo, variational_parameters, learnable_parameters_prior=None,
                            learnable_parameters=None, flags=FLAGS):
    """Optimise the ELBO with every learning rate of the sweep and keep the best run
    (reference inference.py:26-154).  All learning rates run concurrently inside one kernel launch, each on a group of
    G x R workgroups that share its Monte-Carlo draws (and, German credit, its observations' row parts)."""
    if learnable_parameters_prior is not None and not isinstance(learnable_parameters_prior, DiscretePrior):
        raise NotImplementedError("the only prior on the learnable parameterisation the engine evaluates is "
                                  "inference.DiscretePrior (the reference's --discrete_prior mixture)")
    spec = elbo.target.spec
    dev = torch.device(flags.device)
    eng = _engine.engine_for(spec, dev)
    a, b = elbo.target.ab
    eng.set_param(0, (a, b))
    lrs = [float(v) for v in flags.learning_rates]
    n_lr, D = len(lrs), spec.D
    rs = np.random.RandomState(flags.seed)
    loc0 = (1e-2 * rs.randn(n_lr, D)).astype(np.float32)
    rho0 = np.full((n_lr, D), -2.0, np.float32)
    loc = torch.as_tensor(loc0, device=dev)
    rho = torch.as_tensor(rho0, device=dev)
    w = wb = None
    if elbo.learn_a:
        w = torch.zeros(n_lr, D, device=dev)
        if not elbo.tied:
            wb = torch.zeros(n_lr, D, device=dev)
    a_group = b_group = None
    if elbo.learn_a and not elbo.tied:
        a_group, b_group = spec.untied_groups()
    use_prior = learnable_parameters_prior is not None and elbo.learn_a
    res = eng.vi_run(lrs, loc, rho, flags.num_optimization_steps, flags.num_mc_samples, which=0, w=w, wb=wb,
                     seed=flags.seed, a_prior=use_prior, a_group=a_group, b_group=b_group, return_prior=use_prior)
    if use_prior:
        # the reference optimises, ranks the learning rates on and returns the timeline of elbo + prior, and
        # subtracts the prior only from the final value (inference.py:50-54, 120-150)
        pure, prior_tl = (t.cpu().numpy().astype(np.float64) for t in res)
        timeline = pure + prior_tl
    else:
        timeline = res.cpu().numpy().astype(np.float64)
        prior_tl = np.zeros_like(timeline)
    loc, rho = loc.cpu().numpy(), rho.cpu().numpy()
    scale = np.where(rho > 20, rho, np.log1p(np.exp(np.minimum(rho, 20))))

    best = None
    best_elbo = None
    for i, lr in enumerate(lrs):
        this = np.mean(timeline[i, -32:])
        util.print_("     finished optimization with elbo {} vs best ELBO {}".format(this, best_elbo))
        if not np.isfinite(this):
            continue
        if best_elbo is None or best_elbo < this:
            best_elbo, best = this, i
    if best is None:
        raise RuntimeError("no learning rate gave a finite ELBO")
    best_pure_elbo = best_elbo - np.mean(prior_tl[best, -32:])   # "a 'pure' ELBO for valid comparisons"  (inference.py:150)
    learned_variational_params = collections.OrderedDict()
    for k, name in enumerate(spec.part_names):
        lo, hi = spec.offsets[k], spec.offsets[k + 1]
        learned_variational_params[name + "_loc"] = loc[best, lo:hi].reshape(spec.part_shapes[k])
        learned_variational_params[name + "_scale"] = scale[best, lo:hi].reshape(spec.part_shapes[k])
    step_size_init = util.get_approximate_step_size(learned_variational_params, num_leapfrog_steps=1)
    learned_reparam = None
    if elbo.learn_a:
        av = 1.0 / (1.0 + np.exp(-w.cpu().numpy()[best]))
        bv = 1.0 / (1.0 + np.exp(-wb.cpu().numpy()[best])) if wb is not None else None
        learned_reparam = collections.OrderedDict()
        for k, name in enumerate(spec.part_names):
            lo, hi = spec.offsets[k], spec.offsets[k + 1]
            if bv is None:
                learned_reparam[name + "_a"] = av[lo:hi].reshape(spec.part_shapes[k]).astype(np.float32)
            else:   # untied: the reference's variable shapes (a shared value is a scalar)
                sa, sb = spec.untied_shape(name, "a"), spec.untied_shape(name, "b")
                learned_reparam[name + "_a"] = (av[lo:hi].reshape(spec.part_shapes[k]) if sa == spec.part_shapes[k]
                                                else av[lo].reshape(())).astype(np.float32)
                learned_reparam[name + "_b"] = (bv[lo:hi].reshape(spec.part_shapes[k]) if sb == spec.part_shapes[k]
                                                else bv[lo].reshape(())).astype(np.float32)
    return (np.float64(best_pure_elbo), list(timeline[best]), lrs[best], step_size_init,
            learned_variational_params, learned_reparam)


def _flat_step(spec, step_size_init, L):
    """inference.py:212-216: step_size_init[i] * ones(part shape) / (L/4)^2, flattened to [D]."""
    out = np.zeros(spec.D, np.float32)
    for k in range(len(spec.part_names)):
        lo, hi = spec.offsets[k], spec.offsets[k + 1]
        v = np.asarray(step_size_init[k], np.float64)
        out[lo:hi] = (np.broadcast_to(v, spec.part_shapes[k]).reshape(-1) / (float(L) / 4.0) ** 2)
    return out


class _DevicePart(object):
    """One latent part of a recorded trace, [S, C, *event], left on the device: indexing (e.g. the
    `[:, :num_chains_to_save]` main.py takes for _traces.npz) and np.asarray() copy to the host only
    what is asked for.  The reference returns whole numpy arrays; at 16 384 x 125 x 2 000 samples that
    copy alone is 16 GB."""

    def __init__(self, tensor):
        self._t = tensor
        self.shape = tuple(tensor.shape)
        self.dtype = np.dtype(np.float32)
        self.ndim = tensor.dim()

    def __getitem__(self, idx):
        return self._t[idx].cpu().numpy()

    def __array__(self, dtype=None, copy=None):
        a = self._t.cpu().numpy()
        return a.astype(dtype) if dtype is not None else a

    def __len__(self):
        return self.shape[0]


class _DeviceAccept(_DevicePart):
    """is_accepted [S, C] left on the device: main.py only ever takes np.sum() of it (acceptance rate), which
    reduces on the GPU; anything else copies as a bool array.  (65 536 chains x 1 000 samples x two inner kernels are
    131 MB of pageable device-to-host copies otherwise -- a quarter of the wall clock of such a run.)"""

    def __init__(self, tensor):
        _DevicePart.__init__(self, tensor)
        self.dtype = np.dtype(bool)

    def sum(self, axis=None, dtype=None, out=None, **kw):
        if axis is None and out is None:
            return np.int64(self._t.sum(dtype=torch.int64).item())
        return np.asarray(self).sum(axis=axis, dtype=dtype, out=out, **kw)

    def __getitem__(self, idx):
        return self._t[idx].cpu().numpy().astype(bool)

    def __array__(self, dtype=None, copy=None):
        a = self._t.cpu().numpy().astype(bool)
        return a.astype(dtype) if dtype is not None else a


def _device_parts(spec, trace):
    """spec.unpack on a device trace, without leaving the device."""
    return [_DevicePart(t) for t in spec.unpack(trace)]


class _LazyOriginalStates(object):
    """states in the sampler's own coordinates, recovered on demand from the
    centred trace (the reference materialises both; main.py never reads these)."""

    def __init__(self, eng, spec, trace, which):
        self._args = (eng, spec, trace, which)
        self._val = None

    def get(self):
        if self._val is None:
            eng, spec, trace, which = self._args
            S, C, D = trace.shape
            flat = eng.transform(trace.reshape(S * C, D), which=which, to_centered=False).reshape(S, C, D)
            self._val = spec.unpack(flat.cpu().numpy())
        return self._val

    def __iter__(self):
        return iter(self.get())

    def __getitem__(self, i):
        return self.get()[i]

    def __len__(self):
        return len(self._args[1].part_names)



class StreamingStats(object):
    """Host-side restatement of the kernels' streaming statistics (arp_hmc_io.stats), kept as their checker.
    Statistics of a trace that is produced chunk by chunk (and then discarded):
    running first/second moments per (chain, element) and means of consecutive
    batches of `batch` samples, from which ESS is estimated by batch means,
    ESS = S * var / (batch * var(batch means)).  Used when the reference's
    [S, C, D] trace (inference.py:228-240 keeps it whole) does not fit in HBM."""

    def __init__(self, C, D, batch, device):
        self.batch = int(batch)
        self.n = 0
        self.s1 = torch.zeros(C, D, dtype=torch.float64, device=device)
        self.s2 = torch.zeros(C, D, dtype=torch.float64, device=device)
        self.bmeans = []
        self._carry = None

    def update(self, chunk):
        x = chunk.to(torch.float64)
        self.n += x.shape[0]
        self.s1 += x.sum(dim=0)
        self.s2 += (x * x).sum(dim=0)
        if self._carry is not None:
            x = torch.cat([self._carry, x], dim=0)
        nb = x.shape[0] // self.batch
        if nb:
            self.bmeans.append(x[: nb * self.batch].reshape(nb, self.batch, *x.shape[1:]).mean(dim=1).to(torch.float32))
        self._carry = x[nb * self.batch:] if x.shape[0] > nb * self.batch else None

    def mean(self):
        return self.s1 / self.n

    def var(self):
        m = self.mean()
        return (self.s2 / self.n - m * m).clamp_min(0) * (self.n / max(self.n - 1, 1))

    def ess(self):
        bm = torch.cat(self.bmeans, dim=0).to(torch.float64)
        if bm.shape[0] < 2:
            raise ValueError("batch-means ESS needs at least two complete batches")
        vb = bm.var(dim=0, unbiased=True)
        ess = self.n * self.var() / (self.batch * vb)
        return torch.minimum(ess, torch.full_like(ess, float(self.n))).to(torch.float32)


def _trace_plan(S, C, D, dev, force_chunk=None):
    """(rows, streaming?) -- the whole trace if it fits in 60 % of the free HBM; `rows` only sets the batch
    length of the batch-means ESS in streaming mode."""
    free, _ = torch.cuda.mem_get_info(dev)
    row = 4.0 * C * D + C
    if force_chunk:
        return int(force_chunk), True
    if S * row <= 0.6 * free:
        return S, False
    rows = int(0.25 * free // row)
    if rows < 64:
        raise MemoryError("not even 64 trace rows of [C=%d, D=%d] fit in HBM" % (C, D))
    return min(rows, S), True


class EssInfo(object):
    """What a sampling run's ESS is: `estimator` ("autocorrelation" = tfp.mcmc.effective_sample_size's definition, the
    reference's; "batch_means(b)" only when not even a small chain subset's trace fits), `chains` = the number of this
    rank's chains it was computed on (all of them with a whole trace; the chains with global id < --ess_chains in
    streaming mode), and -- streaming mode -- `batch_means`, the batch-means ESS [C, D] of EVERY local chain from the
    in-kernel accumulators with its batch length, reported next to the autocorrelation figure, never instead of it."""

    def __init__(self, estimator, chains, batch_means=None, batch=None, kernel_ms=None):
        self.estimator, self.chains, self.batch_means, self.batch, self.kernel_ms = estimator, chains, batch_means, batch, kernel_ms


def _ess_subset(S, C, D, dev, chain_offset, ess_chains):
    """(k_total, k_local): the chains whose whole trace a streaming run keeps for the autocorrelation ESS are those with
    GLOBAL id < k_total, so the subset -- hence the reported figure -- does not depend on how many ranks share the job;
    k_local of them are this rank's.  k_total = --ess_chains, halved until [S, k_total, D] fits in 40 % of the device's
    memory (a function of the device model only: every rank of a job takes the same decision); 0 when not even 64 chains
    fit, or --ess_chains=0 asks for batch means only."""
    k = max(int(ess_chains), 0)
    cap = 0.4 * torch.cuda.get_device_properties(dev).total_memory
    while k > 64 and 4.0 * S * k * D > cap:
        k //= 2
    if 4.0 * S * k * D > cap:
        k = 0
    return k, int(min(C, max(k - int(chain_offset), 0)))


def _sample(run_segment, st, S, B, thin, C, D, dev, keep_chains, n_acc, chunk_rows=None, chain_offset=0, ess_chains=1024):
    """Drive `run_segment(n_steps, n_burnin, trace, accept_buffers, **extra)` over the whole
    sample_chain schedule (result r after transition 1 + B + r*thin).  Whole-trace mode keeps the
    reference's [S, C, D] trace and takes tfp's autocorrelation ESS of every series (arp_ess).  When that does not
    fit in HBM (or --trace_chunk_rows forces it) the run STREAMS: the kernels accumulate the per-chain moments and batch
    means themselves (arp_hmc_io.stats) for all chains, and the chains with global id < --ess_chains keep their whole
    [S, k, D] trace on the device, on which the same autocorrelation ESS is taken -- the reference's estimator on a
    chain subset (its mean over chains has a standard error of ~ 1/sqrt(k) of the spread between chains), with the
    batch-means figure of all chains next to it.
    Returns (trace or None, kept trace (device, [S, k, D]) or None, accept arrays, ess [k, D], EssInfo, moments):
    moments = (mean, var) [C, D] float64 from the in-kernel accumulators in streaming mode, None with a whole trace."""
    rows, streaming = _trace_plan(S, C, D, dev, chunk_rows)
    total = 1 + B + thin * (S - 1)
    if C == 0:
        # a rank of a job with fewer chains than ranks: nothing to launch, empty blocks for the end-of-run gathers
        total = 0
    if not streaming:
        trace = torch.empty(rows, C, D, dtype=torch.float32, device=dev)
        accs = [torch.empty(rows, C, dtype=torch.uint8, device=dev) for _ in range(n_acc)]
        done = 0
        while done < total:
            n = min(_MAX_STEPS_PER_LAUNCH, total - done)
            run_segment(n, B, trace, accs)
            done += n
        ess = util.effective_sample_size(trace) if C > 0 else torch.empty(0, D, dtype=torch.float32, device=dev)
        # the whole trace is returned: moments are the caller's to take
        return trace, None, [_DeviceAccept(a) for a in accs], ess, EssInfo("autocorrelation", C), None
    batch = max(8, min(rows, S) // 8)
    k_total, k_ess = _ess_subset(S, C, D, dev, chain_offset, ess_chains)
    k = max(k_ess, keep_chains)            # --num_chains_to_save chains keep their trace on every rank, as before
    stats = torch.zeros(6, C, D, dtype=torch.float32, device=dev)
    kept = torch.empty(S, k, D, dtype=torch.float32, device=dev)      # every row is written by the run (25.6 GB at config 3: no fill)
    racc = [torch.zeros(C, dtype=torch.int32, device=dev) for _ in range(n_acc)]
    extra = dict(stats=stats, stats_batch=batch, n_samples=S, trace_chains=k)
    for j in range(n_acc):
        extra["rec_accept%d" % j] = racc[j]
    done = 0
    while done < total:
        n = min(_MAX_STEPS_PER_LAUNCH, total - done)
        run_segment(n, B, kept, [None] * n_acc, **extra)
        done += n
    mean, var, ess_bm = _engine.stats_summary(stats, S, batch)
    ess_bm = ess_bm.to(torch.float32)
    if k_total > 0:
        # [k_ess, D]; empty on a rank that owns none of the subset (the summaries gather variable-length blocks)
        ess = util.effective_sample_size(kept[:, :k_ess]) if k_ess > 0 else torch.empty(0, D, dtype=torch.float32, device=dev)
        info = EssInfo("autocorrelation", k_ess, batch_means=ess_bm, batch=batch)
    else:
        # --ess_chains=0, or not even 64 chains' traces fit next to the run: batch means are all there is (the key says so)
        ess = ess_bm
        info = EssInfo("batch_means(%d)" % batch, C, batch_means=ess_bm, batch=batch)
    # per-chain posterior mean / variance of every (centred) element from the in-kernel accumulators, [C, D] float64
    # (build-specific: the reference would take them from the [S, C, D] trace this mode does not materialise)
    return None, kept, [a.cpu().numpy()[np.newaxis, :] for a in racc], ess, info, (mean, var)


def _check_trace_fits(S, C, D, dev):
    need = 4.0 * S * C * D
    free, _ = torch.cuda.mem_get_info(dev)
    if need > 0.8 * free:
        raise MemoryError("the trace [S=%d, C=%d, D=%d] needs %.1f GB of HBM (%.1f GB free); lower --num_samples "
                          "or --num_chains" % (S, C, D, need / 1e9, free / 1e9))


def _initial_rows(spec, parts, dev):
    """The reference's list of [C, *event] host arrays (main.py:310-313) as the engine's [C, D] device rows: every part goes
    to the device as it is and the rows are put together there (interleaving 18 MB of columns on the host cost 5 ms of the
    headline flow's 75)."""
    cols = []
    for p in parts:
        t = p if torch.is_tensor(p) else torch.from_numpy(np.ascontiguousarray(p, np.float32))
        t = t.to(device=dev, dtype=torch.float32)
        cols.append(t.reshape(t.shape[0], int(np.prod(t.shape[1:], dtype=np.int64))))     # (-1 is ambiguous for a rank without chains)
    q0 = torch.cat(cols, dim=1).contiguous()
    if q0.shape[1] != spec.D:
        raise ValueError("initial states do not have the model's parts: %d columns, the model has D = %d" % (q0.shape[1], spec.D))
    return q0


def hmc(target, model_config, step_size_init, initial_states, reparam, flags=FLAGS, chain_offset=0):
    """Batched HMC with dual-averaging step-size adaptation (reference inference.py:198-242).

    Returns (states_orig, kernel_results, states_transformed, ess) with
    states_* = list of [S, C, *event] arrays, kernel_results.inner_results.is_accepted
    [S, C], ess = list of [C, *event] (tfp.mcmc.effective_sample_size of the
    centred states).  `reparam` is accepted for signature compatibility; the
    parameterisation is the target's."""
    spec = target.spec
    dev = torch.device(flags.device)
    eng = _engine.engine_for(spec, dev)
    eng.set_param(0, target.ab)
    L = int(flags.num_leapfrog_steps)
    q0 = _initial_rows(spec, initial_states, dev)
    C = q0.shape[0]
    S, B = int(flags.num_samples), int(flags.num_burnin_steps)
    eps0 = _flat_step(spec, step_size_init, L)
    thin = 2                                      # num_steps_between_results=1 (inference.py:234)
    st = _engine.ChainState(q0)

    def run_segment(n, n_burnin, trace, accs, rec_accept0=None, **extra):
        eng.hmc_run(st, eps0, L, n, which=0, seed=flags.seed, chain_offset=chain_offset,
                    adapt_kind=_lib.ADAPT_DUAL, n_adapt=int(flags.num_adaptation_steps), adapt_target=0.75,
                    n_burnin=n_burnin, thin=thin, trace=trace, trace_accept=accs[0], trace_centered=True,
                    lanes=flags.lanes_per_chain, rec_accept=rec_accept0, **extra)

    keep = max(1, int(flags.num_chains_to_save))
    trace, kept, accs, ess_flat, info, moments = _sample(run_segment, st, S, B, thin, C, spec.D, dev, min(keep, C), 1,
                                                        getattr(flags, "trace_chunk_rows", None), chain_offset,
                                                        getattr(flags, "ess_chains", 1024))
    torch.cuda.synchronize(dev)
    eng.check()                                   # a relay hand-over that timed out inside a launch surfaces here
    ess = spec.unpack(ess_flat.cpu().numpy())
    step_mult = st.adapt[:, 0].cpu().numpy()
    kernel_results = KernelResults(HmcInnerResults(accs[0]), step_mult, st.step, info, moments)
    if trace is not None:
        states_transformed = _device_parts(spec, trace)
        states_orig = _LazyOriginalStates(eng, spec, trace, 0)
    else:
        # streaming run: only the chains with global id < --ess_chains keep their trace ([S, k, D], left on the device),
        # is_accepted holds per-chain counts ([1, C]; np.sum is unchanged) and `ess` is [k, *event]: the reference's
        # autocorrelation ESS of those chains (EssInfo; the batch-means figure of all chains rides along in it)
        states_transformed = _device_parts(spec, kept)
        states_orig = None
    return states_orig, kernel_results, states_transformed, ess


def hmc_interleaved(model_config, target_cp, target_ncp, num_leapfrog_steps_cp, num_leapfrog_steps_ncp,
                    step_size_cp, step_size_ncp, initial_states_cp, flags=FLAGS, chain_offset=0):
    """Interleaved CP/NCP HMC with SimpleStepSizeAdaptation(0.05, 0.75) on each inner
    kernel (reference inference.py:258-329, interleaved.py:113-155).

    Returns (states, kernel_results, ess); states are in CP coordinates."""
    spec = target_cp.spec
    dev = torch.device(flags.device)
    eng = _engine.engine_for(spec, dev)
    eng.set_param(0, target_cp.ab)
    eng.set_param(1, target_ncp.ab)
    q0 = _initial_rows(spec, initial_states_cp, dev)
    C = q0.shape[0]
    S, B = int(flags.num_samples), int(flags.num_burnin_steps)
    e_cp = _flat_step(spec, step_size_cp, num_leapfrog_steps_cp)
    e_ncp = _flat_step(spec, step_size_ncp, num_leapfrog_steps_ncp)
    thin = 2
    st = _engine.ChainState(q0)

    def run_segment(n, n_burnin, trace, accs, **extra):
        eng.interleaved_run(st, e_cp, e_ncp, int(num_leapfrog_steps_cp), int(num_leapfrog_steps_ncp), n,
                            seed=flags.seed, chain_offset=chain_offset, adapt_kind=_lib.ADAPT_SIMPLE,
                            n_adapt=int(flags.num_adaptation_steps), adapt_target=0.75, adapt_rate=0.05,
                            n_burnin=n_burnin, thin=thin, trace=trace, trace_accept0=accs[0], trace_accept1=accs[1],
                            trace_centered=False, lanes=flags.lanes_per_chain, **extra)

    keep = max(1, int(flags.num_chains_to_save))
    trace, kept, accs, ess_flat, info, moments = _sample(run_segment, st, S, B, thin, C, spec.D, dev, min(keep, C), 2,
                                                        getattr(flags, "trace_chunk_rows", None), chain_offset,
                                                        getattr(flags, "ess_chains", 1024))
    torch.cuda.synchronize(dev)
    eng.check()
    states = _device_parts(spec, trace if trace is not None else kept)
    ess = spec.unpack(ess_flat.cpu().numpy())
    kr = InterleavedKernelResults(
        cp_results=KernelResults(HmcInnerResults(accs[0]), st.adapt[:, 0].cpu().numpy(), st.step),
        ncp_results=KernelResults(HmcInnerResults(accs[1]), st.adapt1[:, 0].cpu().numpy(), st.step),
        ess_info=info, moments=moments)
    return states, kr, ess
