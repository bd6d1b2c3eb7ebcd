"""Device-side handle around the C ABI: one Engine per (process, GPU, model).

torch is used only for device memory and streams; every number on the hot path is
produced by the HIP kernels in libautoreparam_hip.so.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class ChainState(object):
    """Per-chain persistent state of a run (the reference's kernel_results)."""

    def __init__(self, q):
        C_, D = q.shape
        dev = q.device
        self.q = q.contiguous().clone()
        self.grad = torch.zeros_like(self.q)
        self.logp = torch.zeros(C_, dtype=torch.float32, device=dev)
        self.adapt = torch.zeros(C_, 4, dtype=torch.float32, device=dev)
        self.rng = torch.zeros(C_, _lib.RNG_SLOTS, 4, dtype=torch.int32, device=dev)
        self.accept_count = torch.zeros(C_, dtype=torch.int32, device=dev)
        # second inner kernel of the interleaved sampler
        self.adapt1 = torch.zeros(C_, 4, dtype=torch.float32, device=dev)
        self.accept_count1 = torch.zeros(C_, dtype=torch.int32, device=dev)
        self.step = 0  # transitions done


class Engine(object):
    def __init__(self, spec, device=None):
        if not torch.cuda.is_available():
            raise RuntimeError("autoreparam_amd.Engine needs a GPU (gfx950); there is no CPU fallback")
        self.spec = spec
        self.device = torch.device(device if device is not None else "cuda:%d" % torch.cuda.current_device())
        self._L = _lib.lib()
        self._h = C.c_void_p(0)
        ds, keep = spec.dataset()
        with torch.cuda.device(self.device):
            _lib.check(self._L.arp_model_create(C.byref(ds), C.byref(self._h)))
        del keep
        self.D = self._L.arp_model_dim(self._h)
        assert self.D == spec.D, (self.D, spec.D)
        self._ab = [None, None]
        self._cache = {}

    def __del__(self):
        try:
            if self._h:
                self._L.arp_model_destroy(self._h)
                self._h = C.c_void_p(0)
        except Exception:
            pass

    # -- parameterisations ------------------------------------------------
    def set_param(self, which, reparam):
        """reparam: 'CP', 'NCP', a dict of ``<rv>_a``/``<rv>_b`` values, or an (a, b) pair of [D] arrays."""
        if isinstance(reparam, tuple):
            a, b = (np.ascontiguousarray(v, np.float32) for v in reparam)
        else:
            a, b = self.spec.ab_from_reparam(reparam)
        assert a.shape == (self.D,) and b.shape == (self.D,)
        with torch.cuda.device(self.device):
            _lib.check(self._L.arp_model_set_param(self._h, which, a.ctypes.data_as(_lib._f32p),
                                                   b.ctypes.data_as(_lib._f32p)))
        self._ab[which] = (a, b)

    def logp_const(self, which=0):
        return self._L.arp_model_logp_const(self._h, which)

    def _dev_cached(self, key, x):
        """Device copy of a small host array that is re-used while its contents do not change (the base step
        sizes of consecutive launches): no host-to-device copy, and no host synchronisation, per launch."""
        a = np.ascontiguousarray(x, np.float32)
        hit = self._cache.get(key)
        if hit is None or hit[0].shape != a.shape or not np.array_equal(hit[0], a):
            hit = (a.copy(), self._dev(a))
            self._cache[key] = hit
        return hit[1]

    # -- density / converters --------------------------------------------
    def _dev(self, x):
        t = torch.as_tensor(x, dtype=torch.float32)
        return t.to(self.device).contiguous()

    def logp_grad(self, x, which=0, lanes=0):
        x = self._dev(x)
        n = x.shape[0]
        logp = torch.empty(n, dtype=torch.float32, device=self.device)
        grad = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _lib.check(self._L.arp_logp_grad(self._h, which, _ptr(x), n, _ptr(logp), _ptr(grad), lanes, _stream()))
        return logp, grad

    def transform(self, x, which=0, to_centered=True):
        x = self._dev(x)
        out = torch.empty_like(x)
        with torch.cuda.device(self.device):
            _lib.check(self._L.arp_transform(self._h, which, 0 if to_centered else 1, _ptr(x), x.shape[0],
                                             _ptr(out), _stream()))
        return out

    # -- HMC ----------------------------------------------------------------
    def hmc_run(self, state, eps0, n_leapfrog, n_steps, which=0, seed=0, chain_offset=0,
                adapt_kind=_lib.ADAPT_NONE, n_adapt=0, adapt_target=0.75, adapt_rate=0.05,
                n_burnin=0, thin=1, trace=None, trace_accept=None, trace_centered=True, lanes=0,
                stats=None, stats_batch=1, n_samples=None, trace_chains=0, rec_accept=None):
        """Advance `state` by n_steps transitions (one kernel launch).

        `stats` ([6, C, D] float32, zeroed before the first call) accumulates the streaming statistics of the
        recorded samples inside the kernel (`stats_summary`), `rec_accept` ([C] int32) the accepted recorded
        transitions; with them a run needs no trace, or only one of the first `trace_chains` chains
        (`trace` is then [S, trace_chains, D]).  `n_samples` = recorded samples of the whole run when no
        trace buffer gives it."""
        cfg = _lib.HmcConfig()
        cfg.n_chains = state.q.shape[0]
        cfg.n_leapfrog = int(n_leapfrog)
        cfg.n_steps = int(n_steps)
        cfg.step_base = int(state.step)
        cfg.chain_offset = int(chain_offset)
        cfg.seed = int(seed)
        cfg.adapt_kind = int(adapt_kind)
        cfg.n_adapt = int(n_adapt)
        cfg.adapt_target = float(adapt_target)
        cfg.adapt_rate = float(adapt_rate)
        cfg.n_burnin = int(n_burnin)
        cfg.thin = int(thin)
        cfg.n_samples = int(n_samples) if n_samples is not None else (int(trace.shape[0]) if trace is not None else (
            int(trace_accept.shape[0]) if trace_accept is not None else 0))
        cfg.trace_centered = 1 if trace_centered else 0
        cfg.lanes_per_chain = int(lanes)
        cfg.stats_batch = int(stats_batch)
        cfg.trace_chains = int(trace_chains)
        io = _lib.HmcIO()
        io.q, io.grad, io.logp = _ptr(state.q), _ptr(state.grad), _ptr(state.logp)
        io.adapt, io.rng, io.accept_count = _ptr(state.adapt), _ptr(state.rng), _ptr(state.accept_count)
        self._eps0 = eps0 if torch.is_tensor(eps0) else self._dev_cached("eps0", eps0)
        io.eps0 = _ptr(self._eps0)
        io.trace, io.trace_accept, io.stats = _ptr(trace), _ptr(trace_accept), _ptr(stats)
        io.rec_accept_count = _ptr(rec_accept)
        with torch.cuda.device(self.device):
            _lib.check(self._L.arp_hmc_run(self._h, which, C.byref(cfg), C.byref(io), _stream()))
        state.step += int(n_steps)
        return state


    def interleaved_run(self, state, eps0_0, eps0_1, n_leapfrog_0, n_leapfrog_1, n_steps, seed=0,
                        chain_offset=0, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=0, adapt_target=0.75,
                        adapt_rate=0.05, n_burnin=0, thin=1, trace=None, trace_accept0=None,
                        trace_accept1=None, trace_centered=True, lanes=0, stats=None, stats_batch=1, n_samples=None,
                        trace_chains=0, rec_accept0=None, rec_accept1=None):
        """Advance `state` by n_steps interleaved steps (parameterisation 0 then 1 per step);
        stats / rec_accept* / trace_chains / n_samples as in hmc_run."""
        cfg = _lib.HmcConfig()
        cfg.n_chains = state.q.shape[0]
        cfg.n_leapfrog = int(n_leapfrog_0)
        cfg.n_steps = int(n_steps)
        cfg.step_base = int(state.step)
        cfg.chain_offset = int(chain_offset)
        cfg.seed = int(seed)
        cfg.adapt_kind = int(adapt_kind)
        cfg.n_adapt = int(n_adapt)
        cfg.adapt_target = float(adapt_target)
        cfg.adapt_rate = float(adapt_rate)
        cfg.n_burnin = int(n_burnin)
        cfg.thin = int(thin)
        cfg.n_samples = int(n_samples) if n_samples is not None else (int(trace.shape[0]) if trace is not None else (
            int(trace_accept0.shape[0]) if trace_accept0 is not None else 0))
        cfg.trace_centered = 1 if trace_centered else 0
        cfg.lanes_per_chain = int(lanes)
        cfg.stats_batch = int(stats_batch)
        cfg.trace_chains = int(trace_chains)
        io = _lib.InterleavedIO()
        io.k0.q = _ptr(state.q)
        io.k0.grad, io.k0.logp = _ptr(state.grad), _ptr(state.logp)   # carried gradient / log density
        io.k0.adapt, io.k0.rng, io.k0.accept_count = _ptr(state.adapt), _ptr(state.rng), _ptr(state.accept_count)
        self._eps0 = eps0_0 if torch.is_tensor(eps0_0) else self._dev_cached("eps0", eps0_0)
        self._eps1 = eps0_1 if torch.is_tensor(eps0_1) else self._dev_cached("eps1", eps0_1)
        io.k0.eps0 = _ptr(self._eps0)
        io.k0.trace, io.k0.trace_accept = _ptr(trace), _ptr(trace_accept0)
        io.k0.stats, io.k0.rec_accept_count, io.rec_accept_count1 = _ptr(stats), _ptr(rec_accept0), _ptr(rec_accept1)
        io.adapt1, io.accept_count1 = _ptr(state.adapt1), _ptr(state.accept_count1)
        io.eps0_1 = _ptr(self._eps1)
        io.trace_accept1 = _ptr(trace_accept1)
        with torch.cuda.device(self.device):
            _lib.check(self._L.arp_interleaved_run(self._h, C.byref(cfg), int(n_leapfrog_1), C.byref(io), _stream()))
        state.step += int(n_steps)
        return state


    # -- mean-field VI ----------------------------------------------------------
    def vi_run(self, lr, loc, rho, n_steps, n_mc, which=0, w=None, tied_b=False, wb=None, seed=0, a_prior=False,
               a_group=None, b_group=None, return_prior=False):
        """Run len(lr) independent Adam optimisations of the mean-field ELBO in one launch.

        loc, rho (and w, the unconstrained VIP parameter, when given) are [n_lr, D]
        device tensors updated in place; returns the ELBO timeline [n_lr, n_steps] (and, with return_prior,
        the per-step log prior of the learnable parameters).  a_group / b_group ([D] int leader indices) make
        untied parameterisation variables shared over a part (arp_vi_io.a_group)."""
        lr_t = self._dev(np.asarray(lr, np.float32))
        n_lr = lr_t.shape[0]
        assert loc.shape == (n_lr, self.D) and rho.shape == (n_lr, self.D)
        elbo = torch.empty(n_lr, int(n_steps), dtype=torch.float32, device=self.device)
        cfg = _lib.ViConfig()
        cfg.n_lr, cfg.n_steps, cfg.n_mc = n_lr, int(n_steps), int(n_mc)
        cfg.learn_a = 1 if w is not None else 0
        cfg.tied_b = 1 if tied_b else 0
        cfg.a_prior = 1 if a_prior else 0   # --discrete_prior on the learnable parameters
        cfg.seed = int(seed)
        io = _lib.ViIO()
        io.lr, io.loc, io.rho, io.w, io.elbo = _ptr(lr_t), _ptr(loc), _ptr(rho), _ptr(w), _ptr(elbo)
        io.wb = _ptr(wb)
        prior = torch.zeros(n_lr, int(n_steps), dtype=torch.float32, device=self.device) if return_prior else None
        io.prior = _ptr(prior)
        ag = torch.as_tensor(np.asarray(a_group, np.int32), device=self.device) if a_group is not None else None
        bg = torch.as_tensor(np.asarray(b_group, np.int32), device=self.device) if b_group is not None else None
        io.a_group, io.b_group = _ptr(ag), _ptr(bg)
        with torch.cuda.device(self.device):
            _lib.check(self._L.arp_vi_run(self._h, which, C.byref(cfg), C.byref(io), _stream()))
        return (elbo, prior) if return_prior else elbo


    def set_option(self, key, value):
        """Per-handle option (arp_model_set_option), e.g. ("german_math", "f32" | "bf16x3" | "auto")."""
        _lib.check(self._L.arp_model_set_option(self._h, key.encode(), value.encode()))

    def check(self):
        """Deferred status of this handle's asynchronous chain launches (arp_model_check): raises if a relay hand-over
        inside one of them timed out.  Call it after synchronising the stream the launches went to."""
        _lib.check(self._L.arp_model_check(self._h))

    def relay_geometry(self):
        """What this thread's last hmc_run / interleaved_run launch did (arp_relay_geometry): relay segments, chain blocks,
        workgroups of the kernel per CU."""
        out = (C.c_int32 * 3)()
        _lib.check(self._L.arp_relay_geometry(out))
        return dict(zip(("segments", "chain_blocks", "workgroups_per_cu"), [int(v) for v in out]))

    def vi_attempts(self):
        """Launches this thread's last vi_run needed (arp_vi_attempts): 1 unless a hand-off ran into its bound and the fit was retaken."""
        out = (C.c_int32 * 1)()
        _lib.check(self._L.arp_vi_attempts(out))
        return int(out[0])

    def vi_geometry(self):
        """Shape of this thread's last vi_run launch (arp_vi_geometry): threads per workgroup, sample groups G and row
        parts R per learning rate, learning rates per launch, workgroups resident together, workgroups one CU holds."""
        out = (C.c_int32 * 6)()
        _lib.check(self._L.arp_vi_geometry(out))
        keys = ("threads_per_workgroup", "sample_groups", "row_parts", "learning_rates_per_launch", "workgroups_resident",
                "workgroups_per_cu")
        return dict(zip(keys, [int(v) for v in out]))

    def adapt_probe(self, log_accept, adapt, kind, n_adapt, step_base=0, target=0.75, rate=0.05):
        """Test hook (arp_adapt_probe): the kernels' step-size recurrence on scripted log acceptance ratios
        `log_accept` [n_steps, n]; `adapt` [n, 4] is updated in place; returns the multipliers [n_steps, n]."""
        la = self._dev(log_accept)
        n_steps, n = la.shape
        out = torch.empty(n_steps, n, dtype=torch.float32, device=self.device)
        cfg = _lib.HmcConfig()
        cfg.n_steps, cfg.step_base = int(n_steps), int(step_base)
        cfg.adapt_kind, cfg.n_adapt = int(kind), int(n_adapt)
        cfg.adapt_target, cfg.adapt_rate = float(target), float(rate)
        with torch.cuda.device(self.device):
            _lib.check(self._L.arp_adapt_probe(C.byref(cfg), _ptr(la), n, _ptr(adapt), _ptr(out), _stream()))
        return out


def clock_probe(device, ms=10.0):
    """GHz the chip holds under a vector-bound load (arp_clock_probe: packed FMAs on every SIMD for ~`ms` milliseconds, one
    wave reading s_memtime / s_memrealtime around its loop).  A measurement hook for bench.py: every roofline figure assumes
    a clock, and the boxes differ in the one they hold."""
    dev = torch.device(device)
    out = torch.zeros(3, dtype=torch.int64, device=dev)
    iters = max(1, int(ms * 1e-3 / (64 * 2 * 4.43 / 2.4e9)))       # 64 packed FMAs per iteration, two waves per SIMD
    with torch.cuda.device(dev):
        _lib.check(_lib.lib().arp_clock_probe(iters, C.c_void_p(out.data_ptr()), _stream()))
        torch.cuda.synchronize(dev)
    cyc, ticks = int(out[0].item()), int(out[1].item())
    return (cyc / ticks) * 0.1 if ticks > 0 else None


def stats_summary(stats, n, batch):
    """(mean, var, ess) [C, D] float64 tensors from an `arp_hmc_io.stats` buffer after `n` recorded samples:
    mean = ref + s1/n, var = (s2 - s1^2/n)/(n-1), ESS by batch means, n var / (batch var(batch means)), capped at n."""
    ref, s1, s2, _, sb1, sb2 = (stats[k].to(torch.float64) for k in range(6))
    mean = ref + s1 / n
    var = ((s2 - s1 * s1 / n) / max(n - 1, 1)).clamp_min(0)
    nb = n // batch
    if nb < 2:
        raise ValueError("batch-means ESS needs at least two complete batches")
    vb = ((sb2 - sb1 * sb1 / nb) / (nb - 1)).clamp_min(0)
    ess = torch.minimum(n * var / (batch * vb), torch.full_like(var, float(n)))
    return mean, var, ess


# ---------------------------------------------------------------------------
# State converters with the reference's calling convention (models.py:56-128):
# callables on lists of [C, *event] arrays, evaluated on the device.
# ---------------------------------------------------------------------------
_engines = {}


def engine_for(spec, device=None):
    """One cached Engine per (model spec, device); device defaults to FLAGS.device."""
    from .flags import FLAGS
    dev = torch.device(device if device is not None else FLAGS.device)
    key = (id(spec), str(dev))
    if key not in _engines:
        _engines[key] = Engine(spec, dev)
    return _engines[key]


def _convert(spec, reparam, to_centered):
    def fn(state_parts):
        eng = engine_for(spec)
        eng.set_param(1, reparam)
        flat = spec.pack([np.asarray(p, np.float32).reshape(1, *np.shape(p)) if np.ndim(p) == len(s) else p
                          for p, s in zip(state_parts, spec.part_shapes)])
        single = all(np.ndim(p) == len(s) for p, s in zip(state_parts, spec.part_shapes))
        out = eng.transform(flat, which=1, to_centered=to_centered).cpu().numpy()
        parts = spec.unpack(out)
        return [p[0] for p in parts] if single else parts
    return fn


def build_make_to_centered(spec):
    def make_to_centered(**centering_kwargs):
        return _convert(spec, centering_kwargs, True)
    return make_to_centered


def build_make_to_partially_noncentered(spec):
    def make_to_partially_noncentered(**centering_kwargs):
        return _convert(spec, centering_kwargs, False)
    return make_to_partially_noncentered
