"""Helpers around the hot path: ESS, step-size and initial-state bookkeeping
(reference util.py:271-276, 394-410, 445-460 and tfp.mcmc.effective_sample_size
as called at inference.py:240, 327)."""
import collections

import numpy as np
import torch


def print_(*args):
    print(*args, flush=True)


def debug_switch(name):
    """Value of a test / experiment environment switch (ARP_SHARE_GPU, ARP_DIST_BACKEND, ARP_LIB_PATH, ...), honoured
    only under ARP_DEBUG=1 and announced on stderr every time it takes effect: a stray variable in a production
    launch must not silently put two ranks on one GPU or move the collectives to the host."""
    import os
    import sys
    v = os.environ.get(name)
    if not v:
        return None
    if os.environ.get("ARP_DEBUG") != "1":
        print("autoreparam_amd: %s=%s IGNORED (test/experiment switch; set ARP_DEBUG=1 to enable it)" % (name, v),
              file=sys.stderr, flush=True)
        return None
    print("autoreparam_amd: DEBUG SWITCH %s=%s is in effect (ARP_DEBUG=1) -- not a production configuration" % (name, v),
          file=sys.stderr, flush=True)
    return v


def get_approximate_step_size(variational_parameters, num_leapfrog_steps):
    """reference util.py:271-276: the variational scales divided by L^2."""
    return [np.asarray(variational_parameters[key]) / num_leapfrog_steps ** 2
            for key in variational_parameters.keys() if key.endswith("_scale")]


def variational_inits_from_params(learned_variational_params, param_names, num_inits, seed=None):
    """Sample initial states from the fitted mean-field Normal (reference util.py:394-410;
    the reference uses the unseeded global numpy RNG, here a seed may be given)."""
    rs = np.random.RandomState(seed) if seed is not None else np.random
    locs, stddevs, samples = collections.OrderedDict(), collections.OrderedDict(), collections.OrderedDict()
    for k, v in learned_variational_params.items():
        if k.endswith("_loc"):
            locs[k[:-4]] = v
        elif k.endswith("_scale"):
            stddevs[k[:-6]] = v
    for k in param_names:
        shape = (num_inits,) + np.asarray(locs[k]).shape
        samples[k] = (rs.randn(*shape) * stddevs[k] + locs[k]).astype(np.float32)
    return samples


def effective_sample_size(states, max_chains_per_batch=None):
    """tfp.mcmc.effective_sample_size with its defaults (filter_threshold=0), per chain
    and element: `states` [S, C, D] (torch) -> [C, D].  A float32 trace on the GPU goes to the
    engine's own kernel (`arp_ess`: direct auto-covariances up to the first negative one);
    anything else (the CPU tests) to the FFT form below."""
    if states.is_cuda and states.dtype == torch.float32:
        import ctypes as C
        from . import _lib
        S, Cn, D = states.shape
        # a leading block of chains of a wider trace ([S, :k, D] of [S, K, D]) is taken in place: rows stay rows, the
        # row stride says how far apart they are
        in_place = states.is_contiguous() or (Cn > 0 and states.stride(2) == 1 and states.stride(1) == D and S > 1)
        x = states if in_place else states.contiguous()
        row_stride = x.stride(0) if S > 1 else Cn * D
        out = torch.empty(Cn, D, dtype=torch.float32, device=x.device)
        L = _lib.lib()
        with torch.cuda.device(x.device):
            # long traces (a streaming run's kept chains at the reference's 50 000 samples): slowly mixing series are
            # finished on the matrix cores out of a workspace -- every series at once when that fits in a third of the
            # free memory, in chunks otherwise
            need = int(L.arp_ess_workspace_bytes(S, Cn * D))
            ws = None
            if need > 0:
                free, _ = torch.cuda.mem_get_info(x.device)
                lists = 28 * Cn * D + 4096
                ws = torch.empty(max(min(need, int(free // 3)), lists + (64 * 4 + 256) * (S + 64)), dtype=torch.uint8,
                                 device=x.device)
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
            _lib.check(L.arp_ess_ws(C.c_void_p(x.data_ptr()), S, Cn * D, row_stride, C.c_void_p(out.data_ptr()),
                                    C.c_void_p(ws.data_ptr() if ws is not None else 0), ws.numel() if ws is not None else 0,
                                    C.c_void_p(torch.cuda.current_stream().cuda_stream)))
            ev[1].record()
            del ws
        # bench.py reads the kernel time of the LAST call off these events (after the caller has synchronised anyway)
        effective_sample_size.last_events = ev + (4.0 * S * Cn * D,)
        return out
    return effective_sample_size_fft(states, max_chains_per_batch)


def effective_sample_size_fft(states, max_chains_per_batch=None):
    """The same statistic by FFT: `states` [S, C, D] (torch, any device) -> [C, D].

    Restated from the published definition: auto-correlation by FFT of the
    mean-removed series, lag k divided by (S - k) and normalised by lag 0; every
    lag from the first negative one on is dropped; ESS = S / (-1 + 2 sum_k (S-k)/S rho_k).
    Runs on the tensor's device (rocFFT through torch.fft), in chain batches so the
    complex work buffers stay bounded.
    """
    S, C, D = states.shape
    out = torch.empty(C, D, dtype=torch.float32, device=states.device)
    n_fft = 1 << int(np.ceil(np.log2(2 * S)))
    if max_chains_per_batch is None:
        max_chains_per_batch = max(1, int(2 ** 27 // (n_fft * D)))   # ~1 GiB of complex64 per batch
    k = torch.arange(S, device=states.device, dtype=torch.float32)
    for c0 in range(0, C, max_chains_per_batch):
        x = states[:, c0:c0 + max_chains_per_batch, :].to(torch.float32)
        x = x - x.mean(dim=0, keepdim=True)
        f = torch.fft.rfft(x, n=n_fft, dim=0)
        ac = torch.fft.irfft(f * f.conj(), n=n_fft, dim=0)[:S]
        ac = ac / (S - k).view(-1, 1, 1)
        ac = ac / ac[:1]
        mask = (ac < 0).to(torch.float32).cumsum(dim=0)
        ac = ac * torch.clamp(1.0 - mask, min=0.0)
        nk = ((S - k) / S).view(-1, 1, 1)
        out[c0:c0 + max_chains_per_batch] = S / (-1.0 + 2.0 * (nk * ac).sum(dim=0))
    return out


def get_min_ess(ess):
    """reference util.py:445-460: per chain the minimum over all elements of all parts,
    then mean and standard error over chains.  `ess` is a list of [C, *event] arrays."""
    ess = [np.nan_to_num(np.asarray(e)) for e in ess]
    num_chains = ess[0].shape[0]
    # (the reference loops over chains in Python; the same minima, one vector pass per part -- 0.5 s at 65 536 chains otherwise)
    if num_chains == 0:
        return np.float32("nan"), np.float32("nan")
    min_ess = np.minimum.reduce([e.reshape(num_chains, int(np.prod(e.shape[1:], dtype=np.int64))).min(axis=1) for e in ess])
    mean_ess = np.mean(min_ess)
    sem_ess = np.std(min_ess) / np.sqrt(len(min_ess))
    return mean_ess, sem_ess


def stddvs_to_mcmc_step_sizes(results, num_leapfrog_steps):
    """reference util.py:308-313: sqrt(2 mean(scale)) / L per `*_scale` entry of a stored fit."""
    L = float(num_leapfrog_steps)
    return [np.sqrt(2 * np.mean(results[k])) / L for k in results.keys() if k.endswith("_scale")]


def reject_outliers(data, m=1.5):
    """reference util.py:418-423: the entries within m standard deviations of the mean (all of them if none is)."""
    data = np.asarray(data)
    kept = data[np.abs(data - np.mean(data)) < m * np.std(data)]
    return kept if kept.size else data


def get_min_ess_other(ess, num_chains=None):
    """reference util.py:426-442: `ess` indexed [chain][part]; per chain the minimum over parts and elements, outliers
    dropped (reject_outliers), then mean and standard error."""
    num_chains = len(ess) if num_chains is None else num_chains
    mins = np.array([min(np.nan_to_num(np.asarray(e)).min() for e in ess[c]) for c in range(num_chains)])
    mins = reject_outliers(mins)
    return np.mean(mins), np.std(mins) / np.sqrt(len(mins))


def estimate_true_mean(sample_groups, esss):
    """reference util.py:316-331: per group, each variable's sample mean weighted by the group's share of the total ESS."""
    total = float(sum(esss))
    return [[0 + w * np.mean(v) / total for v in group] for group, w in zip(sample_groups, esss)]


def compute_V_cp(q, v):
    """reference util.py:65-67: posterior covariance of the two-variable Gaussian hierarchy, centred coordinates."""
    return np.array([[1.0 + v, 1.0], [1.0, q * v + 1.0]]) / (v * q + q + 1.0)


def compute_V_ncp(q, v):
    """reference util.py:70-72: the same covariance in non-centred coordinates."""
    off = -np.sqrt(v) * q
    return np.array([[q + 1, off], [off, v * q + 1]]) * (1 / (v * q + q + 1))


def _condition_number(q, v, shared):
    t = v * q + 1
    root = 2 * np.sqrt(t * t - shared * (v * q + q + 1) * t)
    return (2 * t + root) / (2 * t - root)


def condition_number_cp(q, v):
    """reference util.py:75-80: ratio of the eigenvalues of the centred posterior precision."""
    return _condition_number(q, v, v / (v + 1))


def condition_number_ncp(q, v):
    """reference util.py:83-88: the same for the non-centred form."""
    return _condition_number(q, v, 1 / (q + 1))
