"""Multi-GPU sharding of the chain axis (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

Chains never interact inside a transition and the reference adapts the step size
per chain, so ranks own contiguous blocks of chains, the RNG
is keyed by the global chain id, and the only communication is the end-of-run
exchange of per-chain statistics: one all-gather of the per-chain minimum ESS
(<= 4 bytes per chain) and one all-reduce of the acceptance counts.
"""
import numpy as np
import torch
import torch.distributed as dist


def _collective_device(device):
    """Where a tensor has to live for the initialised backend: the rank's GPU for RCCL ("nccl"), the host for gloo
    (the CPU tests of the rank logic)."""
    if dist.get_backend() == "gloo":
        return None
    if device is None:                       # RCCL moves device memory only: default to the rank's current GPU
        return torch.device("cuda", torch.cuda.current_device())
    return device


def live():
    """True when a process group exists: the exchange then runs through the backend's collectives -- also for a group of
    ONE rank (a single-GPU launch under torch.distributed.run), which is how the RCCL calls below are exercised on a
    one-GPU box (tests/test_gpu_scaling.py); without a group every function returns its local values."""
    return dist.is_available() and dist.is_initialized()


def world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def barrier():
    """All ranks of the job (no-op for a single process)."""
    if live():
        dist.barrier()


def shard_bounds(n, rank, world_size):
    """Contiguous block [lo, hi) of `n` chains owned by `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_states(parts, rank, world_size):
    """Slice a list of [C, *event] arrays along the chain axis; returns (local parts, chain_offset)."""
    n = parts[0].shape[0]
    lo, hi = shard_bounds(n, rank, world_size)
    return [p[lo:hi] for p in parts], lo


def all_gather_chains(local, n_total, device=None):
    """Concatenate per-chain values (first axis = local chains) from all ranks, in rank order.  `n_total` = chains of
    the whole job when every rank contributes its shard; None when the blocks have lengths only their owners know (the
    ESS chain subset of a streaming run: the chains with global id < --ess_chains) -- the lengths are gathered first."""
    rank, ws = world()
    t = torch.as_tensor(local)
    if not live():
        return t
    device = _collective_device(device)
    if device is not None:
        t = t.to(device)
    if n_total is None:
        mine = torch.tensor([t.shape[0]], dtype=torch.int64, device=t.device)
        got = [torch.zeros_like(mine) for _ in range(ws)]
        dist.all_gather(got, mine)
        sizes = [int(g.item()) for g in got]
    else:
        sizes = [shard_bounds(n_total, r, ws)[1] - shard_bounds(n_total, r, ws)[0] for r in range(ws)]
    mx = max(sizes)
    pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    pad[: t.shape[0]] = t
    out = [torch.empty_like(pad) for _ in range(ws)]
    dist.all_gather(out, pad)
    return torch.cat([o[:s] for o, s in zip(out, sizes)], dim=0)


def all_reduce_sum(value, device=None):
    rank, ws = world()
    t = torch.as_tensor(value, dtype=torch.float64)
    if not live():
        return t
    device = _collective_device(device)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return t


def min_ess_per_chain(ess_parts, num_chains_total, device=None):
    """Per chain the minimum over all elements of all parts (reference util.get_min_ess), gathered over the ranks.
    `num_chains_total` None: variable-length blocks (see all_gather_chains)."""
    parts = [np.nan_to_num(np.asarray(e)) for e in ess_parts]
    n = parts[0].shape[0]
    local_min = np.min(np.stack([p.reshape(n, -1).min(axis=1) if n else np.zeros(0, p.dtype) for p in parts]), axis=0)
    return all_gather_chains(torch.as_tensor(local_min, dtype=torch.float32), num_chains_total, device).cpu().numpy()


def summarize(normalized_ess_parts, is_accepted, num_samples, num_chains_total, device=None, ess_chains_total=-1):
    """ess_min, sem_min (reference util.get_min_ess) and acceptance rate in percent
    (main.py:372-373) over ALL chains, from each rank's local chains.  `ess_chains_total` = None when the ESS parts
    cover a chain subset whose per-rank sizes differ (streaming runs); default: the same chains as everything else."""
    mins = min_ess_per_chain(normalized_ess_parts, num_chains_total if ess_chains_total == -1 else ess_chains_total, device)
    acc = float(all_reduce_sum(float(np.sum(is_accepted)), device).item())
    ess_min = float(np.mean(mins))
    sem_min = float(np.std(mins) / np.sqrt(len(mins)))
    return ess_min, sem_min, acc * 100.0 / float(num_samples * num_chains_total), mins


def gather_parts(parts, num_chains_total, device=None):
    """Every rank's [C_local, *event] arrays -> the [C, *event] arrays of the whole job, in chain order (the per-element
    ESS that rank 0 writes to _ess.npz: <= 4 bytes per chain and element, once per run)."""
    if not live():
        return [np.asarray(p) for p in parts]
    return [all_gather_chains(torch.as_tensor(np.asarray(p)), num_chains_total, device).cpu().numpy() for p in parts]


def gather_leading_chains(samples, k, chain_offset, device=None):
    """Every rank's [S, C_local, *event] sample arrays -> [S, min(k, C), *event]: the traces of the JOB's first k chains
    (global id < k), which rank 0 writes to _traces.npz (--num_chains_to_save) -- the same chains however many ranks share
    the job, also when rank 0's own block is shorter than k.  A collective: every rank calls it (most contribute nothing)."""
    k = max(0, int(k))
    if not live():
        return [np.asarray(s[:, :k]) for s in samples]
    out = []
    for s in samples:
        kl = max(0, min(int(s.shape[1]), k - int(chain_offset)))
        mine = np.ascontiguousarray(np.moveaxis(np.asarray(s[:, :kl]), 1, 0))            # chain-major: [kl, S, *event]
        got = all_gather_chains(torch.as_tensor(mine), None, device).cpu().numpy()
        out.append(np.ascontiguousarray(np.moveaxis(got, 0, 1)))
    return out


def mean_sem(x):
    x = np.asarray(x, np.float64)
    return float(np.mean(x)), float(np.std(x) / np.sqrt(len(x)))
