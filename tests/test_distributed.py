"""CPU, world_size 2 over gloo: chain sharding and the end-of-run statistics
exchange (the only collective of the path) give the single-process answer."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from autoreparam_amd import parallel, util


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, ws, port, C, S, L, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    rs = np.random.RandomState(0)
    ess = [rs.rand(C) * 50, rs.rand(C, 5) * 50]           # parts [C] and [C, 5]
    ess[1][3, 2] = np.nan                                  # nan_to_num path
    acc = rs.rand(S, C) < 0.8
    lo, hi = parallel.shard_bounds(C, rank, ws)
    parts, off = parallel.shard_states(ess, rank, ws)
    assert off == lo and parts[0].shape[0] == hi - lo
    e, s, a, mins = parallel.summarize(parts, acc[:, lo:hi], S, C)
    g = parallel.all_gather_chains(torch.arange(lo, hi), C)
    assert torch.equal(g, torch.arange(C))
    np.save(os.path.join(out_dir, "r%d.npy" % rank), np.array([e, s, a]))
    dist.destroy_process_group()


def test_shard_bounds_cover_everything():
    for n in (1, 7, 64, 65537):
        for ws in (1, 2, 3, 8):
            b = [parallel.shard_bounds(n, r, ws) for r in range(ws)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(ws - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1


def test_two_rank_statistics_match_single_process(tmp_path):
    C, S, L = 7, 20, 4                                     # uneven split: 4 + 3 chains
    port = _free_port()
    mp.spawn(_worker, args=(2, port, C, S, L, str(tmp_path)), nprocs=2, join=True)
    rs = np.random.RandomState(0)
    ess = [rs.rand(C) * 50, rs.rand(C, 5) * 50]
    ess[1][3, 2] = np.nan
    acc = rs.rand(S, C) < 0.8
    e_ref, s_ref = util.get_min_ess(ess)
    a_ref = np.sum(acc) * 100.0 / (S * C)
    for r in range(2):
        e, s, a = np.load(os.path.join(str(tmp_path), "r%d.npy" % r))
        assert abs(e - e_ref) < 1e-5 and abs(s - s_ref) < 1e-5 and abs(a - a_ref) < 1e-9


# ---------------------------------------------------------------------------
# The CLI's own rank logic (autoreparam_amd/main.py) on two gloo ranks, with the engine calls replaced by
# deterministic stand-ins keyed by the GLOBAL chain id: what is under test is everything main.py does around them --
# VI on rank 0 only (the other ranks leave without a process group of their own), shard_states + chain_offset,
# summarize over the real result objects, rank 0 writing the JSON / ESS files -- against the one-process run.
# ---------------------------------------------------------------------------
def _fake_engine(monkeypatch_target):
    import collections
    from autoreparam_amd import inference

    def fake_vi(elbo, vp, learnable_parameters_prior=None, learnable_parameters=None, flags=None):
        spec = elbo.target.spec
        lv = collections.OrderedDict()
        for k, name in enumerate(spec.part_names):
            lv[name + "_loc"] = np.full(spec.part_shapes[k], 0.1 * (k + 1), np.float32)
            lv[name + "_scale"] = np.full(spec.part_shapes[k], 0.5, np.float32)
        return (np.float64(-12.5), [-13.0] * 40, 0.1, util.get_approximate_step_size(lv, 1), lv, None)

    def _ess_parts(spec, g):
        return [(10.0 + (g % 7)[(slice(None),) + (None,) * len(sh)] + k + np.zeros((len(g),) + tuple(sh))).astype(np.float32)
                for k, sh in enumerate(spec.part_shapes)]

    def fake_hmc(target, model_config, step_size_init, initial_states, reparam, flags=None, chain_offset=0):
        spec = target.spec
        Cl, S = initial_states[0].shape[0], int(flags.num_samples)
        g = chain_offset + np.arange(Cl)
        acc = ((np.arange(S)[:, None] + g[None, :]) % 3 != 0)
        samples = [np.broadcast_to(g.reshape((1, Cl) + (1,) * len(sh)).astype(np.float32), (S, Cl) + tuple(sh)).copy()
                   for sh in spec.part_shapes]
        kr = inference.KernelResults(inference.HmcInnerResults(acc), np.ones(Cl, np.float32), S,
                                     inference.EssInfo("autocorrelation", Cl))
        ess = _ess_parts(spec, g)
        if getattr(flags, "trace_chunk_rows", None):
            # a streaming run: the autocorrelation ESS covers the chains with GLOBAL id < --ess_chains (this rank's block
            # of them may be empty), the batch-means ESS every local chain
            import torch
            k = int(min(Cl, max(int(flags.ess_chains) - chain_offset, 0)))
            ess = [e[:k] for e in ess]
            samples = [x[:, :max(k, int(flags.num_chains_to_save))] for x in samples]
            bm = torch.as_tensor(np.concatenate([e.reshape(Cl, int(np.prod(e.shape[1:]))) for e in _ess_parts(spec, g)], axis=1) * 0.5)
            kr = kr._replace(ess_info=inference.EssInfo("autocorrelation", k, batch_means=bm, batch=4))
        return None, kr, samples, ess

    def fake_inter(model_config, target_cp, target_ncp, num_leapfrog_steps_cp, num_leapfrog_steps_ncp, step_size_cp,
                   step_size_ncp, initial_states_cp, flags=None, chain_offset=0):
        _, kr, samples, ess = fake_hmc(target_cp, model_config, step_size_cp, initial_states_cp, None, flags, chain_offset)
        acc1 = ~np.asarray(kr.inner_results.is_accepted)
        return samples, inference.InterleavedKernelResults(kr._replace(ess_info=None), inference.KernelResults(
            inference.HmcInnerResults(acc1), kr.new_step_size, kr.step), ess_info=kr.ess_info), ess

    inference.find_best_learning_rate = fake_vi
    inference.hmc = fake_hmc
    inference.hmc_interleaved = fake_inter


def _cli_sequence(results_dir, chains=7, ess_small=3, ess_mid=5):
    from autoreparam_amd import main as cli
    from autoreparam_amd.flags import FLAGS
    base = ["--model=8schools", "--results_dir=%s" % results_dir, "--seed=3", "--num_chains=%d" % chains, "--num_samples=12",
            "--num_burnin_steps=4", "--num_adaptation_steps=3", "--num_chains_to_save=2"]
    out = {}

    def phase(args):
        # on the command line every phase is its own launch: it has ended -- rank 0 has written its files -- before the next
        # one starts and reads them (the tuning runs decide the leapfrog count of the HMC run on EVERY rank)
        r = cli.main(base + args, flags=FLAGS.copy())
        if dist.is_initialized():
            dist.barrier()
        return r
    for m in ("CP", "NCP"):
        out["vi_" + m] = phase(["--inference=VI", "--method=" + m])
        for L in (2, 4):
            phase(["--inference=HMCtuning", "--method=" + m, "--num_leapfrog_steps=%d" % L])
    out["hmc"] = phase(["--inference=HMC", "--method=CP"])
    out["inter"] = phase(["--inference=HMC", "--method=i"])
    # streaming runs: the ESS chain subset lies on rank 0 alone (3 of 7 chains) / on the first two ranks (5 = 4 + 1); at
    # world size 8 every other rank owns NONE of the subset (inference._ess_subset -> k_local = 0: empty blocks in the
    # variable-length gathers)
    out["hmc_s3"] = phase(["--inference=HMC", "--method=CP", "--trace_chunk_rows=8", "--ess_chains=%d" % ess_small])
    out["hmc_s5"] = phase(["--inference=HMC", "--method=NCP", "--trace_chunk_rows=8", "--ess_chains=%d" % ess_mid])
    out["inter_s5"] = phase(["--inference=HMC", "--method=i", "--trace_chunk_rows=8", "--ess_chains=%d" % ess_mid])
    return out


def _cli_worker(rank, ws, port, results_dir, chains=7, ess_small=3, ess_mid=5):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WORLD_SIZE=str(ws), RANK=str(rank), LOCAL_RANK=str(rank))
    _fake_engine(None)
    # VI first, BEFORE any process group exists: ranks != 0 must leave on RANK alone, rank 0 runs without one
    from autoreparam_amd import main as cli
    from autoreparam_amd.flags import FLAGS
    r = cli.main(["--model=8schools", "--results_dir=%s" % results_dir, "--inference=VI", "--method=CP"], flags=FLAGS.copy())
    assert (r is None) == (rank != 0) and not dist.is_initialized()
    dist.init_process_group("gloo", rank=rank, world_size=ws)     # main() keeps an initialised group (RCCL on the GPU box)
    dist.barrier()
    out = _cli_sequence(results_dir, chains, ess_small, ess_mid)
    assert (out["vi_NCP"] is None) == (rank != 0)
    np.save(os.path.join(results_dir, "ret%d.npy" % rank), np.array([out["hmc"][0], out["hmc"][1], out["hmc"][2],
                                                                      out["inter"][0], out["inter"][2], out["inter"][3],
                                                                      out["hmc_s3"][0], out["hmc_s5"][0], out["hmc_s5"][1],
                                                                      out["inter_s5"][0]], np.float64))
    dist.barrier()
    dist.destroy_process_group()


def _results(d):
    import json
    out = {}
    for f in sorted(os.listdir(d)):
        if f.endswith(".json"):
            r = json.load(open(os.path.join(d, f)))
            for k in ("mcmc_time_sec", "variational_fit_time_secs"):
                r.pop(k, None)
            for t in r.get("tuning_runs", []):
                t.pop("mcmc_time", None)
            out[f] = r
        elif f.endswith("_ess.npz") or f.endswith("_traces.npz"):
            z = np.load(os.path.join(d, f))
            out[f] = {k: z[k] for k in z.files}
    return out


import pytest  # noqa: E402


@pytest.mark.parametrize("ws,chains,ess_small,ess_mid", [
    (2, 7, 3, 5),            # 4 + 3 chains
    (8, 1003, 3, 130),       # what the round-end 8-GPU run looks like: shards of 126 and 125 chains; the ESS subset on rank 0
                             # alone / on ranks 0 and 1 (126 + 4), six or seven ranks contributing empty blocks
    (8, 5, 3, 5),            # fewer chains than ranks: ranks 5 - 7 own no chain at all
])
def test_cli_rank_logic_n_ranks_equal_one(tmp_path, ws, chains, ess_small, ess_mid):
    one, two = str(tmp_path / "one"), str(tmp_path / "two")
    os.makedirs(one); os.makedirs(two)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        os.environ.pop(k, None)
    import importlib
    from autoreparam_amd import inference
    saved = (inference.find_best_learning_rate, inference.hmc, inference.hmc_interleaved)
    try:
        _fake_engine(None)
        ref = _cli_sequence(one, chains, ess_small, ess_mid)
    finally:
        inference.find_best_learning_rate, inference.hmc, inference.hmc_interleaved = saved
    mp.spawn(_cli_worker, args=(ws, _free_port(), two, chains, ess_small, ess_mid), nprocs=ws, join=True)
    a, b = _results(one), _results(two)
    assert sorted(a) == sorted(b) and any(k.endswith("_ess.npz") for k in a)
    for k in a:
        for kk in a[k]:
            va, vb = a[k][kk], b[k][kk]
            if isinstance(va, np.ndarray):
                assert np.array_equal(va, vb), (k, kk)
            else:
                assert va == vb or np.allclose(va, vb, rtol=1e-6), (k, kk, va, vb)
    want = np.array([ref["hmc"][0], ref["hmc"][1], ref["hmc"][2], ref["inter"][0], ref["inter"][2], ref["inter"][3],
                     ref["hmc_s3"][0], ref["hmc_s5"][0], ref["hmc_s5"][1], ref["inter_s5"][0]])
    # the streaming runs' extra keys: estimator, subset size, the batch-means figure of all 7 chains
    import json
    r = json.load(open(os.path.join(two, "CP_tied.json")))
    assert r["ess_estimator"] == ["autocorrelation", "autocorrelation"] and r["ess_chains"] == [chains, ess_small]
    # one entry per run in every list: None for the whole-trace run, the figure for the streaming one
    assert r["ess_min_batch_means"][0] is None and r["ess_min_batch_means"][1] > 0 and r["batch_means_batch"] == [None, 4]
    assert len(r["ess_min"]) == len(r["ess_estimator"]) == len(r["sem_min_batch_means"]) == 2
    assert a["CP_tied_ess.npz"]["theta"].shape == (ess_small, 8)             # the last run's: the 3-chain subset
    assert a["NCP_tied_ess.npz"]["theta"].shape == (ess_mid, 8) and np.array_equal(a["NCP_tied_ess.npz"]["theta"], b["NCP_tied_ess.npz"]["theta"])
    for r in range(ws):   # every rank returns the statistics over ALL chains
        np.testing.assert_allclose(np.load(os.path.join(two, "ret%d.npy" % r)), want, rtol=1e-6)
