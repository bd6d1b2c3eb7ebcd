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
