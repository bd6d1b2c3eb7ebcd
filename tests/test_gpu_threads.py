"""GPU: the boundary's threading contract (include/autoreparam.h, SURVEY.md 8b "Threading"): one handle per thread, several
threads of a process driving the same device concurrently, each on a stream of its own -- every result is bit for bit the
one the thread gets alone.  ctypes releases the GIL for the duration of a call, so the calls really overlap."""
import threading

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


def _vi(eng, gpu, sp, seed, learn):
    rs = np.random.RandomState(0)
    lrs = [0.02, 0.05, 0.1]
    loc = torch.as_tensor((1e-2 * rs.randn(3, sp.D)).astype(np.float32), device=gpu); rho = torch.full((3, sp.D), -2.0, device=gpu)
    w = torch.zeros(3, sp.D, device=gpu) if learn else None
    e = eng.vi_run(lrs, loc, rho, 120, 256, w=w, seed=seed)
    return [t.cpu().numpy() for t in (e, loc, rho)]


def _hmc(eng, gpu, sp, seed):
    from autoreparam_amd import engine, _lib
    st = engine.ChainState(torch.as_tensor(helpers.states(sp, 3001, seed=seed, scale=0.05), device=gpu))
    tr = torch.zeros(24, 3001, sp.D, device=gpu)      # 3 launches x 16 transitions, every second one recorded
    eps = np.full(sp.D, 2e-3, np.float32)
    for _ in range(3):
        eng.hmc_run(st, eps, 4, 16, seed=seed, adapt_kind=_lib.ADAPT_DUAL, n_adapt=40, thin=2, trace=tr)
    return [st.q.cpu().numpy(), st.accept_count.cpu().numpy(), tr.cpu().numpy()]


def _job(kind, mname, gpu, seed):
    from autoreparam_amd import engine
    sp = helpers.spec(mname)
    eng = engine.Engine(sp, gpu)                  # a handle of this thread's own
    eng.set_param(0, (np.full(sp.D, 0.5, np.float32), np.ones(sp.D, np.float32)) if kind == "cvip" else "NCP")
    out = []
    for rep in range(3):
        out.append(_hmc(eng, gpu, sp, seed + rep) if kind == "hmc" else _vi(eng, gpu, sp, seed + rep, kind == "cvip"))
    return out


JOBS = [("vi", "german", 11), ("cvip", "election", 12), ("hmc", "radon_PA", 13), ("vi", "radon_MN", 14), ("hmc", "german", 15),
        ("cvip", "german", 16)]


def test_handles_driven_from_concurrent_threads(gpu):
    alone = [_job(k, m, gpu, s) for k, m, s in JOBS]
    torch.cuda.synchronize()
    got, errs = [None] * len(JOBS), []

    def work(i):
        try:
            with torch.cuda.stream(torch.cuda.Stream(device=gpu)):
                got[i] = _job(*JOBS[i][:2], gpu, JOBS[i][2])
                torch.cuda.current_stream().synchronize()
        except Exception as e:                    # noqa: BLE001 -- reported below, in the main thread
            errs.append((JOBS[i], repr(e)))

    for _round in range(2):
        ts = [threading.Thread(target=work, args=(i,)) for i in range(len(JOBS))]
        for t in ts: t.start()
        for t in ts: t.join(timeout=300)
        assert not any(t.is_alive() for t in ts), "a thread hangs"
        assert not errs, errs
        for job, a, b in zip(JOBS, alone, got):
            for ra, rb in zip(a, b):
                for x, y in zip(ra, rb):
                    assert np.array_equal(x, y, equal_nan=True), job


@pytest.mark.parametrize("C", [65536, 70001])
def test_one_handle_two_streams_overlapping_launches(gpu, C):
    """One handle, launches enqueued back to back on TWO streams for two different chain populations (calls serialised by the
    caller, the work overlaps on the device): the relay segments of a launch use flag words and a ticket counter of their own
    (a stream-ordered allocation per launch), so each population's run is bit for bit the one it gets alone.  A workgroup's
    (segment, block) comes from the ticket it draws when it starts, so the block it waits for was started before it whatever
    the dispatcher interleaves from the other launch -- 70 001 chains: a block count that is no multiple of the XCD count,
    the case round 5's blockIdx-ordered form could deadlock on."""
    from autoreparam_amd import engine, _lib
    sp = helpers.spec("radon_PA")
    eng = engine.Engine(sp, gpu)
    eng.set_param(0, "CP"); eng.set_param(1, "NCP")
    e = np.full(sp.D, 0.05, np.float32); e[2] = 0.012
    T = 384
    q0 = [helpers.states(sp, C, seed=s, scale=0.1) for s in (1, 2)]

    def go(st, seed):
        eng.interleaved_run(st, e, e, 3, 3, T, seed=seed, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=500)

    alone = []
    for k in range(2):
        st = engine.ChainState(torch.as_tensor(q0[k], device=gpu))
        for _ in range(3): go(st, 40 + k)
        torch.cuda.synchronize()
        alone.append([t.cpu().numpy() for t in (st.q, st.grad, st.rng, st.accept_count, st.accept_count1)])
    streams = [torch.cuda.Stream(device=gpu) for _ in range(2)]
    sts = [engine.ChainState(torch.as_tensor(q0[k], device=gpu)) for k in range(2)]
    torch.cuda.synchronize()
    for _ in range(3):
        for k in range(2):
            with torch.cuda.stream(streams[k]):
                go(sts[k], 40 + k)
    torch.cuda.synchronize()
    eng.check()
    for k in range(2):
        got = [t.cpu().numpy() for t in (sts[k].q, sts[k].grad, sts[k].rng, sts[k].accept_count, sts[k].accept_count1)]
        for x, y in zip(alone[k], got):
            assert np.array_equal(x, y)


def test_relay_hand_over_time_out_is_an_error_code_not_a_trap(gpu, monkeypatch):
    """kernels.h: relay_begin bounds its wait.  With segments that never raise their flag (ARP_RELAY_FAULT, a test hook) and a
    20 ms time-out, the waiting workgroups mark the launch failed and leave; the process and its GPU context live on, the
    failure is reported ONCE -- by arp_model_check, or by the next chain launch on the handle -- with a message, and the
    handle then works as before (round 5 ended such a wait with __builtin_trap after a minute)."""
    import time
    from autoreparam_amd import engine, _lib
    sp = helpers.spec("radon_PA")
    eng = engine.Engine(sp, gpu)
    eng.set_param(0, "CP"); eng.set_param(1, "NCP")
    e = np.full(sp.D, 0.05, np.float32); e[2] = 0.012
    C, T = 8192, 256
    q0 = helpers.states(sp, C, seed=5, scale=0.1)

    def go(st):
        eng.interleaved_run(st, e, e, 3, 3, T, seed=3, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=500)

    ref = engine.ChainState(torch.as_tensor(q0, device=gpu))
    go(ref); torch.cuda.synchronize(); eng.check()
    monkeypatch.setenv("ARP_DEBUG", "1")
    monkeypatch.setenv("ARP_SEGMENTS", "4")
    for how in ("check", "next_launch"):
        monkeypatch.setenv("ARP_RELAY_FAULT", "1")
        monkeypatch.setenv("ARP_RELAY_TIMEOUT_MS", "20")
        st = engine.ChainState(torch.as_tensor(q0, device=gpu))
        t0 = time.time()
        go(st)
        torch.cuda.synchronize()
        assert time.time() - t0 < 10.0                      # every waiter left at the first time-out, not one after another
        assert eng.relay_geometry()["segments"] == 4
        monkeypatch.delenv("ARP_RELAY_FAULT"); monkeypatch.delenv("ARP_RELAY_TIMEOUT_MS")
        st2 = engine.ChainState(torch.as_tensor(q0, device=gpu))
        if how == "check":
            with pytest.raises(RuntimeError, match="relay hand-over"):
                eng.check()
        else:
            with pytest.raises(RuntimeError, match="relay hand-over"):
                go(st2)
            assert st2.step == 0 or np.array_equal(st2.q.cpu().numpy(), q0)     # the refused call launched nothing
            st2 = engine.ChainState(torch.as_tensor(q0, device=gpu))
        eng.check()                                         # reported once
        go(st2); torch.cuda.synchronize(); eng.check()      # the handle and the context are intact: 4 segments, same bits
        assert np.array_equal(st2.q.cpu().numpy(), ref.q.cpu().numpy())
