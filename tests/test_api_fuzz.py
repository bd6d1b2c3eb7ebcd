"""CPU: argument validation of every arp_* entry point, driven without a GPU.

Under ARP_DEBUG=1 ARP_HOST_ONLY=1 (a test hook of csrc/arp_api.hip that announces itself) arp_model_create builds a
handle whose tables live in host memory, so the host side of the library -- sufficient statistics, lane selection,
argument checks, work-list and workspace sizing -- runs here for every model.  Nothing can be computed with such a
handle: a call that gets past validation fails with HIP's own "no device" error.  The property held over a few thousand
randomised calls: NEVER a crash, and every call either returns 0 for a documented no-op (n_steps == 0) or returns
non-zero with a message in arp_last_error()."""
import ctypes as C
import os
import random

import numpy as np
import pytest

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def host_lib():
    import torch
    if torch.cuda.is_available():
        # with a device present a call that passes validation would LAUNCH on the stand-in pointers
        pytest.skip("argument fuzzing with host-only handles runs on GPU-less machines only")
    import __graft_entry__ as ge
    ge.build()
    from autoreparam_amd import _lib
    old = {k: os.environ.get(k) for k in ("ARP_DEBUG", "ARP_HOST_ONLY")}
    os.environ["ARP_DEBUG"] = "1"
    os.environ["ARP_HOST_ONLY"] = "1"
    yield _lib
    for k, v in old.items():
        if v is None:
            os.environ.pop(k, None)
        else:
            os.environ[k] = v


def _handle(_lib, name):
    sp = helpers.spec(name)
    ds, keep = sp.dataset()
    h = C.c_void_p(0)
    rc = _lib.lib().arp_model_create(C.byref(ds), C.byref(h))
    assert rc == 0, _lib.lib().arp_last_error()
    return sp, h, keep


MODELS = ["8schools", "radon_MN", "radon_PA", "election", "german", "radon_sd_MN", "funnel", "electric", "time_series"]


@pytest.mark.parametrize("name", MODELS)
def test_host_only_handles_build_and_validate(host_lib, name):
    L = host_lib.lib()
    sp, h, keep = _handle(host_lib, name)
    assert L.arp_model_dim(h) == sp.D
    assert np.isfinite(L.arp_model_logp_const(h, 0))
    a = np.full(sp.D, 0.5, np.float32)
    f32p = host_lib._f32p
    assert L.arp_model_set_param(h, 0, a.ctypes.data_as(f32p), a.ctypes.data_as(f32p)) == 0
    assert L.arp_model_set_param(h, 2, a.ctypes.data_as(f32p), a.ctypes.data_as(f32p)) != 0
    assert L.arp_model_set_param(h, 0, None, a.ctypes.data_as(f32p)) != 0
    # options: german_math on german credit only (the reference's data have 7 columns that need three bf16 pieces: every
    # value is accepted); unknown keys, values, NULLs and other models fail with a message
    if name == "german":
        for v in (b"f32", b"bf16x3", b"auto"):
            assert L.arp_model_set_option(h, b"german_math", v) == 0
        assert L.arp_model_set_option(h, b"german_math", b"fp8") != 0 and b"german_math" in L.arp_last_error()
    else:
        assert L.arp_model_set_option(h, b"german_math", b"f32") != 0
    assert L.arp_model_set_option(h, b"no_such_key", b"1") != 0 and len(L.arp_last_error()) > 0
    assert L.arp_model_set_option(h, None, b"1") != 0 and L.arp_model_set_option(None, b"german_math", b"f32") != 0
    assert L.arp_model_destroy(h) == 0


def _fake(n_bytes):
    """An address range of our own to hand over as a `device` pointer (never dereferenced: no launch succeeds)."""
    buf = np.zeros(max(8, n_bytes), np.uint8)
    return buf, C.c_void_p(buf.ctypes.data)


def test_randomised_calls_never_crash_and_always_explain(host_lib):
    L = host_lib.lib()
    rnd = random.Random(20261002)
    handles = {n: _handle(host_lib, n) for n in MODELS}
    calls = failures = 0
    keepalive = []

    def ptr(ok_p=0.8, n=1 << 16):
        if rnd.random() < ok_p:
            b, p = _fake(n)
            keepalive.append(b)
            return p
        return C.c_void_p(0)

    def err():
        return L.arp_last_error()

    for it in range(3000):
        name = rnd.choice(MODELS)
        sp, h, _ = handles[name]
        hh = h if rnd.random() < 0.9 else C.c_void_p(0)
        kind = rnd.choice(["logp", "transform", "hmc", "inter", "vi", "ess", "adapt", "clock"])
        del keepalive[:]
        L.arp_model_set_param(h, 0, np.ones(sp.D, np.float32).ctypes.data_as(host_lib._f32p),
                              np.ones(sp.D, np.float32).ctypes.data_as(host_lib._f32p))
        if kind == "logp":
            rc = L.arp_logp_grad(hh, rnd.choice([0, 1, -1, 2]), ptr(), rnd.choice([0, -5, 1, 7, 4096]), ptr(), ptr(),
                                 rnd.choice([0, 1, 2, 3, 4, 8, 16, 32, -1]), None)
        elif kind == "transform":
            rc = L.arp_transform(hh, rnd.choice([0, 1, 5]), rnd.choice([0, 1, 2, -1]), ptr(), rnd.choice([0, 1, 100]), ptr(), None)
        elif kind in ("hmc", "inter"):
            cfg = host_lib.HmcConfig()
            cfg.n_chains = rnd.choice([0, -1, 1, 64, 70, 4096])
            cfg.n_leapfrog = rnd.choice([0, 1, 4, -2])
            cfg.n_steps = rnd.choice([-1, 0, 1, 16])
            cfg.step_base = rnd.choice([0, 5, -1])
            cfg.adapt_kind = rnd.choice([0, 1, 2, 3, -1])
            cfg.n_adapt = rnd.choice([0, 10])
            cfg.adapt_target = rnd.choice([0.75, 0.0, 1.0, 1.5, -0.1])
            cfg.adapt_rate = rnd.choice([0.05, 0.0, -1.0])
            cfg.thin = rnd.choice([0, 1, 2, -1])
            cfg.n_samples = rnd.choice([0, 10])
            cfg.lanes_per_chain = rnd.choice([0, 1, 2, 4, 8, 16, 3, 64, -4])
            cfg.stats_batch = rnd.choice([0, 1, 8, -1])
            cfg.trace_chains = rnd.choice([0, 1, 10 ** 6, -3])        # more than n_chains: clamped, never an overrun
            io = host_lib.HmcIO()
            io.q, io.grad, io.logp, io.adapt = ptr(), ptr(), ptr(), ptr()
            io.rng, io.accept_count, io.eps0 = ptr(), ptr(), ptr()
            io.trace, io.trace_accept, io.stats, io.rec_accept_count = ptr(0.3), ptr(0.3), ptr(0.3), ptr(0.3)
            if kind == "hmc":
                rc = L.arp_hmc_run(hh, rnd.choice([0, 1, 2]), C.byref(cfg) if rnd.random() < 0.95 else None,
                                   C.byref(io) if rnd.random() < 0.95 else None, None)
                if rc == 0:
                    assert cfg.n_steps == 0, "a launch cannot have succeeded without a device"
            else:
                io2 = host_lib.InterleavedIO()
                io2.k0 = io
                io2.adapt1, io2.accept_count1, io2.eps0_1 = ptr(), ptr(), ptr()
                io2.trace_accept1, io2.rec_accept_count1 = ptr(0.3), ptr(0.3)
                rc = L.arp_interleaved_run(hh, C.byref(cfg), rnd.choice([0, 4, -1]), C.byref(io2), None)
                if rc == 0:
                    assert cfg.n_steps == 0
        elif kind == "vi":
            cfg = host_lib.ViConfig()
            cfg.n_lr = rnd.choice([0, 1, 5, -1, 600])
            cfg.n_steps = rnd.choice([0, 1, 100])
            cfg.n_mc = rnd.choice([0, 1, 37, 256, 4096, 4097, -8])
            cfg.learn_a = rnd.choice([0, 1])
            cfg.tied_b = rnd.choice([0, 1])
            cfg.a_prior = rnd.choice([0, 1])
            io = host_lib.ViIO()
            io.lr, io.loc, io.rho, io.w, io.wb = ptr(), ptr(), ptr(), ptr(0.5), ptr(0.3)
            io.elbo, io.prior = ptr(), ptr(0.3)
            rc = L.arp_vi_run(hh, rnd.choice([0, 1, 3]), C.byref(cfg), C.byref(io), None)
        elif kind == "ess":
            S = rnd.choice([0, 1, 100, 2300, 5000, 10 ** 7])
            n = rnd.choice([0, 1, 25, 64, 1000, 1 << 31])
            need = L.arp_ess_workspace_bytes(S, n)
            assert need >= 0
            ws_bytes = rnd.choice([0, 1024, max(0, need - 64), need])
            wsb, wsp = _fake(min(ws_bytes, 1 << 20))
            keepalive.append(wsb)
            if rnd.random() < 0.3:
                wsp = C.c_void_p(wsp.value + 4)                        # misaligned workspace
            rc = L.arp_ess_ws(ptr(), S, n, rnd.choice([n, n - 1, 2 * n + 3]), ptr(),
                              wsp if ws_bytes else None, ws_bytes, None)
        elif kind == "adapt":
            cfg = host_lib.HmcConfig()
            cfg.n_steps = rnd.choice([-1, 0, 4])
            cfg.adapt_kind = rnd.choice([0, 1, 2, 7])
            cfg.adapt_target = rnd.choice([0.75, 2.0])
            cfg.adapt_rate = rnd.choice([0.05, -0.5])
            rc = L.arp_adapt_probe(C.byref(cfg), ptr(), rnd.choice([0, 3]), ptr(), ptr(0.5), None)
        else:
            rc = L.arp_clock_probe(rnd.choice([0, -1, 5]), ptr(0.5), ptr(0.5))
        calls += 1
        if rc != 0:
            failures += 1
            assert len(err()) > 0, (kind, name)
    assert failures >= 0.95 * calls            # all but the n_steps == 0 no-ops
    for n_, (sp, h, _) in handles.items():
        assert L.arp_model_destroy(h) == 0


def test_ess_workspace_size_is_a_whole_number_of_row_blocks(host_lib):
    """include/autoreparam.h: arp_ess_workspace_bytes sizes the workspace at which every listed series fits at once --
    64-row blocks, at least one (a single german-credit chain has 125 series, one parameter block of it 25)."""
    L = host_lib.lib()
    assert L.arp_ess_workspace_bytes(1000, 10 ** 6) == 0
    for S in (2400, 5000, 50000):
        sizes = [L.arp_ess_workspace_bytes(S, n) for n in (1, 25, 63, 64, 65, 128)]
        assert sizes[0] > 64 * 4 * S                                   # one whole 64-row block even for one series
        assert sizes[4] - sizes[3] >= 64 * 4 * S                       # the 65th series opens a second block
        assert all(b >= a for a, b in zip(sizes, sizes[1:]))
