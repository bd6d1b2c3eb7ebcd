"""GPU: edge cases of the C ABI -- empty / tiny / ragged batches, zero-step calls,
non-finite proposals, bad arguments, the largest BASELINE chain count."""

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


def _eng(mname, gpu):
    from autoreparam_amd import engine
    return engine.Engine(helpers.spec(mname), gpu)


def test_bad_arguments_raise(gpu):
    from autoreparam_amd import engine, _lib
    sp = helpers.spec("radon_MN")
    eng = _eng("radon_MN", gpu)
    x0 = torch.zeros(0, sp.D, device=gpu)
    with pytest.raises(RuntimeError):
        eng.logp_grad(x0)                               # empty batch
    st = engine.ChainState(torch.zeros(4, sp.D, device=gpu))
    eps = np.full(sp.D, 0.1, np.float32)
    with pytest.raises(RuntimeError):
        eng.hmc_run(st, eps, 0, 1)                      # zero leapfrog steps
    with pytest.raises(RuntimeError):
        eng.hmc_run(st, eps, 2, 1, thin=0)
    with pytest.raises(RuntimeError):
        eng.hmc_run(st, eps, 2, 1, adapt_kind=7)
    with pytest.raises(RuntimeError):
        eng.hmc_run(st, eps, 2, 1, lanes=3)
    eng.hmc_run(st, eps, 2, 0)                          # zero transitions: a no-op
    assert st.step == 0 and torch.equal(st.q, torch.zeros_like(st.q))
    # the step-size recurrences compare log alpha with log(target): a target outside (0, 1) is refused, not mis-adapted
    for bad in (0.0, 1.0, 1.5, float("nan")):
        with pytest.raises(RuntimeError, match="adapt_target"):
            eng.hmc_run(st, eps, 2, 1, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=5, adapt_target=bad)
        with pytest.raises(RuntimeError, match="adapt_target"):
            eng.adapt_probe(np.zeros((2, 3), np.float32), torch.ones(3, 4, device=gpu), _lib.ADAPT_DUAL, 2, target=bad)
    with pytest.raises(RuntimeError, match="adapt_rate"):
        eng.hmc_run(st, eps, 2, 1, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=5, adapt_rate=0.0)
    eng.hmc_run(st, eps, 2, 1, adapt_kind=_lib.ADAPT_NONE, adapt_target=7.0)      # not adapting: the field is not read
    # shared-parameter groups of the untied VI: every element must point at the first element of a contiguous group
    lr = [0.1]
    loc = torch.zeros(1, sp.D, device=gpu); rho = torch.full((1, sp.D), -2.0, device=gpu)
    w = torch.zeros(1, sp.D, device=gpu); wb = torch.zeros(1, sp.D, device=gpu)
    good = np.arange(sp.D); good[4:9] = 4
    eng.vi_run(lr, loc, rho, 2, 16, w=w, wb=wb, a_group=good, b_group=np.arange(sp.D))
    for grp in (np.full(sp.D, sp.D + 3), np.roll(np.arange(sp.D), 1), np.where(np.arange(sp.D) == 6, 4, np.arange(sp.D))):
        with pytest.raises(RuntimeError, match="a_group"):
            eng.vi_run(lr, loc, rho, 2, 16, w=w, wb=wb, a_group=grp, b_group=np.arange(sp.D))


@pytest.mark.parametrize("mname", ["8schools", "radon_PA", "election", "german", "electric"])
def test_single_chain_and_ragged_tail(oracle_lib, gpu, mname):
    """C = 1 and C = 67 (neither a multiple of the chains per wave) run and agree with the oracle."""
    from autoreparam_amd import engine
    sp = helpers.spec(mname)
    eng = _eng(mname, gpu)
    orc = oracle_lib.OracleModel(sp)
    a, b = helpers.params(sp, "NCP")
    eng.set_param(0, (a, b))
    eps = np.full(sp.D, 1e-3, np.float32)
    for lanes in {"8schools": [8], "radon_PA": [8], "election": [16], "german": [4, 8, 16], "electric": [16]}[mname]:
        for Cn in (1, 67):
            q0 = helpers.states(sp, Cn, seed=Cn, scale=0.05)
            st = engine.ChainState(torch.as_tensor(q0, device=gpu))
            tr = torch.full((2, Cn, sp.D), 7.0, device=gpu)
            eng.hmc_run(st, eps, 2, 3, seed=3, n_burnin=0, thin=2, trace=tr, lanes=lanes)
            so = oracle_lib.new_state(q0, np.float32)
            tro = np.zeros((2, Cn, sp.D), np.float32)
            orc.hmc_run(so, a, b, eps, 2, 3, seed=3, n_burnin=0, thin=2, trace=tro, lanes=lanes)
            scale = np.abs(so["q"]).max() + 1
            assert np.abs(st.q.cpu().numpy() - so["q"]).max() <= 1e-4 * scale
            assert np.abs(tr.cpu().numpy() - tro).max() <= 1e-4 * scale      # every row written, nothing else


def test_non_finite_proposals_are_rejected(gpu):
    """inference.py:323-324: numerical failure becomes NaN -> the proposal is rejected."""
    from autoreparam_amd import engine
    sp = helpers.spec("8schools")
    eng = _eng("8schools", gpu)
    eng.set_param(0, "CP")
    q0 = helpers.states(sp, 256, seed=1)
    st = engine.ChainState(torch.as_tensor(q0, device=gpu))
    eng.hmc_run(st, np.full(sp.D, 80.0, np.float32), 4, 6, seed=2)   # exp(log_tau) overflows on the way
    assert torch.isfinite(st.q).all() and torch.isfinite(st.logp).all()
    assert (st.accept_count <= 6).all()
    # most proposals blow up, so most chains never move
    assert (st.q.cpu().numpy() == q0).all(axis=1).mean() > 0.5


def test_election_at_config5_size(gpu):
    """BASELINE config 5 shape: election, 131 072 chains (VIP parameterisation), finite and mixing."""
    from autoreparam_amd import engine, _lib
    sp = helpers.spec("election")
    eng = _eng("election", gpu)
    a = np.full(sp.D, 0.5, np.float32); b = np.ones(sp.D, np.float32)   # the reference's tied cVIP: b = 1
    eng.set_param(0, (a, b))
    Cn = 131072
    q0 = helpers.states(sp, Cn, seed=3, scale=0.05)
    st = engine.ChainState(torch.as_tensor(q0, device=gpu))
    eps = np.full(sp.D, 0.01, np.float32); eps[[0, 1, 53, 54]] = 1e-3
    eng.hmc_run(st, eps, 8, 60, seed=5, adapt_kind=_lib.ADAPT_DUAL, n_adapt=50)
    assert torch.isfinite(st.q).all()
    acc = st.accept_count.double().mean().item() / st.step
    assert 0.4 < acc < 0.98


def test_electric_rejects_data_its_cell_collapse_cannot_hold(gpu):
    """The electric lane collapses each pair to (control, treated) cells: pairs must not mix grades and
    treatment must be a 0/1 indicator; anything else fails loudly at model creation."""
    import copy
    from autoreparam_amd import engine
    sp = helpers.spec("electric")
    bad = copy.copy(sp); bad.raw = dict(sp.raw)
    g = sp.raw["grade"].copy(); g[0] = g[0] % 4 + 1        # first observation's pair now mixes two grades
    bad.raw["grade"] = g
    with pytest.raises(RuntimeError, match="share a grade"):
        engine.Engine(bad, gpu)
    bad = copy.copy(sp); bad.raw = dict(sp.raw)
    t = sp.raw["treatment"].copy(); t[3] = 0.5
    bad.raw["treatment"] = t
    with pytest.raises(RuntimeError, match="0/1"):
        engine.Engine(bad, gpu)


def test_native_ess_matches_oracle_and_ar1(gpu):
    """arp_ess (direct auto-covariances, cut at the first negative one) against the ORACLE's float64
    tfp.mcmc.effective_sample_size restatement (oracle/ess_ref.py: FFT form, TFP defaults; inference.py:240, 327) on the
    same traces, and against the AR(1) known answer ESS / S -> (1 - rho) / (1 + rho)."""
    from oracle import ess_ref
    from autoreparam_amd import util
    S, Cn, D = 600, 37, 5
    rho = np.array([0.0, 0.3, 0.6, 0.9, -0.4])
    x64 = ess_ref.ar1(S, (Cn, D), rho, seed=5) * [1.0, 10.0, 0.1, 3.0, 1.0] + [0.0, 100.0, -5.0, 1e3, 0.0]   # scales / offsets
    x = torch.as_tensor(x64, dtype=torch.float32)
    xd = x.to(gpu)
    native = util.effective_sample_size(xd).cpu().numpy()
    ref = ess_ref.ess_fft(x.numpy())                      # the float32 trace the kernel saw, statistic in float64
    np.testing.assert_allclose(native, ref, rtol=1e-3)
    # AR(1): a long run pins the estimator itself (ESS/S within 3 % of (1-rho)/(1+rho), averaged over 64 series)
    long = ess_ref.ar1(20000, (64, 4), rho[:4], seed=6)
    got = util.effective_sample_size(torch.as_tensor(long, dtype=torch.float32).to(gpu)).cpu().numpy().mean(axis=0) / 20000
    expect = (1 - rho[:4]) / (1 + rho[:4])
    assert np.all(np.abs(got / expect - 1) < 0.03), (got, expect)
    assert np.allclose(native[:, 4], S)                   # negative first lag: the sum stops at lag 0
    # a strided view (sub-range of chains) and a constant series
    sub = util.effective_sample_size(xd[:, 3:11, :]).cpu().numpy()
    np.testing.assert_allclose(sub, native[3:11], rtol=1e-6)
    const = torch.ones(50, 2, 3, device=gpu)
    assert torch.isnan(util.effective_sample_size(const)).all()
    # a series whose first samples sit far from where it settles (the one-pass form centres on the first 16 samples and
    # must notice)
    drift = x.clone()
    drift[:20] += torch.tensor([50.0, 5e3, 3.0, 2e4, -80.0])
    nat = util.effective_sample_size(drift.to(gpu)).cpu().numpy()
    np.testing.assert_allclose(nat, ess_ref.ess_fft(drift.numpy()), rtol=5e-3)
    short = x[:9].contiguous()            # fewer samples than the window of lags
    np.testing.assert_allclose(util.effective_sample_size(short.to(gpu)).cpu().numpy(), ess_ref.ess_fft(short.numpy()), rtol=2e-3)
    # series slow enough to need lag sweeps past the first window (rho = 0.98: ~ 100 positive lags)
    slow = torch.as_tensor(ess_ref.ar1(3000, (16, 3), [0.98, 0.95, 0.5], seed=7), dtype=torch.float32)
    np.testing.assert_allclose(util.effective_sample_size(slow.to(gpu)).cpu().numpy(), ess_ref.ess_fft(slow.numpy()), rtol=2e-3)


def test_long_series_ess_on_the_matrix_cores(gpu):
    """arp_ess_ws: series longer than the one-kernel path holds in LDS (S + 72 > 2 304) that are still positively
    correlated after the coalesced sweeps are finished by ess_tail_kernel (Toeplitz blocks on v_mfma_f32_16x16x4_f32)
    out of a caller-owned workspace -- against the float64 FFT oracle, against the workspace-free per-lane route of
    arp_ess, with the listed series taken in one chunk and in many, and for a leading block of a wider trace."""
    import ctypes as C
    from oracle import ess_ref
    from autoreparam_amd import util, _lib
    S, Cn, D = 6000, 24, 5
    rho = np.array([0.995, 0.97, 0.5, 0.9, -0.2])              # cut lags from ~ 1 000 down to 0
    x64 = ess_ref.ar1(S, (Cn, D), rho, seed=11) * [1.0, 4.0, 0.1, 30.0, 1.0] + [0.0, -20.0, 5.0, 1e3, 0.0]
    x = torch.as_tensor(x64, dtype=torch.float32)
    xd = x.to(gpu)
    ref = ess_ref.ess_fft(x.numpy())
    got = util.effective_sample_size(xd).cpu().numpy()          # with a workspace (util allocates it)
    np.testing.assert_allclose(got, ref, rtol=2e-3)
    L = _lib.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    n = Cn * D
    out = torch.empty(Cn, D, device=gpu)
    _lib.check(L.arp_ess(C.c_void_p(xd.data_ptr()), S, n, n, C.c_void_p(out.data_ptr()), stream))   # no workspace
    np.testing.assert_allclose(out.cpu().numpy(), got, rtol=1e-4)
    # a workspace that holds the lists and 64 rows: the 72 listed series go in two chunks -- bitwise the one-chunk result
    need = int(L.arp_ess_workspace_bytes(S, n))
    assert need > 4 * n * S
    small = torch.empty(28 * n + 4096 + 64 * 4 * ((S + 63) // 64 * 64) + 4096, dtype=torch.uint8, device=gpu)
    out2 = torch.empty(Cn, D, device=gpu)
    _lib.check(L.arp_ess_ws(C.c_void_p(xd.data_ptr()), S, n, n, C.c_void_p(out2.data_ptr()), C.c_void_p(small.data_ptr()),
                            small.numel(), stream))
    assert np.array_equal(out2.cpu().numpy(), got)
    tiny = torch.empty(1024, dtype=torch.uint8, device=gpu)
    assert L.arp_ess_ws(C.c_void_p(xd.data_ptr()), S, n, n, C.c_void_p(out2.data_ptr()), C.c_void_p(tiny.data_ptr()),
                        tiny.numel(), stream) != 0 and b"workspace too small" in L.arp_last_error()
    assert L.arp_ess_workspace_bytes(1000, 10 ** 6) == 0        # short series never need one
    # leading block of chains of a wider trace, taken in place (a streaming run's kept trace)
    sub = util.effective_sample_size(xd[:, :7, :]).cpu().numpy()
    assert np.array_equal(sub, got[:7])
    # one slowly mixing series per wave (64 series per chain, one of them slow): the wave skips its dense continuation and
    # lists the series at lag 16 -- the tail starts from block 1 instead of block 3
    rho64 = np.zeros(64); rho64[5] = 0.99; rho64[40] = 0.6
    y = torch.as_tensor(ess_ref.ar1(4000, (3, 64), rho64, seed=13), dtype=torch.float32)
    np.testing.assert_allclose(util.effective_sample_size(y.to(gpu)).cpu().numpy(), ess_ref.ess_fft(y.numpy()), rtol=2e-3)
    # a series that never decorrelates within the trace (a trend): every lag positive until the weights run out
    t = torch.linspace(0, 1, 2500, device=gpu).reshape(-1, 1, 1) + 0.01 * torch.randn(2500, 2, 3, device=gpu)
    np.testing.assert_allclose(util.effective_sample_size(t).cpu().numpy(), ess_ref.ess_fft(t.cpu().numpy()), rtol=5e-3)


def test_workspace_size_function_is_enough_for_few_series(gpu):
    """arp_ess_workspace_bytes alone sizes the workspace (include/autoreparam.h): one german-credit chain (25 series
    of a long trace, fewer than the 64 rows the gather works in) must not be refused as `workspace too small`, and a
    count that is not a multiple of 64 must go in one chunk (same figures as the chunked and the workspace-free routes)."""
    import ctypes as C
    from oracle import ess_ref
    from autoreparam_amd import _lib
    L = _lib.lib()
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for n, S in ((25, 5000), (1, 3000), (70, 2600)):
        rho = np.linspace(0.99, 0.9, n)
        x = torch.as_tensor(ess_ref.ar1(S, (1, n), rho, seed=3 + n), dtype=torch.float32)
        xd = x.to(gpu)
        need = int(L.arp_ess_workspace_bytes(S, n))
        assert need > 0
        ws = torch.empty(need, dtype=torch.uint8, device=gpu)
        out = torch.empty(1, n, device=gpu)
        _lib.check(L.arp_ess_ws(C.c_void_p(xd.data_ptr()), S, n, n, C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                need, stream))
        np.testing.assert_allclose(out.cpu().numpy(), ess_ref.ess_fft(x.numpy()), rtol=2e-3)
        out0 = torch.empty(1, n, device=gpu)
        _lib.check(L.arp_ess(C.c_void_p(xd.data_ptr()), S, n, n, C.c_void_p(out0.data_ptr()), stream))
        np.testing.assert_allclose(out.cpu().numpy(), out0.cpu().numpy(), rtol=1e-4)


def test_matrix_core_tail_at_the_reference_trace_length_and_beyond(gpu):
    """The tail's float32 accumulators are folded every 65 536 samples (ess_tail.h: kEssTailFlush): the reference's
    50 000-sample trace (one fold) and a 150 000-sample one (three folds) against the float64 FFT oracle, slowly mixing
    series with a large offset included."""
    from oracle import ess_ref
    from autoreparam_amd import util
    for S in (50000, 150000):
        rho = np.array([0.999, 0.99, 0.9, 0.3])
        x64 = ess_ref.ar1(S, (2, 4), rho, seed=S) * [1.0, 5.0, 0.2, 1.0] + [0.0, 40.0, -3.0, 0.0]
        x = torch.as_tensor(x64, dtype=torch.float32)
        got = util.effective_sample_size(x.to(gpu)).cpu().numpy()
        np.testing.assert_allclose(got, ess_ref.ess_fft(x.numpy()), rtol=2e-3)


def test_clock_probe_reports_a_plausible_shader_clock(gpu):
    """arp_clock_probe (measurement hook of bench.py): shader cycles over 100 MHz ticks of a vector-bound load -- an MI355X
    holds 1.9 - 2.4 GHz under it; bad arguments fail loudly."""
    import ctypes as C
    from autoreparam_amd import engine, _lib
    ghz = engine.clock_probe(gpu, 5.0)
    assert ghz is not None and 1.2 < ghz < 2.6, ghz
    assert _lib.lib().arp_clock_probe(0, C.c_void_p(0), C.c_void_p(0)) != 0


@pytest.mark.parametrize("sampler", ["interleaved", "hmc"])
def test_trace_past_two_to_the_32_elements(gpu, sampler):
    """A trace of 4.65e9 floats (18.6 GB: 1 000 rows x 65 536 chains x 71) and acceptance flags: rows written beyond element
    2^32 (byte 2^34) are where they belong -- the last rows equal those of a second run of the same chains that records only the
    tail (same seeds, same transitions, burn-in moved), and the rows in front of them are not touched by it."""
    from autoreparam_amd import engine, _lib
    sp = helpers.spec("radon_PA")
    eng = _eng("radon_PA", gpu)
    eng.set_param(0, "CP"); eng.set_param(1, "NCP")
    C, S, T = 65536, 1000, 2000                         # a row every second transition
    assert S * C * sp.D > 2 ** 32
    q0 = helpers.states(sp, C, seed=6, scale=0.1)
    e = np.full(sp.D, 0.05, np.float32); e[2] = 0.012

    def run(n_burn, rows):
        st = engine.ChainState(torch.as_tensor(q0, device=gpu))
        tr = torch.full((rows, C, sp.D), -7.0, device=gpu)
        acc = torch.full((rows, C), 9, dtype=torch.uint8, device=gpu)
        for _ in range(2):                              # two launches of 1 000 transitions
            if sampler == "interleaved":
                eng.interleaved_run(st, e, e, 1, 1, T // 2, seed=3, adapt_kind=_lib.ADAPT_SIMPLE, n_adapt=100, n_burnin=n_burn,
                                    thin=2, trace=tr, trace_accept0=acc, trace_centered=False)
            else:
                eng.hmc_run(st, e, 1, T // 2, seed=3, adapt_kind=_lib.ADAPT_DUAL, n_adapt=100, n_burnin=n_burn, thin=2, trace=tr,
                            trace_accept=acc, trace_centered=False)
        torch.cuda.synchronize()
        return tr, acc

    tr, acc = run(0, S)
    assert not (tr[-1] == -7.0).any() and not (acc[-1] == 9).any() and torch.isfinite(tr[-3:]).all()
    tail_rows = 5
    tr_t, acc_t = run(2 * (S - tail_rows), tail_rows)       # the same transitions; result r is taken after transition 1 + burn-in + 2 r
    assert torch.equal(tr[S - tail_rows:], tr_t) and torch.equal(acc[S - tail_rows:], acc_t)
    # rows on both sides of element 2^32 hold chain states (not the fill value), row by row different
    k = 2 ** 32 // (C * sp.D)
    assert not (tr[k - 1:k + 2] == -7.0).any() and not torch.equal(tr[k], tr[k + 1])
    del tr, acc, tr_t, acc_t
    torch.cuda.empty_cache()


@pytest.mark.parametrize("streaming", [False, True])
def test_a_rank_without_chains_returns_empty_blocks(gpu, streaming):
    """A job with fewer chains than ranks (8 GPUs, --num_chains=5) leaves some ranks no chain: inference.hmc /
    hmc_interleaved launch nothing there and hand back empty [S, 0, ...] / [0, ...] blocks for the end-of-run gathers
    (tests/test_distributed.py drives main.py's side of it at world size 8)."""
    from autoreparam_amd import flags as flags_mod, graphs, inference, models
    cfg = models.get_model_by_name("radon", "MN")
    sp = cfg.model
    f = flags_mod.FlagValues()
    f.num_chains, f.num_samples, f.num_burnin_steps, f.num_adaptation_steps, f.num_leapfrog_steps = 5, 40, 10, 8, 3
    f.device = str(gpu)
    if streaming:
        f.trace_chunk_rows, f.ess_chains = 16, 4
    target, *_ = graphs.make_cp_graph(cfg, flags=f)
    init = [np.zeros((0,) + tuple(s), np.float32) for s in sp.part_shapes]
    step = [0.15] * 3 + [np.full(85, 0.3)]
    _, kr, st, ess = inference.hmc(target, cfg, step, init, "CP", flags=f, chain_offset=5)
    assert kr.ess_info.chains == 0 and np.sum(kr.inner_results.is_accepted) == 0
    assert all(np.asarray(e).shape[0] == 0 for e in ess) and all(s.shape[1] == 0 for s in st)
    if streaming:
        assert kr.ess_info.batch_means.shape == (0, sp.D)
    t_cp, t_ncp = target, graphs.make_ncp_graph(cfg, flags=f)[0]
    st_i, kr_i, ess_i = inference.hmc_interleaved(cfg, t_cp, t_ncp, 2, 2, step, step, init, flags=f, chain_offset=5)
    assert kr_i.ess_info.chains == 0 and all(np.asarray(e).shape[0] == 0 for e in ess_i)


def _ar1(rs, S, n, rho):
    x = np.empty((S, n), np.float32)
    prev = rs.randn(n)
    rho = np.broadcast_to(np.asarray(rho, np.float64), (n,))
    for t in range(S):
        prev = rho * prev + np.sqrt(1 - rho * rho) * rs.randn(n)
        x[t] = prev
    return x


@pytest.mark.parametrize("S,n", [(32, 1), (33, 31), (100, 33), (999, 1000), (1000, 4097), (1024, 64), (1000, 65)])
def test_one_pass_tile_ess_matches_oracle_and_the_two_sweep_kernel(gpu, monkeypatch, S, n):
    """ess.hip: ess_tile_kernel (32 <= S <= 1 024: a workgroup keeps 32 series in LDS for their whole length, the trace is
    read once) against the oracle's float64 tfp.mcmc.effective_sample_size restatement (oracle/ess_ref.py; inference.py:240)
    and against the two-sweep kernel (ARP_ESS_TILE=0): fast and slowly mixing series side by side (cuts from lag 1 to
    several hundred), an offset 10^4 standard deviations from zero, a constant series (NaN as in the FFT form), a trending one,
    ragged tile widths, a strided view of a wider trace, and the same bits run after run."""
    from oracle import ess_ref
    from autoreparam_amd import util
    rs = np.random.RandomState(S * 7 + n)
    rho = rs.choice([0.0, 0.3, 0.6, 0.9, 0.97, 0.995], size=n)
    x = _ar1(rs, S, n, rho)
    x[:, 0] += 1e4 * 1.0                                   # far from zero: the mean is taken about each thread's first value
    if n > 5:
        x[:, 3] = 2.5                                     # constant
        x[:, 4] += np.linspace(0, 20, S)                   # trend: positive at every lag up to ~ S / 3
    xd = torch.as_tensor(x).to(gpu).reshape(S, n, 1)
    monkeypatch.setenv("ARP_DEBUG", "1"); monkeypatch.setenv("ARP_ESS_TILE", "1")
    got = util.effective_sample_size(xd).cpu().numpy().reshape(n)
    got2 = util.effective_sample_size(xd).cpu().numpy().reshape(n)
    assert np.array_equal(got, got2, equal_nan=True)
    want = ess_ref.ess_fft(x.astype(np.float64).reshape(S, n, 1)).reshape(n)
    if n > 5:
        assert np.isnan(got[3]) and np.isnan(want[3])
    ok = ~np.isnan(want)
    np.testing.assert_allclose(got[ok], want[ok], rtol=2e-3)
    monkeypatch.setenv("ARP_ESS_TILE", "0")
    old = util.effective_sample_size(xd).cpu().numpy().reshape(n)
    monkeypatch.setenv("ARP_ESS_TILE", "1")
    np.testing.assert_allclose(got[ok], old[ok], rtol=2e-3)
    # a leading block of a wider trace, in place (row stride > series count)
    if n >= 33:
        k = 17
        sub = util.effective_sample_size(xd.reshape(S, n, 1)[:, :k, :]).cpu().numpy().reshape(k)
        assert np.array_equal(sub, got[:k], equal_nan=True)
