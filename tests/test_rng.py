"""CPU: known-answer and distribution tests of the sampler's RNG specification
(Philox4x32-10 seeding, MWC64X streams, Box-Muller normals)."""
import ctypes as C

import numpy as np


def _philox(oracle, ctr, key):
    c = (C.c_uint32 * 4)(*ctr)
    oracle.lib().orc_philox(c, C.c_uint32(key[0]), C.c_uint32(key[1]))
    return [int(v) for v in c]


def test_philox_known_answers(oracle_lib):
    # Random123 kat_vectors, philox4x32 10 rounds
    assert _philox(oracle_lib, [0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert _philox(oracle_lib, [0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert _philox(oracle_lib, [0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def _mwc64x(x, c, n):
    """Independent restatement (python ints) of D. B. Thomas' MWC64X: output x ^ c, then
    (x, c) <- (lo, hi) of A x + c with A = 4294883355."""
    out = []
    for _ in range(n):
        out.append(x ^ c)
        t = x * 4294883355 + c
        x, c = t & 0xffffffff, t >> 32
    return out


def test_mwc64x_known_answer():
    # first outputs from state (x, c) = (1, 0): 1, then A, then (A*A mod 2^32) ^ (A*A >> 32), ...
    A = 4294883355
    o = _mwc64x(1, 0, 3)
    assert o[0] == 1 and o[1] == A and o[2] == ((A * A) & 0xffffffff) ^ ((A * A) >> 32)
    # (0, 0) and (2^32 - 1, A - 1) are the generator's fixed points
    assert _mwc64x(0, 0, 2) == [0, 0]
    assert _mwc64x(0xffffffff, A - 1, 2)[0] == _mwc64x(0xffffffff, A - 1, 2)[1]


def test_stream_is_philox_seeded_mwc64x(oracle_lib):
    seed, chain, slot, lanes = 0x123456789abcdef, 4242, 3, 8
    out = np.zeros(64, np.uint32)
    oracle_lib.lib().orc_stream(C.c_uint64(seed), C.c_uint64(chain), C.c_uint32(slot), C.c_uint32(lanes), 64,
                                out.ctypes.data_as(C.c_void_p))
    st = _philox(oracle_lib, [chain & 0xffffffff, chain >> 32, slot, lanes], [seed & 0xffffffff, seed >> 32])
    assert [int(v) for v in out] == _mwc64x(st[0], st[1] >> 1, 64)


def test_word_stream_is_equidistributed(oracle_lib):
    """Bit frequencies and byte chi-square of one stream (the normals consume the high bits)."""
    n = 1 << 18
    out = np.zeros(n, np.uint32)
    oracle_lib.lib().orc_stream(C.c_uint64(99), C.c_uint64(7), C.c_uint32(1), C.c_uint32(4), n,
                                out.ctypes.data_as(C.c_void_p))
    bits = ((out[:, None] >> np.arange(32, dtype=np.uint32)) & 1).mean(axis=0)
    assert np.abs(bits - 0.5).max() < 5 * 0.5 / np.sqrt(n)
    for shift in (0, 8, 16, 24):
        cnt = np.bincount((out >> shift) & 0xff, minlength=256)
        chi2 = ((cnt - n / 256.0) ** 2 / (n / 256.0)).sum()
        assert chi2 < 255 + 5 * np.sqrt(2 * 255)
    # serial correlation of successive words
    u = out.astype(np.float64) / 2 ** 32
    assert abs(np.corrcoef(u[:-1], u[1:])[0, 1]) < 5 / np.sqrt(n)


def test_streams_differ_by_chain_slot_and_seed(oracle_lib):
    def first(seed, chain, slot, lanes):
        out = np.zeros(4, np.uint32)
        oracle_lib.lib().orc_stream(C.c_uint64(seed), C.c_uint64(chain), C.c_uint32(slot), C.c_uint32(lanes), 4,
                                    out.ctypes.data_as(C.c_void_p))
        return tuple(int(v) for v in out)
    base = first(1, 0, 0, 4)
    assert len({base, first(2, 0, 0, 4), first(1, 1, 0, 4), first(1, 0, 1, 4), first(1, 0, 0, 8)}) == 5


def test_normals_moments(oracle_lib):
    n_pairs = 200000
    z = np.zeros(2 * n_pairs, np.float32)
    oracle_lib.lib().orc_normals(C.c_uint64(7), C.c_uint64(0), C.c_uint32(0), C.c_uint32(4), n_pairs,
                                 z.ctypes.data_as(C.c_void_p))
    z = z.astype(np.float64)
    n = z.size
    assert abs(z.mean()) < 4 / np.sqrt(n)
    assert abs(z.var() - 1) < 4 * np.sqrt(2.0 / n)
    assert abs((z ** 3).mean()) < 4 * np.sqrt(15.0 / n)
    assert abs((z ** 4).mean() - 3) < 4 * np.sqrt(96.0 / n)
    # the two outputs of a pair are uncorrelated, and so are successive pairs
    assert abs(np.corrcoef(z[0::2], z[1::2])[0, 1]) < 4 / np.sqrt(n_pairs)
    assert abs(np.corrcoef(z[:-2:2], z[2::2])[0, 1]) < 4 / np.sqrt(n_pairs)
    assert np.abs(z).max() < 6.8  # u >= 2^-33: |z| <= sqrt(2*33*ln2) = 6.76


def test_mwc64x_jump_ahead_is_n_single_steps():
    """arp_device.h: rng_jump.  With v = c 2^32 + x a step is v -> A v mod m, m = A 2^32 - 1, so n steps are ONE modular
    multiplication by A^n; the device forms it as a Montgomery product (two rounds of the generator's own reduction on
    the 128-bit product).  Restated here in 64-bit-wrapping Python integers, operation for operation, and held to n
    single steps (and to pow() for long jumps); the intermediate bounds the device code relies on are asserted."""
    import random
    A, M64 = 4294883355, (1 << 64) - 1
    m = (A << 32) - 1
    one = (0 - m) & M64
    assert one == (1 << 64) % m
    mont_a = (A * one) % m

    def montmul(a, b):
        p = a * b
        lo, hi = p & M64, p >> 64
        t = (lo & 0xffffffff) * A
        l1, h1 = ((hi << 32) & M64) | (lo >> 32), hi >> 32
        l1 = (l1 + t) & M64
        h1 += 1 if l1 < t else 0
        t = (l1 & 0xffffffff) * A
        c2 = h1 >> 32
        assert ((h1 << 32) & M64) + (l1 >> 32) <= M64
        l2 = (((h1 << 32) & M64) + (l1 >> 32) + t) & M64
        c2 += 1 if l2 < t else 0
        assert c2 <= 1
        if c2:
            assert l2 + one <= M64
            l2 += one
        for _ in range(2):
            if l2 >= m:
                l2 -= m
        assert l2 < m
        return l2

    def power(n):
        r, base = one, mont_a
        while n:
            if n & 1:
                r = montmul(r, base)
            base = montmul(base, base)
            n >>= 1
        return r

    rnd = random.Random(1)
    cases = [(0xffffffff, A - 2), (1, 0), (0, 1), (0xffffffff, 0x7fffffff), (12345, 0)]
    cases += [(rnd.getrandbits(32), rnd.getrandbits(31)) for _ in range(400)]
    for x, c in cases:
        for n in (0, 1, 2, 10, 34, 68, 70, 1000, 4095 * 34):
            if n <= 1000:
                xs, cs = x, c
                for _ in range(n):
                    t = xs * A + cs
                    xs, cs = t & 0xffffffff, t >> 32
            else:
                v = ((c << 32) | x) * pow(A, n, m) % m
                xs, cs = v & 0xffffffff, v >> 32
            v = montmul((c << 32) | x, power(n))
            assert (v & 0xffffffff, v >> 32) == (xs, cs), (x, c, n)
