"""CPU: the layout contract between the German-credit tile image (arp_api.hip: build_german) and the matrix-core
likelihood's LDS addresses (model_german.h: likelihood_mfma), restated in Python: every operand address a lane forms
must hit the element the host stored there, and both ds_read_b128 access patterns must be bank-conflict free under the
gfx950 rules (4 groups of 16 lanes, bank = dword address mod 64).  The GPU tests check the kernel itself against the oracle
(tests/test_gpu_density.py::test_german_observation_tiles); this one pins the index arithmetic both sides are written from."""
import numpy as np

# lane groups of ds_read_b128 (MI355X_MICROARCH: LDS)
G128 = [list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
        list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
        list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
        list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64))]
ROWS, COLS, BLK_WORDS = 128, 64, 16 * 64


def host_word(r, f):
    """word of element (row r of the tile, feature f) in the image: chunk c = f / 4 at chunk position c ^ (r & 11)"""
    return r * COLS + ((((f >> 2) ^ (r & 11)) & 15) << 2) + (f & 3)


def forward_words(lane, blk, i):
    """ds_read_b128 number i of the forward operand: lane (g, j) reads row j of block blk, columns 16 g + 4 i .. + 3"""
    g, j = lane >> 4, lane & 15
    abase = j * 256 + (((g << 2) ^ (j & 8)) << 4)
    a = abase + ((i ^ (j & 3)) << 4) + blk * 4096
    return [a // 4 + w for w in range(4)], [(16 * blk + j, 16 * g + 4 * i + w) for w in range(4)]


def backward_words(lane, blk, s):
    """ds_read_b128 number s of the backward operand: lane (g, j) reads row 4 g + s of block blk, columns 4 j .. 4 j + 3"""
    g, j = lane >> 4, lane & 15
    b = g * 1024 + s * 256 + ((j ^ s ^ ((g & 2) << 2)) << 4) + blk * 4096
    return [b // 4 + w for w in range(4)], [(16 * blk + 4 * g + s, 4 * j + w) for w in range(4)]


def test_image_is_a_permutation_and_addresses_hit_their_elements():
    img = {}
    for r in range(ROWS):
        for f in range(COLS):
            img[host_word(r, f)] = (r, f)
    assert sorted(img) == list(range(ROWS * COLS))          # a permutation of the tile: nothing overwritten, no holes
    for blk in range(ROWS // 16):
        for lane in range(64):
            for k in range(4):
                for words, want in (forward_words(lane, blk, k), backward_words(lane, blk, k)):
                    assert [img[w] for w in words] == want, (blk, lane, k)


def _extra_cycles(read):
    extra = 0
    for k in range(4):
        for grp in G128:
            banks = {}
            for lane in grp:
                for w in read(lane, 0, k)[0]:
                    banks.setdefault(w % 64, set()).add(w)
            extra += max(len(v) for v in banks.values()) - 1
    return extra


def test_operand_reads_are_bank_conflict_free():
    assert _extra_cycles(forward_words) == 0
    assert _extra_cycles(backward_words) == 0
    # the un-swizzled image would not be: 16 rows x the same chunk land on one 16-byte slot
    plain = lambda lane, blk, i: ([((lane & 15) * 256 + (4 * (lane >> 4) + i) * 16) // 4 + w for w in range(4)], None)
    assert _extra_cycles(plain) > 0


def test_backward_accumulators_cover_sixteen_consecutive_features():
    """accumulator k of the backward product takes feature 4 j + k on output row j; a lane (g', chain) of the result holds
    output rows 4 g' + r, i.e. features 16 g' + 4 r + k: the 16 consecutive features of state slot g'"""
    for gp in range(4):
        feats = sorted(4 * (4 * gp + r) + k for r in range(4) for k in range(4))
        assert feats == list(range(16 * gp, 16 * gp + 16))
