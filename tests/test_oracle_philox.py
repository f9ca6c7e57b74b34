"""Pin oracle/philox.py against the Random123 known-answer vectors (kat_vectors, philox4x32 10 rounds)."""
import numpy as np

from oracle import philox

KAT = [
    ((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
    ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
    ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
     (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1)),
]


def test_known_answer_vectors():
    for ctr, key, exp in KAT:
        out = philox.philox4x32_10(*ctr, *key)
        assert tuple(int(x) for x in out) == exp


def test_vectorised_matches_scalar():
    ids = np.arange(5, dtype=np.uint32)
    out = philox.philox4x32_10(ids, 7, 1, 2, 123, 456)
    for i in range(5):
        one = philox.philox4x32_10(i, 7, 1, 2, 123, 456)
        assert [int(o[i]) for o in out] == [int(x) for x in one]


def test_unit_float_ranges_and_moments():
    u = philox.reset_uniforms(42, np.arange(20000), 3)
    assert u.dtype == np.float32 and u.shape == (20000, 12)
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 5e-3
    z = philox.normals(42, np.arange(20000), 3, philox.STREAM_OBS_NOISE, 18)
    assert z.shape == (20000, 18) and np.isfinite(z).all()
    assert abs(z.mean()) < 5e-3 and abs(z.std() - 1.0) < 5e-3


def test_streams_and_ticks_are_independent():
    a = philox.raw_blocks(1, np.arange(8), 0, 0, 1)
    b = philox.raw_blocks(1, np.arange(8), 0, 1, 1)
    c = philox.raw_blocks(1, np.arange(8), 1, 0, 1)
    d = philox.raw_blocks(2, np.arange(8), 0, 0, 1)
    assert not (a == b).any() and not (a == c).any() and not (a == d).any()
    # env id is global: a shard starting at 4 sees the same numbers
    e = philox.raw_blocks(1, np.arange(4, 8), 0, 0, 1)
    assert (e == a[4:]).all()
