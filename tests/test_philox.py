"""Philox4x32-10 known-answer tests (Random123 kat_vectors) for the NumPy twin of the device generator."""
import numpy as np

from oracle import philox


def _kat(c, k):
    r = philox.philox4x32_10(*[np.uint32(x) for x in c], *[np.uint32(x) for x in k])
    return [int(x) for x in r]


def test_random123_known_answers():
    assert _kat((0, 0, 0, 0), (0, 0)) == [0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8]
    assert _kat((0xFFFFFFFF,) * 4, (0xFFFFFFFF,) * 2) == [0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD]
    assert _kat((0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344), (0xA4093822, 0x299F31D0)) == \
        [0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1]


def test_layer_normals_are_shard_invariant_and_standard():
    full = philox.layer_normals(7, 11, 2, 0, 64, 33)
    lo = philox.layer_normals(7, 11, 2, 0, 40, 33)
    hi = philox.layer_normals(7, 11, 2, 40, 24, 33)
    assert np.array_equal(full, np.concatenate([lo, hi]))           # global chain ids: sharding changes nothing
    z = philox.layer_normals(3, 5, 0, 0, 4096, 256).astype(np.float64).reshape(-1)
    assert abs(z.mean()) < 5 / np.sqrt(z.size) and abs(z.var() - 1) < 5 * np.sqrt(2 / z.size)
    assert not np.array_equal(philox.layer_normals(3, 5, 0, 0, 8, 8), philox.layer_normals(3, 6, 0, 0, 8, 8))
    assert not np.array_equal(philox.layer_normals(3, 5, 0, 0, 8, 8), philox.layer_normals(3, 5, 1, 0, 8, 8))
