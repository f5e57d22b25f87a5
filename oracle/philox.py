"""Philox4x32-10 + Box-Muller, NumPy restatement.  TEST INFRASTRUCTURE ONLY.

This file is part of ``oracle/``: it may be imported by ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` and by
nothing else.  The product path (``montecarlopredictivecoding_amd``) never imports it.

What it restates
----------------
The reference draws its Langevin noise with ``Tensor.normal_`` inside
``random_step`` (/root/reference/utils/model.py:35-44).  torch's CPU mt19937
stream cannot be reproduced on a GPU, so the engine defines its own counter-based
generator (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11:
Philox4x32 with 10 rounds) and this module is the bit-exact CPU twin of the HIP
device function ``philox4x32_10`` in ``csrc/mcpc_device.h``.

Counter / key layout (identical on both sides, documented in DESIGN.md):

    key     = (seed_lo, seed_hi)
    counter = (global_chain, (layer << 24) | (unit // 4), step_lo, step_hi)

One call yields four u32 -> four standard normals for units 4g..4g+3 of one
layer of one chain at one step, which makes trajectories independent of how the
chains are sharded over GPUs.

Pinned against the Random123 known-answer vectors in tests/test_philox.py.
"""
import numpy as np

PHILOX_M0 = np.uint64(0xD2511F53)
PHILOX_M1 = np.uint64(0xCD9E8D57)
PHILOX_W0 = np.uint32(0x9E3779B9)
PHILOX_W1 = np.uint32(0xBB67AE85)
_MASK32 = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised Philox4x32-10.  All arguments broadcastable uint32 arrays.

    Returns four uint32 arrays.
    """
    c0, c1, c2, c3, k0, k1 = np.broadcast_arrays(
        *[np.asarray(a, dtype=np.uint32) for a in (c0, c1, c2, c3, k0, k1)])
    c0 = c0.copy(); c1 = c1.copy(); c2 = c2.copy(); c3 = c3.copy()
    k0 = k0.copy(); k1 = k1.copy()
    with np.errstate(over="ignore"):
        for _ in range(10):
            p0 = PHILOX_M0 * c0.astype(np.uint64)
            p1 = PHILOX_M1 * c2.astype(np.uint64)
            hi0 = (p0 >> np.uint64(32)).astype(np.uint32)
            lo0 = (p0 & _MASK32).astype(np.uint32)
            hi1 = (p1 >> np.uint64(32)).astype(np.uint32)
            lo1 = (p1 & _MASK32).astype(np.uint32)
            n0 = hi1 ^ c1 ^ k0
            n1 = lo1
            n2 = hi0 ^ c3 ^ k1
            n3 = lo0
            c0, c1, c2, c3 = n0, n1, n2, n3
            k0 = k0 + PHILOX_W0
            k1 = k1 + PHILOX_W1
    return c0, c1, c2, c3


def u32_to_unit_open0(x):
    """(0, 1]: ((x >> 8) + 1) * 2^-24, exact in fp32 (argument of the log)."""
    return ((x >> np.uint32(8)).astype(np.float32) + np.float32(1.0)) * np.float32(2.0 ** -24)


def u32_to_unit_half_open(x):
    """[0, 1): (x >> 8) * 2^-24, exact in fp32 (fraction of a full turn)."""
    return (x >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)


def box_muller(xa, xb):
    """Two u32 -> two fp32 standard normals (r cos, r sin)."""
    u1 = u32_to_unit_open0(xa)
    u2 = u32_to_unit_half_open(xb)
    r = np.sqrt(np.float32(-2.0) * np.log(u1)).astype(np.float32)
    ang = (np.float32(2.0 * np.pi) * u2).astype(np.float32)
    return (r * np.cos(ang)).astype(np.float32), (r * np.sin(ang)).astype(np.float32)


def layer_normals(seed, step, layer, chain0, n_chains, n_units):
    """Standard normals xi[n_chains, n_units] for one layer at one step.

    ``seed``/``step`` are 64-bit integers, ``layer`` the 0-based latent layer
    index, ``chain0`` the global id of the first chain.  Bit-identical u32
    stream to the device generator; normals agree to fp32 round-off of
    log/sin/cos.
    """
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    step = int(step) & 0xFFFFFFFFFFFFFFFF
    n_groups = (n_units + 3) // 4
    chains = (np.arange(n_chains, dtype=np.uint64) + np.uint64(chain0)).astype(np.uint32)[:, None]
    groups = np.arange(n_groups, dtype=np.uint32)[None, :]
    c1 = (np.uint32(layer) << np.uint32(24)) | groups
    r0, r1, r2, r3 = philox4x32_10(
        chains, c1, np.uint32(step & 0xFFFFFFFF), np.uint32(step >> 32),
        np.uint32(seed & 0xFFFFFFFF), np.uint32(seed >> 32))
    z0, z1 = box_muller(r0, r1)
    z2, z3 = box_muller(r2, r3)
    out = np.stack([z0, z1, z2, z3], axis=-1).reshape(n_chains, n_groups * 4)
    return np.ascontiguousarray(out[:, :n_units])


def layer_u32(seed, step, layer, chain0, n_chains, n_units):
    """Raw u32 stream in the same element order as ``layer_normals`` (for bit-exact tests)."""
    seed = int(seed) & 0xFFFFFFFFFFFFFFFF
    step = int(step) & 0xFFFFFFFFFFFFFFFF
    n_groups = (n_units + 3) // 4
    chains = (np.arange(n_chains, dtype=np.uint64) + np.uint64(chain0)).astype(np.uint32)[:, None]
    groups = np.arange(n_groups, dtype=np.uint32)[None, :]
    c1 = (np.uint32(layer) << np.uint32(24)) | groups
    r = philox4x32_10(chains, c1, np.uint32(step & 0xFFFFFFFF), np.uint32(step >> 32),
                      np.uint32(seed & 0xFFFFFFFF), np.uint32(seed >> 32))
    out = np.stack(r, axis=-1).reshape(n_chains, n_groups * 4)
    return np.ascontiguousarray(out[:, :n_units])


def uniform_pm(seed, stream, shape, lo, hi):
    """Deterministic U(lo, hi) fp32 array (weights / x0 generator for synthetic configs)."""
    n = int(np.prod(shape))
    idx = np.arange((n + 3) // 4, dtype=np.uint64)
    r = philox4x32_10((idx & np.uint64(0xFFFFFFFF)).astype(np.uint32),
                      (idx >> np.uint64(32)).astype(np.uint32),
                      np.uint32(stream), np.uint32(0x5EED),
                      np.uint32(seed & 0xFFFFFFFF), np.uint32((seed >> 32) & 0xFFFFFFFF))
    u = np.stack([u32_to_unit_half_open(x) for x in r], axis=-1).reshape(-1)[:n]
    return (np.float32(lo) + np.float32(hi - lo) * u).astype(np.float32).reshape(shape)
